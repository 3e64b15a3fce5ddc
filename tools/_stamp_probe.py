import sys, torch
sys.path.insert(0, '/root/repo')
import gcgcn_amd
from gcgcn_amd import functional as F_
dev = torch.device('cuda:0')
B, N, D, L, H = 32, 64, 256, 2, 8
torch.manual_seed(0)
hops = gcgcn_amd.GraphHops(D, L, H).to(dev).eval()
x = torch.randn(B, N, D, device=dev, requires_grad=True)
e1 = torch.randn(B, N, N, D, device=dev, requires_grad=True)
e2 = torch.randn(B, N, N, D, device=dev, requires_grad=True)
for it in range(3):
    out = hops(x, [e1, e2])[-1]
    out.sum().backward()
    torch.cuda.synchronize()
F_._dbg_list = []
F_._dbg_rinv = []
out = hops(x, [e1, e2])[-1]
torch.cuda.synchronize()
fw = [r.clone() for r in F_._dbg_rinv]
out.sum().backward()
torch.cuda.synchronize()
for d in F_._dbg_list:
    print('gcn_bwd drow shape', tuple(d.shape))
    v = d.reshape(-1, 64)[:, :14]
    print(' mean over WGs (us):', [round(float(t), 2) for t in v.mean(0)])
    print(' wave4 mean      (us):', [round(float(t), 2) for t in d.reshape(-1, 64)[:, 16:30].mean(0)])
    print(' WG0:', [round(float(t), 2) for t in v[0]])
    print(' max:', [round(float(t), 2) for t in v.max(0).values])

torch.cuda.synchronize()
