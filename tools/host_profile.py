#!/usr/bin/env python3
"""Host-side cost of one eager step (cProfile over 300 steps at cfg 1, where the GPU is far ahead of the host)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcgcn_amd
from bench import CONFIGS, synth
from gcgcn_amd.dist import FlatGradBucket

dev = torch.device("cuda:0")
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c1"]; B, N, D, L, H = (cfg[k] for k in "BNDLH")
hops = gcgcn_amd.GraphHops(D, L, H).to(dev).train()
gcgcn_amd.manual_seed(1, dev)
x, e1, e2, adj = synth(cfg, 1, dev)
for t in (x, e1, e2): t.requires_grad_()
cot = torch.ones(B, N, D, device=dev)
bucket = FlatGradBucket(hops)

def step():
    x.grad = e1.grad = e2.grad = None
    bucket.zero_grad()
    out = hops(x, [e1, e2], adj)[-1]
    torch.autograd.backward(out, cot)

for _ in range(50): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host issue time {1e3 * (t1 - t0) / 300:.3f} ms/step, with sync {1e3 * (time.perf_counter() - t0) / 300:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
