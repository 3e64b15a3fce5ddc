#!/usr/bin/env python3
"""Cost of the gradient all-reduce variants on ONE GPU (1-rank RCCL group): ms/step of the cfg-2 step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import gcgcn_amd
from bench import CONFIGS, synth

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
cfg = CONFIGS["c2"]; B, N, D, L, H = (cfg[k] for k in "BNDLH")
hops = gcgcn_amd.GraphHops(D, L, H).to(dev).train()
gcgcn_amd.manual_seed(1, dev)
x, e1, e2, adj = synth(cfg, 1, dev)
for t in (x, e1, e2): t.requires_grad_()
cot = torch.ones(B, N, D, device=dev)
params = [p for n, p in hops.named_parameters() if not n.endswith("flat_k")]
arena = torch.empty(sum(p.numel() for p in params), device=dev)

def fwd_bwd():
    x.grad = e1.grad = e2.grad = None
    for p in params: p.grad = None
    out = hops(x, [e1, e2], adj)[-1]
    torch.autograd.backward(out, cot)

def v_none(): fwd_bwd()
def v_async4():
    fwd_bwd(); ws = [dist.all_reduce(p.grad, async_op=True) for p in params]
    for w in ws: w.wait()
def v_sync4():
    fwd_bwd()
    for p in params: dist.all_reduce(p.grad)
def v_one_async():
    fwd_bwd(); dist.all_reduce(arena, async_op=True).wait()
def v_one_sync():
    fwd_bwd(); dist.all_reduce(arena)
def v_coalesced():
    fwd_bwd()
    with dist._coalescing_manager(device=dev, async_ops=False):
        for p in params: dist.all_reduce(p.grad)

for name, fn in [("none", v_none), ("async x4", v_async4), ("sync x4", v_sync4), ("one async", v_one_async),
                 ("one sync", v_one_sync), ("coalesced x4", v_coalesced), ("none", v_none)]:
    try:
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); print(f"{name:14s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step", flush=True)
    except Exception as ex:
        print(name, "failed:", repr(ex)[:200], flush=True)
dist.destroy_process_group()
