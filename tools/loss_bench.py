#!/usr/bin/env python3
"""SURVEY 8 row f2 measurement: the fused pair-BCE loss (forward + backward) on MI355X against its HBM roofline, with
the trainer's own formulation (N^2 - N nn.BCELoss calls in a Python loop, config/Config.py:355-366) timed beside it on
the host CPU for one document.  Algorithmic bytes: forward reads logits + labels (8 B per element), backward reads both
and writes dlogits (12 B per element)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcgcn_amd
from oracle import gcgcn_oracle as O

dev = torch.device("cuda:0")
B, N, R = 32, 64, 97
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, N, N, R, generator=g) * 3).to(dev).requires_grad_()
y = (torch.rand(B, N, N, R, generator=g) < 0.03).float().to(dev)
w = torch.ones(B, device=dev)

def step():
    x.grad = None
    gcgcn_amd.pair_bce_loss(x, y).backward(w)

for _ in range(20): step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): step()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 200 * 1e3
nbytes = 20.0 * B * N * N * R
print(f"HIP pair_bce fwd+bwd: B={B} N={N} R={R}: {us:.1f} us/step = {B / us * 1e6:.0f} docs/s, "
      f"{nbytes / us / 1e3:.0f} GB/s of algorithmic traffic = {nbytes / us / 1e3 / 8000:.3f} of the 8 TB/s HBM roofline")
xc, yc = x.detach()[0].cpu().requires_grad_(), y[0].cpu()
t0 = time.perf_counter()
O.pair_bce_loss_loop(xc, yc).backward()
dt = time.perf_counter() - t0
print(f"trainer's loop on the host CPU ({torch.get_num_threads()} threads), one document N={N}: {dt:.2f} s = {1 / dt:.2f} docs/s")
