#!/usr/bin/env python3
"""Micro-benchmark of the fp32-MFMA GEMM (gcgcn_gemm) on the shapes the CAGGC/MAGGC stack issues,
beside torch.matmul (rocBLAS/hipBLASLt fp32) as a known-good reference on the same device.
Usage: python tools/gemm_bench.py [c2|c3|c5]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcgcn_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def bench(name, M, N, K, a_kc, b_kc, batch=1, tiles=(1,), splits=(0, 1)):
    A = torch.randn(batch, M, K, device=dev)
    B = torch.randn(batch, K, N, device=dev)
    Ad = (A if a_kc else A.transpose(1, 2)).contiguous()
    Bd = (B.transpose(1, 2) if b_kc else B).contiguous()
    C = torch.empty(batch, M, N, device=dev)
    ws = torch.empty(8 << 20, device=dev)
    flop = 2.0 * M * N * K * batch
    res = []
    for tile in tiles:
        for sp in splits:
            def run():
                _lib.call("gcgcn_gemm", M, N, K, p(Ad), K if a_kc else M, a_kc, p(Bd), K if b_kc else N, b_kc, p(C), N,
                          batch, M * K, K * N, M * N, 1.0, None, 0, 0, tile, sp, p(ws), ws.numel(), None)
            us = timeit(run)
            res.append(f"t{tile}/s{sp}: {us:7.1f}us {flop / us / 1e6:6.1f}TF")
    At = A if a_kc else Ad.transpose(1, 2)
    Bt = Bd.transpose(1, 2) if b_kc else B
    us = timeit(lambda: torch.matmul(At, Bt, out=C))
    ref = torch.matmul(A, B)
    run()
    err = (C - ref).abs().max().item()
    print(f"{name:28s} M={M:5d} N={N:5d} K={K:5d} b={batch:3d} {'NT'[0] if a_kc else 'T'}{'T' if b_kc else 'N'} | "
          + " | ".join(res) + f" | torch {us:7.1f}us {flop / us / 1e6:6.1f}TF | err {err:.1e}", flush=True)


cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
B_, N_, D_, L_, H_ = {"c2": (32, 64, 256, 2, 8), "c3": (32, 64, 768, 4, 4), "c5": (32, 256, 512, 2, 8)}[cfg]
M_ = B_ * N_
HD, gh, dh = H_ * D_, D_ // L_, D_ // H_
bench("Pn=X.WnX (maggc)", M_, HD, D_, 1, 0)
bench("Pn=X.WnX (caggc)", M_, D_, D_, 1, 0)
bench("out=HO.Wlin^T (maggc)", M_, D_, HD, 1, 1)
bench("dHO=dout.Wlin", M_, HD, D_, 1, 0)
bench("dW=X^T.dP (maggc)", D_, HD, M_, 0, 0)
bench("dW=X^T.dP (caggc)", D_, D_, M_, 0, 0)
bench("dX=dP.WnX^T (maggc)", M_, D_, HD, 1, 1)
bench("Y.Wd dense (per head)", M_, gh, gh, 1, 0, batch=H_, splits=(1,))
bench("A_h.Pn_l (b,h)", N_, gh, N_, 1, 0, batch=B_ * H_, splits=(1,))
bench("A_h^T.dM (b,h)", N_, gh, N_, 0, 0, batch=B_ * H_, splits=(1,))
bench("dM.Pn^T (b,h)", N_, N_, gh, 1, 1, batch=B_ * H_, splits=(1,))
bench("Q_h.Q_h^T (b,h)", N_, N_, dh, 1, 1, batch=B_ * H_, splits=(1,))
bench("dWd=Y^T.dP (per head)", gh, gh, M_, 0, 0, batch=H_)
