#!/usr/bin/env python3
"""In-kernel phase timestamps of the chain kernels (workgroup 0, 100 MHz wall clock), from a trace build of the library:

    make -C gcgcn_amd/csrc -j8 trace                # build/trace.so: chain.hip and the chain_t units with -DGC_T_TRACE
    GCGCN_LIB=$PWD/build/trace.so python tools/trace_chain.py --config c2      (on the GPU box)

Prints microseconds since the kernel's first stamp per slot (slot meanings: the TR / TRB / TS calls in chain_t.hpp / chain.hip).
profiles/r04_chain_phase_trace_before.txt was made this way."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcgcn_amd  # noqa: E402
from gcgcn_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--ragged", action="store_true", help="DocRED-like n_valid (bench.py --ragged); workgroup 0 is document 0")
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
dev = torch.device("cuda:0")
torch.manual_seed(1)
hops = gcgcn_amd.GraphHops(cfg["D"], cfg["L"], cfg["H"]).to(dev).train()
gcgcn_amd.manual_seed(5, dev)
x, e1, e2, adj = bench.synth(cfg, 3, dev)
n_valid = None
if args.ragged:
    g = torch.Generator().manual_seed(4242)
    B, N = x.shape[0], x.shape[1]
    n_valid = torch.clamp(torch.round(torch.randn(B, generator=g) * 6.0 + 19.5), 2, min(42, N)).to(torch.int32).to(dev)
    with torch.no_grad():
        x.mul_((torch.arange(N, device=dev)[None, :] < n_valid[:, None]).unsqueeze(-1).float())
    print("n_valid[0] =", int(n_valid[0]))
for t in (x, e1, e2):
    t.requires_grad_()
h = _lib.lib()
if not hasattr(h, "gcgcn_debug_trace_s"):
    sys.exit("not a trace build: set GCGCN_LIB to a library built by `make -C gcgcn_amd/csrc trace` (chain.hip and the chain_t units compiled with -DGC_T_TRACE)")
buf = (ctypes.c_longlong * 256)()


def show(tag, v, lo, hi):
    if v[lo]:
        print(f"  {tag:28s}", " ".join(f"{i - lo}:{(v[i] - v[lo]) / 100.0:.1f}" for i in range(lo, hi) if v[i]))


for it in range(args.iters):
    hops(x, [e1, e2], adj, n_valid=n_valid)[-1].sum().backward()
    torch.cuda.synchronize()
    print("iteration", it)
    h.gcgcn_debug_trace_s(buf)
    v = list(buf)
    show("chain_s fwd MAGGC", v, 0, 20), show("chain_s bwd MAGGC", v, 20, 64)
    show("chain_s fwd CAGGC", v, 64, 84), show("chain_s bwd CAGGC", v, 84, 128)
    for u in range(4):      # one trace buffer per translation unit of the column-strip kernels (chain_t_u<u>.hip)
        fn = getattr(h, f"gcgcn_debug_trace_t_{u}", None)
        if fn is None:
            continue
        fn(buf)
        v = list(buf)
        show(f"chain_t[u{u}] fwd (last launch)", v, 0, 64), show(f"chain_t[u{u}] bwd MAGGC", v, 100, 164), show(f"chain_t[u{u}] bwd CAGGC", v, 164, 228)
