#!/usr/bin/env python3
"""Throughput of the classifier head on MI355X: documents/second forward + backward through
gcgcn_amd.ClassifierHead, and the MFMA roofline of the bilinear kernels (gc::head_bil3_kernel<1..3> / gc::head_dw_kernel at bench
size, gc::head_gemm_kernel<1..4> below 32 768 pairs; fp32 MFMA, operands generated in registers).

    python tools/head_bench.py [--B 32] [--N 64] [--steps 10] [--cpu] [--ragged [--dense]]

--ragged: DocRED-like entity counts n_valid ~ clip(round(N(19.5, 6^2)), 2, min(42, N)) padded to N; the pair passes then run on
the pairs that exist (compacted rows); --dense switches that off (every pair slot of the padded batch, the round-3 path).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_F32_PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--R", type=int, default=97)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--dense", action="store_true", help="with --ragged: compute every pair slot (option head_compact = 0)")
    ap.add_argument("--mode", default="both", choices=["eager", "graph", "both"],
                    help="graph: the step captured in one hipGraph and replayed (how bench.py times the hot path); eager: launches from Python")
    a = ap.parse_args()
    import gcgcn_amd
    from gcgcn_amd import _lib
    dev = torch.device("cuda:0")
    B, N, R = a.B, a.N, a.R
    g = torch.Generator(device=dev).manual_seed(1337)
    head = gcgcn_amd.ClassifierHead(relation_num=R).to(dev)
    feats = [(torch.rand(B, N, 128, generator=g, device=dev) * 2 - 1).requires_grad_() for _ in range(3)]
    ner = (torch.randn(7, 20, generator=g, device=dev) * 0.3).requires_grad_()
    dis = (torch.randn(21, 20, generator=g, device=dev) * 0.3).requires_grad_()
    ntype = torch.randint(0, 7, (B, N), generator=g, device=dev)
    rel = torch.randint(-10, 11, (B, N, N), generator=g, device=dev)
    cot = torch.randn(B, N, N, R, generator=g, device=dev)
    n_valid, real_pairs = None, B * N * N
    if a.ragged:
        gc_ = torch.Generator().manual_seed(4242)
        n_valid = torch.clamp(torch.round(torch.randn(B, generator=gc_) * 6.0 + 19.5), 2, min(42, N)).to(torch.int32).to(dev)
        real_pairs = int((n_valid.long() ** 2).sum().item())
        if a.dense:
            _lib.call("gcgcn_set_option", b"head_compact", 0)

    def step():
        for t in feats + [ner, dis, head.flat]:
            t.grad = None
        out = head(feats, ntype, rel, ner, dis, n_valid=n_valid)
        torch.autograd.backward(out, cot)

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _graph_mode
    graph = _graph_mode.capture(step) if a.mode != "eager" else None
    ms_mode = _graph_mode.timed(step, graph, a.steps, a.mode)
    dt = ms_mode.get("graph", ms_mode.get("eager")) / 1e3      # the headline is the graph-mode figure when there is one
    shares = {}
    for f in ("head_bilinear", "head_gemm", "head_feat", "gemm_splitk_reduce", "colsum"):
        _lib.call("gcgcn_prof_start", f.encode(), 256)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        _lib.call("gcgcn_prof_stop", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        if n.value:
            shares[f] = {"ms_per_step": round(ms.value / 3, 4), "launches_per_step": round(n.value / 3, 1), "work": w.value / 3}
    hb = shares.get("head_bilinear")
    pairs = B * N * N
    computed = real_pairs if (a.ragged and not a.dense and 64 < R <= 97) else pairs
    if hb and computed != pairs:          # the host-side counter assumes every pair slot; the compacted path runs the real ones
        hb["work"] *= computed / pairs
    useful = 2.0 * real_pairs * 128 * 128 * R * 4 + 2.0 * real_pairs * 256 * R * 1     # four bilinear passes + the linear part, unpadded, real pairs
    line = {"metric": "docs/sec fwd+bwd through the classifier head (SURVEY 8 f3)", "value": round(B / dt, 1), "unit": "docs/s",
            "ms_per_step": round(dt * 1e3, 3), "ms_per_step_by_mode": {k: round(v, 3) for k, v in ms_mode.items()},
            "config": {"workload": f"ClassifierHead fwd+bwd, B={B} N={N} ({pairs} pair slots" +
                                   (f", ragged: {real_pairs} real pairs, {'every slot computed' if a.dense else 'compacted rows'}" if a.ragged else "") +
                                   f") hidden=128 R={R}, " + ("one hipGraph per step (replays)" if "graph" in ms_mode else "eager launches")},
            "time_shares_ms_per_step": {k: {kk: vv for kk, vv in v.items() if kk != "work"} for k, v in shares.items()},
            "roofline": None if not hb else {
                "bound": "mfma", "kernel": ("gc::head_bil3_kernel<1..3> + gc::head_dw_kernel" if (pairs >= 32768 or (a.ragged and not a.dense)) else
                                             "gc::head_gemm_kernel<1..4>") + " (bilinear passes)", "launches": hb["launches_per_step"],
                "achieved": round(hb["work"] / (hb["ms_per_step"] * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(hb["work"] / (hb["ms_per_step"] * 1e-3) / MFMA_F32_PEAK, 4),
                "work": "executed fp32 flops (2MNK incl. the padding of R to 128 columns)",
                "useful_frac": round(useful / (hb["ms_per_step"] * 1e-3) / MFMA_F32_PEAK, 4)}}
    if a.cpu:
        from oracle import gcgcn_oracle as O
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        sd = {k: v.cpu() for k, v in head.state_dict().items()}
        sd["ner_emb.weight"], sd["dis_embed.weight"] = ner.detach().cpu(), dis.detach().cpu()
        sd = {k: v.requires_grad_() for k, v in sd.items()}
        fc = [f[0].detach().cpu().requires_grad_() for f in feats]

        def one():
            t = time.perf_counter()
            O.classifier_head(fc, ntype[0].cpu(), rel[0].cpu(), sd).sum().backward()
            return time.perf_counter() - t
        one()
        tw = min(one(), one())
        line["cpu_baseline"] = {"kind": "port", "cores": torch.get_num_threads(), "value": round(1 / tw, 3), "unit": "docs/s",
                                "sample": f"one document N={N}, reference op sequence (nn.Bilinear as an einsum), 2 timed runs"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
