#!/usr/bin/env python3
"""Throughput of the edge-feature producer on MI355X: documents/second forward + backward through
gcgcn_amd.EdgeFeatureProducer on DocRED-shaped synthetic batches, with per-kernel-family HIP-event times and the
roofline of the kernel that bounds it.

    python tools/producer_bench.py [--B 32] [--N 42] [--S 5] [--T 512] [--H 128] [--live 0.15] [--steps 20] [--cpu]

--live: fraction of sentence slots that contain token 0 -- the only slots the reference keeps (glove:305); DocRED-like
~0.1-0.2 (pairs that co-occur in the first sentence), 1.0 = the dense stress case.
Prints one JSON line (kept under profiles/)."""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12


def synth(B, N, S, T, Hd, P, live, dev, seed=1337):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, device=dev)
    ctx = torch.tanh(torch.randn(B, T, Hd, generator=g, device=dev))
    node = r(B, N, Hd) * 2 - 1
    table = torch.randn(21, P, generator=g, device=dev) * 0.5
    ns = torch.randint(0, S + 1, (B, N, N), generator=g, device=dev)
    slot_on = torch.arange(S, device=dev) < ns.unsqueeze(-1)                          # [B,N,N,S]
    ln = torch.randint(10, 60, (B, N, N, S), generator=g, device=dev)
    first = r(B, N, N, S) < live
    t0 = torch.where(first, torch.zeros_like(ln), torch.randint(1, T - 60, (B, N, N, S), generator=g, device=dev))
    tt = torch.arange(T, device=dev)
    sen = slot_on.unsqueeze(-1) & (tt >= t0.unsqueeze(-1)) & (tt < (t0 + ln).unsqueeze(-1))
    ph = torch.randint(0, 21, sen.shape, generator=g, device=dev) * sen
    pt = torch.randint(0, 21, sen.shape, generator=g, device=dev) * sen
    return ctx, node, table, sen, ph, pt


def word_roofline(wbytes, rng_tokens, word_ms):
    """The word-attention family (prod_word_fwd, prod_word_bwd_rows, prod_word_bwd_tok) against the HBM roof -- by the bytes the
    HBM-side counters saw (the committed rocprofv3 --pmc passes of this tool, profiles/r*_producer_pmc.json: FETCH_SIZE x 2 +
    WRITE_SIZE per launch), not by the bytes its loads ask for: the token states of a document (T x hidden x 4 = 262 KB) and the
    rows' gradients stay on-die across their many readers.  `algorithmic` = what one pass over the live slots' token ranges has
    to move; `requested_over_traffic` shows how much of what the kernels request is served by the caches."""
    import glob
    import re
    r = {"bound": "hbm", "kernel": "gc::prod_word_fwd_kernel + prod_word_bwd_rows_kernel + prod_word_bwd_tok_kernel",
         "algorithmic_bytes_per_step": wbytes, "algorithmic": f"3 passes x 2 sides x {int(rng_tokens)} tokens in live slots' ranges x 4 x hidden bytes",
         "ms_per_step": word_ms, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None}
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_producer_pmc.json")))
    if pm and word_ms:
        ks = json.load(open(pm[-1]))["kernels"]
        per = {k: v["hbm_bytes_per_launch_corrected"] for k, v in ks.items() if re.search(r"prod_word_(fwd|bwd_rows|bwd_tok)_kernel", k)
               and "hbm_bytes_per_launch_corrected" in v}
        if per:
            traffic = float(sum(per.values()))
            r.update(traffic=traffic, traffic_by_kernel={re.sub(r".*(prod_word_\w+?)_kernel.*", r"\1", k): v for k, v in per.items()},
                     pmc_source=os.path.relpath(pm[-1], ROOT), achieved=round(traffic / (word_ms * 1e-3) / 1e9, 1),
                     frac=round(traffic / (word_ms * 1e-3) / HBM_PEAK, 4), requested_over_traffic=round(wbytes / traffic, 2),
                     note="achieved = HBM-side bytes of the three launches (counters) / their time; prod_word_bwd_tok reads each live "
                          "slot's gradient row (2 x hidden floats) once per token of the slot -- one workgroup per (document, token) "
                          "collects every row that covers its token, in a fixed order (deterministic, no atomics) -- which is why its "
                          "counter traffic is ~30 x the rows' own bytes")
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--N", type=int, default=42)
    ap.add_argument("--S", type=int, default=5)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--P", type=int, default=20)
    ap.add_argument("--live", type=float, default=0.15)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--ids", default="int64", choices=["int64", "uint8"])
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (reference op sequence) on a reduced document")
    ap.add_argument("--mode", default="both", choices=["eager", "graph", "both"],
                    help="graph: the step captured in one hipGraph and replayed (how bench.py times the hot path); eager: launches from Python")
    a = ap.parse_args()
    import gcgcn_amd
    from gcgcn_amd import _lib, functional as F_
    dev = torch.device("cuda:0")
    B, N, S, T, Hd, P = a.B, a.N, a.S, a.T, a.H, a.P
    ctx, node, table, sen, ph, pt = synth(B, N, S, T, Hd, P, a.live, dev)
    if a.ids == "uint8":
        ph, pt = ph.to(torch.uint8), pt.to(torch.uint8)
    prod = gcgcn_amd.EdgeFeatureProducer(Hd, P).to(dev)
    for t in (ctx, node, table):
        t.requires_grad_()
    rows, pairs = F_.producer_live_counts(sen.view(torch.uint8))
    cot = torch.randn(B, N, N, Hd, device=dev)

    def step():
        ctx.grad = node.grad = table.grad = prod.flat.grad = None
        e = prod(ctx, sen, ph, pt, node, table, max_live_slots=rows, max_live_pairs=pairs)   # capacities known: no host sync
        torch.autograd.backward(e, cot)

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _graph_mode
    graph = _graph_mode.capture(step) if a.mode != "eager" else None
    ms_mode = _graph_mode.timed(step, graph, a.steps, a.mode)
    dt = ms_mode.get("graph", ms_mode.get("eager")) / 1e3      # the headline is the graph-mode figure when there is one

    fams = ["gemm_single", "prod_gemm", "gemm_dyn", "gemm_splitk_reduce", "prod_index", "prod_table", "prod_word", "prod_sent",
            "prod_expand", "prod_gather", "prod_colsum", "colsum"]
    shares = {}
    for f in fams:
        _lib.call("gcgcn_prof_start", f.encode(), 64 * 8)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        _lib.call("gcgcn_prof_stop", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        if n.value:
            shares[f] = {"ms_per_step": round(ms.value / 3, 4), "launches_per_step": round(n.value / 3, 1)}
    ebytes = 4.0 * B * N * N * Hd
    exp_ms = shares.get("prod_expand", {}).get("ms_per_step", 0)
    # The dominant kernels are the word-attention ones (prod_word_fwd, prod_word_bwd_rows, prod_word_bwd_tok).  Their algorithmic
    # traffic: every live slot's token range [first .. last set token] of the token states, 4 * hidden bytes per token, for
    # both sides (head / tail distance embeddings) -- read once forward, once by each of the two backward passes.
    sl = sen[..., 0]                                             # live slots: token 0 belongs to them (glove:305)
    pos = torch.arange(T, device=dev)
    last = (sen.to(torch.int32) * (pos + 1)).amax(-1)            # last set token + 1 (0 = empty slot)
    first = torch.where(sen, pos, torch.full_like(pos, T)).amin(-1)
    rng_tokens = ((last - first).clamp_min(0) * sl).sum().item()
    wbytes = 3 * 2 * rng_tokens * 4.0 * Hd
    word_ms = shares.get("prod_word", {}).get("ms_per_step", 0)
    line = {"metric": "docs/sec fwd+bwd through the edge-feature producer (SURVEY 8 f1)", "value": round(B / dt, 1), "unit": "docs/s",
            "ms_per_step": round(dt * 1e3, 4), "ms_per_step_by_mode": {k: round(v, 4) for k, v in ms_mode.items()},
            "config": {"workload": f"EdgeFeatureProducer fwd+bwd, B={B} N={N} S={S} T={T} hidden={Hd} dis_size={P}, "
                                   f"{rows} live slots of {B * N * N * S} ({rows / (B * N * N * S):.1%}), {pairs} live pairs of {B * N * N}, "
                                   f"position ids {a.ids}, " + ("one hipGraph per step (replays)" if "graph" in ms_mode else "eager launches")},
            "input_bytes_per_doc": int((sen.element_size() * sen.numel() + 2 * ph.element_size() * ph.numel()) / B),
            "reference_materialises_bytes_per_doc": 4 * N * N * S * T * Hd,
            "time_shares_ms_per_step": shares,
            "roofline": word_roofline(wbytes, rng_tokens, word_ms),
            "roofline_expand": {"bound": "hbm", "kernel": "gc::prod_expand_kernel (writes E[B,N,N,hidden])", "work_per_launch": ebytes,
                         "achieved": round(ebytes / (exp_ms * 1e-3) / 1e9, 1) if exp_ms else None, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(ebytes / (exp_ms * 1e-3) / HBM_PEAK, 4) if exp_ms else None}}
    if a.cpu:
        from oracle import gcgcn_oracle as O
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        n2, s2, t2 = 12, 3, 128                                     # the as-written op sequence needs [N,N,S,T,H] tensors
        c2, nd2, tb2, se2, p2, q2 = (t.cpu() for t in synth(1, n2, s2, t2, Hd, P, a.live, dev, seed=7))
        sd = {f"{k.split('.', 1)[0]}.0.{k.split('.', 1)[1]}": v.cpu().requires_grad_() for k, v in prod.state_dict().items()}
        c2 = c2.detach().requires_grad_()

        def one(fn, args):
            t = time.perf_counter()
            fn(*args).sum().backward()
            return time.perf_counter() - t
        one(O.edge_features, (c2[0], se2[0], p2[0], q2[0], nd2[0].detach(), tb2.detach(), sd, 0))
        tw = one(O.edge_features, (c2[0], se2[0], p2[0], q2[0], nd2[0].detach(), tb2.detach(), sd, 0))
        line["cpu_baseline"] = {"kind": "port", "cores": torch.get_num_threads(), "value": round(1 / tw, 3), "unit": "docs/s",
                                "sample": f"ONE reduced document N={n2} S={s2} T={t2} hidden={Hd}, reference op sequence "
                                          f"(materialises [N,N,S,T,hidden]); the full N={N} S={S} T={T} document needs "
                                          f"{4 * N * N * S * T * Hd / 1e9:.1f} GB per intermediate tensor"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
