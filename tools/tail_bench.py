#!/usr/bin/env python3
"""The whole post-encoder model (gcgcn_amd.GraphModelTail: f1 producer -> CAGGC -> f1 -> MAGGC -> f3 head) + the trainer's loss
(f2), forward + backward, on DocRED-shaped synthetic batches: documents/second and where the time goes.

    python tools/tail_bench.py [--B 32] [--N 42] [--S 5] [--T 512] [--live 0.15] [--steps 10] [--ragged] [--layers 2] [--heads 8]

--ragged: DocRED-like entity counts n_valid ~ clip(round(N(19.5, 6^2)), 2, N) in a batch padded to N (what a real batch looks like;
the default computes N entities in every document).  --layers / --heads: sub-layers and heads of the graph blocks (the GloVe model:
2 / 8, glove:250-251; the BERT model: 4 / 4, bert:247-248 -- 32 features per sub-layer).
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
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--N", type=int, default=42)
    ap.add_argument("--S", type=int, default=5)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--live", type=float, default=0.15)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--compact", action="store_true", help="the producers hand over compact rows: E / dE of the hops are never written")
    ap.add_argument("--skip-dead-hop", action="store_true", help="do not compute the hop whose output never reaches the classifier")
    ap.add_argument("--mode", default="both", choices=["eager", "graph", "both"],
                    help="graph: the whole step (producer -> CAGGC -> producer -> MAGGC -> head -> loss, forward + backward) captured in ONE "
                         "hipGraph and replayed, as bench.py times the hot path; eager: launches issued from Python; both (default)")
    a = ap.parse_args()
    import gcgcn_amd
    from gcgcn_amd import _lib
    from producer_bench import synth
    dev = torch.device("cuda:0")
    B, N, S, T, Hd, P, R = a.B, a.N, a.S, a.T, 128, 20, 97
    ctx, node, table, sen, ph, pt = synth(B, N, S, T, Hd, P, a.live, dev)
    ph, pt = ph.to(torch.uint8), pt.to(torch.uint8)
    g = torch.Generator(device=dev).manual_seed(7)
    ner = (torch.randn(7, 20, generator=g, device=dev) * 0.3).requires_grad_()
    table.requires_grad_(), ctx.requires_grad_(), node.requires_grad_()
    ntype = torch.randint(0, 7, (B, N), generator=g, device=dev)
    rel = torch.randint(-10, 11, (B, N, N), generator=g, device=dev)
    labels = (torch.rand(B, N, N, R, generator=g, device=dev) < 0.03).float()
    tail = gcgcn_amd.GraphModelTail(layer_num=a.layers, head_num=a.heads).to(dev).train()
    n_valid = None
    if a.ragged:
        gr = torch.Generator().manual_seed(4242)
        n_valid = torch.clamp(torch.round(torch.randn(B, generator=gr) * 6.0 + 19.5), 2, N).to(torch.int32).to(dev)
        with torch.no_grad():                       # padding rows of the node features must be zero (include/gcgcn.h)
            node.mul_((torch.arange(N, device=dev)[None, :] < n_valid[:, None]).unsqueeze(-1).float())
    tail.compact_edges, tail.skip_dead_hop = a.compact, a.skip_dead_hop
    gcgcn_amd.manual_seed(1337, dev)

    from gcgcn_amd import functional as F_
    rows, pairs = F_.producer_live_counts(sen.view(torch.uint8), n_valid)   # capacities known up front: no host read inside a step

    def step():
        for t in [ctx, node, table, ner] + list(tail.parameters()):
            t.grad = None
        logits = tail(ctx, node, None, sen, ph, pt, ntype, rel, table, ner, n_valid=n_valid, max_live_slots=rows, max_live_pairs=pairs)
        loss = gcgcn_amd.pair_bce_loss(logits, labels, n_valid=n_valid).sum() / B
        loss.backward()

    import _graph_mode
    graph = _graph_mode.capture(step) if a.mode != "eager" else None
    ms_mode = _graph_mode.timed(step, graph, a.steps, a.mode)
    dt = ms_mode.get("graph", ms_mode.get("eager")) / 1e3          # the headline is the graph-mode figure when there is one
    groups = {"producer (f1)": ["prod_", "gemm_dyn"], "classifier head (f3)": ["head_"], "loss (f2)": ["pair_bce"],
              "graph blocks (hot path)": ["gemm_group", "gemm_single", "gcn_chain", "edge_", "mha_core", "head_sum", "gat_", "node_score",
                                          "mask_rows", "softmax", "rowsum", "relu_norm", "dropout"],
              "shared small kernels": ["gemm_splitk_reduce", "colsum"]}
    shares = {}
    for name, prefs in groups.items():
        tot = 0.0
        for f in prefs:
            _lib.call("gcgcn_prof_start", f.encode(), 1024)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
            _lib.call("gcgcn_prof_stop", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
            tot += ms.value / 2
        shares[name] = round(tot, 3)
    print(json.dumps({"metric": "docs/sec fwd+bwd through the whole post-encoder model + loss", "value": round(B / dt, 1), "unit": "docs/s",
                      "ms_per_step": round(dt * 1e3, 3), "ms_per_step_by_mode": {k: round(v, 3) for k, v in ms_mode.items()},
                      "config": {"workload": f"GraphModelTail (2 hops, hidden 128, L={a.layers}, H={a.heads}, R=97) + pair_bce_loss, train mode, B={B} N={N} "
                                             + (f"(ragged: mean n_valid {n_valid.float().mean().item():.1f}) " if a.ragged else "") +
                                             f"S={S} T={T}, {a.live:.0%} of the sentence slots start at token 0, uint8 position ids, " + ("one hipGraph per step (replays)" if "graph" in ms_mode else "eager launches")
                                             + (", compact edge rows (no E / dE tensors)" if a.compact else "") + (", dead last hop skipped" if a.skip_dead_hop else "")},
                      "gpu_ms_per_step_by_part": shares}))


if __name__ == "__main__":
    main()
