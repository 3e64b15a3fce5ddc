#!/usr/bin/env python3
"""End-to-end training step of the whole model on MI355X, the reference trainer's inner loop (config/Config.py:339-374) as
ONE batched step:

    packed documents --collate (f4, one launch)--> the ten forward tensors --gcgcn_amd.models.GCGCN_glove--> logits
    --pair_bce_loss (f2)--> total_loss / batch_size --backward--> FusedAdam.step() (one launch)

reported as documents/second with the per-part split.  Synthetic DocRED-shaped documents (no dataset in the image): 512
tokens, entity counts ~ clip(round(N(19.5, 6^2)), 2, 42), one to three mentions per entity, an edge for every ordered pair
of entities sharing a sentence (up to max_num = 5 sentence slots per pair), 3 % positive labels.

    python tools/train_step_bench.py [--B 32] [--steps 10] [--skip-dead-hop] [--torch-adam]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


class Cfg:
    """The attributes GCGCN_glove(config) reads (glove:222-279, 306-339); values of config/Config.py:58-126."""
    entity_type_size, coref_size, max_length, keep_prob, graph_hop = 20, 20, 512, 0.8, 2
    dis_size, dis_num, dis_plus, relation_num, alpha = 20, 21, 10, 97, 1.0

    def __init__(self, vocab):
        self.data_word_vec = (np.random.RandomState(1337).randn(vocab, 100) * 0.1).astype(np.float32)


def synth_doc(rs, vocab, T=512, R=97, max_num=5):
    from gcgcn_amd.data import PackedDoc
    I32 = np.int32
    n = int(np.clip(round(rs.normal(19.5, 6.0)), 2, 42))
    bounds = np.sort(rs.choice(np.arange(8, T - 8), size=rs.randint(6, 14), replace=False))
    sents = list(zip([0] + bounds.tolist(), bounds.tolist() + [T]))           # the first sentence starts at token 0
    tokens = np.zeros((T, 3), I32)
    tokens[:, 0] = rs.randint(1, vocab, T)
    mentions, men_ptr, where = [], [0], []
    for e in range(n):
        ss = rs.choice(len(sents), size=rs.randint(1, 4), replace=False)
        for si in ss:
            a, b = sents[si]
            st = rs.randint(a, max(a + 1, b - 3))
            en = min(b, st + rs.randint(1, 4))
            mentions.append((st, en))
            tokens[st:en, 1] = e + 1                                           # coreference id
            tokens[st:en, 2] = rs.randint(1, 7)                                # entity type
            where.append((e, si, st, en))
        men_ptr.append(len(mentions))
    slots, edges, count = [], [], {}
    for si, (a, b) in enumerate(sents):
        here = [(e, st, en) for (e, s2, st, en) in where if s2 == si]
        for (u, hs, he) in here:
            for (v, ts, te) in here:
                if u == v:
                    continue
                j = count.get((u, v), 0)
                if j >= max_num:
                    continue
                if j == 0:
                    edges.append((u, v))
                count[(u, v)] = j + 1
                slots.append((u, v, j, a, b, hs, he, ts, te))
    npos = max(1, int(0.03 * n * n))
    lab = np.stack([rs.randint(0, n, npos), rs.randint(0, n, npos), rs.randint(1, R, npos)], 1).astype(I32)
    lab = lab[lab[:, 0] != lab[:, 1]]
    return PackedDoc(tokens, rs.randint(1, 7, n).astype(I32), np.asarray(men_ptr, I32), np.asarray(mentions, I32).reshape(-1, 2),
                     np.asarray(slots, I32).reshape(-1, 9), np.asarray(edges, I32).reshape(-1, 2), lab, R,
                     max(count.values()) if count else 1, "synthetic")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4, help="resident synthetic batches rotated through the steps")
    ap.add_argument("--vocab", type=int, default=20000)
    ap.add_argument("--skip-dead-hop", action="store_true", help="do not compute the hop whose output never reaches the classifier")
    ap.add_argument("--torch-adam", action="store_true", help="A/B: torch.optim.Adam instead of FusedAdam")
    ap.add_argument("--mode", default="both", choices=["eager", "graph", "both"],
                    help="graph: additionally, forward -> loss -> backward -> optimiser step of each resident batch captured in one hipGraph "
                         "(collate stays on the host, outside the graph) and the replays timed; eager: launches from Python only")
    a = ap.parse_args()
    import gcgcn_amd
    from gcgcn_amd import _lib, functional as F_
    from gcgcn_amd.data import collate
    from gcgcn_amd.models import GCGCN_glove
    from gcgcn_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(7)
    batches = [[synth_doc(rs, a.vocab) for _ in range(a.B)] for _ in range(a.batches)]
    torch.manual_seed(1337)
    model = GCGCN_glove(Cfg(a.vocab)).to(dev).train()
    model.skip_dead_hop = a.skip_dead_hop
    gcgcn_amd.manual_seed(1337, dev)
    F_.check_ids = False                     # the ids of these batches come from collate (validated once below)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = (torch.optim.Adam if a.torch_adam else FusedAdam)(params, lr=1e-4)          # Config.py:300, learn_rate 1e-4 (:72)
    order = ("document", "document_ner", "document_pos", "adj_matrix", "sen_matrix", "pos_matrix_h", "pos_matrix_t", "node_pos",
             "node_type", "node_relative_pos")
    parts = {}

    def timed(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        parts.setdefault(name, []).append((e0, e1))
        return out

    def step(k):
        bt = timed("collate (f4)", lambda: collate(batches[k % len(batches)], dev))
        logits = timed("forward", lambda: model(*[bt[n] for n in order], n_valid=bt["n_valid"]))
        loss = timed("loss (f2)", lambda: gcgcn_amd.pair_bce_loss(logits, bt["label_matrix"], n_valid=bt["n_valid"]).sum() / a.B)
        opt.zero_grad(set_to_none=True)
        timed("backward", lambda: loss.backward())
        timed("optimiser step", lambda: opt.step())
        return loss

    F_.check_ids = True
    step(0)                                  # one checked step: every id in range
    F_.check_ids = False
    for k in range(3):
        step(k)
    torch.cuda.synchronize()
    parts.clear()
    t0 = time.perf_counter()
    losses = [step(k) for k in range(a.steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    split = {n: round(sum(e0.elapsed_time(e1) for e0, e1 in evs) / a.steps, 3) for n, evs in parts.items()}
    ms_mode = {"eager": round(dt * 1e3, 3)}
    if a.mode != "eager":                    # the device side of the step as one hipGraph per resident batch
        import _graph_mode
        resident = [collate(b, dev) for b in batches]

        def model_step(bt):
            logits = model(*[bt[n] for n in order], n_valid=bt["n_valid"])
            loss = gcgcn_amd.pair_bce_loss(logits, bt["label_matrix"], n_valid=bt["n_valid"]).sum() / a.B
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

        graphs = [_graph_mode.capture(lambda bt=bt: model_step(bt)) for bt in resident]
        if all(g is not None for g in graphs):
            cnt = [0]

            def replay():
                graphs[cnt[0] % len(graphs)].replay()
                cnt[0] += 1
            for _ in range(3):
                replay()
            ms_mode["graph (collate outside)"] = round(_graph_mode.wall_ms(replay, a.steps), 3)
    # GPU time of the model's parts inside forward + backward (HIP events around the library's own launches)
    groups = {"encoder (PyTorch: embeddings, BiLSTM, linear_re) + everything outside the library": None,
              "edge-feature producers (f1)": ["prod_", "gemm_dyn"], "classifier head (f3)": ["head_"],
              "graph blocks (hot path)": ["gemm_group", "gemm_single", "gcn_chain", "edge_", "mha_core", "head_sum", "gat_", "node_score",
                                          "mask_rows", "softmax", "rowsum", "relu_norm", "dropout", "gemm_splitk_reduce", "colsum"]}
    by_part = {}
    for name, prefs in groups.items():
        if prefs is None:
            continue
        tot = 0.0
        for f in prefs:
            _lib.call("gcgcn_prof_start", f.encode(), 4096)
            for k in range(2):
                step(k)
            torch.cuda.synchronize()
            ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
            _lib.call("gcgcn_prof_stop", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
            tot += ms.value / 2
        by_part[name] = round(tot, 3)
    nv = np.mean([d.n for b in batches for d in b])
    print(json.dumps({
        "metric": "docs/sec, full training step of GCGCN_glove (collate -> forward -> loss -> backward -> Adam)", "value": round(a.B / dt, 1),
        "unit": "docs/s", "ms_per_step": round(dt * 1e3, 3), "ms_per_step_by_mode": ms_mode, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"B={a.B} DocRED-shaped documents (T=512, mean {nv:.1f} entities, padded per batch), vocabulary {a.vocab}, "
                               f"train mode, {'FusedAdam (one launch)' if not a.torch_adam else 'torch.optim.Adam'}, eager launches, "
                               f"{a.batches} resident packed batches rotated" + (", dead last hop skipped" if a.skip_dead_hop else "")},
        "ms_per_step_by_stage (stream time between HIP events, includes launch gaps)": split,
        "gpu_ms_per_step_by_part (library kernels in forward + backward)": by_part,
        "loss_first_last": [round(losses[0].item(), 5), round(losses[-1].item(), 5)]}))


if __name__ == "__main__":
    main()
