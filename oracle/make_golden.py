#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes on CPU.

Build-container only: imports /root/reference/models/GCGCN_glove.py by file path (the
package import is broken upstream) behind a stub for the absent
``pytorch_pretrained_bert`` module.  Nothing from the reference is copied: the output
files hold only numeric inputs, parameters, outputs and gradients.  The reference does
not exist on the GPU box; tests read the committed .npz files instead.

Usage:  python oracle/make_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF_ROOT = "/root/reference"
REF_FILE = REF_ROOT + "/models/GCGCN_glove.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    sys.dont_write_bytecode = True          # the reference tree is read-only
    stub = types.ModuleType("pytorch_pretrained_bert")
    stub.BertModel = object                 # only needed for the import line (glove:13)
    sys.modules["pytorch_pretrained_bert"] = stub
    spec = importlib.util.spec_from_file_location("ref_glove", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def _np(t):
    return t.detach().cpu().numpy()


def _pack(inputs, module, out, cot, grads_in, extra=None):
    d = {}
    for k, v in inputs.items():
        d["in." + k] = _np(v)
    for k, v in module.state_dict().items():
        d["sd." + k] = _np(v)
    d["out"] = _np(out)
    d["cot"] = _np(cot)
    for k, v in grads_in.items():
        d["grad.in." + k] = _np(v)
    for k, p in module.named_parameters():
        if p.grad is not None:
            d["grad.sd." + k] = _np(p.grad)
    for k, v in (extra or {}).items():
        d[k] = np.asarray(v)
    return d


def _inputs(n, d, seed, zero_rows=()):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, d, generator=g) * 2 - 1).requires_grad_()
    adj01 = (torch.rand(n, n, generator=g) < 0.3).float() * (1 - torch.eye(n))
    e = (torch.randn(n, n, d, generator=g) * 0.5 * adj01.unsqueeze(-1)).requires_grad_()
    e2 = (torch.randn(n, n, d, generator=g) * 0.5).requires_grad_()
    a = torch.rand(n, n, generator=g)
    for r in zero_rows:
        a[r] = 0
    a = a.requires_grad_()
    return g, x, e, e2, adj01, a


def block_cases(ref):
    shapes = [(5, 8, 2, 2), (16, 128, 2, 8), (7, 12, 4, 4), (16, 64, 4, 4), (1, 8, 2, 2), (2, 8, 2, 2)]
    for (n, d, L, H) in shapes:
        for seed in (1337, 0):
            if (n <= 2 or d >= 64) and seed == 0:
                continue      # big / degenerate shapes: one seed keeps the fixtures small
            tag = f"n{n}_d{d}_l{L}_h{H}_s{seed}"
            torch.manual_seed(seed)
            meta = {"meta.n": n, "meta.d": d, "meta.l": L, "meta.h": H}

            # GraphConv with an all-zero adjacency row (exercises glove:47-49)
            g, x, e, e2, adj01, a = _inputs(n, d, seed, zero_rows=(0,) if n > 1 else ())
            m = ref.GraphConv(d, d, d // L).eval()
            out = m(x, e, a)
            cot = torch.randn(out.shape, generator=g)
            out.backward(cot)
            yield f"graphconv_{tag}", _pack({"x": x, "e": e, "adj": a}, m, out, cot,
                                             {"x": x.grad, "e": e.grad, "adj": a.grad}, meta)

            # GATAttention: mask given (must be a no-op) — glove:154-168
            g, x, e, e2, adj01, a = _inputs(n, d, seed)
            m = ref.GATAttention(d, d).eval()
            mask = torch.eq(adj01, 0)
            out = m(x, e, mask)
            out_nomask = m(x, e, None)
            out_allmask = m(x, e, torch.ones(n, n, dtype=torch.bool))
            assert torch.equal(out, out_nomask) and torch.equal(out, out_allmask)
            cot = torch.randn(out.shape, generator=g)
            out.backward(cot)
            yield f"gat_{tag}", _pack({"x": x, "e": e, "mask": mask}, m, out, cot,
                                       {"x": x.grad, "e": e.grad}, meta)

            # GraphConvolution (CAGGC conv) — glove:52-80
            g, x, e, e2, adj01, a = _inputs(n, d, seed, zero_rows=(n - 1,) if n > 2 else ())
            m = ref.GraphConvolution(L, d, d).eval()
            out = m(x, e, a)
            cot = torch.randn(out.shape, generator=g)
            out.backward(cot)
            yield f"caggc_{tag}", _pack({"x": x, "e": e, "adj": a}, m, out, cot,
                                         {"x": x.grad, "e": e.grad, "adj": a.grad}, meta)

            # MultiHeadAttention (E passed in the ignored slot, as glove:336 does)
            g, x, e, e2, adj01, a = _inputs(n, d, seed)
            m = ref.MultiHeadAttention(H, d).eval()
            outs = m(x, e2)
            cots = [torch.randn(o.shape, generator=g) for o in outs]
            torch.autograd.backward(outs, cots)
            assert e2.grad is None
            yield f"mha_{tag}", _pack({"x": x}, m, torch.stack(outs), torch.stack(cots),
                                       {"x": x.grad}, meta)

            # MultiGraphConvolution (MAGGC conv) — glove:82-120
            g, x, e, e2, adj01, a = _inputs(n, d, seed)
            al = [torch.rand(n, n, generator=g).requires_grad_() for _ in range(H)]
            m = ref.MultiGraphConvolution(L, H, d, d).eval()
            out = m(x, e2, al)
            cot = torch.randn(out.shape, generator=g)
            out.backward(cot)
            yield f"maggc_{tag}", _pack({"x": x, "e": e2, "adj": torch.stack(al)}, m, out, cot,
                                         {"x": x.grad, "e": e2.grad,
                                          "adj": torch.stack([t.grad for t in al])}, meta)

            # chained GAT -> CAGGC conv -> MHA -> MAGGC conv (glue of glove:329-341, alpha=1)
            g, x, e, e2, adj01, a = _inputs(n, d, seed)
            gat = ref.GATAttention(d, d).eval()
            cag = ref.GraphConvolution(L, d, d).eval()
            mha = ref.MultiHeadAttention(H, d).eval()
            mag = ref.MultiGraphConvolution(L, H, d, d).eval()
            a0 = gat(x, e, torch.eq(adj01, 0))
            x1 = cag(x, e, a0)
            al = mha(x1, e2)
            x2 = mag(x1, e2, al)
            cot = torch.randn(x2.shape, generator=g)
            x2.backward(cot)
            holder = torch.nn.Module()
            holder.get_weighted_adj_matrix = gat
            holder.get_adj_matrix = torch.nn.ModuleList([mha])
            holder.graphcnn = torch.nn.ModuleList([cag, mag])
            pk = _pack({"x": x, "e1": e, "e2": e2, "adj": adj01}, holder, x2, cot,
                       {"x": x.grad, "e1": e.grad, "e2": e2.grad}, meta)
            pk["mid.a0"] = _np(a0)
            pk["mid.x1"] = _np(x1)
            pk["mid.al"] = _np(torch.stack(al))
            yield f"stack_{tag}", pk


class _Cfg:
    """Duck-typed config for GCGCN_glove(config) (attributes read at glove:222-279,306-339)."""
    entity_type_size = 20
    coref_size = 20
    max_length = 512
    keep_prob = 1.0
    graph_hop = 2
    dis_size = 20
    dis_num = 21
    dis_plus = 10
    relation_num = 97
    alpha = 1.0

    def __init__(self, vocab):
        rs = np.random.RandomState(1337)
        self.data_word_vec = rs.randn(vocab, 100).astype(np.float32) * 0.1


def full_model_case(ref, docs=8, n=16, t=40, s=3, vocab=200):
    """cfg 1: the full GCGCN_glove forward on 8 synthetic docs; hooks record what the four
    hot-path modules saw and returned inside the real model."""
    torch.manual_seed(1337)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.GCGCN_glove(_Cfg(vocab)).eval()
    rec = {}

    def hook(name):
        def f(mod, inp, out):
            rec[name + ".in"] = [i for i in inp]
            rec[name + ".out"] = out
        return f
    model.get_weighted_adj_matrix.register_forward_hook(hook("gat"))
    model.graphcnn[0].register_forward_hook(hook("cag"))
    model.get_adj_matrix[0].register_forward_hook(hook("mha"))
    model.graphcnn[1].register_forward_hook(hook("mag"))

    g = torch.Generator().manual_seed(1337)
    pk = {"meta.n": n, "meta.d": 128, "meta.l": 2, "meta.h": 8, "meta.docs": docs}
    hot = ("get_weighted_adj_matrix.", "get_adj_matrix.", "graphcnn.")
    for k, v in model.state_dict().items():
        if k.startswith(hot):
            pk["sd." + k] = _np(v)
    for di in range(docs):
        document = torch.randint(1, vocab, (t,), generator=g)
        ner = torch.randint(0, 7, (t,), generator=g)
        pos = torch.randint(0, n + 1, (t,), generator=g)
        adj = (torch.rand(n, n, generator=g) < 0.3).float() * (1 - torch.eye(n))
        sen = torch.zeros(n, n, s, t, dtype=torch.bool)
        for i in range(n):
            for j in range(n):
                if adj[i, j] > 0:
                    for k in range(int(torch.randint(1, s + 1, (1,), generator=g))):
                        a0 = int(torch.randint(0, t - 8, (1,), generator=g))
                        sen[i, j, k, a0:a0 + 8] = True
        ph = torch.randint(0, 21, (n, n, s, t), generator=g)
        pt = torch.randint(0, 21, (n, n, s, t), generator=g)
        node_pos = torch.zeros(n, t)
        for i in range(n):
            a0 = int(torch.randint(0, t - 3, (1,), generator=g))
            node_pos[i, a0:a0 + 3] = 1.0 / 3
        node_type = torch.randint(0, 7, (n,), generator=g)
        rel = torch.randint(-10, 11, (n, n), generator=g)
        with torch.no_grad():
            logits = model(document, ner, pos, adj, sen, ph, pt, node_pos, node_type, rel)
        p = f"doc{di}."
        pk[p + "adj"] = _np(adj)
        pk[p + "x0"] = _np(rec["gat.in"][0])
        pk[p + "e1"] = _np(rec["gat.in"][1])
        pk[p + "a0"] = _np(rec["gat.out"])
        pk[p + "x1_new"] = _np(rec["cag.out"])
        pk[p + "x1"] = _np(rec["mha.in"][0])          # eval: dropout identity, alpha=1 -> == x1_new
        pk[p + "e2"] = _np(rec["mag.in"][1])
        pk[p + "al"] = _np(torch.stack(rec["mha.out"]))
        pk[p + "x2_new"] = _np(rec["mag.out"])
        pk[p + "logits_sum"] = np.float64(logits.double().sum().item())
    return "model_c1", pk


def producer_cases(ref):
    """SURVEY 8 row f1 (groundwork): the edge-feature producer run with the reference's own WordAttention /
    SentenceAttention classes in the order of the model's forward (glove:300-330).  Inputs, parameters (under the model's
    key names for hop 0), E and the gradients of sum(E * cot) w.r.t. the token states, the node features, the distance
    table and every parameter."""
    for (n, s, t, hd, p, seed) in ((5, 3, 11, 16, 6, 1), (4, 2, 7, 8, 4, 2)):
        torch.manual_seed(seed)
        wa = ref.WordAttention(hd, hd, position_dim=p)
        sa = ref.SentenceAttention(hd, hd)
        lw, ls = torch.nn.Linear(2 * hd, hd), torch.nn.Linear(2 * hd, hd)
        g = torch.Generator().manual_seed(seed)
        table = (torch.randn(21, p, generator=g) * 0.5).requires_grad_()
        ctx = torch.tanh(torch.randn(1, t, hd, generator=g)).requires_grad_()
        node = torch.randn(n, hd, generator=g).requires_grad_()
        sen = torch.rand(n, n, s, t, generator=g) < 0.4
        sen[:, :, 0, 0] = True                       # first sentence slot is real everywhere
        sen[:, :, s - 1, :] = False                  # last slot is padding ...
        sen[0, 1, s - 1, 0] = True                   # ... except for one pair: no padded slot -> division by 1e-10 (glove:212)
        ph = torch.randint(0, 21, (n, n, s, t), generator=g)
        pt = torch.randint(0, 21, (n, n, s, t), generator=g)
        wpad = ~sen.unsqueeze(4)                                                               # glove:302
        ctxe = ctx.unsqueeze(0).unsqueeze(0).expand(n, n, s, -1, -1)                           # glove:303
        spad = ~sen[:, :, :, 0:1]                                                              # glove:305
        cw = torch.cat([wa(wpad, ctxe, table[ph]), wa(wpad, ctxe, table[pt])], 3)              # glove:317-320
        cwa = lw(cw)                                                                           # glove:321
        nh = node.unsqueeze(0).unsqueeze(2).expand_as(cwa)                                     # glove:324
        nt = node.unsqueeze(1).unsqueeze(2).expand_as(cwa)                                     # glove:325
        e = ls(torch.cat([sa(spad, cwa, nh), sa(spad, cwa, nt)], 2))                           # glove:327-330
        cot = torch.randn(e.shape, generator=g) * (e.abs() < 1e3).float()                      # keep the 1e10-scaled pair out of the loss
        (e * cot).sum().backward()
        d = {"ctx": _np(ctx[0]), "node": _np(node), "table": _np(table), "sen": _np(sen), "pos_h": _np(ph), "pos_t": _np(pt),
             "out": _np(e), "cot": _np(cot), "grad.ctx": _np(ctx.grad[0]), "grad.node": _np(node.grad), "grad.table": _np(table.grad)}
        for pre, m in (("word_attention.0", wa), ("sentence_attention.0", sa), ("linear_word_att.0", lw), ("linear_sentence_att.0", ls)):
            for k, v in m.state_dict().items():
                d[f"sd.{pre}.{k}"] = _np(v)
            for k, v in m.named_parameters():
                d[f"grad.sd.{pre}.{k}"] = _np(v.grad)
        yield f"producer_n{n}_s{s}_t{t}_h{hd}", d


def tail_case(ref, docs=2, n=12, t=48, s=3, vocab=120):
    """The whole post-encoder model on the REAL GCGCN_glove: forward hooks / a wrapped ``linear_re`` record the token states
    (``context_output``, glove:292) of each document; the fixture holds them, the raw inputs, every parameter from the
    producers to the classifier (the 6.4 MB bilinear weight as a seed, see head_bilinear_weight) and the model's logits.
    tests/ run EdgeFeatureProducer -> GAT / CAGGC -> producer -> MHA / MAGGC -> ClassifierHead on the same inputs."""
    import contextlib, io
    torch.manual_seed(4242)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.GCGCN_glove(_Cfg(vocab)).eval()
    with torch.no_grad():
        model.bili_layer_01.weight.copy_(head_bilinear_weight(4242))
    rec = {}
    model.linear_re.register_forward_hook(lambda m, i, o: rec.__setitem__("pre_tanh", o))
    g = torch.Generator().manual_seed(4242)
    pk = {"meta.docs": docs, "meta.bili_seed": np.int64(4242)}
    tail = ("get_weighted_adj_matrix.", "get_adj_matrix.", "graphcnn.", "word_attention.", "sentence_attention.", "linear_word_att.",
            "linear_sentence_att.", "dense_layer.", "classification_layer_01.", "bili_layer_01.bias", "dis_embed.", "ner_emb.")
    for k, v in model.state_dict().items():
        if k.startswith(tail):
            pk["sd." + k] = _np(v)
    for di in range(docs):
        document = torch.randint(1, vocab, (t,), generator=g)
        ner = torch.randint(0, 7, (t,), generator=g)
        pos = torch.randint(0, n + 1, (t,), generator=g)
        adj = (torch.rand(n, n, generator=g) < 0.3).float() * (1 - torch.eye(n))
        sen = torch.zeros(n, n, s, t, dtype=torch.bool)
        for i in range(n):
            for j in range(n):
                if adj[i, j] > 0:
                    for k in range(int(torch.randint(1, s + 1, (1,), generator=g))):
                        ln = int(torch.randint(4, 12, (1,), generator=g))
                        a0 = 0 if torch.rand(1, generator=g).item() < 0.5 else int(torch.randint(0, t - ln, (1,), generator=g))
                        sen[i, j, k, a0:a0 + ln] = True               # half of the sentences start at token 0: live slots (glove:305)
        ph = torch.randint(0, 21, (n, n, s, t), generator=g)
        pt = torch.randint(0, 21, (n, n, s, t), generator=g)
        node_pos = torch.zeros(n, t)
        for i in range(n):
            a0 = int(torch.randint(0, t - 3, (1,), generator=g))
            node_pos[i, a0:a0 + 3] = 1.0 / 3
        node_type = torch.randint(0, 7, (n,), generator=g)
        rel = torch.randint(-10, 11, (n, n), generator=g)
        with torch.no_grad():
            logits = model(document, ner, pos, adj, sen, ph, pt, node_pos, node_type, rel)
        ctx = torch.tanh(rec["pre_tanh"])[0]                                           # glove:292
        p = f"doc{di}."
        pk[p + "ctx"], pk[p + "node_pos"], pk[p + "adj"] = _np(ctx), _np(node_pos), _np(adj)
        pk[p + "sen"], pk[p + "pos_h"], pk[p + "pos_t"] = _np(sen), _np(ph.to(torch.uint8)), _np(pt.to(torch.uint8))
        pk[p + "node_type"], pk[p + "rel"], pk[p + "logits"] = _np(node_type), _np(rel), _np(logits)
    return "tail_c1", pk


def model_step_case(ref, docs=2, n=10, t=40, s=3, vocab=90):
    """The reference's model AND its trainer's step arithmetic, end to end: a real ``GCGCN_glove(config)`` (toy vocabulary) runs
    ``docs`` synthetic documents from their RAW inputs (token / entity-type / coreference ids and the graph tensors of
    ``Config.from_list_to_tensor``); every document's loss comes from the trainer's own statements (``config/Config.py:302,
    355-364``, ``trainer_loss_lines``), the documents' losses are summed and divided by ``batch_size`` and ONE ``backward()``
    runs, exactly as ``Config.py:366-372`` does with ``batch_size = docs``.  The fixture holds the full ``state_dict`` (the
    6.4 MB bilinear weight as a seed), the raw inputs, logits, losses and the gradient of every parameter (``None`` for the
    dead last hop).  eval() mode: the model's dropout draws come from torch's global generator and cannot be
    replayed on another implementation; ``keep_prob = 1`` anyway switches the encoder's LockedDropout off."""
    import contextlib, io
    from torch.autograd import Variable
    torch.manual_seed(777)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.GCGCN_glove(_Cfg(vocab)).eval()
    with torch.no_grad():
        model.bili_layer_01.weight.copy_(head_bilinear_weight(777))
        # the sentence-level weights are relu(score) (glove:211): with the default initialisation every score of these toy
        # documents is negative, E is the bias everywhere and the producers' gradients are exactly zero -- lift the bias so
        # that the fixture exercises them
        for sa in model.sentence_attention:
            sa.attention_all.bias.fill_(0.4)
    code = compile(trainer_loss_lines(), "Config.py:302,355-364", "exec")
    g = torch.Generator().manual_seed(777)
    pk = {"meta.docs": docs, "meta.bili_seed": np.int64(777), "meta.vocab": np.int64(vocab)}
    for k, v in model.state_dict().items():
        if k != "bili_layer_01.weight":
            pk["sd." + k] = _np(v)
    pk["names.keys"] = np.array(list(model.state_dict().keys()))          # the reference's key order
    total_loss = 0
    for di in range(docs):
        document = torch.randint(1, vocab, (t,), generator=g)
        document[t - 6:] = 0                                              # padding tokens at the end, as from_list_to_tensor leaves them
        ner = torch.randint(0, 7, (t,), generator=g)
        pos = torch.randint(0, n + 1, (t,), generator=g)
        adj = (torch.rand(n, n, generator=g) < 0.3).float() * (1 - torch.eye(n))
        sen = torch.zeros(n, n, s, t, dtype=torch.bool)
        for i in range(n):
            for j in range(n):
                if adj[i, j] > 0:
                    for k in range(int(torch.randint(1, s + 1, (1,), generator=g))):
                        ln = int(torch.randint(4, 12, (1,), generator=g))
                        a0 = 0 if torch.rand(1, generator=g).item() < 0.5 else int(torch.randint(0, t - ln, (1,), generator=g))
                        sen[i, j, k, a0:a0 + ln] = True
        ph = torch.randint(0, 21, (n, n, s, t), generator=g)
        pt = torch.randint(0, 21, (n, n, s, t), generator=g)
        node_pos = torch.zeros(n, t)
        for i in range(n):
            a0 = int(torch.randint(0, t - 3, (1,), generator=g))
            node_pos[i, a0:a0 + 3] = 1.0 / 3
        node_type = torch.randint(0, 7, (n,), generator=g)
        rel = torch.randint(-10, 11, (n, n), generator=g)
        labels = (torch.rand(n, n, 97, generator=g) < 0.04).float()
        logits = model(document, ner, pos, adj, sen, ph, pt, node_pos, node_type, rel)
        ns = {"torch": torch, "nn": nn, "Variable": Variable, "predict_re": logits, "label_matrix": labels}
        exec(code, ns)                                                     # Config.py:355-364
        total_loss = total_loss + ns["temp_loss"]                          # Config.py:366
        p = f"doc{di}."
        pk[p + "document"], pk[p + "ner"], pk[p + "pos"] = _np(document), _np(ner), _np(pos)
        pk[p + "node_pos"], pk[p + "adj"], pk[p + "sen"] = _np(node_pos), _np(adj), _np(sen)
        pk[p + "pos_h"], pk[p + "pos_t"] = _np(ph.to(torch.uint8)), _np(pt.to(torch.uint8))
        pk[p + "node_type"], pk[p + "rel"], pk[p + "labels"] = _np(node_type), _np(rel), _np(labels.to(torch.uint8))
        pk[p + "logits"], pk[p + "loss"] = _np(logits), _np(ns["temp_loss"].squeeze(0))
    total_loss = total_loss / docs                                         # Config.py:369 (batch_size = docs)
    model.zero_grad()
    total_loss.backward()                                                  # Config.py:371
    pk["total_loss"] = _np(total_loss.squeeze(0))
    none = []
    for k, prm in model.named_parameters():
        if prm.grad is None:
            none.append(k)
        elif k == "bili_layer_01.weight":
            pk["gradpart.bili.r"] = np.array(HEAD_BILI_SLICES)
            pk["gradpart.bili.slices"] = _np(prm.grad[HEAD_BILI_SLICES])
            pk["gradpart.bili.sum_r"] = _np(prm.grad.sum(0))
        else:
            pk["grad.sd." + k] = _np(prm.grad)
    pk["names.grad_none"] = np.array(none)
    return "model_step_c1", pk


def tensorise_cases():
    """SURVEY 8 row f4, pinned on the reference's own function: ``Config.from_list_to_tensor`` (config/Config.py:162-233) and the
    ``dis2idx`` table (:106-116) are read as text and executed on synthetic documents (dicts with a networkx.DiGraph, the
    reference's pickle format, gen_data_extend_graph.py:289-307).  Writes the documents in the packed format
    (tests/golden/tensorise_docs.npz, gcgcn_amd.data.PackedDocs) and the reference's tensors for each of them
    (tensorise_ref.npz).  Cases: overlapping mentions, a document longer than max_length, more sentence slots than max_num,
    an edge without sentences, tokens inside / left / right of a mention, far-apart entities."""
    import textwrap
    import networkx as nx
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from gcgcn_amd.data import PackedDocs, pack_document
    lines = open(os.path.join(REF_ROOT, "config", "Config.py"), encoding="utf-8").read().split("\n")
    assert lines[161].strip().startswith("def from_list_to_tensor(self,item,train = True):"), lines[161]
    assert lines[232].strip().startswith("return new_item"), lines[232]
    assert lines[105].strip().startswith("self.dis2idx = np.zeros((1024)"), lines[105]
    ns = {"torch": torch, "np": np}
    exec(compile(textwrap.dedent("\n".join(lines[161:233])), "Config.py:162-233", "exec"), ns)

    class Self:
        pass
    me = Self()
    exec(compile(textwrap.dedent("\n".join(lines[105:116])), "Config.py:106-116", "exec"), {"self": me, "np": np})
    me.dis_plus = 10
    rs = np.random.RandomState(1337)
    docs, refs, cfgs = [], {}, []
    for di, (n, tlen, maxlen, maxnum) in enumerate(((4, 50, 512, 5), (6, 90, 64, 2), (1, 12, 512, 5), (7, 700, 512, 5))):
        g = nx.DiGraph()
        sents = sorted(set([0, tlen] + list(rs.randint(1, tlen, size=max(2, tlen // 12)))))
        for node in range(n):
            cnt = rs.randint(1, 4)
            spans = []
            for _ in range(cnt):
                a0 = int(rs.randint(0, tlen - 3))
                spans.append((a0, a0 + int(rs.randint(1, 4))))
            if node == 0 and cnt > 1:
                spans[1] = (spans[0][0], spans[0][1] + 1)                      # overlapping mentions: the later one overwrites
            g.add_node(node, exist_pos=spans, type=[int(rs.randint(0, 7))])
        smax = 0
        for u in range(n):
            for v in range(n):
                if u == v or rs.rand() > 0.5:
                    continue
                k = int(rs.randint(0, 5))                                      # 0: an edge without any sentence
                ss, ps = [], []
                for _ in range(k):
                    i = int(rs.randint(0, len(sents) - 1))
                    s0, s1 = int(sents[i]), int(sents[i + 1])
                    hp = g.nodes[u]["exist_pos"][0]
                    tp = g.nodes[v]["exist_pos"][0]
                    ss.append((s0, s1)), ps.append((hp[0], hp[1], tp[0], tp[1]))
                g.add_edge(u, v, sentences=ss, position=ps)
                smax = max(smax, k)
        g.graph["max_sentence_num"] = max(smax, 1)
        lab = (rs.rand(n, n, 97) < 0.02).astype(np.float32)
        item = {"document": list(rs.randint(1, 200, size=tlen)), "document_pos": list(rs.randint(0, n + 1, size=tlen)),
                "document_ner": list(rs.randint(0, 7, size=tlen)), "graph": g, "label_matrix": lab, "label_mask": None,
                "title": f"doc{di}"}
        me.max_length, me.max_num = maxlen, maxnum
        out = ns["from_list_to_tensor"](me, item)                                                  # the reference's function
        for k, v in out.items():
            if isinstance(v, torch.Tensor):
                refs[f"doc{di}.{k}"] = v.numpy()
        docs.append(pack_document(item))
        cfgs.append((maxlen, maxnum))
    PackedDocs(docs).save(os.path.join(OUT_DIR, "tensorise_docs.npz"))
    refs["cfg"] = np.asarray(cfgs, dtype=np.int64)
    return [("tensorise_ref", refs)]


HEAD_BILI_SLICES = [0, 48, 96]


def head_bilinear_weight(seed: int, r: int = 97, h: int = 128):
    """The bilinear weight used by the head fixtures: U(-1/sqrt(h), 1/sqrt(h)) from torch.Generator().manual_seed(1000 + seed)
    (tests/ regenerate it from the stored seed)."""
    g = torch.Generator().manual_seed(1000 + int(seed))
    return (torch.rand(r, h, h, generator=g) * 2 - 1) / (h ** 0.5)


def head_cases(ref):
    """SURVEY 8 row f3, pinned on the reference's own statements: a real ``GCGCN_glove`` is built (its ``dense_layer``,
    ``bili_layer_01``, ``classification_layer_01``, ``ner_emb``, ``dis_embed`` with the reference's initialisers) and
    lines 306-307 and 344-358 of GCGCN_glove.py are executed verbatim (read as text at generation time, nothing of them
    is written to the repo) on synthetic ``node_feats`` / ``node_type`` / ``node_relative_pos``."""
    import contextlib, io, textwrap
    lines = open(REF_FILE, encoding="utf-8").read().split("\n")
    pre = lines[305:307]
    assert pre[0].strip().startswith("node_relative_pos_h = self.dis_embed(self.config.dis_plus + node_relative_pos)"), pre[0]
    body = lines[343:358]
    assert body[0].strip() == "node_feats = torch.cat(node_feats,1)", body[0]
    assert lines[359].strip() == "return relation_before_softmax_01", lines[359]
    code = compile(textwrap.dedent("\n".join(pre)) + "\n" + textwrap.dedent("\n".join(body)) + "\n", "GCGCN_glove.py:306-307,344-358", "exec")
    keep = ("ner_emb.", "dis_embed.", "dense_layer.", "bili_layer_01.", "classification_layer_01.")
    for n, seed in ((1, 0), (2, 1), (5, 2), (16, 1337)):
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            model = ref.GCGCN_glove(_Cfg(50)).eval()
        # The bilinear weight (97 x 128 x 128 = 6.4 MB) is an INPUT of the statements under test: it is drawn from a
        # documented generator (nn.Bilinear's own bound 1/sqrt(in1_features)) so that the fixture stores a seed, not the tensor.
        with torch.no_grad():
            model.bili_layer_01.weight.copy_(head_bilinear_weight(seed))
        g = torch.Generator().manual_seed(seed + 7)
        feats = [(torch.rand(n, 128, generator=g) * 2 - 1).requires_grad_() for _ in range(3)]     # [nf0, nf0', nf1]: graph_hop + 1
        node_type = torch.randint(0, 7, (n,), generator=g)
        rel = torch.randint(-10, 11, (n, n), generator=g)
        ns = {"torch": torch, "self": model, "node_feats": list(feats), "node_type": node_type, "node_relative_pos": rel, "node_num": n}
        exec(code, ns)                                                              # the model's own statements
        out = ns["relation_before_softmax_01"]
        cot = torch.randn(out.shape, generator=g)
        (out * cot).sum().backward()
        d = {"node_type": _np(node_type), "rel": _np(rel), "out": _np(out), "cot": _np(cot), "meta.bili_seed": np.int64(seed)}
        for i, f in enumerate(feats):
            d[f"in.f{i}"] = _np(f)
            d[f"grad.in.f{i}"] = _np(f.grad)
        for k, v in model.state_dict().items():
            if k.startswith(keep) and k != "bili_layer_01.weight":
                d["sd." + k] = _np(v)
        for k, p in model.named_parameters():
            if k.startswith(keep) and p.grad is not None:
                if k == "bili_layer_01.weight":      # three relation slices + the sum over relations instead of 6.4 MB
                    d["gradpart.bili.r"] = np.array(HEAD_BILI_SLICES)
                    d["gradpart.bili.slices"] = _np(p.grad[HEAD_BILI_SLICES])
                    d["gradpart.bili.sum_r"] = _np(p.grad.sum(0))
                else:
                    d["grad.sd." + k] = _np(p.grad)
        yield f"head_n{n}_s{seed}", d


def trainer_loss_lines():
    """The trainer's own loss statements, read as text from /root/reference/config/Config.py at generation time (the module
    itself is not importable: it needs torch_geometric, and the loss is a loop inside ``train``, not a function):
    line 302 (``BCE = nn.BCELoss(reduction='mean')``) and lines 355-364 (sigmoid, the double loop of BCE calls, the
    division by N^2 - N), dedented, with the literal ``.cuda()`` calls removed (no GPU in the build container).  Returns
    source text to ``exec``; nothing of it is written to the repo."""
    import textwrap
    lines = open(os.path.join(REF_ROOT, "config", "Config.py"), encoding="utf-8").read().split("\n")
    bce = lines[301]
    assert bce.strip() == "BCE = nn.BCELoss(reduction='mean')", bce
    body = lines[354:364]
    assert body[0].strip() == "predict_re = torch.sigmoid(predict_re)", body[0]
    assert body[-1].strip() == "temp_loss = temp_loss/(node_num*node_num-node_num)", body[-1]
    return textwrap.dedent(bce) + "\n" + textwrap.dedent("\n".join(body)).replace(".cuda()", "") + "\n"


def loss_cases():
    """SURVEY 8 row f2, pinned on the reference's own lines: ``config/Config.py:302, 355-364`` are executed verbatim
    (``trainer_loss_lines``) on synthetic ``predict_re`` / ``label_matrix``; loss and autograd gradient become the fixture.
    The oracle's transcription (``gcgcn_oracle.pair_bce_loss_loop``) and its vectorised restatement are checked against
    these files by tests/test_oracle_golden.py."""
    from torch.autograd import Variable
    src = trainer_loss_lines()
    code = compile(src, "Config.py:302,355-364", "exec")
    for n, scale, seed in ((2, 1.0, 0), (5, 3.0, 1), (16, 1.0, 1337), (9, 40.0, 2)):   # scale 40: saturated sigmoids
        g = torch.Generator().manual_seed(seed)
        logits = (torch.randn(n, n, 97, generator=g) * scale).requires_grad_()
        labels = (torch.rand(n, n, 97, generator=g) < 0.04).float()
        ns = {"torch": torch, "nn": nn, "Variable": Variable, "predict_re": logits, "label_matrix": labels}
        exec(code, ns)                                                                  # the trainer's statements
        loss = ns["temp_loss"].squeeze(0)
        grad, = torch.autograd.grad(loss, logits)
        yield f"pair_bce_n{n}_s{int(scale)}", {"logits": _np(logits), "labels": _np(labels), "loss": _np(loss),
                                              "dlogits": _np(grad)}


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    total = 0
    if "--loss-only" in sys.argv:
        cases = list(loss_cases())
    elif "--producer-only" in sys.argv:
        cases = list(producer_cases(load_reference()))
    elif "--head-only" in sys.argv:
        cases = list(head_cases(load_reference()))
    elif "--tensorise-only" in sys.argv:
        cases = tensorise_cases()
    elif "--tail-only" in sys.argv:
        cases = [tail_case(load_reference())]
    elif "--model-only" in sys.argv:
        cases = [model_step_case(load_reference())]
    else:
        ref = load_reference()
        cases = list(block_cases(ref)) + [full_model_case(ref)] + list(loss_cases()) + list(producer_cases(ref)) + list(head_cases(ref)) + tensorise_cases() + [tail_case(ref), model_step_case(ref)]
    for name, pk in cases:
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **pk)
        total += os.path.getsize(path)
        print(f"{name:40s} {os.path.getsize(path) / 1024:9.1f} KiB")
    print(f"total {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
