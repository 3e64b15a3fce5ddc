"""Model-level drop-ins (gcgcn_amd.models) against the REAL reference model and the reference trainer's own step arithmetic:
tests/golden/model_step_c1.npz was produced by ``GCGCN_glove(config)`` run from RAW inputs on two documents, each document's
loss by the trainer's statements (config/Config.py:302, 355-364), ``total_loss / batch_size`` and one ``backward()``
(Config.py:366-372).  Here: same constructor, the reference's checkpoint loaded strict, logits from the ten raw tensors, the
loss kernel, one backward -- every parameter gradient, and ``None`` exactly where the reference leaves ``None``.  Then the
optimiser: FusedAdam (one launch) against torch.optim.Adam, the reference's optimiser (Config.py:300)."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
import gcgcn_amd
from gcgcn_amd import models as M
from gcgcn_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu


class Cfg:
    """Duck-typed config, the attributes GCGCN_glove(config) reads (glove:222-279, 306-339)."""
    entity_type_size, coref_size, max_length, keep_prob, graph_hop = 20, 20, 512, 1.0, 2
    dis_size, dis_num, dis_plus, relation_num, alpha = 20, 21, 10, 97, 1.0

    def __init__(self, vocab):
        self.data_word_vec = np.zeros((vocab, 100), np.float32)


def head_bilinear_weight(seed, r=97, h=128):
    g = torch.Generator().manual_seed(1000 + int(seed))
    return (torch.rand(r, h, h, generator=g) * 2 - 1) / (h ** 0.5)


def _load(dev):
    g = load_golden(golden_files("model_step")[0])
    sd = dict(g["sd"])
    sd["bili_layer_01.weight"] = head_bilinear_weight(g["meta"]["bili_seed"])
    model = M.GCGCN_glove(Cfg(g["meta"]["vocab"])).to(dev).eval()
    model.rnn.train()        # MIOpen's LSTM backward insists on training mode; with keep_prob = 1 the encoder has no dropout either way
    res = model.load_state_dict(sd, strict=True)                       # a reference checkpoint, strict
    assert not res.missing_keys and not res.unexpected_keys
    return g, sd, model


def _doc(r, di, dev):
    p = f"doc{di}."
    t = lambda k: torch.from_numpy(r[p + k]).to(dev)
    return dict(document=t("document").long(), document_ner=t("ner").long(), document_pos=t("pos").long(), adj_matrix=t("adj"),
                sen_matrix=t("sen"), pos_matrix_h=t("pos_h"), pos_matrix_t=t("pos_t"), node_pos=t("node_pos"),
                node_type=t("node_type").long(), node_relative_pos=t("rel").long())


def test_state_dict_is_the_reference_checkpoint(gpu_device):
    g, sd, model = _load(gpu_device)
    ref_keys = [str(k) for k in g["raw"]["names.keys"]]
    mine = model.state_dict()
    assert list(mine.keys()) == ref_keys                               # same keys, the reference's order
    for k in ref_keys:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
        torch.testing.assert_close(mine[k].cpu(), sd[k], rtol=0, atol=0)
    # what Config.train does with the model (Config.py:297-300): .cuda(), parameters() into optim.Adam
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4)
    assert sum(p.numel() for gr in opt.param_groups for p in gr["params"]) >= sum(v.numel() for v in sd.values())


def test_logits_from_raw_inputs_match_the_real_model(gpu_device):
    g, sd, model = _load(gpu_device)
    r = g["raw"]
    docs = [_doc(r, di, gpu_device) for di in range(g["meta"]["docs"])]
    order = ("document", "document_ner", "document_pos", "adj_matrix", "sen_matrix", "pos_matrix_h", "pos_matrix_t", "node_pos",
             "node_type", "node_relative_pos")
    with torch.no_grad():
        for di, d in enumerate(docs):                                  # the reference's call: ten positional tensors, one document
            out = model(*[d[k] for k in order])
            torch.testing.assert_close(out.cpu(), torch.from_numpy(r[f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)
        batch = {k: torch.stack([d[k] for d in docs]) for k in order}   # extension: all documents as one batch
        out = model(**batch)
        for di in range(len(docs)):
            torch.testing.assert_close(out[di].cpu(), torch.from_numpy(r[f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)
        model.skip_dead_hop = True                                     # opt-in: the last hop never reaches the classifier
        out2 = model(**batch)
        model.skip_dead_hop = False
        assert torch.equal(out, out2)


@pytest.mark.parametrize("batched", [False, True])
def test_training_step_gradients_match_the_reference_trainer(gpu_device, batched):
    """Config.py:339-372 with batch_size = 2: per-document loss, total_loss / batch_size, one backward."""
    g, sd, model = _load(gpu_device)
    r = g["raw"]
    nd = g["meta"]["docs"]
    docs = [_doc(r, di, gpu_device) for di in range(nd)]
    labels = [torch.from_numpy(r[f"doc{di}.labels"]).float().to(gpu_device) for di in range(nd)]
    model.zero_grad()
    if batched:
        batch = {k: torch.stack([d[k] for d in docs]) for k in docs[0]}
        losses = gcgcn_amd.pair_bce_loss(model(**batch), torch.stack(labels))
        total = losses.sum() / nd
    else:                                                              # the trainer's own pattern: one model call per document
        losses = torch.stack([gcgcn_amd.pair_bce_loss(model(**d), lb) for d, lb in zip(docs, labels)])
        total = losses.sum() / nd
    total.backward()
    for di in range(nd):
        torch.testing.assert_close(losses[di].cpu(), torch.from_numpy(r[f"doc{di}.loss"]), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(total.cpu(), torch.from_numpy(r["total_loss"]), rtol=1e-4, atol=1e-6)
    # gradients under the reference's names
    got = {}
    for name, p in model.named_parameters():
        if "." in name and name.rsplit(".", 1)[1] in ("flat", "flat_k"):
            continue
        got[name] = p.grad
    for i, pr in enumerate(model.producers):
        for k, v in pr.named_grads().items():
            head, rest = k.split(".", 1)
            got[f"{head}.{i}.{rest}"] = v
    for k, v in model.get_weighted_adj_matrix.named_grads().items():
        got["get_weighted_adj_matrix." + k] = v
    for i, m in enumerate(model.get_adj_matrix):
        for k, v in m.named_grads().items():
            got[f"get_adj_matrix.{i}.{k}"] = v
    for i, m in enumerate(model.graphcnn):
        for k, v in m.named_grads().items():
            got[f"graphcnn.{i}.{k}"] = v
    for k, v in model.head.named_grads().items():
        got[k] = v
    want_none = set(str(k) for k in r["names.grad_none"])
    for k, want in g["grad_sd"].items():
        assert got.get(k) is not None, f"no gradient for {k}"
        # relative to each tensor's own scale (the producers' gradients are ~1e-7 on these toy documents): 2e-3 of its largest entry, + 1e-10 for the
        # gradients that are rounding residue of an exact zero (the attention biases: a softmax ignores shifts)
        torch.testing.assert_close(got[k].cpu(), want, rtol=2e-3, atol=2e-3 * want.abs().max().item() + 1e-10, msg=lambda m: f"grad {k}: {m}")
    gb = got["bili_layer_01.weight"].cpu()
    torch.testing.assert_close(gb[torch.from_numpy(r["gradpart.bili.r"])], torch.from_numpy(r["gradpart.bili.slices"]), rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(gb.sum(0), torch.from_numpy(r["gradpart.bili.sum_r"]), rtol=2e-3, atol=1e-5)
    for k in want_none:                                                # the dead last hop, linears_k
        assert got.get(k) is None, f"{k} has a gradient; the reference leaves None"
    assert len(want_none) > 0


def test_fused_adam_is_torch_adam(gpu_device):
    """gcgcn_adam_step against torch.optim.Adam (the reference's optimiser, Config.py:300) over 5 steps: odd sizes (tails,
    unaligned views), a parameter that gets no gradient in some steps (its own step count), checkpoint round trip."""
    gen = torch.Generator().manual_seed(0)
    shapes = [(1000, 100), (7,), (1, 1), (513, 3), (4096,), (33, 31)]
    base = [torch.randn(*s, generator=gen) for s in shapes]
    big = torch.randn(2001, generator=gen)
    pa = [b.clone().to(gpu_device).requires_grad_() for b in base] + [big.clone().to(gpu_device)[1:].requires_grad_()]   # unaligned
    pb = [b.clone().to(gpu_device).requires_grad_() for b in base] + [big.clone().to(gpu_device)[1:].requires_grad_()]
    oa, ob = FusedAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2)
    for step in range(5):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and step in (1, 2):
                a.grad = b.grad = None                                # skipped: keeps its own step count
                continue
            gr = torch.randn(a.shape, generator=gen).to(gpu_device) * (10.0 ** (i - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        if step == 2:                                                  # checkpoint round trip through torch's Adam state format
            oa2 = FusedAdam(pa, lr=1e-2)
            oa2.load_state_dict(oa.state_dict())
            oa = oa2
    for a, b in zip(pa, pb):              # (a few ulp of O(1) parameters: the two evaluate sqrt / divide / fused multiply-adds differently)
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-6)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sb:
        assert int(sa[k]["step"]) == int(sb[k]["step"])
        torch.testing.assert_close(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_fused_adam_ring_wraps_and_checkpoints_move_both_ways(gpu_device):
    """Round-3 ADVICE: step() must not drain the stream, and a checkpoint of torch's Adam that uses a feature FusedAdam lacks
    must raise instead of training on with other arithmetic.  Eight steps back to back (the staging ring of three pinned
    buffers wraps twice) against torch.optim.Adam; FusedAdam's state_dict loads into torch's Adam and continues identically."""
    gen = torch.Generator().manual_seed(1)
    base = [torch.randn(300, 7, generator=gen), torch.randn(129, generator=gen)]
    pa = [b.clone().to(gpu_device).requires_grad_() for b in base]
    pb = [b.clone().to(gpu_device).requires_grad_() for b in base]
    oa, ob = FusedAdam(pa, lr=3e-3), torch.optim.Adam(pb, lr=3e-3)
    grads = [[torch.randn(b.shape, generator=gen).to(gpu_device) for b in base] for _ in range(8)]
    for gs in grads:                                  # no synchronisation between the steps: the ring hands out fresh staging
        for a, b, g_ in zip(pa, pb, gs):
            a.grad, b.grad = g_.clone(), g_.clone()
        oa.step()
        ob.step()
    assert len(oa._ring) == FusedAdam._RING and oa._ring_next == 8
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-6)
    for key in ("weight_decay", "amsgrad", "maximize"):
        assert key in oa.state_dict()["param_groups"][0]
    oc = torch.optim.Adam(pa, lr=3e-3)                 # FusedAdam -> torch
    oc.load_state_dict(oa.state_dict())
    for a, b, g_ in zip(pa, pb, grads[0]):
        a.grad, b.grad = g_.clone(), g_.clone()
    oc.step()
    ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-6)
    od = torch.optim.Adam(pb, lr=3e-3, weight_decay=0.01)   # torch (with weight decay) -> FusedAdam: loud
    oe = FusedAdam(pa, lr=3e-3)
    oe.load_state_dict(od.state_dict())
    with pytest.raises(RuntimeError, match="weight_decay"):
        oe.step()


def test_model_trains_with_fused_adam(gpu_device):
    """Three optimiser steps on the fixture's documents (eval-mode forward: no dropout noise).  FusedAdam over the model's own
    parameters (one flat tensor per block, the dead hop's without gradient) against torch.optim.Adam fed the SAME gradients
    (model B takes model A's: two independent backward passes differ by an ulp in PyTorch's own embedding / LSTM backward,
    which Adam's division by sqrt(v) + eps turns into a fraction of lr wherever a gradient is ~1e-8): the reference-named
    tensors move alike, parameters without a gradient do not move, and the loss goes down."""
    g, sd, model_a = _load(gpu_device)
    _, _, model_b = _load(gpu_device)
    r = g["raw"]
    docs = [_doc(r, di, gpu_device) for di in range(g["meta"]["docs"])]
    batch = {k: torch.stack([d[k] for d in docs]) for k in docs[0]}
    labels = torch.stack([torch.from_numpy(r[f"doc{di}.labels"]).float().to(gpu_device) for di in range(len(docs))])
    pa = [p for p in model_a.parameters() if p.requires_grad]
    pb = [p for p in model_b.parameters() if p.requires_grad]
    oa, ob = FusedAdam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    hist = []
    for _ in range(3):
        oa.zero_grad()
        loss = gcgcn_amd.pair_bce_loss(model_a(**batch), labels).sum() / len(docs)
        loss.backward()
        for a_, b_ in zip(pa, pb):
            b_.grad = None if a_.grad is None else a_.grad.clone()
        oa.step()
        ob.step()
        hist.append(loss.item())
    assert hist[-1] < hist[0]
    a, b = model_a.state_dict(), model_b.state_dict()
    moved, still = 0, 0
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")
        same = torch.equal(a[k].cpu(), sd[k])
        moved += int(not same)
        still += int(same)
    want_none = set(str(k) for k in r["names.grad_none"])
    # Biases in front of a softmax (GATAttention's wt.bias, the word attention's attention_all.bias): a softmax ignores a shift, their
    # gradient is the rounding residue of an exact zero (1e-12 .. 1e-14 in the reference's own backward, the fixture's grad_sd) and
    # whether that residue is +-1e-12 or exactly 0.0 depends on the ulps of the step before (PyTorch's LSTM / embedding backward use
    # atomics).  Adam moves the parameter by ~lr in the first case and not at all in the second: both are right, neither is asserted.
    residue = set(k for k, v in g["grad_sd"].items() if float(v.abs().max()) < 1e-9)
    unmoved = sorted(k for k in a if torch.equal(a[k].cpu(), sd[k]) and k not in want_none and k not in residue)
    assert not unmoved, f"parameters with a gradient that did not move: {unmoved}"
    assert still >= len(want_none)                                               # the dead hop and linears_k stay where they were


class _StubBert(torch.nn.Module):
    """Stands in for pytorch_pretrained_bert.BertModel (third-party, absent here): same call contract as bert:275 --
    ``bert(document, output_all_encoded_layers=False) -> (states[B,T,768], pooled[B,768])``."""

    def __init__(self, vocab):
        super().__init__()
        self.emb = torch.nn.Embedding(vocab, 768)
        self.mix = torch.nn.Linear(768, 768)

    def forward(self, document, output_all_encoded_layers=True):
        assert output_all_encoded_layers is False
        h = torch.tanh(self.mix(self.emb(document)))
        return h, h[:, 0]


def _bert_oracle(sd, d, L=4, H=4):
    """bert:275-346 restated on the CPU with the oracle's blocks (four sub-layers, four heads: bert:247-248)."""
    from oracle import gcgcn_oracle as O
    lin = lambda x, k: x @ sd[k + ".weight"].t() + sd[k + ".bias"]
    states = torch.tanh(lin(sd["bert.emb.weight"][d["document"]], "bert.mix"))               # the stub encoder
    cls_feat = states[0]
    doc = torch.cat([states, sd["entity_embed.weight"][d["document_pos"]], sd["ner_emb.weight"][d["document_ner"]]], dim=-1)
    ctx = torch.tanh(lin(doc, "linear_re"))                                                  # bert:283
    x = d["node_pos"] @ ctx
    feats = [x]
    for i in range(2):
        e = O.edge_features_folded(ctx, d["sen_matrix"], d["pos_matrix_h"].long(), d["pos_matrix_t"].long(), x, sd["dis_embed.weight"], sd, i)
        if i == 0:
            a = O.gat_attention(x, e, O.sub(sd, "get_weighted_adj_matrix"), torch.eq(d["adj_matrix"], 0))
            new = O.graph_convolution(x, e, a, O.sub(sd, "graphcnn.0"), L)
        else:
            al = O.multi_head_attention(x, O.sub(sd, "get_adj_matrix.0"), H)
            new = O.multi_graph_convolution(x, e, al, O.sub(sd, "graphcnn.1"), L, H)
        feats.append(x)
        x = new
    return O.classifier_head(feats, d["node_type"], d["node_relative_pos"], sd) + lin(cls_feat, "linear_cls")   # bert:345-346


def test_bert_variant_wiring_with_a_stub_encoder(gpu_device):
    """GraphCNN_multihead_bert_gate_cls: constructor (bert:219-271: four sub-layers, four heads, 768 + 20 + 20 inputs to
    linear_re, linear_cls), the ten-tensor forward and the [CLS] term on every pair, single document and ragged batch, against
    the oracle's blocks chained as bert:275-346 chains them.  The encoder itself is third-party: a stub with BertModel's call
    contract stands in (the reference class cannot be imported without pytorch_pretrained_bert)."""
    g = load_golden(golden_files("model_step")[0])
    r = g["raw"]
    vocab = g["meta"]["vocab"]
    torch.manual_seed(5)
    model = M.GraphCNN_multihead_bert_gate_cls(Cfg(vocab), bert=_StubBert(vocab)).to(gpu_device).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():                         # lift the near-zero default scores off the relu's edge
            if n.endswith("attention_all.bias"):
                p.fill_(0.4)
    keys = list(model.state_dict().keys())
    assert keys[0].startswith("bert.") and "linear_cls.weight" in keys and "graphcnn.1.graphconv.15.weights_node" in keys
    assert not any(k.startswith("rnn.") for k in keys)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    docs = [_doc(r, di, gpu_device) for di in range(g["meta"]["docs"])]
    order = ("document", "document_ner", "document_pos", "adj_matrix", "sen_matrix", "pos_matrix_h", "pos_matrix_t", "node_pos",
             "node_type", "node_relative_pos")
    refs = [_bert_oracle(sd, {k: v.cpu() for k, v in d.items()}) for d in docs]
    with torch.no_grad():
        for d, ref in zip(docs, refs):
            out = model(*[d[k] for k in order])
            torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=2e-4)
        batch = {k: torch.stack([d[k] for d in docs]) for k in order}
        out = model(**batch)
        for di, ref in enumerate(refs):
            torch.testing.assert_close(out[di].cpu(), ref, rtol=1e-4, atol=2e-4)
    # one training step through the trainer's loss: every parameter of the path receives a finite gradient
    model.train()
    lab = torch.from_numpy(r["doc0.labels"]).to(gpu_device).float()
    loss = gcgcn_amd.pair_bce_loss(model(*[docs[0][k] for k in order]), lab).sum()
    loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    assert model.linear_cls.weight.grad is not None and model.graphcnn[0].flat.grad is not None
    assert model.graphcnn[1].flat.grad is None         # the last hop's output never reaches the classifier (bert:333 as glove:338)
