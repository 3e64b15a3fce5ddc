"""The CPU oracle (oracle/gcgcn_oracle.py) against golden vectors produced by the reference's
own classes (oracle/make_golden.py).  This is what pins the oracle; tolerance 1e-5."""
import pytest
import torch

from conftest import golden_files, ids, load_golden, tail_oracle
from oracle import gcgcn_oracle as O

TOL = dict(rtol=1e-5, atol=1e-5)


def _leaf(t):
    return t.clone().requires_grad_()


def _sd_leaf(sd):
    return {k: _leaf(v) for k, v in sd.items()}


def _check_grads(g, ins, sd):
    for k, ref in g["grad_in"].items():
        assert ins[k].grad is not None, k
        torch.testing.assert_close(ins[k].grad, ref, **TOL, msg=lambda m: f"grad in.{k}: {m}")
    for k, ref in g["grad_sd"].items():
        assert sd[k].grad is not None, k
        torch.testing.assert_close(sd[k].grad, ref, **TOL, msg=lambda m: f"grad sd.{k}: {m}")
    # parameters the reference leaves without a gradient must stay untouched here too
    for k, v in sd.items():
        if k not in g["grad_sd"]:
            assert v.grad is None, f"{k} has a gradient but the reference's has none"


@pytest.mark.parametrize("path", golden_files("graphconv"), ids=ids(golden_files("graphconv")))
def test_graphconv(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items()}
    sd = _sd_leaf(g["sd"])
    out = O.graph_conv(ins["x"], ins["e"], ins["adj"], sd["weights_edge"], sd["weights_node"])
    torch.testing.assert_close(out, g["out"], **TOL)
    out.backward(g["cot"])
    _check_grads(g, ins, sd)


@pytest.mark.parametrize("path", golden_files("gat"), ids=ids(golden_files("gat")))
def test_gat(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items() if k != "mask"}
    sd = _sd_leaf(g["sd"])
    out = O.gat_attention(ins["x"], ins["e"], sd, mask=g["in"]["mask"])
    torch.testing.assert_close(out, g["out"], **TOL)
    out.backward(g["cot"])
    _check_grads(g, ins, sd)


@pytest.mark.parametrize("path", golden_files("caggc"), ids=ids(golden_files("caggc")))
def test_caggc_conv(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items()}
    sd = _sd_leaf(g["sd"])
    out = O.graph_convolution(ins["x"], ins["e"], ins["adj"], sd, g["meta"]["l"])
    torch.testing.assert_close(out, g["out"], **TOL)
    out.backward(g["cot"])
    _check_grads(g, ins, sd)


@pytest.mark.parametrize("path", golden_files("mha"), ids=ids(golden_files("mha")))
def test_mha(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items()}
    sd = _sd_leaf(g["sd"])
    outs = torch.stack(O.multi_head_attention(ins["x"], sd, g["meta"]["h"]))
    torch.testing.assert_close(outs, g["out"], **TOL)
    outs.backward(g["cot"])
    _check_grads(g, ins, sd)          # also asserts linears_k.* got no gradient


@pytest.mark.parametrize("path", golden_files("maggc"), ids=ids(golden_files("maggc")))
def test_maggc_conv(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items()}
    sd = _sd_leaf(g["sd"])
    out = O.multi_graph_convolution(ins["x"], ins["e"], list(ins["adj"].unbind(0)), sd,
                                    g["meta"]["l"], g["meta"]["h"])
    torch.testing.assert_close(out, g["out"], **TOL)
    out.backward(g["cot"])
    _check_grads(g, ins, sd)


@pytest.mark.parametrize("path", golden_files("stack"), ids=ids(golden_files("stack")))
def test_stack(path):
    g = load_golden(path)
    ins = {k: _leaf(v) for k, v in g["in"].items() if k != "adj"}
    sd = _sd_leaf(g["sd"])
    feats = O.hop_stack(ins["x"], [ins["e1"], ins["e2"]], g["in"]["adj"], sd,
                        g["meta"]["l"], g["meta"]["h"])
    torch.testing.assert_close(feats[1], g["mid"]["x1"], **TOL)
    torch.testing.assert_close(feats[2], g["out"], **TOL)
    feats[2].backward(g["cot"])
    _check_grads(g, ins, sd)


def test_model_c1_hooks():
    """cfg 1: what the four hot-path modules saw/returned inside the real GCGCN_glove forward."""
    g = load_golden(golden_files("model")[0])
    raw, sd = g["raw"], g["sd"]
    L, H = g["meta"]["l"], g["meta"]["h"]
    for di in range(g["meta"]["docs"]):
        p = f"doc{di}."
        t = lambda k: torch.from_numpy(raw[p + k])
        feats = O.hop_stack(t("x0"), [t("e1"), t("e2")], t("adj"), sd, L, H)
        a0 = O.gat_attention(t("x0"), t("e1"), O.sub(sd, "get_weighted_adj_matrix"))
        torch.testing.assert_close(a0, t("a0"), **TOL)
        torch.testing.assert_close(feats[1], t("x1_new"), **TOL)
        torch.testing.assert_close(feats[1], t("x1"), **TOL)
        al = torch.stack(O.multi_head_attention(t("x1"), O.sub(sd, "get_adj_matrix.0"), H))
        torch.testing.assert_close(al, t("al"), **TOL)
        torch.testing.assert_close(feats[2], t("x2_new"), **TOL)


def test_dropout_mask_convention():
    x = torch.ones(4, 4)
    keep = torch.tensor([[1, 0, 1, 0]] * 4, dtype=torch.bool)
    y = O._drop(x, keep, 0.2)
    assert torch.allclose(y[:, 0], torch.full((4,), 1.25)) and (y[:, 1] == 0).all()
    assert O._drop(x, None, 0.2) is x


@pytest.mark.parametrize("path", golden_files("pair_bce"), ids=ids(golden_files("pair_bce")))
def test_pair_bce_loss(path):
    """the oracle's transcription of the trainer's loop and its vectorised restatement (value and
    ATen-faithful gradient) against fixtures produced by EXECUTING the reference's own statements, config/Config.py:302 and
    :355-364 (oracle/make_golden.py::loss_cases), including saturated sigmoids."""
    r = load_golden(path)["raw"]
    logits, labels = torch.from_numpy(r["logits"]), torch.from_numpy(r["labels"])
    torch.testing.assert_close(O.pair_bce_loss(logits, labels), torch.from_numpy(r["loss"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(O.pair_bce_loss_grad(logits, labels), torch.from_numpy(r["dlogits"]), rtol=1e-5, atol=1e-9)
    x = logits.clone().requires_grad_()
    loop = O.pair_bce_loss_loop(x, labels)
    gl, = torch.autograd.grad(loop, x)
    torch.testing.assert_close(loop.detach(), torch.from_numpy(r["loss"]), rtol=0, atol=0)      # same ops, same order: bitwise
    torch.testing.assert_close(gl, torch.from_numpy(r["dlogits"]), rtol=0, atol=0)


def test_pair_bce_loss_loop_and_ragged():
    """The loop itself on a fresh case, and n_valid = the same loss on the leading sub-block."""
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(6, 6, 11, generator=g) * 2).requires_grad_()
    y = (torch.rand(6, 6, 11, generator=g) < 0.2).float()
    ref = O.pair_bce_loss_loop(x, y)
    gref, = torch.autograd.grad(ref, x)
    torch.testing.assert_close(O.pair_bce_loss(x.detach(), y), ref.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(O.pair_bce_loss_grad(x.detach(), y), gref, rtol=1e-5, atol=1e-9)
    sub = O.pair_bce_loss_loop(x[:4, :4].detach(), y[:4, :4])
    torch.testing.assert_close(O.pair_bce_loss(x.detach(), y, n_valid=4), sub, rtol=1e-6, atol=1e-6)
    gr = O.pair_bce_loss_grad(x.detach(), y, n_valid=4)
    assert gr[4:].abs().sum() == 0 and gr[:, 4:].abs().sum() == 0


@pytest.mark.parametrize("path", golden_files("producer"), ids=ids(golden_files("producer")))
def test_edge_feature_producer(path):
    """SURVEY 8 f1 groundwork: the restatement of WordAttention + SentenceAttention + the two Linear layers (glove:300-330)
    and the FOLDED algorithm a kernel would run ([21, T] score table, no [N,N,S,T,Hd] tensor) against the reference's own
    classes, forward and gradients -- including the pair whose divisor is 1e-10 (no padded sentence slot, glove:205,212)."""
    g = load_golden(path)
    r = g["raw"]
    sd = g["sd"]
    for fn in (O.edge_features, O.edge_features_folded):
        ctx = torch.from_numpy(r["ctx"]).requires_grad_()
        node = torch.from_numpy(r["node"]).requires_grad_()
        table = torch.from_numpy(r["table"]).requires_grad_()
        sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
        e = fn(ctx, torch.from_numpy(r["sen"]), torch.from_numpy(r["pos_h"]), torch.from_numpy(r["pos_t"]), node, table, sdl, 0)
        ref = g["out"]
        assert ((~torch.from_numpy(r["sen"])[:, :, :, 0]).sum(2) == 0).any()   # a pair without any padded slot is in the fixture
        torch.testing.assert_close(e, ref, rtol=2e-5, atol=1e-6)
        (e * g["cot"]).sum().backward()
        torch.testing.assert_close(ctx.grad, torch.from_numpy(r["grad.ctx"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(node.grad, torch.from_numpy(r["grad.node"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(table.grad, torch.from_numpy(r["grad.table"]), rtol=1e-4, atol=1e-6)
        for k, ref_g in g["grad_sd"].items():
            torch.testing.assert_close(sdl[k].grad, ref_g, rtol=1e-4, atol=1e-6, msg=lambda m: f"grad {k}: {m}")


def head_bilinear_weight(seed, r=97, h=128):
    """Same generator as oracle/make_golden.py::head_bilinear_weight (the fixtures store the seed, not the 6.4 MB tensor)."""
    g = torch.Generator().manual_seed(1000 + int(seed))
    return (torch.rand(r, h, h, generator=g) * 2 - 1) / (h ** 0.5)


@pytest.mark.parametrize("path", golden_files("head"), ids=ids(golden_files("head")))
def test_classifier_head(path):
    """the restatement of the classifier head against fixtures produced by EXECUTING the reference's own
    statements (GCGCN_glove.py:306-307, 344-358, oracle/make_golden.py::head_cases) on a real GCGCN_glove's layers: logits
    and every gradient, incl. ner_emb's padding row (no gradient) and N = 1."""
    g = load_golden(path)
    r = g["raw"]
    sd = {k: v.clone().requires_grad_() for k, v in g["sd"].items()}
    sd["bili_layer_01.weight"] = head_bilinear_weight(g["meta"]["bili_seed"]).requires_grad_()
    feats = [g["in"][f"f{i}"].clone().requires_grad_() for i in range(3)]
    out = O.classifier_head(feats, torch.from_numpy(r["node_type"]), torch.from_numpy(r["rel"]), sd)
    torch.testing.assert_close(out, g["out"], rtol=1e-5, atol=1e-5)
    (out * g["cot"]).sum().backward()
    for i, f in enumerate(feats):
        torch.testing.assert_close(f.grad, g["grad_in"][f"f{i}"], rtol=1e-4, atol=1e-5)
    for k, want in g["grad_sd"].items():
        torch.testing.assert_close(sd[k].grad, want, rtol=1e-4, atol=1e-5 * max(1.0, want.abs().max().item()), msg=lambda m: f"grad {k}: {m}")
    gb = sd["bili_layer_01.weight"].grad
    torch.testing.assert_close(gb[torch.from_numpy(r["gradpart.bili.r"])], torch.from_numpy(r["gradpart.bili.slices"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gb.sum(0), torch.from_numpy(r["gradpart.bili.sum_r"]), rtol=1e-4, atol=1e-4)
    assert sd["ner_emb.weight"].grad[0].abs().sum() == 0


def test_tensorise_document_and_packed_format(tmp_path):
    """the restatement of Config.from_list_to_tensor on packed records against the tensors the reference's own
    function produced (oracle/make_golden.py::tensorise_cases), bit for bit incl. dtypes; the packed file format round-trips."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from gcgcn_amd.data import PackedDocs
    docs = PackedDocs.load(os.path.join(GOLDEN, "tensorise_docs.npz"))
    ref = np.load(os.path.join(GOLDEN, "tensorise_ref.npz"))
    assert len(docs) == 4
    for i, d in enumerate(docs.docs):
        ml, mn = (int(v) for v in ref["cfg"][i])
        out = O.tensorise_document(d, ml, mn)
        for k, v in out.items():
            r = ref[f"doc{i}.{k}"]
            assert r.shape == v.shape and r.dtype == v.dtype and np.array_equal(r, v), (i, k)
    p = str(tmp_path / "again.npz")
    docs.save(p)
    again = PackedDocs.load(p)
    for a, b in zip(docs.docs, again.docs):
        for f in ("tokens", "node_type", "men_ptr", "mentions", "slots", "edges", "labels"):
            assert np.array_equal(getattr(a, f), getattr(b, f))
        assert (a.n_rel, a.max_sentence_num, a.title) == (b.n_rel, b.max_sentence_num, b.title)
    assert [int(v) for v in O.dis2idx_table()[[0, 1, 2, 3, 4, 7, 8, 511, 512, 1023]]] == [0, 1, 2, 2, 3, 3, 4, 9, 10, 10]


def test_post_encoder_model_chain():
    """The oracle functions chained as the model chains them reproduce the REAL GCGCN_glove's logits from its own token
    states (fixture tail_c1: oracle/make_golden.py::tail_case)."""
    g = load_golden(golden_files("tail")[0])
    sd = dict(g["sd"])
    sd["bili_layer_01.weight"] = head_bilinear_weight(g["meta"]["bili_seed"])
    for di in range(g["meta"]["docs"]):
        out = tail_oracle(g["raw"], sd, di)
        torch.testing.assert_close(out, torch.from_numpy(g["raw"][f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)
