"""The whole post-encoder model on the HIP kernels (gcgcn_amd.GraphModelTail: f1 producer -> CAGGC -> f1 -> MAGGC -> f3 head)
against the REAL GCGCN_glove's logits computed from the same token states (tests/golden/tail_c1.npz), one document per call
(the reference's call shape) and both documents as one batch; then through the trainer's loss (f2) with gradients."""
import pytest
import torch

from conftest import golden_files, load_golden, tail_oracle
import gcgcn_amd
from oracle import gcgcn_oracle as O

pytestmark = pytest.mark.gpu


def head_bilinear_weight(seed, r=97, h=128):
    g = torch.Generator().manual_seed(1000 + int(seed))
    return (torch.rand(r, h, h, generator=g) * 2 - 1) / (h ** 0.5)


def _setup(dev):
    g = load_golden(golden_files("tail")[0])
    sd = dict(g["sd"])
    sd["bili_layer_01.weight"] = head_bilinear_weight(g["meta"]["bili_seed"])
    tail = gcgcn_amd.GraphModelTail().to(dev).eval()
    own = {k: v for k, v in sd.items() if not k.startswith(("dis_embed.", "ner_emb."))}
    res = tail.load_state_dict(own, strict=True)                                       # the model's own key names
    assert not res.missing_keys and not res.unexpected_keys
    assert set(tail.state_dict().keys()) == set(own.keys())
    return g, sd, tail


def _doc(r, di, dev):
    p = f"doc{di}."
    t = lambda k: torch.from_numpy(r[p + k]).to(dev)
    ctx = t("ctx")
    return dict(context_output=ctx, node_feat=t("node_pos") @ ctx, adj_matrix=t("adj"), sen_matrix=t("sen"), pos_matrix_h=t("pos_h"),
                pos_matrix_t=t("pos_t"), node_type=t("node_type"), node_relative_pos=t("rel"))


def test_tail_matches_the_real_model(gpu_device):
    g, sd, tail = _setup(gpu_device)
    dis, ner = sd["dis_embed.weight"].to(gpu_device), sd["ner_emb.weight"].to(gpu_device)
    docs = [_doc(g["raw"], di, gpu_device) for di in range(g["meta"]["docs"])]
    with torch.no_grad():
        for di, d in enumerate(docs):
            out = tail(dis_embed_weight=dis, ner_emb_weight=ner, **d)
            torch.testing.assert_close(out.cpu(), torch.from_numpy(g["raw"][f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)
        batch = {k: torch.stack([d[k] for d in docs]) for k in docs[0]}
        out = tail(dis_embed_weight=dis, ner_emb_weight=ner, **batch)
        for di in range(len(docs)):
            torch.testing.assert_close(out[di].cpu(), torch.from_numpy(g["raw"][f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)


def test_tail_trains_through_the_loss(gpu_device):
    """logits -> trainer loss (f2) -> backward through head, blocks and producers: parameter gradients against autograd
    through the oracle chain (the reference's dead last hop included: its parameters get no gradient)."""
    g, sd, tail = _setup(gpu_device)
    r = g["raw"]
    dis, ner = sd["dis_embed.weight"].to(gpu_device).requires_grad_(), sd["ner_emb.weight"].to(gpu_device).requires_grad_()
    d = _doc(r, 0, gpu_device)
    n = d["node_feat"].shape[0]
    labels = (torch.rand(n, n, 97, generator=torch.Generator().manual_seed(1)) < 0.05).float()
    out = tail(dis_embed_weight=dis, ner_emb_weight=ner, **d)
    loss = gcgcn_amd.pair_bce_loss(out, labels.to(gpu_device))
    loss.backward()
    sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = O.pair_bce_loss_loop(tail_oracle(r, sdl, 0), labels)
    ref.backward()
    torch.testing.assert_close(loss.cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
    got = {}
    for i, pr in enumerate(tail.producers):
        for k, v in pr.named_grads().items():
            head, rest = k.split(".", 1)
            got[f"{head}.{i}.{rest}"] = v
    for k, v in tail.get_weighted_adj_matrix.named_grads().items():
        got["get_weighted_adj_matrix." + k] = v
    for k, v in tail.graphcnn[0].named_grads().items():
        got["graphcnn.0." + k] = v
    for k, v in tail.head.named_grads().items():
        got[k] = v
    got["dis_embed.weight"], got["ner_emb.weight"] = dis.grad, ner.grad
    checked = 0
    for k, v in sdl.items():
        if v.grad is None:
            continue
        assert k in got, f"no gradient for {k}"
        scale = max(1e-6, v.grad.abs().max().item())
        torch.testing.assert_close(got[k].cpu(), v.grad, rtol=2e-3, atol=2e-4 * scale, msg=lambda m: f"grad {k}: {m}")
        checked += 1
    assert checked > 30
    # the last hop is dead in the reference (node_feats records pre-update features): no gradient reaches it
    dead = [k for k, v in sdl.items() if v.grad is None and not k.startswith(("get_adj_matrix.0.linears_k",))]
    assert any(k.startswith("graphcnn.1.") for k in dead) and any(k.startswith("word_attention.1.") for k in dead)
    assert tail.graphcnn[1].flat.grad is None and tail.producers[1].flat.grad is None
