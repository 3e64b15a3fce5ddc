"""SURVEY 8 row f3 on the GPU: the classifier head kernels (csrc/head.hip) against fixtures produced by executing the
reference's own statements (GCGCN_glove.py:306-307, 344-358) and against the CPU oracle on batched / ragged inputs; the
head's logits feeding the trainer's loss kernel (f2) end to end."""
import pytest
import torch

from conftest import golden_files, ids, load_golden
import gcgcn_amd
from gcgcn_amd import _lib, params as P_
from oracle import gcgcn_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["by_size", "gen1", "gen2"])
def head_generation(request):
    """The outer-product passes exist in two generations chosen by problem size (head.hip head_v1: the second one only from
    32 641 pairs up).  Every parity test runs under the size rule AND with each generation forced, so the kernels the
    benchmarks time (head_bil3_kernel<1..3>, head_dw_kernel, head_bil2_kernel) are the kernels the fixtures check."""
    _lib.call("gcgcn_set_option", b"head_v1", {"by_size": -1, "gen1": 1, "gen2": 0}[request.param])
    yield request.param
    _lib.call("gcgcn_set_option", b"head_v1", -1)


def head_bilinear_weight(seed, r=97, h=128):
    g = torch.Generator().manual_seed(1000 + int(seed))
    return (torch.rand(r, h, h, generator=g) * 2 - 1) / (h ** 0.5)


@pytest.mark.parametrize("path", golden_files("head"), ids=ids(golden_files("head")))
def test_head_golden(gpu_device, path, head_generation):
    g = load_golden(path)
    r = g["raw"]
    sd = dict(g["sd"])
    sd["bili_layer_01.weight"] = head_bilinear_weight(g["meta"]["bili_seed"])
    head = gcgcn_amd.ClassifierHead().to(gpu_device)
    head.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("ner_emb", "dis_embed"))}, strict=True)
    dev = lambda t: t.to(gpu_device)
    feats = [dev(g["in"][f"f{i}"]).requires_grad_() for i in range(3)]
    ner, dis = dev(sd["ner_emb.weight"]).requires_grad_(), dev(sd["dis_embed.weight"]).requires_grad_()
    out = head(feats, dev(torch.from_numpy(r["node_type"])), dev(torch.from_numpy(r["rel"])), ner, dis)
    torch.testing.assert_close(out.cpu(), g["out"], rtol=1e-4, atol=1e-4)
    (out * dev(g["cot"])).sum().backward()
    for i, f in enumerate(feats):
        torch.testing.assert_close(f.grad.cpu(), g["grad_in"][f"f{i}"], rtol=1e-3, atol=1e-4)
    grads = head.named_grads()
    for k, want in g["grad_sd"].items():
        got = ner.grad if k == "ner_emb.weight" else dis.grad if k == "dis_embed.weight" else grads[k]
        torch.testing.assert_close(got.cpu(), want, rtol=1e-3, atol=1e-4 * max(1.0, want.abs().max().item()), msg=lambda m: f"grad {k}: {m}")
    gb = grads["bili_layer_01.weight"].cpu()
    torch.testing.assert_close(gb[torch.from_numpy(r["gradpart.bili.r"])], torch.from_numpy(r["gradpart.bili.slices"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gb.sum(0), torch.from_numpy(r["gradpart.bili.sum_r"]), rtol=1e-3, atol=1e-3)
    assert ner.grad[0].abs().sum().item() == 0           # padding_idx = 0 (glove:241)


@pytest.mark.parametrize("B,N,R,hop", [(3, 9, 97, 2), (2, 64, 97, 2), (2, 13, 5, 1)])
def test_head_batched_ragged_matches_oracle(gpu_device, B, N, R, hop, head_generation):
    g = torch.Generator().manual_seed(B * 10 + N)
    head = gcgcn_amd.ClassifierHead(graph_hop=hop, relation_num=R).to(gpu_device)
    sd = {k: v.cpu().clone().requires_grad_() for k, v in head.state_dict().items()}
    sd["ner_emb.weight"] = (torch.randn(7, 20, generator=g) * 0.3).requires_grad_()
    sd["dis_embed.weight"] = (torch.randn(21, 20, generator=g) * 0.3).requires_grad_()
    feats = [torch.rand(B, N, 128, generator=g) * 2 - 1 for _ in range(hop + 1)]
    ntype = torch.randint(0, 7, (B, N), generator=g)
    rel = torch.randint(-10, 11, (B, N, N), generator=g)
    nv = torch.tensor(([N, max(1, N // 2), 3] * B)[:B], dtype=torch.int32)
    cot = torch.randn(B, N, N, R, generator=g)
    dev = lambda t: t.to(gpu_device)
    fg = [dev(f).requires_grad_() for f in feats]
    ner, dis = dev(sd["ner_emb.weight"].detach()).requires_grad_(), dev(sd["dis_embed.weight"].detach()).requires_grad_()
    out = head(fg, dev(ntype), dev(rel), ner, dis, n_valid=dev(nv))
    (out * dev(cot)).sum().backward()
    fr = [f.clone().requires_grad_() for f in feats]
    loss = 0
    for b in range(B):
        n = int(nv[b])
        ref = O.classifier_head([f[b, :n] for f in fr], ntype[b, :n], rel[b, :n, :n], sd)
        torch.testing.assert_close(out[b, :n, :n].detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
        loss = loss + (ref * cot[b, :n, :n]).sum()
    loss.backward()
    if 64 < R <= 97:      # served by the compacted rows (only pairs of existing entities are computed): every other slot is exactly zero
        okm = torch.arange(N)[None, :] < nv[:, None]
        assert float(out.detach().cpu()[~(okm[:, :, None] & okm[:, None, :])].abs().max()) == 0.0 if not bool(okm.all()) else True
    for a, b_ in zip(fg, fr):
        torch.testing.assert_close(a.grad.cpu(), b_.grad, rtol=1e-3, atol=1e-4 * max(1.0, b_.grad.abs().max().item()))
    grads = head.named_grads()
    for k in grads:
        want = sd[k].grad
        torch.testing.assert_close(grads[k].cpu(), want, rtol=1e-3, atol=2e-4 * max(1.0, want.abs().max().item()), msg=lambda m: f"grad {k}: {m}")
    torch.testing.assert_close(ner.grad.cpu(), sd["ner_emb.weight"].grad, rtol=1e-3, atol=1e-4 * max(1.0, sd["ner_emb.weight"].grad.abs().max().item()))
    torch.testing.assert_close(dis.grad.cpu(), sd["dis_embed.weight"].grad, rtol=1e-3, atol=1e-4 * max(1.0, sd["dis_embed.weight"].grad.abs().max().item()))
    # determinism (no atomics on this path)
    flat_grad_1 = head.flat.grad.clone()
    head.zero_grad()
    fg2 = [dev(f).requires_grad_() for f in feats]
    out2 = head(fg2, dev(ntype), dev(rel), ner.detach(), dis.detach(), n_valid=dev(nv))
    (out2 * dev(cot)).sum().backward()
    assert torch.equal(out, out2) and torch.equal(head.flat.grad, flat_grad_1) and all(torch.equal(a.grad, b_.grad) for a, b_ in zip(fg, fg2))


def test_head_generations_agree_at_bench_size(gpu_device):
    """More than 32 640 pairs, not a multiple of 128 (B = 17 ragged documents of up to 45 entities: 34 425 pairs) -- the
    problem size at which the library itself picks the register-generated kernels (head_bil3_kernel<1..3>, head_dw_kernel).
    Every switchable variant of the second generation against the first generation on the GPU (logits and all gradients),
    one document of the batch against the CPU oracle, and the size rule picks what it says it picks."""
    B, N, R = 17, 45, 97
    g = torch.Generator().manual_seed(1745)
    head = gcgcn_amd.ClassifierHead().to(gpu_device)
    feats = [torch.rand(B, N, 128, generator=g) * 2 - 1 for _ in range(3)]
    ntype, rel = torch.randint(0, 7, (B, N), generator=g), torch.randint(-10, 11, (B, N, N), generator=g)
    nv = torch.randint(2, N + 1, (B,), generator=g).to(torch.int32)
    nv[0] = N
    cot = torch.randn(B, N, N, R, generator=g)
    ner0, dis0 = torch.randn(7, 20, generator=g) * 0.3, torch.randn(21, 20, generator=g) * 0.3
    dev = lambda t: t.to(gpu_device)

    def run():
        head.zero_grad()
        fg = [dev(f).requires_grad_() for f in feats]
        ner, dis = dev(ner0).requires_grad_(), dev(dis0).requires_grad_()
        out = head(fg, dev(ntype), dev(rel), ner, dis, n_valid=dev(nv))
        (out * dev(cot)).sum().backward()
        return [out.detach()] + [f.grad for f in fg] + [ner.grad, dis.grad, head.flat.grad.clone()]

    # (head_compact = 0: every pair slot of the padded batch is computed, the path of a batch without n_valid; the compacted
    # path -- only the pairs that exist, the default for a ragged batch -- is the last two variants)
    variants = {"gen1": dict(head_v1=1), "gen2": dict(head_v1=0), "gen2 bil2 forward": dict(head_v1=0, head_bil3=0),
                "gen2 bil2 backward": dict(head_v1=0, head_bil3_bwd=0), "gen2 gemm dW": dict(head_v1=0, head_dw3=0),
                "by size": dict(), "compact": dict(head_compact=1), "compact bil2": dict(head_compact=1, head_bil3=0, head_bil3_bwd=0)}
    dflt = {"head_v1": -1, "head_bil3": 1, "head_bil3_bwd": 1, "head_dw3": 1, "head_compact": 0}
    res = {}
    try:
        for name, opts in variants.items():
            for k, d in dflt.items():
                _lib.call("gcgcn_set_option", k.encode(), opts.get(k, d))
            res[name] = run()
    finally:
        for k, d in dflt.items():
            _lib.call("gcgcn_set_option", k.encode(), 1 if k == "head_compact" else d)
    names = ["logits", "d f0", "d f1", "d f2", "d ner_emb", "d dis_embed", "d flat"]
    ok = torch.arange(N)[None, :] < nv[:, None]
    real = (ok[:, :, None] & ok[:, None, :]).to(gpu_device)                              # pairs of two existing entities
    for name, r in res.items():
        for what, a, b_ in zip(names, r, res["gen1"]):
            if what == "logits" and name.startswith("compact"):
                assert float(a[~real].abs().max()) == 0.0, f"{name}: logits of pairs with a padding entity must be exactly zero"
                a, b_ = a[real], b_[real]
            torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-4 * max(1.0, b_.abs().max().item()), msg=lambda m: f"{name}, {what}: {m}")
    assert all(torch.equal(a, b_) for a, b_ in zip(res["by size"], res["gen2"]))      # 34 425 pairs: the size rule = generation 2
    assert all(torch.equal(a, b_) for a, b_ in zip(run(), res["compact"]))            # the default for a ragged batch = compacted rows
    # one ragged document of the batch against the CPU oracle (logits + feature gradients; the sums over all documents are
    # tied to generation 1 above, which test_head_batched_ragged_matches_oracle ties to the oracle)
    b = 1
    n = int(nv[b])
    sd = {k: v.cpu() for k, v in head.state_dict().items()}
    sd["ner_emb.weight"], sd["dis_embed.weight"] = ner0, dis0
    fr = [f[b, :n].clone().requires_grad_() for f in feats]
    ref = O.classifier_head(fr, ntype[b, :n], rel[b, :n, :n], sd)
    (ref * cot[b, :n, :n]).sum().backward()
    torch.testing.assert_close(res["gen2"][0][b, :n, :n].cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    for i in range(3):
        torch.testing.assert_close(res["gen2"][1 + i][b, :n].cpu(), fr[i].grad, rtol=1e-3, atol=1e-4 * max(1.0, fr[i].grad.abs().max().item()))


@pytest.mark.parametrize("B,N,full", [(20, 45, True),      # 38 k real pairs: above the one-tile-per-compute-unit threshold -> plain 128-pair tiles
                                      (20, 45, False),     # ~13 k real pairs: every tile split four ways over the k sequence + combine
                                      (3, 20, False)])     # a slab smaller than the threshold (capacity 1 200 pairs)
def test_head_compacted_rows_at_both_launch_shapes(gpu_device, B, N, full):
    """The compacted path picks its launch shape from a count that lives on the device (both shapes are launched, one leaves):
    logits and every gradient against the path that computes every pair slot, on both sides of the threshold."""
    R = 97
    g = torch.Generator().manual_seed(B * 100 + N + int(full))
    head = gcgcn_amd.ClassifierHead().to(gpu_device)
    feats = [torch.rand(B, N, 128, generator=g) * 2 - 1 for _ in range(3)]
    ntype, rel = torch.randint(0, 7, (B, N), generator=g), torch.randint(-10, 11, (B, N, N), generator=g)
    nv = torch.full((B,), N, dtype=torch.int32) if full else torch.randint(2, N + 1, (B,), generator=g).to(torch.int32)
    nv[-1] = max(2, N // 3)
    cot = torch.randn(B, N, N, R, generator=g)
    ner0, dis0 = torch.randn(7, 20, generator=g) * 0.3, torch.randn(21, 20, generator=g) * 0.3
    dev = lambda t: t.to(gpu_device)
    real = int((nv.long() ** 2).sum())
    assert (real >= 32768) == full

    def run():
        head.zero_grad()
        fg = [dev(f).requires_grad_() for f in feats]
        ner, dis = dev(ner0).requires_grad_(), dev(dis0).requires_grad_()
        out = head(fg, dev(ntype), dev(rel), ner, dis, n_valid=dev(nv))
        (out * dev(cot)).sum().backward()
        return [out.detach()] + [f.grad for f in fg] + [ner.grad, dis.grad, head.flat.grad.clone()]
    try:
        _lib.call("gcgcn_set_option", b"head_compact", 0)
        _lib.call("gcgcn_set_option", b"head_v1", 0)
        dense = run()
        _lib.call("gcgcn_set_option", b"head_compact", 1)
        comp, comp2 = run(), run()
    finally:
        _lib.call("gcgcn_set_option", b"head_compact", 1)
        _lib.call("gcgcn_set_option", b"head_v1", -1)
    ok = torch.arange(N)[None, :] < nv[:, None]
    pairs_ok = (ok[:, :, None] & ok[:, None, :]).to(gpu_device)
    assert float(comp[0][~pairs_ok].abs().max()) == 0.0
    names = ["logits", "d f0", "d f1", "d f2", "d ner_emb", "d dis_embed", "d flat"]
    for what, a, b_ in zip(names, comp, dense):
        if what == "logits":
            a, b_ = a[pairs_ok], b_[pairs_ok]
        torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-4 * max(1.0, b_.abs().max().item()), msg=lambda m: f"{what}: {m}")
    assert all(torch.equal(a, b_) for a, b_ in zip(comp, comp2))                     # deterministic


def test_head_rejects_what_it_cannot_index(gpu_device):
    """ner_emb must have the 7 rows the kernels index (glove:241: nn.Embedding(7, ...)); ids out of range raise like the
    reference's embedding lookups (IndexError) when checking is on (the default outside graph capture)."""
    head = gcgcn_amd.ClassifierHead().to(gpu_device)
    dev = lambda t: t.to(gpu_device)
    feats = [dev(torch.rand(1, 4, 128)) for _ in range(3)]
    ntype, rel = torch.zeros(1, 4, dtype=torch.int64), torch.zeros(1, 4, 4, dtype=torch.int64)
    with pytest.raises(ValueError, match="7 rows"):
        head(feats, dev(ntype), dev(rel), dev(torch.randn(5, 20)), dev(torch.randn(21, 20)))
    bad = ntype.clone()
    bad[0, 2] = 9
    with pytest.raises(IndexError, match="node_type"):
        head(feats, dev(bad), dev(rel), dev(torch.randn(7, 20)), dev(torch.randn(21, 20)))
    bad = rel.clone()
    bad[0, 1, 3] = 11
    with pytest.raises(IndexError, match="node_relative_pos"):
        head(feats, dev(ntype), dev(bad), dev(torch.randn(7, 20)), dev(torch.randn(21, 20)))


def test_head_feeds_the_trainer_loss(gpu_device):
    """f3 -> f2: logits from the head into the trainer's per-document loss kernel, gradients back into the head, against
    the oracle chain (classifier_head -> pair_bce_loss_loop)."""
    g = torch.Generator().manual_seed(5)
    N = 6
    head = gcgcn_amd.ClassifierHead().to(gpu_device)
    sd = {k: v.cpu().clone().requires_grad_() for k, v in head.state_dict().items()}
    sd["ner_emb.weight"] = (torch.randn(7, 20, generator=g) * 0.3)
    sd["dis_embed.weight"] = (torch.randn(21, 20, generator=g) * 0.3)
    feats = [torch.rand(N, 128, generator=g) * 2 - 1 for _ in range(3)]
    ntype, rel = torch.randint(0, 7, (N,), generator=g), torch.randint(-10, 11, (N, N), generator=g)
    labels = (torch.rand(N, N, 97, generator=g) < 0.05).float()
    dev = lambda t: t.to(gpu_device)
    out = head([dev(f) for f in feats], dev(ntype), dev(rel), dev(sd["ner_emb.weight"]), dev(sd["dis_embed.weight"]))
    loss = gcgcn_amd.pair_bce_loss(out, dev(labels))
    loss.backward()
    ref = O.pair_bce_loss_loop(O.classifier_head(feats, ntype, rel, sd), labels)
    ref.backward()
    torch.testing.assert_close(loss.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    for k, gk in head.named_grads().items():
        torch.testing.assert_close(gk.cpu(), sd[k].grad, rtol=1e-3, atol=1e-6 * max(1.0, sd[k].grad.abs().max().item() * 1e2))
