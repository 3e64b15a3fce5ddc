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


def _compact_case(dev, B=3, N=11, S=3, T=40, Hd=128, P=20, seed=11):
    """Synthetic batch for the compact-row consumers: ragged entity counts, some pairs with a live slot, one document with none."""
    g = torch.Generator().manual_seed(seed)
    ctx = torch.tanh(torch.randn(B, T, Hd, generator=g))
    node = torch.rand(B, N, Hd, generator=g) * 2 - 1
    table = torch.randn(21, P, generator=g) * 0.5
    sen = torch.zeros(B, N, N, S, T, dtype=torch.bool)
    live = torch.rand(B, N, N, S, generator=g) < 0.25
    live[B - 1] = False                                          # a document without any live slot
    for idx in live.nonzero().tolist():
        b, i, j, s_ = idx
        sen[b, i, j, s_, : int(torch.randint(3, 12, (1,), generator=g))] = True
    other = (torch.rand(B, N, N, S, generator=g) < 0.3) & ~live   # slots that do not start at token 0: padded (glove:305)
    for idx in other.nonzero().tolist():
        b, i, j, s_ = idx
        sen[b, i, j, s_, 5:15] = True
    ph, pt = torch.randint(0, 21, (B, N, N, S, T), generator=g).to(torch.uint8), torch.randint(0, 21, (B, N, N, S, T), generator=g).to(torch.uint8)
    nv = torch.tensor([N, N - 4, N - 1][:B], dtype=torch.int32)
    node = node * (torch.arange(N)[None, :] < nv[:, None]).unsqueeze(-1).float()
    return [t.to(dev) for t in (ctx, node, table, sen, ph, pt, nv)]


def test_compact_rows_equal_the_dense_edge_tensor(gpu_device):
    """EdgeFeatureProducer(compact=True): the handle's dense() is the dense call's E; GATAttention and the edge mean on the
    handle equal the same blocks on the dense tensor, forward and every gradient (node features, token states, distance table,
    producer parameters incl. linear_sentence_att's bias, GATAttention's parameters); capacities that are too small are loud."""
    from gcgcn_amd import functional as F_
    ctx, node, table, sen, ph, pt, nv = _compact_case(gpu_device)
    B, N, Hd = node.shape
    prod = gcgcn_amd.EdgeFeatureProducer(Hd, 20).to(gpu_device)
    gat = gcgcn_amd.GATAttention(Hd, Hd).to(gpu_device).eval()
    cotA = torch.randn(B, N, N, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    cotM = torch.randn(B, N, Hd, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    res = {}
    for compact in (False, True):
        prod.zero_grad(), gat.zero_grad()
        c_, n_, t_ = (t.clone().requires_grad_() for t in (ctx, node, table))
        e = prod(c_, sen, ph, pt, n_, t_, n_valid=nv, compact=compact)
        if compact:
            assert isinstance(e, F_.CompactEdges) and e.shape == (B, N, N, Hd)
            torch.testing.assert_close(e.dense().detach(), res[False]["E"], rtol=1e-5, atol=1e-6)
        a, xa = gat(n_, e, n_valid=nv, return_input_alias=True)
        ebar = F_.take_edge_mean(e, nv)                                  # parked by GATAttention's pass
        assert ebar is not None
        ebar2 = F_.edge_mean(e if compact else e, nv)                    # the mean-only consumer (what a MAGGC hop runs)
        ((a * cotA).sum() + (ebar * cotM).sum() + (ebar2 * cotM).sum() * 0.5 + (xa * 0.1).sum()).backward()
        res[compact] = dict(E=(e.dense() if compact else e).detach(), A=a.detach(), ebar=ebar.detach(), ebar2=ebar2.detach(), dctx=c_.grad,
                            dnode=n_.grad, dtable=t_.grad, dprod=prod.flat.grad.clone(), dgat=gat.flat.grad.clone())
    for k in res[False]:
        want = res[False][k]
        torch.testing.assert_close(res[True][k], want, rtol=1e-4, atol=1e-5 * max(1.0, want.abs().max().item()), msg=lambda m: f"{k}: {m}")
    assert res[True]["dprod"].abs().sum() > 0 and res[True]["dctx"].abs().sum() > 0
    # capacities one short: NaN everywhere a real pair contributes, also through the compact consumers
    r, q = F_.producer_live_counts(sen.view(torch.uint8), nv)
    e_bad = prod(ctx, sen, ph, pt, node, table, n_valid=nv, max_live_slots=r, max_live_pairs=q - 1, compact=True)
    assert prod.last_counts.tolist()[2] == 1
    a_bad = gat(node, e_bad, n_valid=nv)
    assert torch.isnan(a_bad[0, 0]).all() and torch.isnan(F_.edge_mean(e_bad, nv)[0, 0]).all()


def test_tail_with_compact_edges(gpu_device):
    """GraphModelTail(compact_edges=True): E and dE of both hops are never written; logits and every gradient equal the default
    (dense) run, and the real model's logits of the fixture."""
    g, sd, tail = _setup(gpu_device)
    r = g["raw"]
    docs = [_doc(r, di, gpu_device) for di in range(g["meta"]["docs"])]
    batch = {k: torch.stack([d[k] for d in docs]) for k in docs[0]}
    labels = (torch.rand(len(docs), batch["node_feat"].shape[1], batch["node_feat"].shape[1], 97, generator=torch.Generator().manual_seed(3)) < 0.05).float().to(gpu_device)
    res = {}
    for compact in (False, True):
        tail.compact_edges = compact
        tail.zero_grad()
        dis, ner = sd["dis_embed.weight"].to(gpu_device).requires_grad_(), sd["ner_emb.weight"].to(gpu_device).requires_grad_()
        b = {k: (v.clone().requires_grad_() if k in ("context_output", "node_feat") else v) for k, v in batch.items()}
        out = tail(dis_embed_weight=dis, ner_emb_weight=ner, **b)
        gcgcn_amd.pair_bce_loss(out, labels).sum().backward()
        res[compact] = [out.detach(), b["context_output"].grad, b["node_feat"].grad, dis.grad, ner.grad] + \
                       [p.grad.clone() for p in tail.parameters() if p.grad is not None]
    tail.compact_edges = False
    assert len(res[True]) == len(res[False])
    for a, b_ in zip(res[True], res[False]):
        torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-6 * max(1.0, b_.abs().max().item() * 10))
    for di in range(len(docs)):
        torch.testing.assert_close(res[True][0][di].cpu(), torch.from_numpy(r[f"doc{di}.logits"]), rtol=1e-4, atol=1e-4)


def test_tail_graph_replay_matches_eager(gpu_device):
    """tools/tail_bench.py --mode graph times replays of ONE hipGraph holding the whole post-encoder step (producer -> CAGGC ->
    producer -> MAGGC -> head -> loss, forward + backward).  Under capture nothing may be read back to the host (the live-slot /
    live-pair / pair-row counts stay on the device) and every replay has to rewrite every result.  Ragged batch, eval mode: the
    replayed logits and loss are BITWISE those of an eager step on the same inputs, on the first and on the third replay; the
    gradients agree to the summation order of the producers' documented fp32 atomics (and bitwise where no atomic is involved:
    the head's and the graph blocks' parameters behind the loss)."""
    ctx, node, table, sen, ph, pt, nv = _compact_case(gpu_device, B=3, N=13, S=3, T=48, seed=23)
    nv = torch.tensor([13, 5, 12], dtype=torch.int32, device=gpu_device)
    node = node * (torch.arange(13, device=gpu_device)[None, :] < nv[:, None]).unsqueeze(-1).float()
    B, N, _ = node.shape
    g = torch.Generator().manual_seed(5)
    ner = (torch.randn(7, 20, generator=g) * 0.3).to(gpu_device).requires_grad_()
    ntype = torch.randint(0, 7, (B, N), generator=g).to(gpu_device)
    rel = torch.randint(-10, 11, (B, N, N), generator=g).to(gpu_device)
    labels = (torch.rand(B, N, N, 97, generator=g) < 0.05).float().to(gpu_device)
    tail = gcgcn_amd.GraphModelTail().to(gpu_device).eval()
    leaves = [ctx.requires_grad_(), node.requires_grad_(), table.requires_grad_(), ner]
    from gcgcn_amd import functional as F_
    rows, pairs = F_.producer_live_counts(sen.view(torch.uint8), nv)      # capacities up front: no host read inside the step
    out = {}

    def step():
        for t in leaves + list(tail.parameters()):
            t.grad = None
        logits = tail(ctx, node, None, sen, ph, pt, ntype, rel, table, ner, n_valid=nv, max_live_slots=rows, max_live_pairs=pairs)
        loss = gcgcn_amd.pair_bce_loss(logits, labels, n_valid=nv).sum() / B
        loss.backward()
        out["logits"], out["loss"] = logits, loss

    def grad_refs():
        refs = {f"leaf{i}": t.grad for i, t in enumerate(leaves)}
        refs.update({n: p.grad for n, p in tail.named_parameters() if p.grad is not None})
        return refs

    # capture first (warm-up on a side stream, as bench.py and tools/_graph_mode.py do), then the eager reference step
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    torch.cuda.synchronize()
    g_logits, g_loss, g_grads = out["logits"], out["loss"], grad_refs()     # the tensors every replay rewrites
    step()                                                                  # eager, same inputs
    torch.cuda.synchronize()
    want = dict(logits=out["logits"].detach().clone(), loss=out["loss"].detach().clone(), grads={k: v.clone() for k, v in grad_refs().items()})
    assert len(want["grads"]) >= 6 and all(torch.isfinite(v).all() for v in want["grads"].values())
    assert want["grads"].keys() == g_grads.keys()
    for rep in range(3):
        g_logits.detach().fill_(float("nan"))                               # poison: the replay has to rewrite everything
        for v in g_grads.values():
            v.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g_logits.detach(), want["logits"]), f"replay {rep}: logits differ from the eager step"
        assert torch.equal(g_loss.detach(), want["loss"]), f"replay {rep}: loss differs from the eager step"
        for k, w_ in want["grads"].items():
            torch.testing.assert_close(g_grads[k], w_, rtol=1e-4, atol=1e-5 * max(1.0, w_.abs().max().item()),
                                       msg=lambda m: f"replay {rep}, gradient {k}: {m}")
        for k in ("head.flat", "graphcnn.0.flat", "get_weighted_adj_matrix.flat"):
            if k in want["grads"]:
                assert torch.equal(g_grads[k], want["grads"][k]), f"replay {rep}: gradient {k} is not bitwise the eager step's"
