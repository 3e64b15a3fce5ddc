"""How the HIP blocks sit inside torch.autograd (GPU): deferred weight gradients under every way a caller can run a
backward pass -- one module used twice in a pass (the reference trainer's own pattern: one model call per document, one
``total_loss.backward()`` per batch_size documents, config/Config.py:340-372), gradient accumulation, another consumer of
the parameter, ``autograd.grad``, ``backward(inputs=...)``, a pass that raises half-way, two models interleaved -- always
against GCGCN_DEFER=0 (products launched inside the block's own backward) and the CPU oracle."""
import pytest
import torch

import gcgcn_amd
from gcgcn_amd import functional as F_
from oracle import gcgcn_oracle as O

pytestmark = pytest.mark.gpu


def _leaf(t, dev):
    return t.to(dev).clone().requires_grad_()


def _grads(hops):
    return [None if p.grad is None else p.grad.clone() for p in hops.parameters()]


def _same(ga, gb, rtol=2e-5, atol=2e-5):
    assert len(ga) == len(gb)
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            torch.testing.assert_close(a, b, rtol=rtol, atol=atol * max(1.0, b.abs().max().item()))


@pytest.fixture
def setup(gpu_device):
    B, N, D, L, H = 2, 64, 128, 2, 4
    sd = O.init_stack_params(D, L, H, seed=71)
    docs = [O.synth_docs(B, N, D, seed=72 + k) for k in range(2)]
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    yield hops, docs, sd, (B, N, D, L, H)
    F_.defer_weight_grads = True


def _loss(hops, doc, dev):
    x, e1, e2, _ = doc
    return hops(_leaf(x, dev), [_leaf(e1, dev), _leaf(e2, dev)])[-1].sum()


def test_one_module_twice_in_one_backward(gpu_device, setup):
    """conv(x1) + conv(x2) through ONE backward: the gradient is the sum of both uses."""
    hops, docs, sd, (B, N, D, L, H) = setup
    res = {}
    for defer in (True, False):
        F_.defer_weight_grads = defer
        hops.zero_grad()
        (_loss(hops, docs[0], gpu_device) + _loss(hops, docs[1], gpu_device)).backward()
        assert not F_._passes
        res[defer] = _grads(hops)
    _same(res[True], res[False])
    # and against the oracle: the sum over both micro-batches
    sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
    tot = 0
    for x, e1, e2, adj in docs:
        for b in range(B):
            tot = tot + O.hop_stack(x[b], [e1[b], e2[b]], None, sdl, L, H)[-1].sum()
    tot.backward()
    ref = {k: v.grad for k, v in sdl.items() if v.grad is not None}
    for mod, pre in ((hops.get_weighted_adj_matrix, "get_weighted_adj_matrix."), (hops.graphcnn[0], "graphcnn.0."),
                     (hops.get_adj_matrix[0], "get_adj_matrix.0."), (hops.graphcnn[1], "graphcnn.1.")):
        for k, gk in mod.named_grads().items():
            torch.testing.assert_close(gk.cpu(), ref[pre + k], rtol=1e-3, atol=2e-4, msg=lambda m: f"{pre + k}: {m}")


def test_accumulation_over_two_backward_passes(gpu_device, setup):
    hops, docs, _, _ = setup
    res = {}
    for defer in (True, False):
        F_.defer_weight_grads = defer
        hops.zero_grad()
        _loss(hops, docs[0], gpu_device).backward()
        _loss(hops, docs[1], gpu_device).backward()        # .grad exists: the hand-over adds
        res[defer] = _grads(hops)
    _same(res[True], res[False])


def test_parameter_with_another_consumer(gpu_device, setup):
    """A regulariser on the flat parameter in the same graph: autograd's own contribution and the parked one both land."""
    hops, docs, _, _ = setup
    res = {}
    for defer in (True, False):
        F_.defer_weight_grads = defer
        hops.zero_grad()
        reg = sum((p * p).sum() for p in hops.parameters())
        (_loss(hops, docs[0], gpu_device) + 0.5 * reg).backward()
        res[defer] = _grads(hops)
    _same(res[True], res[False])
    conv = hops.graphcnn[1]
    assert (conv.flat.grad - conv.flat.detach()).abs().max() > 0     # more than the regulariser's gradient arrived


def test_autograd_grad_and_backward_inputs(gpu_device, setup):
    """torch.autograd.grad w.r.t. the parameters (captured, not accumulated) and backward(inputs=[x]) (parameter gradients
    not wanted): nothing is parked, results equal the plain pass."""
    hops, docs, _, _ = setup
    hops.zero_grad()
    _loss(hops, docs[0], gpu_device).backward()
    want = _grads(hops)
    hops.zero_grad()
    params = [p for p in hops.parameters() if not any(p is m.flat_k for m in hops.get_adj_matrix)]
    got = torch.autograd.grad(_loss(hops, docs[0], gpu_device), params)
    assert not F_._passes and all(p.grad is None for p in hops.parameters())
    _same([g for g in want if g is not None], list(got))
    x, e1, e2, _ = docs[0]
    xs = _leaf(x, gpu_device)
    hops(xs, [e1.to(gpu_device), e2.to(gpu_device)])[-1].sum().backward(inputs=[xs])
    assert xs.grad is not None and all(p.grad is None for p in hops.parameters()) and not F_._passes


def test_failed_backward_leaves_nothing_behind(gpu_device, setup):
    """A hook that raises after the convolutions have parked their products: the pass object dies with
    its graph task -- no parked operands kept alive, no stale queue entries -- and the next step is a normal step."""
    hops, docs, _, _ = setup
    hops.zero_grad()
    _loss(hops, docs[1], gpu_device).backward()
    want = _grads(hops)
    import gc
    for _ in range(3):
        hops.zero_grad()
        x, e1, e2, _ = docs[0]

        def boom(g):
            raise ValueError("boom")
        feats = hops(_leaf(x, gpu_device), [_leaf(e1, gpu_device), _leaf(e2, gpu_device)])
        feats[1].register_hook(boom)       # fires after the MAGGC hop's backward has parked its products, before any carrier
        out = feats[-1].sum()
        with pytest.raises(ValueError, match="boom"):
            out.backward()
        del out, feats
        gc.collect()
        assert all(r() is None for r in F_._passes.values())        # the failed pass is gone
    hops.zero_grad()
    _loss(hops, docs[1], gpu_device).backward()
    assert not F_._passes
    _same(_grads(hops), want, rtol=0, atol=0)              # bitwise the same step as before the failures


def test_two_models_interleaved(gpu_device, setup):
    """Two GraphHops instances, forwards interleaved, one backward over both / two backward passes in either order:
    modules are independent objects (glove:254-262), no state is shared between them."""
    hops, docs, sd, (B, N, D, L, H) = setup
    other = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    other.load_state_dict(O.init_stack_params(D, L, H, seed=99), strict=True)
    singles = []
    for m, d in ((hops, docs[0]), (other, docs[1])):
        m.zero_grad()
        _loss(m, d, gpu_device).backward()
        singles.append(_grads(m))
    # interleaved forwards (GAT of one model between GAT and convolution of the other), one backward
    hops.zero_grad(), other.zero_grad()
    xa, xb = (_leaf(d[0], gpu_device) for d in docs[:2])
    ea = [_leaf(docs[0][1], gpu_device), _leaf(docs[0][2], gpu_device)]
    eb = [_leaf(docs[1][1], gpu_device), _leaf(docs[1][2], gpu_device)]
    a_a = hops.get_weighted_adj_matrix(xa, ea[0])
    a_b = other.get_weighted_adj_matrix(xb, eb[0])
    ya = hops.graphcnn[0](xa, ea[0], a_a)                  # its edge mean is still parked although `other` came in between
    yb = other.graphcnn[0](xb, eb[0], a_b)
    za = hops.graphcnn[1](ya, ea[1], hops.get_adj_matrix[0](ya))
    zb = other.graphcnn[1](yb, eb[1], other.get_adj_matrix[0](yb))
    (za.sum() + zb.sum()).backward()
    assert not F_._passes
    _same(_grads(hops), singles[0])
    _same(_grads(other), singles[1])
    # two backward passes, reverse order of the forwards
    hops.zero_grad(), other.zero_grad()
    la, lb = _loss(hops, docs[0], gpu_device), _loss(other, docs[1], gpu_device)
    lb.backward()
    la.backward()
    _same(_grads(hops), singles[0])
    _same(_grads(other), singles[1])


def test_backward_from_another_thread(gpu_device, setup):
    """The rng scope and the edge-mean hand-off are per thread, the deferral queue per graph task."""
    import threading
    hops, docs, _, _ = setup
    hops.zero_grad()
    _loss(hops, docs[0], gpu_device).backward()
    want = _grads(hops)
    hops.zero_grad()
    err = []

    def work():
        try:
            torch.cuda.set_device(gpu_device)
            _loss(hops, docs[0], gpu_device).backward()
        except Exception as e:  # noqa: BLE001
            err.append(e)
    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert not err, err
    _same(_grads(hops), want)


def test_head_sum_from_forward_or_backward(gpu_device):
    """gcgcn_gcn_fwd's by-product wsum (sum over heads of the output projection, used by the fused chain backward) against
    the same sum computed inside gcgcn_gcn_bwd when the caller kept none: identical gradients."""
    B, N, D, L, H = 8, 64, 256, 2, 8
    sd = O.init_stack_params(D, L, H, seed=81)
    x, e1, e2, _ = O.synth_docs(B, N, D, seed=82)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train()
    hops.load_state_dict(sd, strict=True)
    res = {}
    try:
        for keep in (True, False):
            F_.head_sum_in_forward = keep
            gcgcn_amd.manual_seed(7)
            hops.zero_grad()
            xs = [_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops(xs[0], [xs[1], xs[2]])[-1].sum().backward()
            res[keep] = [t.grad.clone() for t in xs] + _grads(hops)
    finally:
        F_.head_sum_in_forward = True
    _same(res[True], res[False], rtol=0, atol=0)
