"""GPU parity tests: the HIP path (through the C ABI, via the drop-in nn.Modules) against
(1) golden vectors produced by the reference's own classes and (2) the CPU oracle on seeded inputs.
Bar: 1e-4 absolute on fp32 outputs and gradients (BASELINE.json north_star)."""
import ctypes
import math

import pytest
import torch

from conftest import golden_files, ids, load_golden
import gcgcn_amd
from gcgcn_amd import _lib, functional as F_
from oracle import gcgcn_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def dev_leaf(t, dev):
    return t.to(dev).clone().requires_grad_()


def close(a, b, what=""):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), **TOL, msg=lambda m: f"{what}: {m}")


def check_param_grads(module, golden_grad_sd, prefix=""):
    grads = module.named_grads()
    for k, ref in golden_grad_sd.items():
        if not k.startswith(prefix):
            continue
        kk = k[len(prefix):]
        assert kk in grads, f"missing gradient {kk}"
        close(grads[kk], ref, f"grad {k}")


# ------------------------------------------------------------------------------------------------------
# raw MFMA GEMM
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 1])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K,batch", [(64, 64, 16, 1), (100, 37, 53, 3), (256, 192, 256, 2), (5, 4, 8, 4),
                                          (130, 257, 18, 1)])
def test_gemm(gpu_device, M, N, K, batch, a_kc, b_kc, tile):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(batch, M, K, generator=g)
    B = torch.randn(batch, K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = torch.relu(0.5 * (A.double() @ B.double()) + bias.double()).float()
    # asymmetric operands + non-square shapes catch row/col swaps in the MFMA C/D map
    Ad = (A if a_kc else A.transpose(1, 2)).contiguous().to(gpu_device)
    Bd = (B.transpose(1, 2) if b_kc else B).contiguous().to(gpu_device)
    C = torch.full((batch, M, N), float("nan"), device=gpu_device)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    biasd = bias.to(gpu_device)
    _lib.call("gcgcn_gemm", M, N, K, p(Ad), lda, a_kc, p(Bd), ldb, b_kc, p(C), N, batch, M * K, K * N, M * N,
              0.5, p(biasd), 1, 0, tile, 1, None, 0, None)
    torch.cuda.synchronize()
    torch.testing.assert_close(C.cpu(), ref, rtol=1e-4, atol=1e-4)
    # accumulate: C += A B (no bias/relu)
    C2 = C.clone()
    _lib.call("gcgcn_gemm", M, N, K, p(Ad), lda, a_kc, p(Bd), ldb, b_kc, p(C2), N, batch, M * K, K * N, M * N,
              1.0, None, 0, 1, tile, 1, None, 0, None)
    torch.testing.assert_close(C2.cpu(), ref + (A.double() @ B.double()).float(), rtol=1e-4, atol=2e-4)


def test_gemm_a_identity_asymmetric_b(gpu_device):
    """A = I with an asymmetric B: output must equal B exactly (bitwise) -- catches transposed C writes."""
    n = 96
    B = torch.arange(n * 40, dtype=torch.float32).view(n, 40)
    A = torch.eye(n)
    C = torch.empty(n, 40, device=gpu_device)
    Ad, Bd = A.to(gpu_device), B.to(gpu_device)      # keep the device copies alive across the call
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for tile in (0, 1):
        _lib.call("gcgcn_gemm", n, 40, n, p(Ad), n, 1, p(Bd), 40, 0, p(C), 40, 1, 0, 0, 0,
                  1.0, None, 0, 0, tile, 1, None, 0, None)
        assert torch.equal(C.cpu(), B)


@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K,batch,splits", [(256, 128, 2048, 1, 8), (128, 64, 512, 3, 4), (2048, 256, 2048, 1, 0),
                                                 (100, 60, 256, 2, 4)])
def test_gemm_split_k(gpu_device, M, N, K, batch, splits, a_kc, b_kc):
    """Split-K through the workspace + reduce/epilogue kernel: same answer as the unsplit kernel to fp32
    summation-order slack, and bitwise reproducible run to run."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(batch, M, K, generator=g)
    B = torch.randn(batch, K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = torch.relu((A.double() @ B.double()) + bias.double()).float()
    Ad = (A if a_kc else A.transpose(1, 2)).contiguous().to(gpu_device)
    Bd = (B.transpose(1, 2) if b_kc else B).contiguous().to(gpu_device)
    biasd = bias.to(gpu_device)
    ws = torch.empty(16 * batch * M * N, device=gpu_device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for _ in range(2):
        C = torch.full((batch, M, N), float("nan"), device=gpu_device)
        _lib.call("gcgcn_gemm", M, N, K, p(Ad), K if a_kc else M, a_kc, p(Bd), K if b_kc else N, b_kc, p(C), N, batch,
                  M * K, K * N, M * N, 1.0, p(biasd), 1, 0, 0, splits, p(ws), ws.numel(), None)
        outs.append(C.cpu())
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=2e-4 * math.sqrt(K / 256))
    assert torch.equal(outs[0], outs[1])


def test_gemm_refuses_tile_bodies_that_no_longer_exist(gpu_device):
    """Rounds 2-3 carried two 128 x 128 tile bodies (tile = 2, 3) that lost every A/B on this path's products; round 4 removed
    them.  Asking for one is an error, not a silent fallback."""
    A, B, C = (torch.zeros(128, 128, device=gpu_device) for _ in range(3))
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for tile in (2, 3):
        with pytest.raises(RuntimeError, match="tile"):
            _lib.call("gcgcn_gemm", 128, 128, 128, p(A), 128, 1, p(B), 128, 0, p(C), 128, 1, 0, 0, 0, 1.0, None, 0, 0, tile, 1, None, 0, None)


# ------------------------------------------------------------------------------------------------------
# blocks against the reference's golden vectors
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("graphconv"), ids=ids(golden_files("graphconv")))
def test_graphconv_leaf_golden(gpu_device, path):
    """GraphConv (glove:18-50) with an all-zero adjacency row (the +1 normaliser branch, glove:47-49)."""
    g = load_golden(path)
    D, L = g["meta"]["d"], g["meta"]["l"]
    m = gcgcn_amd.GraphConv(D, D, D // L).to(gpu_device)
    m.load_state_dict(g["sd"], strict=True)
    ins = {k: dev_leaf(v, gpu_device) for k, v in g["in"].items()}
    out = m(ins["x"], ins["e"], ins["adj"])
    close(out, g["out"], "graphconv out")
    out.backward(g["cot"].to(gpu_device))
    for k in ("x", "e", "adj"):
        close(ins[k].grad, g["grad_in"][k], f"d{k}")
    close(m.weights_edge.grad, g["grad_sd"]["weights_edge"], "dWe")
    close(m.weights_node.grad, g["grad_sd"]["weights_node"], "dWn")


def test_graphconv_leaf_bias(gpu_device):
    """bias=True (never used by the reference model, but part of the leaf's signature): bias joins before the
    division by the row sum (glove:45-50)."""
    torch.manual_seed(3)
    N, D, G = 9, 16, 8
    m = gcgcn_amd.GraphConv(D, D, G, bias=True).to(gpu_device)
    with torch.no_grad():
        m.bias.copy_(torch.randn(G))
    x, e, a = torch.randn(N, D), torch.randn(N, N, D), torch.rand(N, N)
    xs = [dev_leaf(t, gpu_device) for t in (x, e, a)]
    out = m(*xs)
    out.sum().backward()
    we, wn, b = (t.detach().cpu().requires_grad_() for t in (m.weights_edge, m.weights_node, m.bias))
    xr = [t.clone().requires_grad_() for t in (x, e, a)]
    r = xr[2].sum(1)
    ref = (torch.einsum("ijk,kp->ijp", xr[1], we).mean(1) + xr[2] @ (xr[0] @ wn) + b) / (r + (r == 0).float()).unsqueeze(1)
    ref.sum().backward()
    close(out, ref, "out")
    close(m.bias.grad, b.grad, "dbias")
    close(xs[2].grad, xr[2].grad, "dA")
    close(xs[1].grad, xr[1].grad, "dE")


@pytest.mark.parametrize("path", golden_files("gat"), ids=ids(golden_files("gat")))
def test_gat_golden(gpu_device, path):
    g = load_golden(path)
    D = g["meta"]["d"]
    m = gcgcn_amd.GATAttention(D, D).to(gpu_device).eval()
    m.load_state_dict(g["sd"])
    x, e = dev_leaf(g["in"]["x"], gpu_device), dev_leaf(g["in"]["e"], gpu_device)
    out = m(x, e, g["in"]["mask"].to(gpu_device))
    close(out, g["out"], "gat out")
    out.backward(g["cot"].to(gpu_device))
    close(x.grad, g["grad_in"]["x"], "dX")
    close(e.grad, g["grad_in"]["e"], "dE")
    check_param_grads(m, g["grad_sd"])


@pytest.mark.parametrize("path", golden_files("caggc"), ids=ids(golden_files("caggc")))
def test_caggc_conv_golden(gpu_device, path):
    g = load_golden(path)
    D, L = g["meta"]["d"], g["meta"]["l"]
    m = gcgcn_amd.GraphConvolution(L, D, D).to(gpu_device).eval()
    m.load_state_dict(g["sd"])
    ins = {k: dev_leaf(v, gpu_device) for k, v in g["in"].items()}
    out = m(ins["x"], ins["e"], ins["adj"])
    close(out, g["out"], "caggc out")
    out.backward(g["cot"].to(gpu_device))
    for k in ("x", "e", "adj"):
        close(ins[k].grad, g["grad_in"][k], f"d{k}")
    check_param_grads(m, g["grad_sd"])


@pytest.mark.parametrize("path", golden_files("mha"), ids=ids(golden_files("mha")))
def test_mha_golden(gpu_device, path):
    g = load_golden(path)
    D, H = g["meta"]["d"], g["meta"]["h"]
    m = gcgcn_amd.MultiHeadAttention(H, D).to(gpu_device).eval()
    m.load_state_dict(g["sd"])
    x = dev_leaf(g["in"]["x"], gpu_device)
    outs = m(x, torch.zeros(1, device=gpu_device))          # 2nd arg ignored, as glove:336 relies on
    assert isinstance(outs, list) and len(outs) == H
    st = torch.stack(outs)
    close(st, g["out"], "mha out")
    st.backward(g["cot"].to(gpu_device))
    close(x.grad, g["grad_in"]["x"], "dX")
    check_param_grads(m, g["grad_sd"])
    assert m.flat_k.grad is None                           # linears_k never get a gradient


@pytest.mark.parametrize("path", golden_files("maggc"), ids=ids(golden_files("maggc")))
def test_maggc_conv_golden(gpu_device, path):
    g = load_golden(path)
    D, L, H = g["meta"]["d"], g["meta"]["l"], g["meta"]["h"]
    m = gcgcn_amd.MultiGraphConvolution(L, H, D, D).to(gpu_device).eval()
    m.load_state_dict(g["sd"])
    ins = {k: dev_leaf(v, gpu_device) for k, v in g["in"].items()}
    out = m(ins["x"], ins["e"], list(ins["adj"].unbind(0)))
    close(out, g["out"], "maggc out")
    out.backward(g["cot"].to(gpu_device))
    for k in ("x", "e", "adj"):
        close(ins[k].grad, g["grad_in"][k], f"d{k}")
    check_param_grads(m, g["grad_sd"])


@pytest.mark.parametrize("path", golden_files("stack"), ids=ids(golden_files("stack")))
def test_stack_golden(gpu_device, path):
    """GAT -> CAGGC conv -> MHA -> MAGGC conv through the hop glue, reference call pattern."""
    g = load_golden(path)
    D, L, H = g["meta"]["d"], g["meta"]["l"], g["meta"]["h"]
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(g["sd"], strict=True)
    x = dev_leaf(g["in"]["x"], gpu_device)
    e1, e2 = dev_leaf(g["in"]["e1"], gpu_device), dev_leaf(g["in"]["e2"], gpu_device)
    feats = hops(x, [e1, e2], g["in"]["adj"].to(gpu_device))
    close(feats[1], g["mid"]["x1"], "x1")
    close(feats[2], g["out"], "x2")
    feats[2].backward(g["cot"].to(gpu_device))
    close(x.grad, g["grad_in"]["x"], "dX")
    close(e1.grad, g["grad_in"]["e1"], "dE1")
    close(e2.grad, g["grad_in"]["e2"], "dE2")
    check_param_grads(hops.get_weighted_adj_matrix, g["grad_sd"], "get_weighted_adj_matrix.")
    check_param_grads(hops.graphcnn[0], g["grad_sd"], "graphcnn.0.")
    check_param_grads(hops.get_adj_matrix[0], g["grad_sd"], "get_adj_matrix.0.")
    check_param_grads(hops.graphcnn[1], g["grad_sd"], "graphcnn.1.")


def test_model_c1_hooks(gpu_device):
    """cfg 1: 8 synthetic docs through the real GCGCN_glove; our blocks fed what the reference's
    blocks saw must return what they returned (batched: all 8 docs in one call)."""
    g = load_golden(golden_files("model")[0])
    raw, L, H, docs = g["raw"], g["meta"]["l"], g["meta"]["h"], g["meta"]["docs"]
    hops = gcgcn_amd.GraphHops(g["meta"]["d"], L, H).to(gpu_device).eval()
    hops.load_state_dict(g["sd"], strict=True)
    st = lambda k: torch.stack([torch.from_numpy(raw[f"doc{i}.{k}"]) for i in range(docs)]).to(gpu_device)
    with torch.no_grad():
        feats = hops(st("x0"), [st("e1"), st("e2")], st("adj"))
        a0 = hops.get_weighted_adj_matrix(st("x0"), st("e1"))
        al = torch.stack(hops.get_adj_matrix[0](st("x1")), dim=1)
    close(a0, st("a0"), "gat A")
    close(feats[1], st("x1_new"), "caggc out")
    close(al, st("al"), "mha A")
    close(feats[2], st("x2_new"), "maggc out")


# ------------------------------------------------------------------------------------------------------
# batched / ragged / train-mode against the CPU oracle
# ------------------------------------------------------------------------------------------------------
def _oracle_stack(x, e1, e2, adj, sd, L, H, nv=None, keeps=None, relus=None, docs=None, traces=None):
    """Per-document loop of the CPU oracle; returns outputs and grads (loss = sum(out * cot)).  ``relus[b]`` / ``traces``:
    the HIP path's relu decisions replayed in the oracle and the oracle's pre-activations (oracle._relu); ``docs``: only
    these documents of the batch (lists stay indexed by position in ``docs``)."""
    B = x.shape[0]
    outs, gx, ge1, ge2 = [], [], [], []
    sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
    for b in (range(B) if docs is None else docs):
        n = x.shape[1] if nv is None else int(nv[b])
        xb = x[b, :n].detach().cpu().clone().requires_grad_()          # (device tensors: only the documents asked for travel)
        e1b = e1[b, :n, :n].detach().cpu().clone().requires_grad_()
        e2b = e2[b, :n, :n].detach().cpu().clone().requires_grad_()
        kb = None if keeps is None else keeps[b]
        tr = None
        if traces is not None:
            tr = {}
            traces.append(tr)
        f = O.hop_stack(xb, [e1b, e2b], None if adj is None else adj[b, :n, :n], sdl, L, H, keeps=kb,
                        relus=None if relus is None else relus[b], trace=tr)
        outs.append(f)
        gx.append(xb), ge1.append(e1b), ge2.append(e2b)
    return outs, gx, ge1, ge2, sdl


class _relu_spy:
    """Collects the relu outputs Y the two convolutions save for backward ([B, N, H, L, gh]; the HIP path's relu decisions
    are Y > 0) by wrapping GcnFn.forward for the duration of a ``with`` block."""

    def __enter__(self):
        self.Y = []
        self._orig = {cls: cls.forward for cls in (F_.GcnFn, F_.MaggcFn)}     # (MaggcFn: a MAGGC hop with its attention fused in)
        spy = self

        def wrap(cls):
            def fwd(ctx, x, ebar, adj_or_flat_mha, flat, n_valid, L, H, *rest):
                out = spy._orig[cls](ctx, x, ebar, adj_or_flat_mha, flat, n_valid, L, H, *rest)
                B, N, D = x.shape
                spy.Y.append(ctx.to_save[5].view(B, N, H, L, D // L))      # save_for_backward(x, ebar, adj, flat, Pn, Y, ...)
                return out
            return staticmethod(fwd)
        for cls in self._orig:
            cls.forward = wrap(cls)
        return self

    def __exit__(self, *exc):
        for cls, f in self._orig.items():
            cls.forward = staticmethod(f)
        return False

    def relus(self, docs):
        """{doc: oracle ``relus`` dict} from the first (CAGGC) and second (MAGGC) convolution call seen."""
        ycag, ymag = (y.detach().cpu() > 0 for y in self.Y[:2])
        L, H = ymag.shape[3], ymag.shape[2]
        return {b: {"cag": [ycag[b, :, 0, l] for l in range(L)],
                    "mag.1": [[ymag[b, :, h, l] for l in range(L)] for h in range(H)]} for b in docs}


def _check_relu_decisions(relus, traces, docs):
    """The HIP path's relu decisions against the oracle's own pre-activations (computed WITH the HIP decisions replayed, so
    both sides saw the same inputs at every layer): they may differ only where the pre-activation is a rounding error away
    from zero, and only in a handful of the millions of elements."""
    total = flips = 0
    worst = 0.0
    for b, tr in zip(docs, traces):
        for key in ("cag", "mag.1"):
            masks = relus[b][key] if key == "cag" else [m for per_head in relus[b][key] for m in per_head]
            for mask, pre in zip(masks, tr[key]):
                dis = (pre > 0) != mask
                total += mask.numel()
                flips += int(dis.sum())
                if dis.any():
                    worst = max(worst, pre[dis].abs().max().item())
    assert worst < 1e-5, f"a relu decision differs from the oracle's at |pre-activation| = {worst:.3e}"
    assert flips <= max(16, int(2e-5 * total)), f"{flips} of {total} relu decisions differ from the oracle's"
    return flips, total


def _check_stack_param_grads(hops, sdl, rtol=1e-3, atol=2e-4):
    """Every parameter gradient of the four blocks against the oracle's (sums over B*N rows: fp32 summation-order slack,
    absolute part relative to the tensor's largest entry)."""
    ref_grads = {k: v.grad for k, v in sdl.items() if v.grad is not None}
    seen = 0
    for mod, pre in ((hops.get_weighted_adj_matrix, "get_weighted_adj_matrix."), (hops.graphcnn[0], "graphcnn.0."),
                     (hops.get_adj_matrix[0], "get_adj_matrix.0."), (hops.graphcnn[1], "graphcnn.1.")):
        for k, gk in mod.named_grads().items():
            ref = ref_grads[pre + k]
            top = max(1.0, ref.abs().max().item())
            torch.testing.assert_close(gk.cpu(), ref, rtol=rtol, atol=atol * top, msg=lambda m: f"grad {pre + k}: {m}")
            seen += 1
    assert seen == len(ref_grads), (seen, len(ref_grads))       # nothing the reference differentiates is missing


@pytest.mark.parametrize("B,N,D,L,H", [(3, 64, 256, 2, 8), (2, 24, 96, 4, 4), (2, 70, 64, 2, 2),
                                       (1, 64, 768, 4, 4),      # cfg 3 (bert) document shape
                                       (1, 256, 512, 2, 8),     # cfg 5 (stress) document shape
                                       (2, 64, 384, 2, 2),      # gh = 192 with two sub-layers, gh = 256: column-strip shapes that
                                       (2, 64, 512, 2, 8),      #   met the oracle nowhere else (round-3 verdict, weak 3)
                                       (2, 33, 128, 4, 4)])     # the BERT model's own width: hidden 128, four sub-layers -> gh = 32 (bert:237,247-248)
def test_batched_matches_per_doc_oracle(gpu_device, B, N, D, L, H):
    sd = O.init_stack_params(D, L, H, seed=1337)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=5)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(3))
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    xg, e1g, e2g = dev_leaf(x, gpu_device), dev_leaf(e1, gpu_device), dev_leaf(e2, gpu_device)
    feats = hops(xg, [e1g, e2g], adj.to(gpu_device))
    (feats[2] * cot.to(gpu_device)).sum().backward()
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H)
    loss = sum((outs[b][2] * cot[b]).sum() for b in range(B))
    loss.backward()
    for b in range(B):
        close(feats[1][b], outs[b][1], f"x1[{b}]")
        close(feats[2][b], outs[b][2], f"x2[{b}]")
        close(xg.grad[b], gx[b].grad, f"dX[{b}]")
        close(e1g.grad[b], ge1[b].grad, f"dE1[{b}]")
        close(e2g.grad[b], ge2[b].grad, f"dE2[{b}]")
    _check_stack_param_grads(hops, sdl)


@pytest.mark.parametrize("B,N,D,L,H,nv", [(4, 32, 64, 2, 4, [32, 7, 19, 2]),
                                          # cfg 2 document shape: the LDS-resident chain kernels and the fused output-
                                          # projection backward on a ragged batch (B % 8 == 0: XCD-aware document order)
                                          (8, 64, 256, 2, 8, [64, 7, 19, 2, 64, 33, 1, 50])])
def test_ragged_batch_matches_truncated_docs(gpu_device, B, N, D, L, H, nv):
    nv = torch.tensor(nv)
    sd = O.init_stack_params(D, L, H, seed=11)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=9)
    for b in range(B):                                     # padding rows of X must be zero (gcgcn.h)
        x[b, nv[b]:] = 0
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(4))
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    xg, e1g, e2g = dev_leaf(x, gpu_device), dev_leaf(e1, gpu_device), dev_leaf(e2, gpu_device)
    nvg = nv.to(gpu_device)
    feats = hops(xg, [e1g, e2g], adj.to(gpu_device), n_valid=nvg)
    (feats[2] * cot.to(gpu_device)).sum().backward()
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H, nv=nv)
    sum((outs[b][2] * cot[b, :nv[b]]).sum() for b in range(B)).backward()
    for b in range(B):
        n = int(nv[b])
        close(feats[2][b, :n], outs[b][2], f"x2[{b}]")
        assert feats[2][b, n:].abs().max().item() == 0 if n < N else True
        close(xg.grad[b, :n], gx[b].grad, f"dX[{b}]")
        close(e1g.grad[b, :n, :n], ge1[b].grad, f"dE1[{b}]")
        close(e2g.grad[b, :n, :n], ge2[b].grad, f"dE2[{b}]")
        if n < N:
            assert e1g.grad[b, n:].abs().max().item() == 0 and e1g.grad[b, :, n:].abs().max().item() == 0
            assert e2g.grad[b, n:].abs().max().item() == 0 and e2g.grad[b, :, n:].abs().max().item() == 0
    _check_stack_param_grads(hops, sdl)


@pytest.mark.parametrize("B,N,D,L,H", [(32, 64, 256, 2, 8),      # the ragged bench: the split data gradients behind MAGGC's chain widen on the device
                                       (8, 64, 768, 4, 4)])      # cfg 3's widths: K = 3072 row-block products get their one host split
def test_ragged_launcher_switches_agree(gpu_device, B, N, D, L, H):
    """Round 5's launcher changes for ragged batches, each against its switch: split row-block products cut finer on the device
    (option split_widen) and the fused hop's attention projection parked (functional.defer_fused_mha_weight_grads).  Same
    snapshots, train mode, NaN-poisoned recycled memory: outputs and every gradient agree to summation-order slack."""
    g = torch.Generator().manual_seed(B + D)
    nv = torch.clamp(torch.round(torch.randn(B, generator=g) * 6.0 + 19.5), 1, N).to(torch.int32)
    nv[0], nv[-1] = N, 1
    sd = O.init_stack_params(D, L, H, seed=5)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=6)
    x = x * (torch.arange(N)[None, :] < nv[:, None]).unsqueeze(-1).float()
    cot = torch.randn(B, N, D, generator=g).to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train(True)
    hops.load_state_dict(sd, strict=True)
    nvg = nv.to(gpu_device)

    def run():
        gcgcn_amd.manual_seed(99, gpu_device)
        xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
        hops.zero_grad()
        junk = torch.full((B * N * H * D * 4,), float("nan"), device=gpu_device)
        del junk
        f = hops(xs[0], [xs[1], xs[2]], adj.to(gpu_device), n_valid=nvg)
        torch.autograd.backward(f[-1], cot)
        return [f[-1].detach(), xs[0].grad, xs[1].grad, xs[2].grad] + [p.grad.clone() for p in hops.parameters() if p.grad is not None]

    try:
        base = run()
        _lib.call("gcgcn_set_option", b"split_widen", 0)
        no_widen = run()
        _lib.call("gcgcn_set_option", b"split_widen", 1)
        F_.defer_fused_mha_weight_grads = False
        no_park = run()
    finally:
        _lib.call("gcgcn_set_option", b"split_widen", 1)
        F_.defer_fused_mha_weight_grads = True
    for name, other in (("split_widen=0", no_widen), ("dWq not parked", no_park)):
        assert len(other) == len(base)
        for k, (a, b_) in enumerate(zip(base, other)):
            assert torch.isfinite(a).all() and torch.isfinite(b_).all(), f"{name}: tensor {k} has non-finite values"
            top = max(a.abs().max().item(), 1e-6)
            torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-5 * top, msg=lambda m: f"{name}, tensor {k}: {m}")


@pytest.mark.parametrize("B,N,D,L,H,train", [(8, 64, 256, 2, 8, False),    # cfg 2's document shape: 2048 rows, split-K, parked tiles
                                             (8, 64, 256, 2, 8, True),     #   ... all six dropout sites on (same snapshots both ways)
                                             (5, 48, 128, 4, 4, False),    # the BERT model's widths, N = 48: three row blocks per document
                                             (32, 64, 256, 2, 8, True),    # the ragged bench itself (bench.py --ragged)
                                             (3, 32, 768, 4, 4, False),    # cfg 3's widths
                                             (136, 64, 64, 2, 2, False)])  # B N = 8704 rows: 544 blocks, past what a tile body keeps in its lanes -> weight gradients dense, the rest on row blocks
def test_row_block_launches_equal_the_dense_products(gpu_device, B, N, D, L, H, train):
    """Ragged batches: the node-phase products run on the LIVE 16-row blocks only (gcgcn_row_blocks; GEMM rows gathered through
    the block list, K = the live rows for weight gradients) against the same step with every row computed (the round-3 path,
    functional.row_block_launches = False): every output and gradient agrees to summation-order slack, padding rows of
    everything that leaves a block are EXACTLY zero, and all of it on NaN-poisoned recycled memory (conftest) -- a product that
    read a row nobody wrote would show."""
    g = torch.Generator().manual_seed(B * N + D)
    nv = torch.clamp(torch.round(torch.randn(B, generator=g) * 6.0 + 19.5), 1, N).to(torch.int32)
    nv[0], nv[-1] = N, 1                                      # a full document and a single entity
    sd = O.init_stack_params(D, L, H, seed=3)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=4)
    x = x * (torch.arange(N)[None, :] < nv[:, None]).unsqueeze(-1).float()
    cot = torch.randn(B, N, D, generator=g).to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train(train)
    hops.load_state_dict(sd, strict=True)
    nvg = nv.to(gpu_device)
    res = []
    seen = []
    orig_rb = F_.row_blocks

    def spy(n_valid, B_, N_):
        r = orig_rb(n_valid, B_, N_)
        seen.append(r)
        return r
    try:
        F_.row_blocks = spy
        for on in (True, False):
            F_.row_block_launches = on
            gcgcn_amd.manual_seed(77, gpu_device)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            junk = torch.full((B * N * H * D * 4,), float("nan"), device=gpu_device)     # poisoned blocks for the workspaces to recycle
            del junk
            f = hops(xs[0], [xs[1], xs[2]], adj.to(gpu_device), n_valid=nvg)
            torch.autograd.backward(f[-1], cot)
            res.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                       [p.grad.clone() for p in hops.parameters() if p.grad is not None])
    finally:
        F_.row_block_launches = True
        F_.row_blocks = orig_rb
    nblk = B * N // 16
    assert seen[0] is not None and seen[0].numel() == 4 + nblk and seen[-1] is None   # the list was built once per hop loop, then not at all
    live = int(seen[0][0].item())
    assert live == int(((nv + 15) // 16).sum())
    blocks = seen[0][4:4 + nblk].cpu().tolist()
    assert sorted(blocks) == list(range(nblk))                                                    # a permutation: live first, then dead
    assert all((blk % (N // 16)) * 16 < int(nv[blk // (N // 16)]) for blk in blocks[:live])
    # both halves ascending: the weight gradients sum over the live list two blocks at a time, in the dense product's row order
    assert blocks[:live] == sorted(blocks[:live]) and blocks[live:] == sorted(blocks[live:])
    assert seen[0][1:4].cpu().tolist() == [0, 0, 0]
    names = ["x1", "x2", "dX", "dE1", "dE2", "d gat", "d mha", "d caggc", "d maggc"]
    assert len(res[0]) == len(res[1]) == 9
    pad = (torch.arange(N)[None, :] >= nv[:, None]).to(gpu_device)
    for nm, a, b_ in zip(names, *res):
        assert torch.isfinite(a).all(), f"{nm}: non-finite values on the row-block path"
        top = max(1.0, b_.abs().max().item())
        torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-5 * top, msg=lambda m: f"{nm}: {m}")
        if nm in ("x1", "x2", "dX"):
            assert float(a[pad].abs().max()) == 0.0 if bool(pad.any()) else True, f"{nm}: padding rows must be exactly zero"


@pytest.mark.parametrize("B,N,D,L,H", [(2, 16, 32, 2, 4),        # guarded (ragged-shape) kernel instantiations
                                       (2, 64, 256, 2, 8),       # cfg 2 document shape: the ALIGNED chain / GEMM dropout
                                       (1, 64, 768, 4, 4),       # epilogues the bench runs; cfg 3 (bert) document shape
                                       (1, 128, 128, 2, 4)])     # more than 64 entities (what cfg 5 runs in train mode): the generic
                                                                 # ALIGNED chain's dropout epilogues, softmax_fwd / _bwd with
                                                                 # dropout, head_sum_drop_bwd, mask_rows
def test_train_mode_matches_oracle_with_replayed_masks(gpu_device, B, N, D, L, H):
    """Dropout on (6 sites, glove:59/74, 111, 131, 152, 341): replay the kernels' keep-masks in the CPU oracle; outputs,
    dX, dE1, dE2 and every parameter gradient of the four blocks must agree."""
    sd = O.init_stack_params(D, L, H, seed=21)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=22)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train()
    hops.load_state_dict(sd, strict=True)
    gcgcn_amd.manual_seed(1234)
    snaps = []
    orig = F_.rng_snapshot

    def spy(dev, lazy=False):
        s = orig(dev, lazy)
        snaps.append(s)
        return s
    F_.rng_snapshot = spy
    try:
        xg, e1g, e2g = dev_leaf(x, gpu_device), dev_leaf(e1, gpu_device), dev_leaf(e2, gpu_device)
        feats = hops(xg, [e1g, e2g], adj.to(gpu_device))
        feats[2].sum().backward()
    finally:
        F_.rng_snapshot = orig
    # call order: gat, caggc conv, glue0, mha, maggc conv, glue1
    assert len(snaps) == 6
    HD = H * D
    k_gat = F_.dropout_keep_mask(snaps[0], _lib.SALT_GAT, 0.1, B * N * N).view(B, N, N).cpu()
    k_cag = F_.dropout_keep_mask(snaps[1], _lib.SALT_GCN, 0.2, B * N * D).view(B, N, 1, L, D // L).cpu()
    k_gl0 = F_.dropout_keep_mask(snaps[2], _lib.SALT_GLUE, 0.2, B * N * D).view(B, N, D).cpu()
    k_mha = F_.dropout_keep_mask(snaps[3], _lib.SALT_MHA, 0.1, B * H * N * N).view(B, H, N, N).cpu()
    k_mag = F_.dropout_keep_mask(snaps[4], _lib.SALT_GCN, 0.2, B * N * HD).view(B, N, H, L, D // L).cpu()
    k_gl1 = F_.dropout_keep_mask(snaps[5], _lib.SALT_GLUE, 0.2, B * N * D).view(B, N, D).cpu()
    for k, p in ((k_mag, 0.2), (k_mha, 0.1)):
        assert abs(k.float().mean().item() - (1 - p)) < 0.03, "keep rate off"
    keeps = []
    for b in range(B):
        keeps.append({"gat": k_gat[b], "cag": [k_cag[b, :, 0, l] for l in range(L)], "glue.0": k_gl0[b],
                      "mha.1": [k_mha[b, h] for h in range(H)],
                      "mag.1": [[k_mag[b, :, h, l] for l in range(L)] for h in range(H)], "glue.1": k_gl1[b]})
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H, keeps=keeps)
    sum(outs[b][2].sum() for b in range(B)).backward()
    for b in range(B):
        close(feats[1][b], outs[b][1], f"x1[{b}]")
        close(feats[2][b], outs[b][2], f"x2[{b}]")
        close(xg.grad[b], gx[b].grad, f"dX[{b}]")
        close(e1g.grad[b], ge1[b].grad, f"dE1[{b}]")
        close(e2g.grad[b], ge2[b].grad, f"dE2[{b}]")
    _check_stack_param_grads(hops, sdl)
    # a second forward draws different masks
    f2 = hops(xg.detach(), [e1g.detach(), e2g.detach()], adj.to(gpu_device))
    assert not torch.equal(f2[2], feats[2])


# ------------------------------------------------------------------------------------------------------
# the step bench.py times: a captured hipGraph, replayed
# ------------------------------------------------------------------------------------------------------
def _capture_step(hops, sets, cot, n_valid=None):
    """One hipGraph per resident input set, captured exactly as bench.py does (a warm-up pass on a side stream, then the
    capture; gradients of the flat parameters are the tensors the capture installed)."""
    from gcgcn_amd.dist import FlatGradBucket
    bucket = FlatGradBucket(hops)
    outs = {}

    def fwd_bwd(k):
        x, e1, e2, adj = sets[k]
        x.grad = e1.grad = e2.grad = None
        bucket.zero_grad()
        f = hops(x, [e1, e2], adj, n_valid=n_valid)
        torch.autograd.backward(f[-1], cot)
        outs[k] = (f[1], f[2])

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for k in range(len(sets)):
            fwd_bwd(k)
    torch.cuda.current_stream().wait_stream(s)
    graphs, results = [], []
    for k in range(len(sets)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd_bwd(k)
        graphs.append(g)
        x, e1, e2, _ = sets[k]
        results.append(dict(x1=outs[k][0], x2=outs[k][1], dx=x.grad, de1=e1.grad, de2=e2.grad,
                            pgrads=[(p, p.grad) for p in bucket.params]))
    return bucket, fwd_bwd, graphs, results


def test_graph_replay_matches_eager(gpu_device):
    """bench.py's timed region is hipGraph replays.  Under capture the step takes code the eager tests never run (the lazy rng
    draw inside gcgcn_gat_fwd, the GAT fold forced on, the end-of-backward hand-over of deferred weight gradients as a captured
    graph task, .grad re-binding).  Eval mode, cfg-2 document shape, two rotating input sets: every replayed output and
    gradient is BITWISE what an eager step on the same inputs gives -- also on the second and third replay."""
    B, N, D, L, H = 4, 64, 256, 2, 8
    sd = O.init_stack_params(D, L, H, seed=1337)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    sets = []
    for k in range(2):
        x, e1, e2, adj = (t.to(gpu_device) for t in O.synth_docs(B, N, D, seed=70 + k))
        sets.append((x.requires_grad_(), e1.requires_grad_(), e2.requires_grad_(), adj))
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(9)).to(gpu_device)
    bucket, fwd_bwd, graphs, results = _capture_step(hops, sets, cot)

    def eager(k):
        x, e1, e2, adj = (t.detach().clone() for t in sets[k])
        for t in (x, e1, e2):
            t.requires_grad_()
        hops.zero_grad()
        f = hops(x, [e1, e2], adj)
        torch.autograd.backward(f[-1], cot)
        return dict(x1=f[1].detach().clone(), x2=f[2].detach().clone(), dx=x.grad, de1=e1.grad, de2=e2.grad,
                    pgrads=[p.grad.clone() for p in bucket.params])

    want = [eager(k) for k in range(2)]
    assert all(g is not None for r in want for g in r["pgrads"])
    for rep in range(3):
        for k in (0, 1):
            for key in ("x1", "x2", "dx", "de1", "de2"):                     # poison: a replay has to rewrite everything
                results[k][key].fill_(float("nan"))
            for _, g in results[k]["pgrads"]:
                g.fill_(float("nan"))
            graphs[k].replay()
            torch.cuda.synchronize()
            for key in ("x1", "x2", "dx", "de1", "de2"):
                assert torch.equal(results[k][key], want[k][key]), f"replay {rep}, set {k}: {key} differs from the eager step"
            for (p, g), w_ in zip(results[k]["pgrads"], want[k]["pgrads"]):
                assert torch.equal(g, w_), f"replay {rep}, set {k}: a flat parameter gradient differs from the eager step"


def test_graph_replay_train_mode_draws_fresh_masks_and_matches_oracle(gpu_device):
    """Train mode under replay: the {seed, counter} snapshots are drawn on the DEVICE inside the captured GATAttention launch,
    so every replay sees new dropout masks.  After each of two replays the snapshot buffer is read back, the six sites' keep
    masks are regenerated from it, and outputs + gradients equal the oracle run with exactly those masks."""
    B, N, D, L, H = 2, 64, 256, 2, 8
    sd = O.init_stack_params(D, L, H, seed=31)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=32)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train()
    hops.load_state_dict(sd, strict=True)
    gcgcn_amd.manual_seed(4321)
    sets = [(dev_leaf(x, gpu_device), dev_leaf(e1, gpu_device), dev_leaf(e2, gpu_device), adj.to(gpu_device))]
    cot = torch.ones(B, N, D, device=gpu_device)
    snaps = []
    orig = F_.rng_snapshot

    def spy(dev, lazy=False):
        s_ = orig(dev, lazy)
        snaps.append(s_)
        return s_
    F_.rng_snapshot = spy
    try:
        bucket, fwd_bwd, graphs, results = _capture_step(hops, sets, cot)
    finally:
        F_.rng_snapshot = orig
    assert len(snaps) == 12                    # 6 sites in the warm-up pass + 6 in the capture
    snaps = snaps[6:]                          # the captured launches read these (views of the scope's buffer in the graph's pool)
    HD = H * D
    r = results[0]
    seen_masks = []
    for rep in range(2):
        graphs[0].replay()
        torch.cuda.synchronize()
        k_gat = F_.dropout_keep_mask(snaps[0], _lib.SALT_GAT, 0.1, B * N * N).view(B, N, N).cpu()
        k_cag = F_.dropout_keep_mask(snaps[1], _lib.SALT_GCN, 0.2, B * N * D).view(B, N, 1, L, D // L).cpu()
        k_gl0 = F_.dropout_keep_mask(snaps[2], _lib.SALT_GLUE, 0.2, B * N * D).view(B, N, D).cpu()
        k_mha = F_.dropout_keep_mask(snaps[3], _lib.SALT_MHA, 0.1, B * H * N * N).view(B, H, N, N).cpu()
        k_mag = F_.dropout_keep_mask(snaps[4], _lib.SALT_GCN, 0.2, B * N * HD).view(B, N, H, L, D // L).cpu()
        k_gl1 = F_.dropout_keep_mask(snaps[5], _lib.SALT_GLUE, 0.2, B * N * D).view(B, N, D).cpu()
        seen_masks.append((k_gat, k_mag))
        keeps = [{"gat": k_gat[b], "cag": [k_cag[b, :, 0, l] for l in range(L)], "glue.0": k_gl0[b],
                  "mha.1": [k_mha[b, h] for h in range(H)],
                  "mag.1": [[k_mag[b, :, h, l] for l in range(L)] for h in range(H)], "glue.1": k_gl1[b]} for b in range(B)]
        outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H, keeps=keeps)
        sum(outs[b][2].sum() for b in range(B)).backward()
        for b in range(B):
            close(r["x1"][b], outs[b][1], f"replay {rep} x1[{b}]")
            close(r["x2"][b], outs[b][2], f"replay {rep} x2[{b}]")
            close(r["dx"][b], gx[b].grad, f"replay {rep} dX[{b}]")
            close(r["de1"][b], ge1[b].grad, f"replay {rep} dE1[{b}]")
            close(r["de2"][b], ge2[b].grad, f"replay {rep} dE2[{b}]")
        for p, g in r["pgrads"]:
            p.grad = g                                                       # what bench.py does after a replay
        _check_stack_param_grads(hops, sdl)
    assert not torch.equal(seen_masks[0][0], seen_masks[1][0]) and not torch.equal(seen_masks[0][1], seen_masks[1][1])


# ------------------------------------------------------------------------------------------------------
# full-size properties (cfg 2: B=32, N=64, D=256, L=2, H=8)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,B,N,D,L,H", [("c2", 32, 64, 256, 2, 8), ("c3", 32, 64, 768, 4, 4), ("c5", 32, 256, 512, 2, 8)])
def test_full_size_properties(gpu_device, cfg, B, N, D, L, H):
    """BASELINE.json's full batch sizes (cfg 2, cfg 3 = bert shape, cfg 5 = stress shape): the code paths only a full
    batch takes (split-K factors, group-launch overflow peeling, weight gradients carried by the CAGGC chain launch and
    the edge pass) checked through size-independent properties + one document against the CPU oracle."""
    sd = O.init_stack_params(D, L, H, seed=1337)
    g = torch.Generator(device=gpu_device).manual_seed(1337)
    x = torch.rand(B, N, D, generator=g, device=gpu_device) * 2 - 1
    adj = (torch.rand(B, N, N, generator=g, device=gpu_device) < 0.3).float() * (1 - torch.eye(N, device=gpu_device))
    e1 = torch.randn(B, N, N, D, generator=g, device=gpu_device) * 0.5 * adj.unsqueeze(-1)
    e2 = torch.randn(B, N, N, D, generator=g, device=gpu_device) * 0.5
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)

    def run(xs, e1s, e2s):
        xg, a, b = (t.clone().requires_grad_() for t in (xs, e1s, e2s))
        hops.zero_grad()        # same state for every run
        f = hops(xg, [a, b])
        f[2].sum().backward()
        return f, xg.grad, a.grad, b.grad, [p.grad.clone() for p in hops.parameters() if p.grad is not None]

    f, dx, de1, de2, pg = run(x, e1, e2)
    f_again, dx2, de1_2, de2_2, pg2 = run(x, e1, e2)
    assert torch.equal(f[2], f_again[2]) and torch.equal(dx, dx2) and torch.equal(de1, de1_2)   # deterministic
    assert torch.equal(de2, de2_2) and all(torch.equal(a, b) for a, b in zip(pg, pg2))
    del f_again, dx2, de1_2, de2_2, pg2
    # batch independence: a document alone (B=1, the reference's call shape) == the same document in the batch
    # (not bitwise: the GEMMs pick a different split-K factor for M = N rows than for M = B N)
    f1, dx1, de1_1, de2_1, _ = run(x[5:6], e1[5:6], e2[5:6])
    for a, b in ((f1[2][0], f[2][5]), (dx1[0], dx[5]), (de1_1[0], de1[5]), (de2_1[0], de2[5])):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-5)
    # attention rows are distributions; dE2 is constant along j
    with torch.no_grad():
        a0 = hops.get_weighted_adj_matrix(x, e1)
    torch.testing.assert_close(a0.sum(-1), torch.ones(B, N, device=gpu_device), rtol=1e-5, atol=1e-5)
    assert torch.equal(de2[:, :, 0, :], de2[:, :, N - 1, :])
    # linearity of the parameter gradients in the batch: grad(all docs) == grad(first half) + grad(second half)
    h = B // 2
    _, _, _, _, pa = run(x[:h], e1[:h], e2[:h])
    _, _, _, _, pb = run(x[h:], e1[h:], e2[h:])
    for whole, a, b in zip(pg, pa, pb):
        # (sums of 2048+ fp32 products in different split-K orders on each side: slack relative to the largest entry)
        torch.testing.assert_close(whole, a + b, rtol=1e-4, atol=5e-4 * max(1.0, whole.abs().max().item()))
    # one document of the full batch against the CPU oracle
    d = B - 1
    xr = x[d].cpu().requires_grad_()
    e1r, e2r = e1[d].cpu().requires_grad_(), e2[d].cpu().requires_grad_()
    ref = O.hop_stack(xr, [e1r, e2r], None, sd, L, H)
    ref[2].sum().backward()
    close(f[1][d], ref[1], f"{cfg} x1[{d}]")
    close(f[2][d], ref[2], f"{cfg} x2[{d}]")
    close(dx[d], xr.grad, f"{cfg} dX[{d}]")
    close(de1[d], e1r.grad, f"{cfg} dE1[{d}]")
    close(de2[d], e2r.grad, f"{cfg} dE2[{d}]")
    # The parameter gradients of the FULL batch (B = 32 launch: its split-K factors, carried weight-gradient tiles, group
    # peeling) against the oracle summed over documents, relu decisions replayed (_relu_spy).  cfg 2 / cfg 3: all 32
    # documents.  cfg 5 (the oracle needs ~10 s per document): the same B = 32 launch with a cotangent that is non-zero on
    # 4 documents only -- the other 28 contribute exact zeros to every sum, the oracle runs those 4.
    docs = list(range(B)) if cfg != "c5" else [0, 9, 18, 31]
    w = torch.zeros(B, device=gpu_device)
    w[docs] = 1.0
    xg, a, b = (t.clone().requires_grad_() for t in (x, e1, e2))
    hops.zero_grad()
    with _relu_spy() as spy:
        fw = hops(xg, [a, b])
    (fw[2] * w[:, None, None]).sum().backward()
    relus = spy.relus(docs)
    traces = []
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, None, sd, L, H, relus=relus, docs=docs, traces=traces)
    sum(o[2].sum() for o in outs).backward()
    flips, total = _check_relu_decisions(relus, traces, docs)
    print(f"{cfg}: {flips} of {total} relu decisions differ from the oracle's")
    for k, d_ in enumerate(docs[:2]):
        close(fw[2][d_], outs[k][2], f"{cfg} x2[{d_}] (relu decisions replayed)")
        close(xg.grad[d_], gx[k].grad, f"{cfg} dX[{d_}]")
    _check_stack_param_grads(hops, sdl)


@pytest.mark.parametrize("B,N,D,L,H", [(16, 64, 768, 4, 4),     # cfg 3 shape: B H <= 64 and a long chain -- the CAGGC chain
                                       (4, 64, 256, 2, 8)])    # launch takes parked products as two-tile passengers
def test_deferred_and_immediate_weight_gradients_agree_at_batch(gpu_device, B, N, D, L, H):
    """The weight gradients parked by the convolutions and carried by a later launch (gemm_take_deferred_pairs in the
    chain launch, gemm_take_deferred in GATAttention's edge pass) against the same products launched at once, and both
    against the per-document CPU oracle summed over the batch."""
    sd = O.init_stack_params(D, L, H, seed=61)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=62)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    res = {}
    try:
        for defer in (False, True):
            F_.defer_weight_grads = defer
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            with _relu_spy() as spy:
                out = hops(xs[0], [xs[1], xs[2]])[-1]
            out.sum().backward()
            assert not F_._passes
            res[defer] = [p.grad.clone() for p in hops.parameters() if p.grad is not None]
    finally:
        F_.defer_weight_grads = True
    for a_, b_ in zip(res[True], res[False]):
        torch.testing.assert_close(a_, b_, rtol=2e-5, atol=2e-5 * max(1.0, b_.abs().max().item()))
    # against the oracle with the HIP path's relu decisions replayed (the last run's: hops' gradients are the deferred ones)
    docs = list(range(B))
    relus, traces = spy.relus(docs), []
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H, relus=relus, traces=traces)
    sum(o[2].sum() for o in outs).backward()
    _check_relu_decisions(relus, traces, docs)
    _check_stack_param_grads(hops, sdl)


@pytest.mark.parametrize("B,N,D,L,H,ragged,train", [(3, 64, 256, 2, 8, False, True),     # cfg 2's shape: head width 32
                                                     (4, 42, 128, 2, 8, True, True),      # the reference's model: head width 16
                                                     (2, 16, 128, 2, 8, False, False),    # cfg 1
                                                     (2, 64, 128, 2, 2, True, False),     # head width 64: backward core as its own launch
                                                     (2, 64, 768, 4, 4, False, True),     # cfg 3: head width 192, the core's Q in two chunks
                                                     (2, 40, 384, 2, 2, True, True),      # head width 192, ragged, N < 64
                                                     (2, 32, 64, 1, 4, False, False)])    # one sub-layer: no room for the core in the chain
                                                                                          # kernel's LDS -> the core keeps its launch
def test_fused_maggc_hop_equals_separate_modules(gpu_device, B, N, D, L, H, ragged, train):
    """GraphHops.fuse_maggc: MultiHeadAttention + MultiGraphConvolution of a hop as ONE autograd node (functional.MaggcFn: the
    query projection as a problem of the convolution's first group launch, the attention core's backward as passenger workgroups
    of its last one) against the two separate module calls -- same dropout draws, outputs and every gradient."""
    assert F_.maggc_fusable(torch.empty(B, N, D, device=gpu_device), H)
    sd = O.init_stack_params(D, L, H, seed=3 * N + H)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=N + H)
    n_valid = None
    if ragged:
        n_valid = torch.randint(2, N + 1, (B,), generator=torch.Generator().manual_seed(N)).to(torch.int32)
        n_valid[0] = N
        x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
        n_valid = n_valid.to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device)
    hops.train(train)
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(6)).to(gpu_device)
    res, nodes = [], []
    try:
        for fuse in (True, False):
            hops.fuse_maggc = fuse
            gcgcn_amd.manual_seed(123, gpu_device)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            f = hops(xs[0], [xs[1], xs[2]], n_valid=n_valid)
            nodes.append(type(f[-1].grad_fn).__name__)
            torch.autograd.backward(f[-1], cot)
            res.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                       [p.grad.clone() for p in hops.parameters() if p.grad is not None])
    finally:
        hops.fuse_maggc = True
    assert "Maggc" in nodes[0] and "Maggc" not in nodes[1], nodes
    assert len(res[0]) == len(res[1]) == 9
    for nm, a, b_ in zip(["x1", "x2", "dX", "dE1", "dE2", "d gat", "d mha", "d caggc", "d maggc"], *res):
        torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-5 * max(1.0, b_.abs().max().item()), msg=lambda m: f"{nm}: {m}")


def test_edge_mean_handoff_is_used_and_safe(gpu_device):
    """GraphConvolution reuses GATAttention's edge mean only for the very same, unmodified tensor."""
    D = 32
    gat = gcgcn_amd.GATAttention(D, D).to(gpu_device).eval()
    conv = gcgcn_amd.GraphConvolution(2, D, D).to(gpu_device).eval()
    x = torch.randn(6, D, device=gpu_device)
    e = torch.randn(6, 6, D, device=gpu_device)
    with torch.no_grad():
        a = gat(x, e)
        assert id(e) in F_._handoffs()
        y1 = conv(x, e, a)
        assert id(e) not in F_._handoffs()              # consumed
        y2 = conv(x, e, a)                              # recomputed from E
        a = gat(x, e)
        e2 = e.clone()
        y3 = conv(x, e2, a)                             # different tensor object: hand-off ignored
        a = gat(x, e)
        e.mul_(2.0)                                     # in-place change bumps the version: ignored
        y4 = conv(x, e, a)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    assert not torch.allclose(y1, y4)


def test_kernel_timer_api(gpu_device):
    """gcgcn_prof_start/enable/stop: HIP-event timing of the launches of one kernel family + their work."""
    import ctypes as ct
    B, N, D = 4, 32, 64
    e = torch.randn(B, N, N, D, device=gpu_device)
    _lib.call("gcgcn_prof_start", b"edge_fwd_mean", 16)
    for _ in range(3):
        F_.edge_mean(e)
    _lib.call("gcgcn_prof_enable", 0)
    F_.edge_mean(e)                      # not recorded
    _lib.call("gcgcn_prof_enable", 1)
    F_.edge_mean(e)
    torch.cuda.synchronize()
    ms, n, w = ct.c_double(0), ct.c_int(0), ct.c_double(0)
    _lib.call("gcgcn_prof_stop", ct.byref(ms), ct.byref(n), ct.byref(w))
    assert n.value == 4 and ms.value > 0
    assert w.value == 4 * 4.0 * B * N * N * D          # algorithmic bytes of the recorded launches
    F_.edge_mean(e)                      # timer off again: nothing recorded, nothing crashes


def test_chain_and_per_product_paths_agree(gpu_device):
    """chain.hip (one persistent workgroup per (doc, head)) against one launch per product: same numbers."""
    B, N, D, L, H = 3, 64, 128, 2, 4
    sd = O.init_stack_params(D, L, H, seed=5)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=6)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    res = []
    try:
        for chain in (1, 0):
            _lib.call("gcgcn_set_option", b"chain", chain)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            out = hops(xs[0], [xs[1], xs[2]])[-1]
            out.sum().backward()
            res.append((out.detach(), xs[0].grad, xs[1].grad, hops.graphcnn[1].flat.grad.clone()))
            hops.zero_grad()
    finally:
        _lib.call("gcgcn_set_option", b"chain", 1)
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="unknown option"):
        _lib.call("gcgcn_set_option", b"bogus", 1)


@pytest.mark.parametrize("N,D,L,H", [(64, 256, 2, 8), (42, 128, 2, 8), (16, 128, 2, 8), (64, 768, 4, 4), (128, 128, 2, 4)])
@pytest.mark.parametrize("train", [False, True])
def test_empty_document_in_a_batch(gpu_device, N, D, L, H, train):
    """n_valid = 0 (a batch padded with an empty document): every output and gradient of that document is exactly zero, nothing
    is NaN, and the other documents come out as they do in a batch without it."""
    sd = O.init_stack_params(D, L, H, seed=3)
    x, e1, e2, adj = O.synth_docs(3, N, D, seed=4)
    nv = torch.tensor([N, 0, 5], dtype=torch.int32)
    x = x * (torch.arange(N)[None, :] < nv[:, None]).unsqueeze(-1).float()
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train(False)   # (dropout draws depend on the position in the batch)
    hops.load_state_dict(sd, strict=True)

    def run(sel):
        xs = [dev_leaf(t[sel], gpu_device) for t in (x, e1, e2)]
        hops.zero_grad()
        hops.train(train and len(sel) == 3)
        f = hops(xs[0], [xs[1], xs[2]], adj[sel].to(gpu_device), n_valid=nv[sel].to(gpu_device))
        f[-1].sum().backward()
        return [f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad], [p.grad.clone() for p in hops.parameters() if p.grad is not None]
    full, pfull = run([0, 1, 2])
    for t in full + pfull:
        assert torch.isfinite(t).all()
    for t in full:
        assert float(t[1].abs().max()) == 0.0
    if not train:
        part, ppart = run([0, 2])
        for a, b_ in zip(full, part):
            top = max(1.0, float(b_.abs().max()))
            torch.testing.assert_close(a[[0, 2]], b_, rtol=2e-4, atol=2e-5 * top)
        for a, b_ in zip(pfull, ppart):
            top = max(1.0, float(b_.abs().max()))
            torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-5 * top)


@pytest.mark.parametrize("B,N,D,L,H,ragged,train", [
    (4, 64, 256, 2, 8, False, True),     # cfg 2's shape: chain_s, fused MAGGC hop, parked tiles
    (4, 64, 256, 2, 8, True, True),      #   ... ragged: row blocks (tiles of four, k-tiles of two)
    (3, 64, 768, 4, 4, False, True),     # cfg 3's shape: chain_t <192, 4>
    (3, 64, 768, 4, 4, True, False),
    (4, 42, 128, 2, 8, True, True),      # the reference's own model
    (3, 42, 128, 4, 4, True, True),      # the BERT model's graph blocks
    (8, 16, 128, 2, 8, False, True),     # cfg 1
    (1, 128, 128, 2, 4, False, True),    # more than 64 entities: per-product launches, softmax kernels
    (2, 80, 64, 2, 2, True, False),      # unaligned shapes (the guarded instantiations, which spill the most)
])
def test_whole_step_is_deterministic(gpu_device, B, N, D, L, H, ragged, train):
    """The library claims bitwise reproducibility (no atomics, fixed summation orders, counter-based dropout): the same hop loop
    run three times from the same seeds returns identical outputs and gradients, on NaN-poisoned recycled memory."""
    sd = O.init_stack_params(D, L, H, seed=11)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=12)
    n_valid = None
    if ragged:
        n_valid = torch.randint(1, N + 1, (B,), generator=torch.Generator().manual_seed(N + B)).to(torch.int32)
        n_valid[0] = N
        if B > 1:
            n_valid[1] = min(N, 39)                    # three row blocks
        x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
        n_valid = n_valid.to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train(train)
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    runs = []
    for _ in range(3):
        gcgcn_amd.manual_seed(123, gpu_device)
        xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
        hops.zero_grad()
        junk = torch.full((B * N * H * D * 4,), float("nan"), device=gpu_device)
        del junk
        f = hops(xs[0], [xs[1], xs[2]], adj.to(gpu_device), n_valid=n_valid)
        torch.autograd.backward(f[-1], cot)
        runs.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                    [p.grad.clone() for p in hops.parameters() if p.grad is not None])
    names = ["x1", "x2", "dX", "dE1", "dE2", "d gat", "d mha", "d caggc", "d maggc"]
    for r in runs[1:]:
        for nm, a, b_ in zip(names, runs[0], r):
            assert torch.equal(a, b_), f"{nm} differs between two runs by {float((a - b_).abs().max()):.3e}"


@pytest.mark.parametrize("gh,L", [(32, 2), (32, 4), (64, 1), (64, 2), (64, 3), (64, 4), (128, 1), (128, 2), (128, 3), (128, 4),
                                  (192, 2), (192, 4), (256, 1), (256, 2)])
def test_chain_t_is_deterministic(gpu_device, gh, L):
    """Every instantiation of the column-strip chain kernels, documents of one to four row blocks, one and two heads, N a
    multiple of 16 and not: a block's forward and backward run three times on the same inputs must agree BITWISE (out, dX,
    dEbar, dA, dflat).  Round 4: one build of the <192, 4, ragged> backward returned a different dA on every run for documents
    of three row blocks; round 5 found why (a spill store read an MFMA result four wait states after the MFMA on the path that
    skipped the fourth row block) -- the build now checks every kernel's assembly for that
    (tools/isa_mfma_hazard_check.py), the kernels have no branch around a matrix instruction left, and none of them spills."""
    D = gh * L
    _scan_chain_t_determinism(gpu_device, gh, L, D, 64)
    if (gh, L) in ((64, 2), (192, 4), (32, 4)):      # rows bounded by the buffer descriptor instead of by N % 16 == 0
        _scan_chain_t_determinism(gpu_device, gh, L, D, 42, ([42, 39, 17, 5], [33, 20], None))


def _scan_chain_t_determinism(gpu_device, gh, L, D, N, cases=([64, 39, 17, 5], [48, 33], None)):
    for H in (1, 2):
        for nvl in cases:
            B = len(nvl) if nvl else 2
            g = torch.Generator().manual_seed(gh + L)
            flat = (torch.randn(_lib.layout("gcn", D, L, H)[5], generator=g) * 0.05).to(gpu_device)
            nv = torch.tensor(nvl if nvl else [N] * B, dtype=torch.int32)
            m = (torch.arange(N)[None, :] < nv[:, None]).float()
            x = (torch.randn(B, N, D, generator=g) * m[..., None]).to(gpu_device)
            ebar = (torch.randn(B, N, D, generator=g) * 0.3 * m[..., None]).to(gpu_device)
            adj = (torch.rand(B, H, N, N, generator=g) * m[:, None, :, None] * m[:, None, None, :]).to(gpu_device)
            cot = torch.randn(B, N, D, generator=g).to(gpu_device)
            runs = []
            for _ in range(3):
                xs = [t.clone().requires_grad_() for t in (x, ebar, adj, flat)]
                junk = torch.full((B * N * H * D * 4,), float("nan"), device=gpu_device)     # poisoned blocks for the workspaces to recycle
                del junk
                o = F_.gcn_stack(xs[0], xs[1], xs[2], xs[3], L, H, n_valid=nv.to(gpu_device) if nvl else None, training=False)
                torch.autograd.backward(o, cot)
                runs.append([o.detach()] + [t.grad.clone() for t in xs])
            for r in runs[1:]:
                for name, a, b_ in zip(("out", "dX", "dEbar", "dA", "dflat"), runs[0], r):
                    assert torch.equal(a, b_), f"gh={gh} L={L} H={H} n_valid={nvl}: {name} differs between two runs by {float((a - b_).abs().max()):.3e}"


@pytest.mark.parametrize("B,N,D,L,H,train", [(2, 128, 128, 2, 4, False), (1, 128, 128, 2, 4, True), (2, 80, 64, 2, 2, False)])
def test_graphs_above_64_entities_chain_kernels_against_per_product_launches(gpu_device, B, N, D, L, H, train):
    """More than 64 entities (what cfg 5 runs): by default only a FORWARD chain launch with an edge mean riding in it uses the
    generic chain kernels, everything else one launch per product (round-4 A/B at cfg 5).  Option chain_big = 1 puts the chain
    kernels back everywhere: same outputs and gradients, same dropout draws.  (N = 80: the unaligned instantiations.)"""
    sd = O.init_stack_params(D, L, H, seed=31)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=32)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).train(train)
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(9)).to(gpu_device)
    res = []
    try:
        for big in (0, 1):
            _lib.call("gcgcn_set_option", b"chain_big", big)
            gcgcn_amd.manual_seed(99, gpu_device)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            f = hops(xs[0], [xs[1], xs[2]], adj.to(gpu_device))
            torch.autograd.backward(f[-1], cot)
            res.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                       [p.grad.clone() for p in hops.parameters() if p.grad is not None])
    finally:
        _lib.call("gcgcn_set_option", b"chain_big", 0)
    assert len(res[0]) == len(res[1]) == 9
    for i, (a, b_) in enumerate(zip(*res)):
        top = max(1.0, b_.abs().max().item())
        torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-5 * top, msg=lambda m: f"tensor {i}: {m}")


@pytest.mark.parametrize("B,N,D,L,H,ragged,train", [
    (3, 64, 768, 4, 4, False, False),    # cfg 3 (bert-sized): gh = 192, four sub-layers, 12 waves
    (2, 64, 768, 4, 4, True, True),      #   ... ragged, dropout on
    (4, 42, 128, 2, 8, True, False),     # the reference's own model: hidden 128 -> gh = 64, N <= 42 ragged (glove:234, 250-251)
    (3, 42, 128, 2, 8, True, True),
    (8, 16, 128, 2, 8, False, True),     # cfg 1
    (2, 64, 256, 2, 8, False, True),     # cfg 2's shape (served by gcn_chain_s_* by default; chain_t = 2 forces these kernels)
    (2, 64, 512, 2, 8, True, False),     # gh = 256: 16 waves
    (2, 30, 192, 3, 2, True, False),     # gh = 64, three sub-layers
    (2, 64, 384, 2, 2, False, False),    # gh = 192, two sub-layers
    (2, 7, 256, 4, 2, False, True),      # gh = 64, four sub-layers, one row block
    (1, 1, 64, 1, 1, False, False),      # a single entity, a single sub-layer
    (3, 42, 128, 4, 4, True, True),      # the BERT model's graph blocks: hidden 128 over four sub-layers -> gh = 32, two waves (bert:237,247-248)
    (2, 64, 128, 4, 4, False, False),    #   ... every row block full
    (2, 20, 64, 2, 2, True, False),      # gh = 32, two sub-layers
])
def test_chain_t_matches_generic_chain(gpu_device, B, N, D, L, H, ragged, train):
    """chain_t.hpp (column strips, chained products, pushed dense connections; N <= 64) against the generic chain kernels on
    the same inputs, same dropout snapshots: outputs, dX, dE and every parameter gradient.  (The generic kernels are tied to
    the oracle by every other test in this file; at cfg 3 / the reference's shape those tests now run chain_t themselves.)"""
    # (seed of the (2, 64, 384, 2, 2) case: with 7 N + L one pre-activation of document 0 lands within an ulp of zero, the two
    # kernel generations round it to different sides and a whole gradient column legitimately differs -- see _check_relu_decisions)
    sd = O.init_stack_params(D, L, H, seed=7 * N + L + (1 if D == 384 else 0))
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=N + D)
    n_valid = None
    if ragged:
        n_valid = torch.randint(1, N + 1, (B,), generator=torch.Generator().manual_seed(N)).to(torch.int32)
        n_valid[0] = N
        x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
        n_valid = n_valid.to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device)
    hops.train(train)
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    res, masks = [], []
    try:
        for mode in (2, 0):
            _lib.call("gcgcn_set_option", b"chain_t", mode)
            gcgcn_amd.manual_seed(99, gpu_device)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            with _relu_spy() as spy:
                f = hops(xs[0], [xs[1], xs[2]], n_valid=n_valid)
            torch.autograd.backward(f[-1], cot)
            res.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                       [p.grad.clone() for p in hops.parameters() if p.grad is not None])
            masks.append([y.detach() > 0 for y in spy.Y])
    finally:
        _lib.call("gcgcn_set_option", b"chain_t", 1)
    flips = sum(int((a != b_).sum()) for a, b_ in zip(*masks))
    assert flips == 0, f"{flips} relu decisions differ between the two kernel generations on this seed (a pre-activation within " \
                       "an ulp of zero): both are valid fp32 evaluations, but their gradients are not comparable -- pick another seed"
    names = ["x1", "x2", "dX", "dE1", "dE2", "d gat", "d mha", "d caggc", "d maggc"]
    assert len(res[0]) == len(res[1]) == 9
    bad = []
    for nm, a, b_ in zip(names, *res):
        top = max(1.0, b_.abs().max().item())
        err = (a - b_).abs()
        if not bool((err <= 2e-5 * top + 2e-4 * b_.abs()).all()):
            bad.append(f"{nm}: max |diff| {err.max().item():.3e} (largest entry {top:.3e}), {int((err > 2e-5 * top + 2e-4 * b_.abs()).sum())} of {err.numel()} off")
    assert not bad, "; ".join(bad)


@pytest.mark.parametrize("N", [3, 15, 16, 17, 31, 33, 47, 48, 49, 63])
@pytest.mark.parametrize("H", [1, 2])
def test_chain_t_bounds_by_descriptor_at_every_row_count(gpu_device, N, H):
    """Round 5: the column-strip kernels bound their strip accesses by buffer descriptors (rows past N read zero, their stores are
    dropped by the hardware's range check) instead of comparing rows with N.  Graphs of N entities on either side of every 16-row
    block boundary, three documents back to back in memory (a store past a document's last row would land in its neighbour), ragged
    n_valid, one head (the fused backward's dXres = dHO path) and two: everything against the generic chain kernels, which test
    rows the old way."""
    B, D, L = 3, 128, 2
    sd = O.init_stack_params(D, L, H, seed=31 * N + H)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=N + 7 * H)
    n_valid = torch.tensor([N, max(1, N - 5), max(1, (N + 1) // 2)], dtype=torch.int32)
    x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    res = []
    try:
        for mode in (2, 0):
            _lib.call("gcgcn_set_option", b"chain_t", mode)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            f = hops(xs[0], [xs[1], xs[2]], n_valid=n_valid.to(gpu_device))
            torch.autograd.backward(f[-1], cot)
            res.append([f[1].detach(), f[2].detach(), xs[0].grad, xs[1].grad, xs[2].grad] +
                       [p.grad.clone() for p in hops.parameters() if p.grad is not None])
    finally:
        _lib.call("gcgcn_set_option", b"chain_t", 1)
    names = ["x1", "x2", "dX", "dE1", "dE2", "d gat", "d mha", "d caggc", "d maggc"]
    assert len(res[0]) == len(res[1]) == 9
    for nm, a, b_ in zip(names, *res):
        assert torch.isfinite(a).all(), f"{nm}: not finite"
        top = max(1.0, b_.abs().max().item())
        err = (a - b_).abs()
        assert bool((err <= 2e-5 * top + 2e-4 * b_.abs()).all()), f"{nm}: max |diff| {err.max().item():.3e} (largest entry {top:.3e})"
    for b in range(B):   # padding rows of the outputs are exact zeros
        assert float(res[0][1][b, int(n_valid[b]):].abs().max() if int(n_valid[b]) < N else 0.0) == 0.0


def test_relu_decision_within_an_ulp_of_zero_is_what_separates_the_kernel_generations(gpu_device):
    """The seed on which test_chain_t_matches_generic_chain[2-64-384-2-2] once failed with a whole gradient column apart
    (round 3; answered then by moving the seed).  The claim -- ONE relu pre-activation lands within rounding of zero and the
    two kernel generations put it on different sides, both being correct fp32 evaluations -- as a test: each generation runs
    against the ORACLE with its own relu decisions replayed and must match it in every output and gradient; each generation's
    decisions may differ from the oracle's pre-activations only where |pre-activation| < 1e-5; and wherever the two
    generations disagree with each other, that same bound holds for the disputed elements."""
    B, N, D, L, H = 2, 64, 384, 2, 2
    sd = O.init_stack_params(D, L, H, seed=7 * N + L)             # the original seed
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=N + D)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(5))
    docs = list(range(B))
    decided, pres = [], []
    try:
        for mode in (2, 0):                                       # column-strip kernels, then the generic chain
            _lib.call("gcgcn_set_option", b"chain_t", mode)
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            hops.zero_grad()
            with _relu_spy() as spy:
                f = hops(xs[0], [xs[1], xs[2]])
            torch.autograd.backward(f[-1], cot.to(gpu_device))
            relus = spy.relus(docs)
            traces = []
            outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, None, sd, L, H, relus=relus, traces=traces)
            sum((outs[b][2] * cot[b]).sum() for b in range(B)).backward()
            _check_relu_decisions(relus, traces, docs)            # |pre-activation| < 1e-5 wherever this generation differs from the oracle
            for b in range(B):
                close(f[1][b], outs[b][1], f"chain_t={mode} x1[{b}]")
                close(f[2][b], outs[b][2], f"chain_t={mode} x2[{b}]")
                close(xs[0].grad[b], gx[b].grad, f"chain_t={mode} dX[{b}]")
                close(xs[1].grad[b], ge1[b].grad, f"chain_t={mode} dE1[{b}]")
                close(xs[2].grad[b], ge2[b].grad, f"chain_t={mode} dE2[{b}]")
            _check_stack_param_grads(hops, sdl)
            decided.append(relus)
            pres.append(traces)
    finally:
        _lib.call("gcgcn_set_option", b"chain_t", 1)
    # the elements on which the two generations decided differently: a rounding error away from zero in BOTH oracle traces
    disputed, worst = 0, 0.0
    for b in docs:
        for key in ("cag", "mag.1"):
            ma = decided[0][b][key] if key == "cag" else [m for ph in decided[0][b][key] for m in ph]
            mb = decided[1][b][key] if key == "cag" else [m for ph in decided[1][b][key] for m in ph]
            for m0, m1, p0, p1 in zip(ma, mb, pres[0][b][key], pres[1][b][key]):
                dis = m0 != m1
                if dis.any():
                    disputed += int(dis.sum())
                    worst = max(worst, p0[dis].abs().max().item(), p1[dis].abs().max().item())
    assert worst < 1e-5, f"the generations disagree on a relu whose pre-activation is {worst:.3e} away from zero"
    assert disputed <= 16, f"{disputed} disputed relu decisions"
    print(f"disputed relu decisions between the kernel generations: {disputed}, largest |pre-activation| {worst:.3e}")


@pytest.mark.parametrize("B,N,D,H,ragged", [(3, 64, 256, 8, False), (2, 42, 128, 4, True), (2, 7, 64, 4, False),
                                              (2, 64, 768, 4, True), (1, 13, 512, 2, False)])
def test_mha_core_and_generic_paths_agree(gpu_device, B, N, D, H, ragged):
    """mha_core.hip (scores of one (doc, head) kept in LDS, N <= 64) against the batched-GEMM + row-softmax path,
    train mode with the same dropout snapshot: same adjacency, same gradients."""
    g = torch.Generator().manual_seed(B * 1000 + N)
    x = torch.randn(B, N, D, generator=g) * 0.5
    n_valid = None
    if ragged:
        n_valid = torch.randint(1, N + 1, (B,), generator=g).to(torch.int32)
        x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
        n_valid = n_valid.to(gpu_device)
    cot = torch.randn(B, H, N, N, generator=g).to(gpu_device)
    mha = gcgcn_amd.MultiHeadAttention(H, D).to(gpu_device).train()
    res = []
    try:
        for core in (1, 0):
            _lib.call("gcgcn_set_option", b"mha_core", core)
            gcgcn_amd.manual_seed(77, gpu_device)
            xs = dev_leaf(x, gpu_device)
            a = torch.stack(mha(xs, None, n_valid=n_valid), 1)
            torch.autograd.backward(a, cot)
            res.append((a.detach(), xs.grad, mha.flat.grad.clone()))
            mha.zero_grad()
    finally:
        _lib.call("gcgcn_set_option", b"mha_core", 1)
    assert (res[0][0] == 0).float().mean() > 0.05          # dropout was on
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,N,D,L,H,ragged", [(3, 64, 128, 2, 4, False), (2, 21, 64, 2, 2, True), (2, 9, 6, 1, 2, False)])
def test_riding_edge_mean_equals_separate_launches(gpu_device, B, N, D, L, H, ragged):
    """The next hop's edge mean / dE broadcast as passengers of the chain launches (chain.hip) against their own
    launches: same outputs, same gradients (D = 6 is not 16-byte aligned: the library falls back to own launches)."""
    sd = O.init_stack_params(D, L, H, seed=11)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=12)
    n_valid = None
    if ragged:
        n_valid = torch.tensor([N, max(1, N // 3)][:B], dtype=torch.int32)
        x = x * (torch.arange(N)[None, :] < n_valid[:, None]).unsqueeze(-1).float()
        n_valid = n_valid.to(gpu_device)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    res = []
    try:
        for ride in (True, False):
            hops.ride_edge_mean = ride
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            out = hops(xs[0], [xs[1], xs[2]], n_valid=n_valid)[-1]
            out.sum().backward()
            res.append((out.detach(), xs[0].grad, xs[1].grad, xs[2].grad, hops.graphcnn[1].flat.grad.clone()))
            hops.zero_grad()
    finally:
        hops.ride_edge_mean = True
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    assert res[0][3].abs().sum() > 0            # dE2 really flowed through the riding broadcast


def test_dropout_draws_are_unbiased_and_uncorrelated(gpu_device):
    """The per-element draw (common.hpp rng_u32): keep rate, neighbour / row / site / step independence on 4M draws."""
    gcgcn_amd.manual_seed(99)
    n, p, W = 1 << 22, 0.2, 2048
    s1, s2 = F_.rng_snapshot(gpu_device), F_.rng_snapshot(gpu_device)        # consecutive steps of one site
    m = [F_.dropout_keep_mask(s, salt, p, n).float() for s, salt in
         ((s1, _lib.SALT_GCN), (s2, _lib.SALT_GCN), (s1, _lib.SALT_GLUE))]
    for k in m:
        assert abs(k.mean().item() - (1 - p)) < 2e-3

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return (a * b).mean().item() / (a.std() * b.std()).item()
    k = m[0]
    assert abs(corr(k[:-1], k[1:])) < 3e-3                 # neighbouring elements
    assert abs(corr(k[:-W], k[W:])) < 3e-3                 # same column, next row
    assert abs(corr(k[:-65536], k[65536:])) < 3e-3         # index bit 16 (the finaliser's first shift)
    assert abs(corr(m[0], m[1])) < 3e-3                    # next step, same site
    assert abs(corr(m[0], m[2])) < 3e-3                    # same step, another site


# ---- the trainer's per-document loss ------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("pair_bce"), ids=ids(golden_files("pair_bce")))
def test_pair_bce_loss_golden(gpu_device, path):
    """HIP loss against the fixtures produced by the trainer's own loop of nn.BCELoss calls (config/Config.py:355-366),
    value and gradient, including saturated sigmoids (scale 40)."""
    r = load_golden(path)["raw"]
    logits = dev_leaf(torch.from_numpy(r["logits"]), gpu_device)
    labels = torch.from_numpy(r["labels"]).to(gpu_device)
    loss = gcgcn_amd.pair_bce_loss(logits, labels)
    loss.backward()
    torch.testing.assert_close(loss.cpu(), torch.from_numpy(r["loss"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(logits.grad.cpu(), torch.from_numpy(r["dlogits"]), rtol=1e-4, atol=1e-9)


def test_pair_bce_loss_batched_ragged(gpu_device):
    """[B,N,N,R] with n_valid against the oracle per document; upstream gradients per document; full-size linearity."""
    B, N, R = 4, 42, 97
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, N, R, generator=g) * 4
    y = (torch.rand(B, N, N, R, generator=g) < 0.03).float()
    nv = torch.tensor([42, 17, 2, 30], dtype=torch.int32)
    w = torch.tensor([1.0, 0.5, -2.0, 3.0])
    xd = dev_leaf(x, gpu_device)
    loss = gcgcn_amd.pair_bce_loss(xd, y.to(gpu_device), n_valid=nv.to(gpu_device))
    (loss * w.to(gpu_device)).sum().backward()
    for b in range(B):
        torch.testing.assert_close(loss[b].cpu(), O.pair_bce_loss(x[b], y[b], int(nv[b])), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(xd.grad[b].cpu(), w[b] * O.pair_bce_loss_grad(x[b], y[b], int(nv[b])), rtol=1e-4, atol=1e-9)
    one = gcgcn_amd.pair_bce_loss(xd.detach()[:, :1, :1], y.to(gpu_device)[:, :1, :1])     # a single entity: no pairs
    assert torch.isnan(one).all()                                                         # 0 / 0, as in the reference
    with pytest.raises(ValueError):
        gcgcn_amd.pair_bce_loss(xd.detach(), y.to(gpu_device)[:, :, :3])


@pytest.mark.parametrize("alpha", [1.0, 0.7])
def test_three_hops_and_alpha_mix_match_oracle(gpu_device, alpha):
    """graph_hop = 3 (two MAGGC hops: the second rides its edge mean in the first's chain launches) and the
    alpha-mix of the hop glue (glove:339; alpha != 1 takes the unfused dropout path), eval mode, against the oracle."""
    B, N, D, L, H, hop = 2, 12, 32, 2, 4, 3
    sd = O.init_stack_params(D, L, H, hops=hop, seed=41)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(B, N, D, generator=g) * 2 - 1
    es = [torch.randn(B, N, N, D, generator=g) * 0.5 for _ in range(hop)]
    hops = gcgcn_amd.GraphHops(D, L, H, graph_hop=hop, alpha=alpha).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    xs = dev_leaf(x, gpu_device)
    eds = [dev_leaf(e, gpu_device) for e in es]
    feats = hops(xs, eds)
    feats[-1].sum().backward()
    for b in range(B):
        xr = x[b].clone().requires_grad_()
        er = [e[b].clone().requires_grad_() for e in es]
        ref = O.hop_stack(xr, er, None, sd, L, H, alpha=alpha)
        ref[-1].sum().backward()
        for k in range(hop + 1):
            torch.testing.assert_close(feats[k][b].detach().cpu(), ref[k].detach(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(xs.grad[b].cpu(), xr.grad, rtol=1e-3, atol=1e-4)
        for k in range(hop):
            torch.testing.assert_close(eds[k].grad[b].cpu(), er[k].grad, rtol=1e-3, atol=1e-5)


def test_deferred_weight_gradients_equal_immediate_ones(gpu_device):
    """MAGGC's weight gradients parked and carried by GATAttention's edge pass (or flushed at the end of backward when
    there is no such pass) against launching them inside the block's own backward: same numbers up to summation order."""
    B, N, D, L, H = 3, 64, 128, 2, 4
    sd = O.init_stack_params(D, L, H, seed=51)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=52)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    res = []
    try:
        for defer in (True, False):
            F_.defer_weight_grads = defer
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            out = hops(xs[0], [xs[1], xs[2]])[-1]
            out.sum().backward()
            assert not F_._passes                       # the pass object lives exactly as long as its backward
            res.append([xs[0].grad, xs[1].grad] + [p.grad.clone() for p in hops.parameters() if p.grad is not None])
            hops.zero_grad()
        # a MAGGC block on its own: nothing carries the parked products, the end-of-backward callback launches them
        conv = hops.graphcnn[1]
        a = torch.softmax(torch.randn(B, H, N, N, generator=torch.Generator().manual_seed(5)), -1).to(gpu_device)
        for defer in (True, False):
            F_.defer_weight_grads = defer
            xs = dev_leaf(x, gpu_device)
            conv(xs, e2.to(gpu_device), a).sum().backward()
            assert not F_._passes
            res[0 if defer else 1].append(conv.flat.grad.clone())
            conv.zero_grad()
        # gradient accumulation: the second backward finds .grad set; the end-of-backward hand-over adds to it
        F_.defer_weight_grads = True
        xs = dev_leaf(x, gpu_device)
        conv(xs, e2.to(gpu_device), a).sum().backward()
        conv(xs, e2.to(gpu_device), a).sum().backward()
        torch.testing.assert_close(conv.flat.grad, 2.0 * res[1][-1], rtol=2e-5, atol=2e-5)
        conv.zero_grad()
    finally:
        F_.defer_weight_grads = True
    assert len(res[0]) == len(res[1])
    for a_, b_ in zip(*res):
        torch.testing.assert_close(a_, b_, rtol=2e-5, atol=2e-5)


# ---- API width: what the reference's constructors accept beyond the one shape its model builds -------------------
@pytest.mark.parametrize("B,N,Din,Dh", [(2, 9, 12, 20), (1, 64, 256, 128), (3, 17, 40, 8)])
def test_gat_rectangular_projection(gpu_device, B, N, Din, Dh):
    """GATAttention(att_input_dim, hidden_dim) with att_input_dim != hidden_dim (nn.Linear(att_input_dim, hidden_dim),
    glove:148-150): output and all gradients against the op-for-op oracle."""
    g = torch.Generator().manual_seed(Din * 31 + Dh)
    m = gcgcn_amd.GATAttention(Din, Dh).to(gpu_device).eval()
    sd = {k: v.cpu().clone().requires_grad_() for k, v in m.state_dict().items()}
    x = torch.rand(B, N, Din, generator=g) * 2 - 1
    e = torch.randn(B, N, N, Din, generator=g) * 0.5
    cot = torch.randn(B, N, N, generator=g)
    xg, eg = dev_leaf(x, gpu_device), dev_leaf(e, gpu_device)
    a = m(xg, eg)
    a.backward(cot.to(gpu_device))
    xr, er = x.clone().requires_grad_(), e.clone().requires_grad_()
    ref = torch.stack([O.gat_attention(xr[b], er[b], sd) for b in range(B)])
    ref.backward(cot)
    close(a, ref, "A")
    close(xg.grad, xr.grad, "dX")
    close(eg.grad, er.grad, "dE")
    for k, gk in m.named_grads().items():
        torch.testing.assert_close(gk.cpu(), sd[k].grad, rtol=1e-3, atol=1e-4, msg=lambda s: f"grad {k}: {s}")


@pytest.mark.parametrize("B,N,D", [(2, 16, 32), (2, 64, 256)])
def test_gat_masked_opt_in(gpu_device, B, N, D):
    """apply_mask=True: the paper-faithful partially connected adjacency (energy -100000 where mask, i.e. what the
    reference's discarded masked_fill, glove:163-164, would have done in place).  Default stays the reference's no-op;
    a fully masked row falls back to the uniform distribution like torch's softmax over equal energies."""
    sd = O.sub(O.init_stack_params(D, 2, 4, seed=3), "get_weighted_adj_matrix")
    x, e1, _, adj = O.synth_docs(B, N, D, seed=4)
    adj[0, 3] = 0                                          # a fully masked row
    mask = torch.eq(adj, 0)
    outs = {}
    for flag in (True, False):
        m = gcgcn_amd.GATAttention(D, D, apply_mask=flag).to(gpu_device).eval()
        m.load_state_dict(sd)
        xg, eg = dev_leaf(x, gpu_device), dev_leaf(e1, gpu_device)
        a = m(xg, eg, mask.to(gpu_device))
        a.square().sum().backward()
        xr, er = x.clone().requires_grad_(), e1.clone().requires_grad_()
        sdl = {k: v.clone().requires_grad_() for k, v in sd.items()}
        ref = torch.stack([O.gat_attention(xr[b], er[b], sdl, mask[b], apply_mask=flag) for b in range(B)])
        ref.square().sum().backward()
        close(a, ref, f"A (apply_mask={flag})")
        close(xg.grad, xr.grad, "dX")
        close(eg.grad, er.grad, "dE")
        for k, gk in m.named_grads().items():
            torch.testing.assert_close(gk.cpu(), sdl[k].grad, rtol=1e-3, atol=1e-4)
        outs[flag] = a.detach()
    part = (mask & ~mask.all(-1, keepdim=True)).to(gpu_device)
    assert part.any() and (outs[True][part] == 0).all()                         # masked pairs carry no weight ...
    torch.testing.assert_close(outs[True][0, 3], torch.full((N,), 1.0 / N, device=gpu_device))   # ... unless all are masked
    assert not torch.allclose(outs[True], outs[False])
    # through the hop loop: GraphHops(apply_mask=True) builds mask = eq(adj, 0) itself (glove:330)
    full = O.init_stack_params(D, 2, 4, seed=3)
    hops = gcgcn_amd.GraphHops(D, 2, 4, apply_mask=True).to(gpu_device).eval()
    hops.load_state_dict(full, strict=True)
    _, _, e2, _ = O.synth_docs(B, N, D, seed=4)
    f = hops(x.to(gpu_device), [e1.to(gpu_device), e2.to(gpu_device)], adj.to(gpu_device))
    a_ref = O.gat_attention(x[1], e1[1], sd, mask[1], apply_mask=True)
    x1_ref = O.graph_convolution(x[1], e1[1], a_ref, O.sub(full, "graphcnn.0"), 2)
    close(f[1][1], x1_ref, "x1 through the masked hop")


def test_many_sublayers(gpu_device):
    """layer_num = 16: more weight-gradient products than one group launch describes (the dWd loop used to stop
    silently at 16 problems when nothing was parked) -- with and without deferral, against the oracle."""
    B, N, D, L, H = 2, 10, 64, 16, 2
    sd = O.init_stack_params(D, L, H, seed=81)
    x, e1, e2, adj = O.synth_docs(B, N, D, seed=82)
    hops = gcgcn_amd.GraphHops(D, L, H).to(gpu_device).eval()
    hops.load_state_dict(sd, strict=True)
    outs, gx, ge1, ge2, sdl = _oracle_stack(x, e1, e2, adj, sd, L, H)
    sum(o[2].sum() for o in outs).backward()
    try:
        for defer in (False, True):
            F_.defer_weight_grads = defer
            hops.zero_grad()
            xs = [dev_leaf(t, gpu_device) for t in (x, e1, e2)]
            f = hops(xs[0], [xs[1], xs[2]])
            f[2].sum().backward()
            for b in range(B):
                close(f[2][b], outs[b][2], f"x2[{b}]")
                close(xs[0].grad[b], gx[b].grad, f"dX[{b}]")
            _check_stack_param_grads(hops, sdl)
    finally:
        F_.defer_weight_grads = True


@pytest.mark.parametrize("N,K,a_kc_b", [(128, 256, 1), (64, 32, 0), (40, 24, 1)])
@pytest.mark.parametrize("count,cap", [(0, 64), (1, 64), (63, 200), (64, 64), (65, 200), (1000, 1500), (1500, 1500)])
def test_gemm_device_side_rows(gpu_device, N, K, a_kc_b, count, cap):
    """gemm_dyn, dyn = 1: C[:count] = A[:count] W^T + bias with the row count in device memory (persistent tile loop); rows
    beyond roundup64(count) are not touched; interior and ragged widths."""
    g = torch.Generator().manual_seed(count * 7 + N)
    R = (cap + 63) // 64 * 64
    A = torch.randn(R, K, generator=g)
    A[count:] = 0
    W = torch.randn(N, K, generator=g) if a_kc_b else torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    C = torch.full((R, N), 7.0, device=gpu_device)
    cnt = torch.tensor([count], dtype=torch.int32, device=gpu_device)
    Ad, Wd, bd = A.to(gpu_device), W.to(gpu_device), bias.to(gpu_device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("gcgcn_gemm_dyn", 0, N, K, p(Ad), K, 1, p(Wd), K if a_kc_b else N, a_kc_b, p(C), N, p(bd), 0, p(cnt), 1, cap, None, 0, None)
    ref = (A[:count].double() @ (W.t() if a_kc_b else W).double() + bias.double()).float()
    torch.testing.assert_close(C[:count].cpu(), ref, rtol=1e-4, atol=1e-4)
    hi = (count + 63) // 64 * 64
    assert (C[hi:] == 7.0).all()


@pytest.mark.parametrize("M,N", [(128, 256), (64, 64), (24, 40)])
@pytest.mark.parametrize("count,cap", [(0, 64), (1, 64), (63, 200), (65, 4096), (3000, 4096), (20000, 20000)])
def test_gemm_device_side_reduction(gpu_device, M, N, count, cap):
    """gemm_dyn, dyn = 2: dW = dY[:count]^T X[:count] with the reduction length in device memory (split-K chosen from the
    capacity, empty slices write zero partials, count == 0 gives zeros), deterministic."""
    g = torch.Generator().manual_seed(count + M)
    R = (cap + 63) // 64 * 64
    dY, X = torch.randn(R, M, generator=g), torch.randn(R, N, generator=g)
    dY[count:] = 0                                     # the zero tail the producer's kernels guarantee
    ws = torch.empty(64 * M * N + 16, device=gpu_device)
    cnt = torch.tensor([count], dtype=torch.int32, device=gpu_device)
    dYd, Xd = dY.to(gpu_device), X.to(gpu_device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), device=gpu_device)
        _lib.call("gcgcn_gemm_dyn", M, N, 0, p(dYd), M, 0, p(Xd), N, 0, p(C), N, None, 0, p(cnt), 2, cap, p(ws), ws.numel(), None)
        outs.append(C.cpu())
    ref = (dY[:count].double().t() @ X[:count].double()).float()
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=2e-4 * max(1.0, (count / 256) ** 0.5))
    assert torch.equal(outs[0], outs[1])


def test_gat_fold_is_kept_while_parameters_are_unchanged(gpu_device):
    """(u, v, c) depends on the parameters only: the second call with unchanged parameters skips the fold kernel (same
    output), an in-place parameter update (optimiser step) or load_state_dict triggers a new fold -- into a NEW buffer, so
    that a graph built earlier keeps the one it saved."""
    D, N = 64, 12
    torch.manual_seed(0)
    m = gcgcn_amd.GATAttention(D, D).to(gpu_device).eval()
    x, e = torch.randn(2, N, D, device=gpu_device), torch.randn(2, N, N, D, device=gpu_device)
    a1 = m(x, e)
    buf1 = m._uvc[1]
    a2 = m(x, e)
    assert m._uvc[1] is buf1 and torch.equal(a1, a2)                       # cached
    with torch.no_grad():
        m.flat.mul_(1.5)                                                   # what an optimiser step does: version bump
    a3 = m(x, e)
    assert m._uvc[1] is not buf1 and not torch.allclose(a3, a1)            # refolded into a new buffer
    m.cache_fold = False
    ref = m(x, e)
    m.cache_fold = True
    assert torch.equal(a3, ref)
    xs = x.clone().requires_grad_()
    m(xs, e).sum().backward()                                              # gradients flow through a cached fold too
    assert torch.isfinite(xs.grad).all() and m.flat.grad is not None
    sd = {k: v * 2 for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    a4 = m(x, e)
    m.cache_fold = False
    assert torch.equal(a4, m(x, e))
