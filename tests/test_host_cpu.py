"""CPU-side tests (no GPU): the C ABI library loads and exports every symbol of include/gcgcn.h,
parameter layouts agree between Python and the library, reference checkpoints round-trip through the
flat parameter buffers, and the product path refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, golden_files, load_golden
import gcgcn_amd
from gcgcn_amd import _lib, params as P


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gcgcn.h")).read()
    declared = set(re.findall(r"\b(gcgcn_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in gcgcn.h but missing from libgcgcn_hip.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.gcgcn_version() == _lib.ABI_VERSION == 3


@pytest.mark.parametrize("D,L,H", [(8, 2, 2), (128, 2, 8), (768, 4, 4), (512, 2, 8), (12, 4, 4)])
def test_layouts_match_library(D, L, H):
    assert P.gat_layout(D) == _lib.layout("gat", D, D)
    assert P.gat_layout(D, 2 * D + 3) == _lib.layout("gat", D, 2 * D + 3)          # att_input_dim != hidden_dim
    assert P.mha_layout(D) == _lib.layout("mha", D)
    assert P.gcn_layout(D, L, H) == _lib.layout("gcn", D, L, H)
    assert P.producer_layout(D, 20) == _lib.layout("producer", D, 20)                 # f1: edge-feature producer
    assert P.head_layout(D, L + 1, 20, 12, 97) == _lib.layout("head", D, L + 1, 20, 12, 97)   # f3: classifier head


def test_layout_errors_are_reported_not_fatal():
    with pytest.raises(RuntimeError, match="not divisible"):
        _lib.layout("gcn", 10, 3, 2)


@pytest.mark.parametrize("path", golden_files("stack"))
def test_reference_checkpoint_roundtrip(path):
    g = load_golden(path)
    m = g["meta"]
    hops = gcgcn_amd.GraphHops(m["d"], m["l"], m["h"])
    res = hops.load_state_dict(g["sd"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    out = hops.state_dict()
    assert list(out.keys()) == list(g["sd"].keys())          # same keys, same order as the reference
    for k, v in g["sd"].items():
        assert out[k].shape == v.shape and torch.equal(out[k], v), k
    # parameter count equals the reference's (linears_k included)
    assert sum(p.numel() for p in hops.parameters()) == sum(v.numel() for v in g["sd"].values())


def test_strict_load_reports_missing_and_unexpected():
    hops = gcgcn_amd.GraphHops(8, 2, 2)
    sd = hops.state_dict()
    sd.pop("graphcnn.1.linear_layer.bias")
    sd["graphcnn.1.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError) as ei:
        hops.load_state_dict(sd, strict=True)
    assert "graphcnn.1.linear_layer.bias" in str(ei.value) and "graphcnn.1.bogus" in str(ei.value)
    bad = hops.state_dict()
    bad["graphcnn.0.linear_layer.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        hops.load_state_dict(bad)


def test_partial_checkpoint_loads_what_matches():
    """strict=False with some keys of a block missing: nn.Module semantics (which the reference follows) load the keys
    that are present and report the others."""
    torch.manual_seed(1)
    hops = gcgcn_amd.GraphHops(8, 2, 2)
    before = {k: v.clone() for k, v in hops.state_dict().items()}
    sd = {"graphcnn.1.graphconv.3.weights_node": torch.full((12, 4), 0.5),
          "get_weighted_adj_matrix.wt.bias": torch.tensor([7.0])}
    res = hops.load_state_dict(sd, strict=False)
    after = hops.state_dict()
    assert "graphcnn.1.linear_layer.bias" in res.missing_keys and not res.unexpected_keys
    for k in before:
        want = sd.get(k, before[k])
        assert torch.equal(after[k], want), k


def test_gat_rectangular_projection_state_dict():
    """GATAttention(att_input_dim, hidden_dim) with att_input_dim != hidden_dim: the reference's shapes (glove:148-151)."""
    m = gcgcn_amd.GATAttention(12, 20)
    sd = m.state_dict()
    assert sd["linear_node_h.weight"].shape == (20, 12) and sd["linear_edge_r.bias"].shape == (20,)
    assert sd["wt.weight"].shape == (1, 60)
    ref = torch.nn.Linear(12, 20)
    sd["linear_node_t.weight"] = ref.weight.detach().clone()
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["linear_node_t.weight"], ref.weight.detach())


def test_constructor_contract():
    with pytest.raises(AssertionError):
        gcgcn_amd.MultiHeadAttention(3, 8)                   # glove:125
    with pytest.raises(ValueError):
        gcgcn_amd.GraphConvolution(3, 8, 8)                  # D % L
    m = gcgcn_amd.MultiGraphConvolution(2, 4, 16, 16)
    # bias=True is accepted and ignored like the reference's blocks (glove:53/60, 83/94): same keys, no bias parameter
    assert list(gcgcn_amd.GraphConvolution(2, 16, 16, bias=True).state_dict()) == list(gcgcn_amd.GraphConvolution(2, 16, 16).state_dict())
    assert list(gcgcn_amd.MultiGraphConvolution(2, 4, 16, 16, bias=True).state_dict()) == list(m.state_dict())
    assert m.flat.numel() == sum(math_prod(s) for s in P.gcn_shapes(16, 2, 4).values())


def math_prod(s):
    n = 1
    for v in s:
        n *= v
    return n


def test_init_matches_reference_initialisers():
    """xavier-uniform bounds for weights_*, Linear default bounds elsewhere (glove:32-34)."""
    torch.manual_seed(0)
    D, L, H = 64, 2, 4
    t = gcgcn_amd.MultiGraphConvolution(L, H, D, D).named_tensors()
    gh = D // L
    we = t["graphconv.1.weights_edge"]
    assert we.abs().max() <= (6.0 / (D + gh)) ** 0.5 + 1e-6 and we.std() > 0.5 * (2.0 / (D + gh)) ** 0.5
    wn = t["graphconv.1.weights_node"]
    assert wn.shape == (D + gh, gh) and wn.abs().max() <= (6.0 / (D + 2 * gh)) ** 0.5 + 1e-6
    assert t["linear_layer.weight"].abs().max() <= 1.0 / (H * D) ** 0.5 + 1e-6


def test_no_cpu_fallback():
    hops = gcgcn_amd.GraphHops(8, 2, 2).eval()
    x, e = torch.zeros(3, 8), torch.zeros(3, 3, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hops(x, [e, e])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gcgcn_amd.GATAttention(8, 8)(x, e)


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "gcgcn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("CPU oracle replay", ""), f"{f} mentions the oracle"
