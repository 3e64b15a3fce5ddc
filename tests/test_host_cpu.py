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
    assert lib.gcgcn_version() == _lib.ABI_VERSION == 7


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


def test_bench_self_launches_its_ranks():
    """``python bench.py --gpus 2`` without a torch.distributed environment starts the two ranks itself (the driver may invoke
    the scaling runs the way it invokes the 1-GPU run).  On a box without GPUs the failure must come from the CHILDREN
    ("needs an MI355X"), not from a launcher hint, and the parent exits non-zero with their status."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the children would run the benchmark")
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert "needs an MI355X" in r.stderr and "launch with" not in r.stderr
    assert r.stdout.strip() == ""                         # no JSON line from a failed run


def test_deferral_is_switched_off_under_ddp():
    """A forward that runs inside torch DistributedDataParallel's forward is recognised (its AccumulateGrad hooks need the
    gradient during backward, which a parked gradient does not give them): functional._under_ddp() is what the blocks ask."""
    import warnings
    import torch.distributed as dist
    from gcgcn_amd import functional as F_
    from torch.nn.parallel import DistributedDataParallel as DDP
    seen = []

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(2))

        def forward(self, x):
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                seen.append((F_._under_ddp(), [str(w.message) for w in rec]))
            return (x * self.w).sum()

    assert F_._under_ddp() is False
    m = Probe()
    m(torch.ones(2))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(so.getsockname()[1])
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        F_._warned_ddp[0] = False
        DDP(m)(torch.ones(2)).backward()
    finally:
        dist.destroy_process_group()
    assert seen[0][0] is False and seen[1][0] is True
    assert any("DistributedDataParallel" in w for w in seen[1][1])
    assert F_._under_ddp() is False
    # a torch build without the private marker: the test fails CLOSED (no parking) once several ranks exist, and stays off alone
    saved = DDP._active_ddp_module
    try:
        del DDP._active_ddp_module
        assert F_._under_ddp() is False                      # no process group: nothing to protect
        real = (dist.is_initialized, dist.get_world_size)
        dist.is_initialized, dist.get_world_size = (lambda: True), (lambda *a, **k: 2)
        try:
            assert F_._under_ddp() is True
        finally:
            dist.is_initialized, dist.get_world_size = real
    finally:
        DDP._active_ddp_module = saved


def test_tail_keys_as_root_and_as_submodule():
    """GraphModelTail's state_dict carries the MODEL's key names (word_attention.{i}.*, dense_layer.*, ...) both as the root
    module and nested in a parent (the documented integration: embeddings / encoder live in the parent), load_state_dict
    accepts them in both positions, strict, and the dict's _metadata survives."""
    torch.manual_seed(0)
    tail = gcgcn_amd.GraphModelTail(hidden_size=16, layer_num=2, head_num=2, dis_size=4, entity_type_size=4, relation_num=5)

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dis_embed = torch.nn.Embedding(21, 4)
            self.tail = gcgcn_amd.GraphModelTail(hidden_size=16, layer_num=2, head_num=2, dis_size=4, entity_type_size=4, relation_num=5)

    root_sd = tail.state_dict()
    assert hasattr(root_sd, "_metadata")
    assert "word_attention.1.attention_sent.weight" in root_sd and "dense_layer.weight" in root_sd
    assert not any(k.startswith(("producers.", "head.")) for k in root_sd)
    par = Parent()
    psd = par.state_dict()
    assert "tail.word_attention.0.attention_all.bias" in psd and "tail.bili_layer_01.weight" in psd and "dis_embed.weight" in psd
    assert not any(".producers." in k or ".head." in k for k in psd)
    assert [k[len("tail."):] for k in psd if k.startswith("tail.")] == list(root_sd)          # same keys, same order
    # round trips: root -> nested, nested -> root, strict
    src = {("tail." + k): v + 1.0 for k, v in root_sd.items()}
    src["dis_embed.weight"] = psd["dis_embed.weight"]
    assert par.load_state_dict(src, strict=True).missing_keys == []
    for k, v in par.tail.state_dict().items():
        torch.testing.assert_close(v, root_sd[k] + 1.0)
    tail.load_state_dict({k[len("tail."):]: v for k, v in par.state_dict().items() if k.startswith("tail.")}, strict=True)
    for k, v in tail.state_dict().items():
        torch.testing.assert_close(v, root_sd[k] + 1.0)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        tail.load_state_dict({**root_sd, "word_attention.7.attention_sent.weight": torch.zeros(1)}, strict=True)


@pytest.mark.parametrize("tiles,others,cohort,pct", [(0, 100, 256, 90), (357, 2176, 256, 0), (1536, 8192, 256, 90), (2208, 8192, 256, 90),
                                                      (2376, 2112, 256, 90), (336, 8192, 128, 85), (224, 2048, 128, 85), (7, 3, 8, 100),
                                                      (1000, 10, 256, 90), (513, 4099, 64, 50), (1, 1, 1, 100)])
def test_tile_passengers_and_rows_each_get_exactly_one_workgroup(tiles, others, cohort, pct):
    """csrc/common.hpp Spread / spread_pick (the function the carrying kernels call, here run on the host): over the
    tiles + others workgroup indices of a launch every tile ordinal and every row ordinal appears exactly once, tiles come in
    cohorts of `cohort` consecutive workgroups, a cohort's first workgroup index is a multiple of 8 whenever the cohort is
    (xcd_remap's low bits), and pct = 0 puts every tile in front."""
    import ctypes
    import numpy as np
    from gcgcn_amd import _lib
    n = tiles + others
    kind = np.full(n, -1, np.int32)
    ordinal = np.full(n, -1, np.int32)
    _lib.call("gcgcn_debug_spread", tiles, others, cohort, pct, kind.ctypes.data_as(ctypes.c_void_p), ordinal.ctypes.data_as(ctypes.c_void_p))
    t_idx = np.flatnonzero(kind == 1)
    r_idx = np.flatnonzero(kind == 0)
    assert len(t_idx) == tiles and len(r_idx) == others
    assert np.array_equal(ordinal[t_idx], np.arange(tiles))            # in order, each exactly once
    assert np.array_equal(ordinal[r_idx], np.arange(others))
    if pct == 0 and tiles:
        assert t_idx[-1] == tiles - 1
    for c0 in range(0, tiles, cohort):                                 # a cohort is one run of consecutive workgroups
        run = t_idx[c0:c0 + cohort]
        assert np.array_equal(run, np.arange(run[0], run[0] + len(run)))
        if cohort % 8 == 0:
            assert run[0] % 8 == 0
