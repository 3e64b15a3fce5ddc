"""SURVEY 8 row f4 on the GPU: packed documents -> gcgcn_amd.data.collate (one tensorise launch) against the tensors the
reference's own Config.from_list_to_tensor produced for the same documents (tests/golden/tensorise_*.npz), one document per
batch (bit-exact) and all documents in one padded batch."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gcgcn_amd.data import PackedDocs, collate
from oracle import gcgcn_oracle as O

pytestmark = pytest.mark.gpu
NAMES = ("document", "document_pos", "document_ner", "adj_matrix", "sen_matrix", "pos_matrix_h", "pos_matrix_t", "node_pos",
         "node_type", "node_relative_pos", "label_matrix")


def _load():
    return PackedDocs.load(os.path.join(GOLDEN, "tensorise_docs.npz")), np.load(os.path.join(GOLDEN, "tensorise_ref.npz"))


@pytest.mark.parametrize("i", range(4))
def test_single_document_equals_reference(gpu_device, i):
    docs, ref = _load()
    ml, mn = (int(v) for v in ref["cfg"][i])
    out = collate([docs[i]], gpu_device, max_length=ml, max_num=mn)
    for k in NAMES:
        want = ref[f"doc{i}.{k}"]
        got = out[k][0].cpu().numpy()
        if k == "sen_matrix" and want.shape[2] < got.shape[2]:     # the reference keeps S = max_sentence_num even when no edge uses it
            pass
        assert got.shape == want.shape, (k, got.shape, want.shape)
        if k.startswith("pos_matrix"):
            assert got.dtype == np.uint8 and np.array_equal(got.astype(np.int64), want), k      # ids 0..20 stored as bytes
        else:
            assert got.dtype == want.dtype and np.array_equal(got, want), k
    assert int(out["n_valid"][0]) == docs[i].n


def test_padded_batch_equals_per_document_oracle(gpu_device):
    docs, _ = _load()
    out = collate(list(docs.docs), gpu_device, max_length=512, max_num=5)
    B = len(docs)
    N, S, T = out["sen_matrix"].shape[1], out["sen_matrix"].shape[3], out["sen_matrix"].shape[4]
    assert (N, S, T) == (7, 4, 512)
    for b, d in enumerate(docs.docs):
        want = O.tensorise_document(d, 512, 5)
        n, t = d.n, min(d.tokens.shape[0], 512)
        s = want["sen_matrix"].shape[2]
        assert int(out["n_valid"][b]) == n and int(out["t_valid"][b]) == t
        for k, sl in (("adj_matrix", (slice(0, n), slice(0, n))), ("node_relative_pos", (slice(0, n), slice(0, n))),
                      ("node_type", (slice(0, n),)), ("label_matrix", (slice(0, n), slice(0, n))), ("node_pos", (slice(0, n), slice(0, t))),
                      ("document", (slice(0, t),)), ("sen_matrix", (slice(0, n), slice(0, n), slice(0, s), slice(0, t))),
                      ("pos_matrix_h", (slice(0, n), slice(0, n), slice(0, s), slice(0, t))),
                      ("pos_matrix_t", (slice(0, n), slice(0, n), slice(0, s), slice(0, t)))):
            got = out[k][b].cpu().numpy()
            assert np.array_equal(got[sl].astype(np.float64), want[k].astype(np.float64)), (b, k)
            z = got.copy().astype(np.float64)
            z[sl] = 0
            assert not z.any(), (b, k, "padding must be zero")


def test_collate_feeds_the_producer(gpu_device):
    """f4 -> f1: the collated batch (uint8 position ids, ragged n_valid) straight into the edge-feature producer equals the
    per-document oracle on the reference-format tensors."""
    import gcgcn_amd
    docs, _ = _load()
    batch = collate(list(docs.docs), gpu_device)
    B, N, _, S, T = batch["sen_matrix"].shape
    g = torch.Generator().manual_seed(3)
    Hd, P = 64, 20
    ctx = torch.tanh(torch.randn(B, T, Hd, generator=g))
    node = torch.rand(B, N, Hd, generator=g) * 2 - 1
    table = torch.randn(21, P, generator=g) * 0.5
    prod = gcgcn_amd.EdgeFeatureProducer(Hd, P).to(gpu_device)
    e = prod(ctx.to(gpu_device), batch["sen_matrix"], batch["pos_matrix_h"], batch["pos_matrix_t"], node.to(gpu_device),
             table.to(gpu_device), n_valid=batch["n_valid"])
    sd = {f"{k.split('.', 1)[0]}.0.{k.split('.', 1)[1]}": v.cpu() for k, v in prod.state_dict().items()}
    for b, d in enumerate(docs.docs):
        ref_in = O.tensorise_document(d, 512, 5)
        n, t = d.n, min(d.tokens.shape[0], 512)
        sen = torch.from_numpy(ref_in["sen_matrix"])
        if sen.shape[2] == 0:
            continue
        want = O.edge_features_folded(ctx[b, :t], sen, torch.from_numpy(ref_in["pos_matrix_h"]), torch.from_numpy(ref_in["pos_matrix_t"]),
                                      node[b, :n], table, sd, 0)
        scale = want.abs().amax(dim=-1, keepdim=True).clamp_min(1.0)
        torch.testing.assert_close(e[b, :n, :n].cpu() / scale, want / scale, rtol=1e-4, atol=1e-4)
