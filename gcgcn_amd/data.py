"""Packed document format and batched device-side tensorisation.

The reference keeps every document as a pickled dict holding a ``networkx.DiGraph`` (gen_data_extend_graph.py:289-307)
and, every epoch, turns it into dense tensors with pure-Python loops of O(N^2 * S * sentence length) per document
(``Config.from_list_to_tensor``, config/Config.py:162-233), then copies int64 ``[N,N,S,T]`` position matrices to the GPU
(77 MB per document at N = 42, S = 5, T = 512) -- one document per step.

Here a document is a handful of small int32 arrays (tokens, mention spans, one 9-int record per (pair, sentence slot),
edges, positive labels): ``PackedDocs`` holds many documents back to back in one ``.npz`` file.  ``collate`` uploads the
records of a batch (a few KB per document) and ONE kernel launch (``gcgcn_tensorise``, csrc/tensorise.hip) expands them on
the GPU into exactly the tensors the reference's ``forward`` takes, batched and padded: ``adj_matrix``, ``sen_matrix``,
``pos_matrix_h/_t`` (uint8: ids are 0..20), ``node_pos``, ``node_type``, ``node_relative_pos``, ``label_matrix``, plus
``n_valid``.  A batch of one document equals ``from_list_to_tensor``'s output (tests/test_data_gpu.py).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ._lib import call

I32 = np.int32
SLOT_W = 9      # u, v, slot j, sentence start, sentence end, head mention start/end, tail mention start/end


@dataclass
class PackedDoc:
    tokens: np.ndarray      # int32 [T_full, 3]: word id, coreference/position id, NER id ('document', 'document_pos', 'document_ner')
    node_type: np.ndarray   # int32 [N]
    men_ptr: np.ndarray     # int32 [N + 1] mention spans of entity n: mentions[men_ptr[n]:men_ptr[n + 1]]
    mentions: np.ndarray    # int32 [M, 2] (start, end) token positions ('exist_pos')
    slots: np.ndarray       # int32 [E, 9] one record per (edge, sentence slot) ('sentences' / 'position' of the edge)
    edges: np.ndarray       # int32 [E2, 2] (u, v): adj_matrix[u, v] = 1
    labels: np.ndarray      # int32 [L, 3] (h, t, r) with label_matrix[h, t, r] == 1
    n_rel: int
    max_sentence_num: int
    title: str = ""

    @property
    def n(self) -> int:
        return int(self.node_type.shape[0])


def pack_document(item: dict) -> PackedDoc:
    """A reference document dict (keys 'document', 'document_pos', 'document_ner', 'graph' -- a networkx.DiGraph with node
    attributes 'exist_pos' / 'type', edge attributes 'sentences' / 'position', graph attribute 'max_sentence_num' --,
    'label_matrix', optional 'title') -> PackedDoc."""
    g = item["graph"]
    n = len(g.nodes())
    tokens = np.stack([np.asarray(item[k], dtype=I32) for k in ("document", "document_pos", "document_ner")], axis=1)
    node_type = np.zeros(n, I32)
    men_ptr, mentions = [0], []
    for node in range(n):
        node_type[node] = g.nodes[node]["type"][0]
        for p in g.nodes[node]["exist_pos"]:
            mentions.append((int(p[0]), int(p[1])))
        men_ptr.append(len(mentions))
    slots, edges = [], []
    for u, v, e in g.edges(data=True):
        edges.append((u, v))
        for j, (sent, pos) in enumerate(zip(e["sentences"], e["position"])):
            slots.append((u, v, j, sent[0], sent[1], pos[0], pos[1], pos[2], pos[3]))
    lab = np.asarray(item["label_matrix"])
    h, t, r = np.nonzero(lab)
    return PackedDoc(tokens, node_type, np.asarray(men_ptr, I32), np.asarray(mentions, I32).reshape(-1, 2),
                     np.asarray(slots, I32).reshape(-1, SLOT_W), np.asarray(edges, I32).reshape(-1, 2),
                     np.stack([h, t, r], 1).astype(I32), int(lab.shape[-1]), int(g.graph["max_sentence_num"]),
                     str(item.get("title", "")))


_FIELDS = ("tokens", "node_type", "men_ptr", "mentions", "slots", "edges", "labels")


class PackedDocs:
    """Many documents in one file: every field concatenated + one pointer array per field."""

    def __init__(self, docs: Sequence[PackedDoc]):
        self.docs = list(docs)

    def __len__(self):
        return len(self.docs)

    def __getitem__(self, i) -> PackedDoc:
        return self.docs[i]

    def save(self, path: str):
        out: Dict[str, np.ndarray] = {}
        for f in _FIELDS:
            arrs = [getattr(d, f) for d in self.docs]
            out[f] = np.concatenate(arrs, 0) if arrs else np.zeros((0,), I32)
            out[f + "_ptr"] = np.cumsum([0] + [a.shape[0] for a in arrs]).astype(np.int64)
        out["n_rel"] = np.asarray([d.n_rel for d in self.docs], I32)
        out["max_sentence_num"] = np.asarray([d.max_sentence_num for d in self.docs], I32)
        out["titles"] = np.frombuffer(json.dumps([d.title for d in self.docs]).encode(), dtype=np.uint8)
        np.savez(path, **out)

    @classmethod
    def load(cls, path: str) -> "PackedDocs":
        z = np.load(path)
        titles = json.loads(bytes(z["titles"]).decode())
        docs = []
        for i in range(len(titles)):
            f = {k: z[k][z[k + "_ptr"][i]:z[k + "_ptr"][i + 1]] for k in _FIELDS}
            docs.append(PackedDoc(f["tokens"], f["node_type"], f["men_ptr"], f["mentions"], f["slots"], f["edges"], f["labels"],
                                  int(z["n_rel"][i]), int(z["max_sentence_num"][i]), titles[i]))
        return cls(docs)


def collate(docs: Sequence[PackedDoc], device, max_length: int = 512, max_num: int = 5, dis_plus: int = 10,
            pad_nodes_to: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Batch of documents -> the reference forward's inputs on ``device`` (config/Config.py:162-233, batched):
    ``document``, ``document_pos``, ``document_ner`` int64 ``[B,T]`` (0-padded), ``adj_matrix`` float ``[B,N,N]``, ``sen_matrix``
    bool ``[B,N,N,S,T]``, ``pos_matrix_h`` / ``pos_matrix_t`` uint8 ``[B,N,N,S,T]``, ``node_pos`` float ``[B,N,T]``, ``node_type``
    int64 ``[B,N]``, ``node_relative_pos`` int64 ``[B,N,N]``, ``label_matrix`` float ``[B,N,N,R]``, ``n_valid`` int32 ``[B]``,
    ``t_valid`` int32 ``[B]``.  T = min(longest document, max_length), S = min(most sentence slots, max_num), N = most entities."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("collate: gcgcn_amd tensorises on the GPU (no CPU fallback)")
    B = len(docs)
    T = min(max(d.tokens.shape[0] for d in docs), max_length)
    S = max(1, min(max(d.max_sentence_num for d in docs), max_num))
    N = max(max(d.n for d in docs), pad_nodes_to or 0)
    R = docs[0].n_rel
    tok = np.zeros((B, T, 3), I32)
    ntype = np.zeros((B, N), I32)
    nvalid = np.zeros(B, I32)
    tvalid = np.zeros(B, I32)
    first = np.zeros((B, N), I32)
    recs = {"mentions": [], "slots": [], "edges": [], "labels": []}     # each record prefixed with its document index
    men_node = []
    for b, d in enumerate(docs):
        t = min(d.tokens.shape[0], T)
        tok[b, :t] = d.tokens[:t]
        ntype[b, :d.n] = d.node_type
        nvalid[b], tvalid[b] = d.n, t
        first[b, :d.n] = d.mentions[d.men_ptr[:-1], 0]
        counts = np.diff(d.men_ptr)
        men_node.append(np.stack([np.full(d.mentions.shape[0], b, I32), np.repeat(np.arange(d.n, dtype=I32), counts),
                                  np.repeat(counts.astype(I32), counts),
                                  np.arange(d.mentions.shape[0], dtype=I32) - np.repeat(d.men_ptr[:-1], counts)], 1))
        for k in ("slots", "edges", "labels"):
            a = getattr(d, k)
            recs[k].append(np.concatenate([np.full((a.shape[0], 1), b, I32), a], 1))
        recs["mentions"].append(d.mentions)

    def up(a, w):
        a = np.concatenate(a, 0) if a else np.zeros((0, w), I32)
        return torch.from_numpy(np.ascontiguousarray(a.reshape(-1, w).astype(I32))).to(dev, non_blocking=True)
    slots, edges, labels = up(recs["slots"], SLOT_W + 1), up(recs["edges"], 3), up(recs["labels"], 4)
    mentions, mnode = up(recs["mentions"], 2), up(men_node, 4)
    tok_d = torch.from_numpy(tok).to(dev)
    out = {"adj_matrix": torch.zeros(B, N, N, device=dev), "sen_matrix": torch.zeros(B, N, N, S, T, dtype=torch.uint8, device=dev),
           "pos_matrix_h": torch.zeros(B, N, N, S, T, dtype=torch.uint8, device=dev),
           "pos_matrix_t": torch.zeros(B, N, N, S, T, dtype=torch.uint8, device=dev),
           "node_pos": torch.zeros(B, N, T, device=dev), "node_relative_pos": torch.zeros(B, N, N, dtype=torch.int64, device=dev),
           "label_matrix": torch.zeros(B, N, N, R, device=dev)}
    nv_d, first_d = torch.from_numpy(nvalid).to(dev), torch.from_numpy(first).to(dev)
    p = lambda t: None if t.numel() == 0 else t.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    call("gcgcn_tensorise", B, N, S, T, R, dis_plus, slots.shape[0], p(slots), edges.shape[0], p(edges), labels.shape[0], p(labels),
         mentions.shape[0], p(mentions), p(mnode), nv_d.data_ptr(), first_d.data_ptr(), out["adj_matrix"].data_ptr(), out["sen_matrix"].data_ptr(),
         out["pos_matrix_h"].data_ptr(), out["pos_matrix_t"].data_ptr(), out["node_pos"].data_ptr(),
         out["node_relative_pos"].data_ptr(), out["label_matrix"].data_ptr(), stream)
    out["sen_matrix"] = out["sen_matrix"].view(torch.bool)
    out["document"], out["document_pos"], out["document_ner"] = (tok_d[:, :, k].long() for k in range(3))
    out["node_type"] = torch.from_numpy(ntype).to(dev).long()
    out["n_valid"], out["t_valid"] = nv_d, torch.from_numpy(tvalid).to(dev)
    out["_keep"] = (slots, edges, labels, mentions, mnode, first_d)      # the launch reads them asynchronously
    return out
