"""Flat parameter buffers of the graph blocks and their mapping to the reference's state_dict.

Each block keeps ONE contiguous fp32 parameter tensor laid out the way the HIP kernels read it
(column-concatenated so that all heads / sub-layers are one GEMM operand); its gradient has the
same layout, so a block's gradient is a single contiguous RCCL all-reduce bucket.  The
reference's parameter names and shapes (GCGCN_glove.py:24-25, 60-61, 94-95, 129-130, 148-151)
are recovered by ``unpack_*`` for ``state_dict()`` and consumed by ``pack_*`` for
``load_state_dict()``, so checkpoints are interchangeable in both directions.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor


# ---- layouts (mirrors of gcgcn_*_layout in csrc/api.hip; checked against the library in tests)
def gat_layout(D: int, Dh: int = None):
    """D = att_input_dim, Dh = hidden_dim (defaults to D, the only shape the reference model builds, glove:254)."""
    Dh = D if Dh is None else Dh
    DD = Dh * D
    o = [0, DD, DD + Dh, 2 * DD + Dh, 2 * DD + 2 * Dh, 3 * DD + 2 * Dh, 3 * DD + 3 * Dh, 3 * DD + 6 * Dh]
    return o + [o[-1] + 1]


def mha_layout(D: int):
    return [0, D * D, D * D + D]


def gcn_layout(D: int, L: int, H: int):
    gh = D // L
    dhd = D * H * D
    wd_head = gh * gh * L * (L - 1) // 2
    o_wd = 2 * dhd
    o_wlin = o_wd + H * wd_head
    o_blin = o_wlin + dhd
    return [0, dhd, o_wd, o_wlin, o_blin, o_blin + D, wd_head]


def wd_offset(D: int, L: int, H: int, h: int, l: int) -> int:
    gh = D // L
    lay = gcn_layout(D, L, H)
    return lay[2] + h * lay[6] + gh * gh * l * (l - 1) // 2


# ---- GATAttention ------------------------------------------------------------------------------
GAT_KEYS = ("linear_node_h.weight", "linear_node_h.bias", "linear_node_t.weight", "linear_node_t.bias",
            "linear_edge_r.weight", "linear_edge_r.bias", "wt.weight", "wt.bias")


def gat_shapes(D: int, Dh: int = None) -> Dict[str, tuple]:
    Dh = D if Dh is None else Dh
    return {"linear_node_h.weight": (Dh, D), "linear_node_h.bias": (Dh,),
            "linear_node_t.weight": (Dh, D), "linear_node_t.bias": (Dh,),
            "linear_edge_r.weight": (Dh, D), "linear_edge_r.bias": (Dh,),
            "wt.weight": (1, 3 * Dh), "wt.bias": (1,)}


def unpack_gat(flat: Tensor, D: int, Dh: int = None) -> Dict[str, Tensor]:
    o = gat_layout(D, Dh)
    shp = gat_shapes(D, Dh)
    return {k: flat[o[i]:o[i + 1]].view(shp[k]) for i, k in enumerate(GAT_KEYS)}


def pack_gat(sd: Dict[str, Tensor], D: int, out: Tensor, Dh: int = None) -> Tensor:
    o = gat_layout(D, Dh)
    for i, k in enumerate(GAT_KEYS):
        out[o[i]:o[i + 1]].copy_(sd[k].reshape(-1))
    return out


# ---- MultiHeadAttention ------------------------------------------------------------------------
def mha_keys(H: int, which: str = "q"):
    ks = []
    for h in range(H):
        ks += [f"linears_{which}.{h}.weight", f"linears_{which}.{h}.bias"]
    return ks


def unpack_mha(flat: Tensor, D: int, H: int, which: str = "q") -> Dict[str, Tensor]:
    dh = D // H
    W = flat[:D * D].view(D, D)
    b = flat[D * D:D * D + D]
    out = {}
    for h in range(H):
        out[f"linears_{which}.{h}.weight"] = W[h * dh:(h + 1) * dh]
        out[f"linears_{which}.{h}.bias"] = b[h * dh:(h + 1) * dh]
    return out


def pack_mha(sd: Dict[str, Tensor], D: int, H: int, out: Tensor, which: str = "q") -> Tensor:
    dh = D // H
    W = out[:D * D].view(D, D)
    b = out[D * D:D * D + D]
    for h in range(H):
        W[h * dh:(h + 1) * dh].copy_(sd[f"linears_{which}.{h}.weight"])
        b[h * dh:(h + 1) * dh].copy_(sd[f"linears_{which}.{h}.bias"])
    return out


# ---- GraphConvolution (H = 1) / MultiGraphConvolution --------------------------------------------
def gcn_keys(L: int, H: int):
    ks = []
    for k in range(H * L):
        ks += [f"graphconv.{k}.weights_edge", f"graphconv.{k}.weights_node"]
    return ks + ["linear_layer.weight", "linear_layer.bias"]


def gcn_shapes(D: int, L: int, H: int) -> Dict[str, tuple]:
    gh = D // L
    s = {}
    for h in range(H):
        for l in range(L):
            k = h * L + l
            s[f"graphconv.{k}.weights_edge"] = (D, gh)
            s[f"graphconv.{k}.weights_node"] = (D + gh * l, gh)
    s["linear_layer.weight"] = (D, H * D)
    s["linear_layer.bias"] = (D,)
    return s


def unpack_gcn(flat: Tensor, D: int, L: int, H: int) -> Dict[str, Tensor]:
    """Reference-named tensors (weights_node needs a cat, so these are copies, not views)."""
    gh = D // L
    lay = gcn_layout(D, L, H)
    HD = H * D
    WnX = flat[lay[0]:lay[1]].view(D, HD)
    We = flat[lay[1]:lay[2]].view(D, HD)
    out = {}
    for h in range(H):
        for l in range(L):
            k = h * L + l
            out[f"graphconv.{k}.weights_edge"] = We[:, k * gh:(k + 1) * gh].clone()
            top = WnX[:, k * gh:(k + 1) * gh]
            if l > 0:
                o = wd_offset(D, L, H, h, l)
                top = torch.cat([top, flat[o:o + l * gh * gh].view(l * gh, gh)], dim=0)
            out[f"graphconv.{k}.weights_node"] = top.clone()
    out["linear_layer.weight"] = flat[lay[3]:lay[4]].view(D, HD).clone()
    out["linear_layer.bias"] = flat[lay[4]:lay[5]].clone()
    return out


def pack_gcn(sd: Dict[str, Tensor], D: int, L: int, H: int, out: Tensor) -> Tensor:
    gh = D // L
    lay = gcn_layout(D, L, H)
    HD = H * D
    WnX = out[lay[0]:lay[1]].view(D, HD)
    We = out[lay[1]:lay[2]].view(D, HD)
    for h in range(H):
        for l in range(L):
            k = h * L + l
            We[:, k * gh:(k + 1) * gh].copy_(sd[f"graphconv.{k}.weights_edge"])
            wn = sd[f"graphconv.{k}.weights_node"]
            WnX[:, k * gh:(k + 1) * gh].copy_(wn[:D])
            if l > 0:
                o = wd_offset(D, L, H, h, l)
                out[o:o + l * gh * gh].view(l * gh, gh).copy_(wn[D:])
    out[lay[3]:lay[4]].view(D, HD).copy_(sd["linear_layer.weight"])
    out[lay[4]:lay[5]].copy_(sd["linear_layer.bias"])
    return out


# ---- reference initialisers ------------------------------------------------------------------------
def _linear_init(weight: Tensor, bias: Tensor):
    """nn.Linear.reset_parameters(): kaiming_uniform(a=sqrt(5)) + U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    torch.nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(weight.shape[1])
    torch.nn.init.uniform_(bias, -bound, bound)


def init_gat(D: int, Dh: int = None) -> Dict[str, Tensor]:
    sd = {k: torch.empty(s) for k, s in gat_shapes(D, Dh).items()}
    for nm in ("linear_node_h", "linear_node_t", "linear_edge_r", "wt"):
        _linear_init(sd[nm + ".weight"], sd[nm + ".bias"])
    return sd


def init_mha(D: int, H: int, which: str = "q") -> Dict[str, Tensor]:
    dh = D // H
    sd = {}
    for h in range(H):
        w, b = torch.empty(dh, D), torch.empty(dh)
        _linear_init(w, b)
        sd[f"linears_{which}.{h}.weight"], sd[f"linears_{which}.{h}.bias"] = w, b
    return sd


def init_gcn(D: int, L: int, H: int) -> Dict[str, Tensor]:
    """xavier_uniform for weights_edge / weights_node (GCGCN_glove.py:32-34), Linear default."""
    sd = {k: torch.empty(s) for k, s in gcn_shapes(D, L, H).items()}
    for k, v in sd.items():
        if k.startswith("graphconv."):
            torch.nn.init.xavier_uniform_(v)
    _linear_init(sd["linear_layer.weight"], sd["linear_layer.bias"])
    return sd


# ---- edge-feature producer (one hop): WordAttention + linear_word_att + SentenceAttention + linear_sentence_att ----
# (GCGCN_glove.py:171-214, constructed at :266-269).  Keys are the model's attribute paths without the hop index.
def producer_shapes(Hd: int, P: int) -> Dict[str, tuple]:
    return {"word_attention.attention_sent.weight": (Hd, Hd), "word_attention.attention_sent.bias": (Hd,),
            "word_attention.attention_pos.weight": (Hd, P), "word_attention.attention_pos.bias": (Hd,),
            "word_attention.attention_all.weight": (1, Hd), "word_attention.attention_all.bias": (1,),
            "linear_word_att.weight": (Hd, 2 * Hd), "linear_word_att.bias": (Hd,),
            "sentence_attention.attention_sent.weight": (Hd, Hd), "sentence_attention.attention_sent.bias": (Hd,),
            "sentence_attention.attention_pos.weight": (Hd, Hd), "sentence_attention.attention_pos.bias": (Hd,),
            "sentence_attention.attention_all.weight": (1, Hd), "sentence_attention.attention_all.bias": (1,),
            "linear_sentence_att.weight": (Hd, 2 * Hd), "linear_sentence_att.bias": (Hd,)}


def producer_layout(Hd: int, P: int):
    o, at = [], 0
    for shp in producer_shapes(Hd, P).values():
        o.append(at)
        n = 1
        for v in shp:
            n *= v
        at += (n + 3) & ~3                     # every piece starts 16-byte aligned (csrc/producer.hip prod_layout)
    return o + [at]


def _numel(shp):
    n = 1
    for v in shp:
        n *= v
    return n


def unpack_producer(flat: Tensor, Hd: int, P: int) -> Dict[str, Tensor]:
    o = producer_layout(Hd, P)
    return {k: flat[o[i]:o[i] + _numel(shp)].view(shp) for i, (k, shp) in enumerate(producer_shapes(Hd, P).items())}


def pack_producer(sd: Dict[str, Tensor], Hd: int, P: int, out: Tensor) -> Tensor:
    o = producer_layout(Hd, P)
    for i, (k, shp) in enumerate(producer_shapes(Hd, P).items()):
        out[o[i]:o[i] + _numel(shp)].copy_(sd[k].reshape(-1))
    return out


def init_producer(Hd: int, P: int) -> Dict[str, Tensor]:
    sd = {k: torch.empty(s) for k, s in producer_shapes(Hd, P).items()}
    for k in list(sd):
        if k.endswith(".weight"):
            _linear_init(sd[k], sd[k[:-6] + "bias"])
    return sd


# ---- classifier head: dense_layer + bili_layer_01 + classification_layer_01 (GCGCN_glove.py:271-276) ----
def head_shapes(Hd: int, nf: int, Pt: int, Pr: int, R: int, HW: int = 128) -> Dict[str, tuple]:
    return {"dense_layer.weight": (HW, Hd * nf + Pt + Pr), "dense_layer.bias": (HW,),
            "classification_layer_01.weight": (R, 2 * HW), "classification_layer_01.bias": (R,),
            "bili_layer_01.bias": (R,), "bili_layer_01.weight": (R, HW, HW)}


HEAD_STATE_ORDER = ("dense_layer.weight", "dense_layer.bias", "bili_layer_01.weight", "bili_layer_01.bias",
                    "classification_layer_01.weight", "classification_layer_01.bias")     # the model's own order


def head_layout(Hd: int, nf: int, Pt: int, Pr: int, R: int):
    o, at = [], 0
    for shp in head_shapes(Hd, nf, Pt, Pr, R).values():
        o.append(at)
        n = 1
        for v in shp:
            n *= v
        at += (n + 3) & ~3                     # every piece starts 16-byte aligned (csrc/head.hip head_layout)
    return o + [at]


def unpack_head(flat: Tensor, Hd: int, nf: int, Pt: int, Pr: int, R: int) -> Dict[str, Tensor]:
    o = head_layout(Hd, nf, Pt, Pr, R)
    shapes = head_shapes(Hd, nf, Pt, Pr, R)
    got = {}
    for i, (k, shp) in enumerate(shapes.items()):
        n = 1
        for v in shp:
            n *= v
        got[k] = flat[o[i]:o[i] + n].view(shp)
    return {k: got[k] for k in HEAD_STATE_ORDER}


def pack_head(sd: Dict[str, Tensor], Hd: int, nf: int, Pt: int, Pr: int, R: int, out: Tensor) -> Tensor:
    o = head_layout(Hd, nf, Pt, Pr, R)
    for i, k in enumerate(head_shapes(Hd, nf, Pt, Pr, R)):
        v = sd[k].reshape(-1)
        out[o[i]:o[i] + v.numel()].copy_(v)
    return out


def init_head(Hd: int, nf: int, Pt: int, Pr: int, R: int, HW: int = 128) -> Dict[str, Tensor]:
    sd = {k: torch.empty(s) for k, s in head_shapes(Hd, nf, Pt, Pr, R).items()}
    _linear_init(sd["dense_layer.weight"], sd["dense_layer.bias"])
    _linear_init(sd["classification_layer_01.weight"], sd["classification_layer_01.bias"])
    bound = 1.0 / math.sqrt(HW)                 # nn.Bilinear.reset_parameters: U(-1/sqrt(in1_features), ...)
    torch.nn.init.uniform_(sd["bili_layer_01.weight"], -bound, bound)
    torch.nn.init.uniform_(sd["bili_layer_01.bias"], -bound, bound)
    return sd
