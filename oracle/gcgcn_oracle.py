"""CPU oracle for the CAGGC + MAGGC hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch fp32 ops on the host, op-for-op in the
reference's order: un-folded einsum+mean, three [N,N,D] GAT linears, per-head /
per-layer loops, one document per call) of

    /root/reference/models/GCGCN_glove.py:18-168   (the five graph blocks)
    /root/reference/models/GCGCN_glove.py:329-341  (the hop-loop glue)

It is NOT part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / the timed
CPU baseline.  ``gcgcn_amd`` never imports it and has no CPU fallback.

Parity pin: ``oracle/make_golden.py`` imports the reference's own classes (by file
path, in the build container only), runs them on seeded inputs and commits
inputs/outputs/gradients as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (<= 1e-5).  The reference has no
tests or golden vectors of its own.

State dicts use the reference's parameter names, so a reference checkpoint's
sub-dict feeds these functions directly.  Dropout is expressed through explicit
keep-masks (already scaled or not, see ``_drop``) so that train-mode results can be
compared with the HIP path bit-for-bit in the mask.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def _relu(pre: Tensor, forced: Optional[Tensor], trace: Optional[list]) -> Tensor:
    """torch.relu, with two test aids.  ``trace``: the pre-activation is appended (to see how close to zero a disputed
    element sits).  ``forced``: a boolean mask that REPLACES ``pre > 0`` -- the HIP path's own relu decisions (its saved
    ``Y > 0``).  Two correct fp32 evaluations in different summation orders can put a pre-activation of a few ulp on
    different sides of zero; one such element shifts a whole weight-gradient column, so full-batch parameter gradients are
    compared with the decisions replayed (like the dropout keep-masks) and the disagreements counted separately."""
    if trace is not None:
        trace.append(pre.detach())
    if forced is None:
        return torch.relu(pre)
    return pre * forced.to(pre.dtype)


def _drop(t: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout(p) in train mode with an explicit boolean keep-mask.

    keep is None  -> eval mode (identity), as nn.Dropout under .eval().
    """
    if keep is None:
        return t
    return t * keep.to(t.dtype) / (1.0 - p)


# --------------------------------------------------------------------------------------
# GraphConv.forward                                          GCGCN_glove.py:36-50
# --------------------------------------------------------------------------------------
def graph_conv(x: Tensor, e: Tensor, adj: Tensor, w_edge: Tensor, w_node: Tensor) -> Tensor:
    """x [N,Din], e [N,N,D], adj [N,N], w_edge [D,gh], w_node [Din,gh] -> [N,gh]."""
    edge = torch.einsum("ijk,kp->ijp", e, w_edge)            # :40
    edge = edge.mean(dim=1)                                  # :41
    # :42 chain_matmul(adj, x, w_node): multi_dot's cost model always picks adj·(x·w_node)
    # here because gh <= Din (N·N·gh <= N·N·Din).
    node = adj @ (x @ w_node)
    out = edge + node                                        # :43
    r = adj.sum(1)                                           # :47
    r = r + (r == 0).to(r.dtype)                             # :48-49
    return out / r.unsqueeze(1)                              # :50


# --------------------------------------------------------------------------------------
# GraphConvolution.forward  (CAGGC conv)                     GCGCN_glove.py:63-80
# --------------------------------------------------------------------------------------
def graph_convolution(x: Tensor, e: Tensor, adj: Tensor, sd: Params, layer_num: int,
                      keep: Optional[Sequence[Tensor]] = None, p: float = 0.2,
                      relu_masks: Optional[Sequence[Tensor]] = None, trace: Optional[list] = None) -> Tensor:
    """sd keys: graphconv.{l}.weights_edge / weights_node, linear_layer.weight / bias.
    relu_masks[l] / trace: see _relu (test aids; default = the reference's relu)."""
    cache = [x]
    outs = []
    cur = x
    for l in range(layer_num):
        y = _relu(graph_conv(cur, e, adj, sd[f"graphconv.{l}.weights_edge"],
                             sd[f"graphconv.{l}.weights_node"]),
                  None if relu_masks is None else relu_masks[l], trace)         # :71
        cache.append(y)
        cur = torch.cat(cache, dim=-1)                                         # :73
        outs.append(_drop(y, None if keep is None else keep[l], p))            # :74
    h = torch.cat(outs, dim=-1) + x                                            # :75-76
    return h @ sd["linear_layer.weight"].t() + sd["linear_layer.bias"]         # :78


# --------------------------------------------------------------------------------------
# MultiGraphConvolution.forward  (MAGGC conv)                GCGCN_glove.py:97-120
# --------------------------------------------------------------------------------------
def multi_graph_convolution(x: Tensor, e: Tensor, adj_list: Sequence[Tensor], sd: Params,
                            layer_num: int, head_num: int,
                            keep: Optional[Sequence[Sequence[Tensor]]] = None,
                            p: float = 0.2, relu_masks: Optional[Sequence[Sequence[Tensor]]] = None,
                            trace: Optional[list] = None) -> Tensor:
    """relu_masks[h][l] / trace (appended in (h, l) order): see _relu."""
    heads = []
    for h in range(head_num):
        cache = [x]
        outs = []
        cur = x
        for l in range(layer_num):
            k = h * layer_num + l                                              # :107
            y = _relu(graph_conv(cur, e, adj_list[h], sd[f"graphconv.{k}.weights_edge"],
                                 sd[f"graphconv.{k}.weights_node"]),
                      None if relu_masks is None else relu_masks[h][l], trace)  # :108
            cache.append(y)
            cur = torch.cat(cache, dim=-1)
            outs.append(_drop(y, None if keep is None else keep[h][l], p))     # :111
        heads.append(torch.cat(outs, dim=-1) + x)                              # :112-113
    cat = torch.cat(heads, dim=-1)                                             # :117
    return cat @ sd["linear_layer.weight"].t() + sd["linear_layer.bias"]       # :118


# --------------------------------------------------------------------------------------
# MultiHeadAttention.forward  (MAGGC adjacency)              GCGCN_glove.py:133-142
# --------------------------------------------------------------------------------------
def multi_head_attention(x: Tensor, sd: Params, head_num: int,
                         keep: Optional[Sequence[Tensor]] = None, p: float = 0.1) -> List[Tensor]:
    """Keys use the *query* projection (reference quirk, :136-137); linears_k unused."""
    d = x.shape[1]
    dh = d // head_num
    out = []
    for h in range(head_num):
        w, b = sd[f"linears_q.{h}.weight"], sd[f"linears_q.{h}.bias"]
        q = x @ w.t() + b                                                      # :136
        k = (x @ w.t() + b).t()                                                # :137
        att = torch.softmax((q @ k) / math.sqrt(dh), dim=-1)                   # :138
        out.append(_drop(att, None if keep is None else keep[h], p))           # :139-140
    return out


# --------------------------------------------------------------------------------------
# GATAttention.forward  (CAGGC adjacency)                    GCGCN_glove.py:154-168
# --------------------------------------------------------------------------------------
def gat_attention(x: Tensor, e: Tensor, sd: Params, mask: Optional[Tensor] = None,
                  keep: Optional[Tensor] = None, p: float = 0.1,
                  apply_mask: bool = False) -> Tensor:
    """mask is ignored by default exactly as in the reference (:163-164 drops the
    out-of-place masked_fill result).  apply_mask=True is the paper-faithful opt-in."""
    n = x.shape[0]
    xh = x.unsqueeze(0).expand(n, n, -1)                                       # :156
    xt = x.unsqueeze(0).expand(n, n, -1)                                       # :157 (same view)
    ah = xh @ sd["linear_node_h.weight"].t() + sd["linear_node_h.bias"]        # :159
    at = xt @ sd["linear_node_t.weight"].t() + sd["linear_node_t.bias"]        # :160
    ar = e @ sd["linear_edge_r.weight"].t() + sd["linear_edge_r.bias"]         # :161
    energy = (torch.cat([ah, at, ar], -1) @ sd["wt.weight"].t() + sd["wt.bias"]).squeeze(-1)  # :162
    if apply_mask and mask is not None:
        energy = energy.masked_fill(mask, -100000.0)
    att = torch.softmax(energy, dim=-1)                                        # :165
    return _drop(att, keep, p)                                                 # :166-167


# --------------------------------------------------------------------------------------
# hop-loop glue                                              GCGCN_glove.py:329-341
# --------------------------------------------------------------------------------------
def sub(sd: Params, prefix: str) -> Params:
    """Sub-dict of a state dict: keys under ``prefix.`` with the prefix removed."""
    pre = prefix + "."
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


def hop_stack(x: Tensor, e_list: Sequence[Tensor], adj: Optional[Tensor], sd: Params,
              layer_num: int, head_num: int, alpha: float = 1.0,
              keeps: Optional[dict] = None, p_glue: float = 0.2,
              relus: Optional[dict] = None, trace: Optional[dict] = None) -> List[Tensor]:
    """The model's hop loop restricted to the graph blocks: hop 0 = GAT + CAGGC conv,
    hop i>=1 = MHA + MAGGC conv.  Returns [x0, x1, ..., x_hops] where x_{i+1} is the node
    feature after hop i (post alpha-mix and glue dropout); the model itself records the
    pre-update features (:338), i.e. the first ``hops`` entries of this list.

    sd uses the model's key names: get_weighted_adj_matrix.*, get_adj_matrix.{i-1}.*,
    graphcnn.{i}.*.  keeps (train mode): dict with optional entries 'gat', 'cag' (list L),
'mha.{i}' (list H), 'mag.{i}' (list H of list L), 'glue.{i}'.
    relus (test aid, see _relu): forced relu decisions, 'cag' (list L of bool [N,gh]) and 'mag.{i}' (list H of list L);
    trace: a dict that receives the pre-activations under the same keys (flat lists in (h, l) order).
    """
    keeps = keeps or {}
    relus = relus or {}
    feats = [x]
    for i, e in enumerate(e_list):
        if i < 1:
            mask = None if adj is None else torch.eq(adj, 0)                   # :330
            a = gat_attention(x, e, sub(sd, "get_weighted_adj_matrix"), mask,
                              keep=keeps.get("gat"))                           # :332
            new = graph_convolution(x, e, a, sub(sd, f"graphcnn.{i}"), layer_num,
                                    keep=keeps.get("cag"), relu_masks=relus.get("cag"),
                                    trace=None if trace is None else trace.setdefault("cag", []))  # :333
        else:
            al = multi_head_attention(x, sub(sd, f"get_adj_matrix.{i - 1}"), head_num,
                                      keep=keeps.get(f"mha.{i}"))              # :336
            new = multi_graph_convolution(x, e, al, sub(sd, f"graphcnn.{i}"), layer_num,
                                          head_num, keep=keeps.get(f"mag.{i}"), relu_masks=relus.get(f"mag.{i}"),
                                          trace=None if trace is None else trace.setdefault(f"mag.{i}", []))  # :337
        x = alpha * new + (1 - alpha) * x                                      # :339
        x = _drop(x, keeps.get(f"glue.{i}"), p_glue)                           # :341
        feats.append(x)
    return feats


# --------------------------------------------------------------------------------------
# parameter construction with the reference's initialisers (for synthetic benchmarks)
# --------------------------------------------------------------------------------------
def init_stack_params(d: int, layer_num: int, head_num: int, hops: int = 2,
                      seed: int = 1337) -> Params:
    """Random parameters in the model's key layout, reference initialisers:
    xavier-uniform for weights_edge / weights_node (GCGCN_glove.py:32-34), nn.Linear
    default (kaiming-uniform a=sqrt(5) + uniform bias) elsewhere."""
    g = torch.Generator().manual_seed(seed)
    gh = d // layer_num
    dh = d // head_num
    sd: Params = {}

    def xavier(fi, fo):
        a = math.sqrt(6.0 / (fi + fo))
        return (torch.rand(fi, fo, generator=g) * 2 - 1) * a

    def linear(name, fo, fi):
        bound = 1.0 / math.sqrt(fi)
        sd[name + ".weight"] = (torch.rand(fo, fi, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(fo, generator=g) * 2 - 1) * bound

    for nm in ("linear_node_h", "linear_node_t", "linear_edge_r"):
        linear("get_weighted_adj_matrix." + nm, d, d)
    linear("get_weighted_adj_matrix.wt", 1, 3 * d)
    for i in range(hops):
        if i == 0:
            for l in range(layer_num):
                sd[f"graphcnn.0.graphconv.{l}.weights_edge"] = xavier(d, gh)
                sd[f"graphcnn.0.graphconv.{l}.weights_node"] = xavier(d + gh * l, gh)
            linear("graphcnn.0.linear_layer", d, d)
        else:
            for h in range(head_num):
                linear(f"get_adj_matrix.{i - 1}.linears_q.{h}", dh, d)
            for h in range(head_num):
                linear(f"get_adj_matrix.{i - 1}.linears_k.{h}", dh, d)
            for h in range(head_num):
                for l in range(layer_num):
                    k = h * layer_num + l
                    sd[f"graphcnn.{i}.graphconv.{k}.weights_edge"] = xavier(d, gh)
                    sd[f"graphcnn.{i}.graphconv.{k}.weights_node"] = xavier(d + gh * l, gh)
            linear(f"graphcnn.{i}.linear_layer", d, d * head_num)
    return sd


def synth_docs(b: int, n: int, d: int, seed: int = 1337, sparsity: float = 0.3):
    """Synthetic DocRED-shaped documents: X ~ U(-1,1) [B,N,D];
    E1,E2 ~ N(0,0.5^2) [B,N,N,D], E1 zeroed where adj==0; adj ~ Bernoulli(0.3), zero diag."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, n, d, generator=g) * 2 - 1
    adj = (torch.rand(b, n, n, generator=g) < sparsity).float()
    adj = adj * (1 - torch.eye(n)).unsqueeze(0)
    e1 = torch.randn(b, n, n, d, generator=g) * 0.5 * adj.unsqueeze(-1)
    e2 = torch.randn(b, n, n, d, generator=g) * 0.5
    return x, e1, e2, adj


# --------------------------------------------------------------------------------------
# per-document loss of the training loop          config/Config.py:355-366
# --------------------------------------------------------------------------------------
def pair_bce_loss_loop(logits: Tensor, labels: Tensor) -> Tensor:
    """The loop exactly as the trainer writes it (config/Config.py:355-366): sigmoid, then one
    ``nn.BCELoss(reduction='mean')`` (Config.py:302) per ordered pair h != t, summed and divided by N^2 - N.
    O(N^2) Python -- small N only; it pins :func:`pair_bce_loss`."""
    bce = torch.nn.BCELoss(reduction="mean")
    p = torch.sigmoid(logits)                                                  # :355
    n = labels.shape[0]
    total = torch.zeros(1)
    for h in range(n):                                                         # :358-363
        for t in range(n):
            if h == t:
                continue
            total = total + bce(p[h][t], labels[h][t])
    return (total / (n * n - n)).squeeze(0)                                    # :364


def pair_bce_loss(logits: Tensor, labels: Tensor, n_valid: Optional[int] = None) -> Tensor:
    """Vectorised restatement of the same loss value: mean over relations of the clamped-log BCE (ATen clamps both
    logs at -100), summed over off-diagonal pairs of the first ``n_valid`` entities, divided by n^2 - n.  Use
    :func:`pair_bce_loss_grad` for the gradient (autograd through the clamp gives NaN where ATen's BCE backward does
    not)."""
    n = labels.shape[0] if n_valid is None else int(n_valid)
    with torch.no_grad():
        p = torch.sigmoid(logits[:n, :n])
        y = labels[:n, :n]
        per = -(y * torch.log(p).clamp_min(-100.0) + (1.0 - y) * torch.log(1.0 - p).clamp_min(-100.0)).mean(-1)
        off = 1.0 - torch.eye(n)
        return (per * off).sum() / (n * n - n)


def pair_bce_loss_grad(logits: Tensor, labels: Tensor, n_valid: Optional[int] = None) -> Tensor:
    """d loss / d logits as autograd produces it for sigmoid followed by ATen's binary_cross_entropy:
    dL/dp = (p - y) / max(p (1 - p), 1e-12) / R (ATen's BCE backward), dp/dx = p (1 - p); zero on the diagonal and
    outside the first ``n_valid`` entities."""
    n = labels.shape[0] if n_valid is None else int(n_valid)
    r = logits.shape[-1]
    g = torch.zeros_like(logits)
    p = torch.sigmoid(logits[:n, :n])
    y = labels[:n, :n]
    q = p * (1.0 - p)
    d = (p - y) / q.clamp_min(1e-12) * q / (r * (n * n - n))
    g[:n, :n] = d * (1.0 - torch.eye(n)).unsqueeze(-1)
    return g


# --------------------------------------------------------------------------------------
# SURVEY 8 row f1 (groundwork for the next round): the edge-feature producer
#   WordAttention glove:171-190, SentenceAttention glove:193-214, call sites glove:314-327
# --------------------------------------------------------------------------------------
def word_attention(pad: Tensor, ctx_exp: Tensor, dis_emb: Tensor, sd: Params) -> Tensor:
    """pad [N,N,S,T,1] bool (True = not a token of that sentence), ctx_exp [N,N,S,T,Hd] (the token states broadcast),
    dis_emb [N,N,S,T,P] -> [N,N,S,Hd]."""
    sent = ctx_exp @ sd["attention_sent.weight"].t() + sd["attention_sent.bias"]                  # :178
    dis = dis_emb @ sd["attention_pos.weight"].t() + sd["attention_pos.bias"]                     # :179
    score = torch.tanh(sent + dis) @ sd["attention_all.weight"].t() + sd["attention_all.bias"]   # :182
    score = score.masked_fill(pad.expand_as(score), -100000.0)                                    # :184-185
    att = torch.softmax(score, dim=3)                                                             # :186
    return (att * ctx_exp).sum(dim=3)                                                             # :187


def sentence_attention(pad: Tensor, cwa: Tensor, node_emb: Tensor, sd: Params) -> Tensor:
    """pad [N,N,S,1] bool, cwa [N,N,S,Hd], node_emb [N,N,S,Hd] -> [N,N,Hd].  Reproduces the reference as written:
    the divisor is the number of PADDED sentence slots (``pad.sum``, :205) plus 1e-10, and the weights are relu(score)."""
    sent = cwa @ sd["attention_sent.weight"].t() + sd["attention_sent.bias"]                      # :201
    dis = node_emb @ sd["attention_pos.weight"].t() + sd["attention_pos.bias"]                    # :202
    score = torch.tanh(sent + dis) @ sd["attention_all.weight"].t() + sd["attention_all.bias"]   # :203
    sent_num = pad.sum(dim=2)                                                                     # :205
    score = score.masked_fill(pad, -100000.0)                                                     # :208
    att = torch.relu(score)                                                                       # :211
    return (att * cwa).sum(dim=2) / (sent_num + 1e-10)                                            # :212


def edge_features(ctx: Tensor, sen_matrix: Tensor, pos_h: Tensor, pos_t: Tensor, node_feat: Tensor, dis_table: Tensor,
                  sd: Params, hop: int) -> Tensor:
    """E = context_sent_att [N,N,Hd] of hop ``hop`` exactly as the model's forward builds it (glove:300-327): ctx [T,Hd]
    token states, sen_matrix [N,N,S,T] bool, pos_h / pos_t [N,N,S,T] distance ids, node_feat [N,Hd], dis_table [21,P]
    (``dis_embed.weight``); sd holds ``word_attention.{hop}.*``, ``linear_word_att.{hop}.*``, ``sentence_attention.{hop}.*``,
    ``linear_sentence_att.{hop}.*``."""
    n, _, s, t = sen_matrix.shape
    wpad = ~sen_matrix.unsqueeze(4)                                                               # :302
    ctx_exp = ctx.unsqueeze(0).unsqueeze(0).unsqueeze(0).expand(n, n, s, -1, -1)                  # :303 (ctx is [1,T,Hd] there)
    spad = ~sen_matrix[:, :, :, 0:1]                                                              # :305
    wa = sub(sd, f"word_attention.{hop}")
    cw_h = word_attention(wpad, ctx_exp, dis_table[pos_h], wa)                                    # :307, 317
    cw_t = word_attention(wpad, ctx_exp, dis_table[pos_t], wa)                                    # :308, 318
    lw = sub(sd, f"linear_word_att.{hop}")
    cwa = torch.cat([cw_h, cw_t], 3) @ lw["weight"].t() + lw["bias"]                              # :320-321
    sa = sub(sd, f"sentence_attention.{hop}")
    ne_h = node_feat.unsqueeze(0).unsqueeze(2).expand_as(cwa)                                     # :324
    ne_t = node_feat.unsqueeze(1).unsqueeze(2).expand_as(cwa)                                     # :325
    cs_h = sentence_attention(spad, cwa, ne_h, sa)                                                # :327
    cs_t = sentence_attention(spad, cwa, ne_t, sa)                                                # :328
    ls = sub(sd, f"linear_sentence_att.{hop}")
    return torch.cat([cs_h, cs_t], 2) @ ls["weight"].t() + ls["bias"]                             # :329-330


def edge_features_folded(ctx: Tensor, sen_matrix: Tensor, pos_h: Tensor, pos_t: Tensor, node_feat: Tensor,
                         dis_table: Tensor, sd: Params, hop: int) -> Tensor:
    """The same E without ever materialising [N,N,S,T,Hd]: the word score depends only on (token t,
    distance id k), so it is a [21, T] table gathered by the position matrices; the attended word context is then a
    [N*N*S, T] x [T, Hd] product, and the node terms of the sentence score are per-entity vectors.  This is the
    algorithm a kernel would implement; it is pinned against :func:`edge_features`."""
    n, _, s, t = sen_matrix.shape
    wa = sub(sd, f"word_attention.{hop}")
    sent = ctx @ wa["attention_sent.weight"].t() + wa["attention_sent.bias"]                       # [T,Hd]
    dis = dis_table @ wa["attention_pos.weight"].t() + wa["attention_pos.bias"]                    # [21,Hd]
    table = (torch.tanh(sent.unsqueeze(0) + dis.unsqueeze(1)) @ wa["attention_all.weight"].t()
             + wa["attention_all.bias"]).squeeze(-1)                                              # [21,T]
    tt = torch.arange(t).expand(n, n, s, t)

    def word(pos):
        score = table[pos, tt].masked_fill(~sen_matrix, -100000.0)                                # [N,N,S,T]
        return (torch.softmax(score, dim=3).reshape(-1, t) @ ctx).view(n, n, s, -1)               # [N,N,S,Hd]
    lw = sub(sd, f"linear_word_att.{hop}")
    cwa = torch.cat([word(pos_h), word(pos_t)], 3) @ lw["weight"].t() + lw["bias"]
    sa = sub(sd, f"sentence_attention.{hop}")
    sfeat = cwa @ sa["attention_sent.weight"].t() + sa["attention_sent.bias"]                      # [N,N,S,Hd]
    nterm = node_feat @ sa["attention_pos.weight"].t() + sa["attention_pos.bias"]                  # [N,Hd]
    spad = ~sen_matrix[:, :, :, 0]                                                                # [N,N,S]
    div = spad.sum(dim=2, keepdim=True) + 1e-10                                                   # padded slots, as the reference

    def sentence(node_term):                                                                      # node_term [N,N,1,Hd]
        score = (torch.tanh(sfeat + node_term) @ sa["attention_all.weight"].t() + sa["attention_all.bias"]).squeeze(-1)
        att = torch.relu(score.masked_fill(spad, -100000.0))
        return (att.unsqueeze(-1) * cwa).sum(dim=2) / div
    cs_h = sentence(nterm.view(1, n, 1, -1))                                                       # head side: indexed by j
    cs_t = sentence(nterm.view(n, 1, 1, -1))                                                       # tail side: indexed by i
    ls = sub(sd, f"linear_sentence_att.{hop}")
    return torch.cat([cs_h, cs_t], 2) @ ls["weight"].t() + ls["bias"]


# --------------------------------------------------------------------------------------
# the classifier head                                 GCGCN_glove.py:306-307, 344-358
# --------------------------------------------------------------------------------------
def classifier_head(node_feats: Sequence[Tensor], node_type: Tensor, node_relative_pos: Tensor, sd: Params,
                    dis_plus: int = 10) -> Tensor:
    """node_feats: the model's list ``[nf_0, ..., nf_hop]`` of [N,Hd] tensors (glove:311, 338), node_type Long[N] in 0..6,
    node_relative_pos Long[N,N] in -dis_plus..dis_plus -> logits [N,N,R].  sd keys (model names): ``ner_emb.weight``,
    ``dis_embed.weight``, ``dense_layer.{weight,bias}``, ``bili_layer_01.{weight,bias}``,
    ``classification_layer_01.{weight,bias}``."""
    n = node_type.shape[0]
    rel_h = sd["dis_embed.weight"][dis_plus + node_relative_pos]                              # :306
    rel_t = sd["dis_embed.weight"][dis_plus - node_relative_pos]                              # :307
    feats = torch.cat(list(node_feats), 1)                                                    # :344
    # ner_emb = nn.Embedding(7, entity_type_size, padding_idx=0) (glove:241): row 0 receives no gradient
    feats = torch.cat([feats, torch.nn.functional.embedding(node_type, sd["ner_emb.weight"], padding_idx=0)], 1)   # :345-347
    with_h = torch.cat([feats.unsqueeze(0).expand(n, -1, -1), rel_h], -1)                     # :351
    with_t = torch.cat([feats.unsqueeze(1).expand(-1, n, -1), rel_t], -1)                     # :352
    w, b = sd["dense_layer.weight"], sd["dense_layer.bias"]
    eh = torch.tanh(with_h @ w.t() + b)                                                       # :354
    et = torch.tanh(with_t @ w.t() + b)                                                       # :355
    ef = torch.cat([eh, et], -1)                                                              # :356
    bil = torch.einsum("ija,rab,ijb->ijr", eh, sd["bili_layer_01.weight"], et) + sd["bili_layer_01.bias"]
    return bil + ef @ sd["classification_layer_01.weight"].t() + sd["classification_layer_01.bias"]   # :358


# --------------------------------------------------------------------------------------
# host tensorisation of one document               config/Config.py:106-116, 162-233
# --------------------------------------------------------------------------------------
def dis2idx_table():
    """config/Config.py:106-116."""
    import numpy as np
    t = np.zeros(1024, dtype="int64")
    t[1] = 1
    for lo, v in ((2, 2), (4, 3), (8, 4), (16, 5), (32, 6), (64, 7), (128, 8), (256, 9), (512, 10)):
        t[lo:] = v
    return t


def tensorise_document(doc, max_length: int = 512, max_num: int = 5, dis_plus: int = 10):
    """``from_list_to_tensor`` (config/Config.py:162-233) restated on the packed record arrays of one document (fields of
    gcgcn_amd.data.PackedDoc: tokens [T,3], node_type [N], men_ptr [N+1], mentions [M,2], slots [E,9], edges [E2,2],
    labels [L,3], n_rel, max_sentence_num).  Returns a dict of numpy arrays with the reference's names / dtypes / shapes."""
    import numpy as np
    d2i = dis2idx_table()
    tfull, n, s = doc.tokens.shape[0], int(doc.node_type.shape[0]), int(doc.max_sentence_num)
    node_pos = np.zeros((n, tfull))                                                    # :169
    for node in range(n):
        m = doc.mentions[doc.men_ptr[node]:doc.men_ptr[node + 1]]
        for p0, p1 in m:
            node_pos[node, p0:p1] = 1.0 / (p1 - p0)                                    # :173
        node_pos[node, :] *= 1.0 / len(m)                                              # :174
    adj = np.zeros((n, n))
    sen = np.zeros((n, n, s, tfull))
    ph = np.zeros((n, n, s, tfull))
    pt = np.zeros((n, n, s, tfull))
    for u, v in doc.edges:
        adj[u, v] = 1                                                                  # :183

    def signed(k, a, b):                                                               # :190-203
        if k - a < 0:
            return -int(d2i[a - k])
        if k - b > 0:
            return int(d2i[k - b])
        return 0
    for u, v, j, s0, s1, h0, h1, t0, t1 in doc.slots:
        sen[u, v, j, s0:s1] = 1                                                        # :187
        for k in range(s0, s1):
            ph[u, v, j, k] = signed(k, h0, h1) + dis_plus
            pt[u, v, j, k] = signed(k, t0, t1) + dis_plus
    rel = np.zeros((n, n))
    for h in range(n):                                                                 # :206-215
        for t in range(n):
            if h == t:
                continue
            r = int(doc.mentions[doc.men_ptr[h], 0]) - int(doc.mentions[doc.men_ptr[t], 0])
            rel[h, t] = -d2i[-r] if r < 0 else d2i[r]
    lab = np.zeros((n, n, doc.n_rel), dtype="float32")
    for h, t, r in doc.labels:
        lab[h, t, r] = 1
    tok = doc.tokens[:max_length].astype("int64")
    return {"document": tok[:, 0], "document_pos": tok[:, 1], "document_ner": tok[:, 2],                  # :164-166
            "adj_matrix": adj.astype("float32"), "sen_matrix": sen[:, :, :max_num, :max_length].astype(bool),          # :217-218
            "pos_matrix_h": ph[:, :, :max_num, :max_length].astype("int64"), "pos_matrix_t": pt[:, :, :max_num, :max_length].astype("int64"),
            "node_pos": node_pos[:, :max_length].astype("float32"), "node_type": doc.node_type.astype("int64"),
            "node_relative_pos": rel.astype("int64"), "label_matrix": lab}
