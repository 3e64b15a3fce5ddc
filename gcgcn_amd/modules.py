"""Drop-in nn.Modules for GCGCN's graph blocks, running on hand-written HIP kernels (gfx950).

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys/shapes as
``/root/reference/models/GCGCN_glove.py:18-168`` (identical copies in
``GraphCNN_multihead_bert_gate_cls.py:18-172``), so the blocks slot into the reference's hop loop
(glove:329-341) and load/save its checkpoints.  Extensions: every ``forward`` also accepts a
leading batch axis (``[B,N,D]``, ``[B,N,N,D]``, ``[B,N,N]`` / ``[B,H,N,N]``) and an optional
``n_valid[B]`` for ragged batches; without a batch axis behaviour is the reference's.

Each module owns ONE flat fp32 ``nn.Parameter`` (``.flat``) in kernel layout (params.py); the
reference-named tensors exist in ``state_dict()`` / ``load_state_dict()`` / ``named_tensors()``.
Element-wise optimisers (Adam, SGD) on ``.flat`` are equivalent to the reference's per-tensor
optimiser state.  Reference quirks reproduced on purpose: the GAT ``mask`` is a
no-op, GAT head/tail are both column-indexed, MHA keys use the query projection and
``linears_k.*`` never receive a gradient.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
from torch import nn

from . import functional as F_
from . import params as P_

Tensor = torch.Tensor


class _FlatBlock(nn.Module):
    """Common state_dict plumbing: reference keys <-> flat kernel-layout parameters."""

    def _ref_tensors(self) -> Dict[str, Tensor]:  # reference-named (detached) tensors
        raise NotImplementedError

    def _load_ref(self, sd: Dict[str, Tensor]):
        raise NotImplementedError

    def _ref_shapes(self) -> Dict[str, tuple]:
        raise NotImplementedError

    def named_tensors(self) -> Dict[str, Tensor]:
        """Parameters under the reference's names (copies/views of ``.flat``, detached)."""
        with torch.no_grad():
            return self._ref_tensors()

    def named_grads(self) -> Dict[str, Optional[Tensor]]:
        """Gradients under the reference's names (None where the reference leaves grad=None)."""
        raise NotImplementedError

    # nn.Module hooks -------------------------------------------------------------------------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_tensors().items():
            destination[prefix + k] = v.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        shapes = self._ref_shapes()
        got = {}
        for k, shp in shapes.items():
            key = prefix + k
            if key not in state_dict:
                missing_keys.append(key)
                continue
            v = state_dict[key]
            if tuple(v.shape) != tuple(shp):
                error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(v.shape)} from "
                                  f"checkpoint, the shape in current model is {tuple(shp)}.")
                continue
            got[k] = v
        if strict:
            for key in state_dict.keys():
                if key.startswith(prefix) and key[len(prefix):] not in shapes:
                    unexpected_keys.append(key)
        if got:
            with torch.no_grad():
                dev = self.flat.device
                full = {k: v.to(device=dev, dtype=torch.float32) for k, v in got.items()}
                if len(got) < len(shapes):      # partial checkpoint (strict=False): nn.Module loads whatever matches
                    cur = self._ref_tensors()
                    full = {k: full.get(k, cur[k]) for k in shapes}
                self._load_ref(full)


def _batched(x: Tensor, nd: int):
    """-> (tensor with a leading batch axis, had_batch)."""
    if x.dim() == nd:
        return x.unsqueeze(0), False
    if x.dim() == nd + 1:
        return x, True
    raise ValueError(f"expected {nd} or {nd + 1} dims, got shape {tuple(x.shape)}")


# ======================================================================================================
class GATAttention(_FlatBlock):
    """CAGGC adjacency (GCGCN_glove.py:144-168): A = dropout(softmax_j(wt.[W_h x_j; W_t x_j; W_r e_ij])).

    ``mask`` is accepted and ignored exactly like the reference (its masked_fill result is discarded,
    glove:163-164).  ``apply_mask=True`` (extension, off by default) is the paper-faithful variant: masked
    pairs get energy -100000 before the softmax, i.e. the partially connected adjacency the in-place
    ``masked_fill_`` would have produced.  The one pass over ``edge_feat`` also produces ``mean_j edge_feat``
    which is parked for the ``GraphConvolution`` call that follows with the same tensor
    (functional.park_edge_mean).
    """

    def __init__(self, att_input_dim: int, hidden_dim: int, dropout: float = 0.1, apply_mask: bool = False):
        super().__init__()
        self.dim = att_input_dim                 # width of node_feat / edge_feat
        self.hidden_dim = hidden_dim             # rows of the three nn.Linear(att_input_dim, hidden_dim), glove:148-150
        self.apply_mask = bool(apply_mask)
        # keep (u, v, c) while the parameters are unchanged (see forward): True = always, False = never, "infer" (default) =
        # only where nothing trains the parameters between calls (eval mode or under no_grad)
        self.cache_fold = "infer"
        self._uvc = None
        self.p = float(dropout) if dropout is not None else 0.0
        self.flat = nn.Parameter(torch.empty(P_.gat_layout(self.dim, hidden_dim)[-1]))
        with torch.no_grad():
            P_.pack_gat(P_.init_gat(self.dim, hidden_dim), self.dim, self.flat, hidden_dim)

    def _ref_shapes(self):
        return P_.gat_shapes(self.dim, self.hidden_dim)

    def _ref_tensors(self):
        return P_.unpack_gat(self.flat.detach(), self.dim, self.hidden_dim)

    def _load_ref(self, sd):
        P_.pack_gat(sd, self.dim, self.flat, self.hidden_dim)

    def named_grads(self):
        return P_.unpack_gat(self.flat.grad, self.dim, self.hidden_dim) if self.flat.grad is not None else {}

    def invalidate_fold(self):
        """Drop the kept (u, v, c).  The cache is keyed on ``flat``'s version counter, which in-place writes through ``.data``
        (``p.data.copy_(ema)``, ``dist.broadcast(p.data)``, ``p.data.normal_()``) do NOT bump: call this after such a write
        (``load_state_dict``, ``.to()`` / ``.cuda()`` and every optimiser step are seen without it)."""
        self._uvc = None

    def _apply(self, fn, *args, **kwargs):
        self._uvc = None
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._uvc = None
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, node_feat: Tensor, edge_feat: Tensor, mask: Optional[Tensor] = None,
                n_valid: Optional[Tensor] = None, return_input_alias: bool = False) -> Tensor:
        x, batched = _batched(node_feat, 2)
        compact = isinstance(edge_feat, F_.CompactEdges)
        if compact and self.apply_mask:
            raise ValueError("GATAttention(apply_mask=True) needs the dense edge tensor (CompactEdges.dense())")
        e = edge_feat if compact else _batched(edge_feat, 3)[0]
        mk = None
        if self.apply_mask and mask is not None:
            mk, _ = _batched(mask, 2)
        # The folded projection (u, v, c) is a function of the parameters only: kept while they are unchanged (several
        # documents per optimiser step -- the reference's gradient accumulation -- and inference); any in-place update of
        # .flat (optimiser step, load_state_dict) bumps its version counter and the fold runs again.
        # Never inside a hipGraph capture (a replay does not re-run this check), and a refold goes into a NEW buffer (an older
        # autograd graph may still hold the previous one for its backward).  By default ("infer") only in eval mode or under
        # no_grad: a training loop may also write parameters through ``.data`` (EMA swaps, broadcasts), which the version
        # counter does not see -- there the fold simply runs every call (one 4-workgroup kernel); see invalidate_fold().
        uvc, valid = None, False
        use_cache = (not (self.training and torch.is_grad_enabled())) if self.cache_fold == "infer" else bool(self.cache_fold)
        if use_cache and x.is_cuda and not torch.cuda.is_current_stream_capturing():
            key = (self.flat._version, self.flat.data_ptr(), x.device)
            if self._uvc is not None and self._uvc[0] == key:
                uvc, valid = self._uvc[1], True
            else:
                uvc = torch.empty(2 * self.dim + 1, device=x.device)
                self._uvc = (key, uvc)
        if compact:      # the producer's compact rows: the edge pass reads Ec[prow] / the bias, E never exists (csrc/compact.hip)
            a, ebar, xa = F_.gat_attention_compact(x, e, self.flat, n_valid, self.p, self.training, hidden_dim=self.hidden_dim,
                                                   uvc=uvc, uvc_valid=valid)
        else:
            a, ebar, xa = F_.gat_attention(x, e, self.flat, n_valid, self.p, self.training, hidden_dim=self.hidden_dim,
                                           mask=mk, uvc=uvc, uvc_valid=valid)
        F_.park_edge_mean(edge_feat, n_valid, ebar)
        a = a if batched else a.squeeze(0)
        # extension: (A, alias of node_feat).  Feeding the alias to the convolution of the same hop routes the
        # convolution's d(node_feat) through this block's backward, which adds it in its own dX kernel
        return (a, xa if batched else xa.squeeze(0)) if return_input_alias else a


# ======================================================================================================
class MultiHeadAttention(_FlatBlock):
    """MAGGC adjacency (GCGCN_glove.py:122-142): per head softmax(Q_h Q_h^T / sqrt(dh)), keys projected
    with the QUERY weights (glove:136-137).  Returns a list of ``head_num`` matrices like the reference
    (views of one ``[B,H,N,N]`` tensor).  ``linears_k.*`` are kept for checkpoint compatibility and never
    receive a gradient, as in the reference.  The second positional argument (``mask`` in the reference,
    which the model feeds with the edge tensor, glove:336) is ignored.
    """

    def __init__(self, head_num: int, att_size: int, dropout: float = 0.1):
        super().__init__()
        assert att_size % head_num == 0          # glove:125
        self.hidden_size = att_size // head_num
        self.head_num = head_num
        self.dim = att_size
        self.p = float(dropout) if dropout is not None else 0.0
        n = P_.mha_layout(att_size)[-1]
        self.flat = nn.Parameter(torch.empty(n))
        self.flat_k = nn.Parameter(torch.empty(n))   # linears_k.*: allocated, saved, never used (glove:130)
        with torch.no_grad():
            P_.pack_mha(P_.init_mha(att_size, head_num, "q"), att_size, head_num, self.flat, "q")
            P_.pack_mha(P_.init_mha(att_size, head_num, "k"), att_size, head_num, self.flat_k, "k")

    def _ref_shapes(self):
        dh, D = self.hidden_size, self.dim
        s = {}
        for w in ("q", "k"):
            for h in range(self.head_num):
                s[f"linears_{w}.{h}.weight"] = (dh, D)
                s[f"linears_{w}.{h}.bias"] = (dh,)
        return s

    def _ref_tensors(self):
        d = P_.unpack_mha(self.flat.detach(), self.dim, self.head_num, "q")
        d.update(P_.unpack_mha(self.flat_k.detach(), self.dim, self.head_num, "k"))
        return d

    def _load_ref(self, sd):
        P_.pack_mha(sd, self.dim, self.head_num, self.flat, "q")
        P_.pack_mha(sd, self.dim, self.head_num, self.flat_k, "k")

    def named_grads(self):
        if self.flat.grad is None:
            return {}
        return P_.unpack_mha(self.flat.grad, self.dim, self.head_num, "q")

    def forward(self, node_feat: Tensor, mask: Optional[Tensor] = None, n_valid: Optional[Tensor] = None,
                return_input_alias: bool = False, rowblk: Optional[Tensor] = None) -> List[Tensor]:
        x, batched = _batched(node_feat, 2)
        a, xa = F_.multi_head_adjacency(x, self.flat, self.head_num, n_valid, self.p, self.training, rowblk=rowblk)  # [B,H,N,N]
        heads = list(a.unbind(1)) if batched else list(a.squeeze(0).unbind(0))
        return (heads, xa if batched else xa.squeeze(0)) if return_input_alias else heads   # see GATAttention


# ======================================================================================================
class _GcnBase(_FlatBlock):
    def _setup(self, layer_num: int, head_num: int, input_dim: int, output_dim: int, bias: bool):
        # `bias` is accepted and ignored exactly like the reference: its constructors take the argument (glove:53, 83) and
        # build every inner GraphConv without it (glove:60, 94), so no bias parameter ever exists in these blocks
        if input_dim != output_dim:
            raise ValueError("residual connection needs input_dim == output_dim (glove:76)")
        if output_dim % layer_num != 0:
            raise ValueError("output_dim must be divisible by layer_num (glove:58,76)")
        self.input_dim = input_dim
        self.layer_num = layer_num
        self.head_num = head_num
        self.dim = output_dim
        self.p = 0.2                                  # self.gcn_dropout = nn.Dropout(0.2), glove:59/90
        self.flat = nn.Parameter(torch.empty(P_.gcn_layout(self.dim, layer_num, head_num)[5]))
        with torch.no_grad():
            P_.pack_gcn(P_.init_gcn(self.dim, layer_num, head_num), self.dim, layer_num, head_num, self.flat)

    def _ref_shapes(self):
        return P_.gcn_shapes(self.dim, self.layer_num, self.head_num)

    def _ref_tensors(self):
        return P_.unpack_gcn(self.flat.detach(), self.dim, self.layer_num, self.head_num)

    def _load_ref(self, sd):
        P_.pack_gcn(sd, self.dim, self.layer_num, self.head_num, self.flat)

    def named_grads(self):
        if self.flat.grad is None:
            return {}
        return P_.unpack_gcn(self.flat.grad, self.dim, self.layer_num, self.head_num)

    def _edge_mean(self, edge_feat: Tensor, n_valid) -> Tensor:
        """mean_j E: taken from the GATAttention call that just streamed this tensor, else computed."""
        ebar = F_.take_edge_mean(edge_feat, n_valid)
        if ebar is None:
            e = edge_feat if isinstance(edge_feat, F_.CompactEdges) else _batched(edge_feat, 3)[0]
            ebar = F_.edge_mean(e, n_valid)
        return ebar

    def _stack(self, x, ebar, adj, n_valid, ride_edge, out_dropout, rowblk=None):
        """The fused block.  Extensions used by GraphHops: ``ride_edge`` is the edge tensor the NEXT hop's
        convolution will be called with -- its mean is computed inside this block's chain launch and parked for that
        call (functional.GcnFn); ``out_dropout`` applies the hop's output dropout (glove:341) in the block's last
        epilogue; ``rowblk``: the hop loop's row-block list of a ragged batch (functional.row_blocks; None: built here)."""
        if ride_edge is None:
            return F_.gcn_stack(x, ebar, adj, self.flat, self.layer_num, self.head_num, n_valid, self.p, self.training,
                                out_dropout=out_dropout, rowblk=rowblk)
        if isinstance(ride_edge, F_.CompactEdges):     # nothing to stream: its mean is a segmented sum, computed when it is asked for
            return F_.gcn_stack(x, ebar, adj, self.flat, self.layer_num, self.head_num, n_valid, self.p, self.training,
                                out_dropout=out_dropout, rowblk=rowblk)
        e_next, _ = _batched(ride_edge, 3)
        out, ebar_next = F_.gcn_stack(x, ebar, adj, self.flat, self.layer_num, self.head_num, n_valid, self.p,
                                      self.training, e_next=e_next, out_dropout=out_dropout, rowblk=rowblk)
        F_.park_edge_mean(ride_edge, n_valid, ebar_next)
        return out


class GraphConv(nn.Module):
    """The leaf layer (GCGCN_glove.py:18-50): ``(mean_j(E) W_e + A X W_n (+ bias)) / rowsum(A)`` with the
    reference's parameter names (``weights_edge``, ``weights_node``, optional ``bias``) and xavier init.
    ``GraphConvolution`` / ``MultiGraphConvolution`` do not instantiate it: they run all their GraphConvs
    fused (params.py); this class exists so that code using the leaf directly keeps working."""

    def __init__(self, input_dim: int, edge_dim: int, output_dim: int, bias: bool = False):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.weights_edge = nn.Parameter(torch.empty(edge_dim, output_dim))
        self.weights_node = nn.Parameter(torch.empty(input_dim, output_dim))
        if bias:
            # the reference leaves this tensor uninitialised (torch.FloatTensor, glove:27); zeros here
            self.bias = nn.Parameter(torch.zeros(output_dim))
        else:
            self.register_parameter("bias", None)
        self.init()

    def init(self):
        nn.init.xavier_uniform_(self.weights_edge.data)          # glove:33
        nn.init.xavier_uniform_(self.weights_node.data)          # glove:34

    def forward(self, inputs: Tensor, edge_inputs: Tensor, adjacency_matrix: Tensor,
                n_valid: Optional[Tensor] = None) -> Tensor:
        x, batched = _batched(inputs, 2)
        adj, _ = _batched(adjacency_matrix, 2)
        ebar = F_.take_edge_mean(edge_inputs, n_valid)
        if ebar is None:
            e, _ = _batched(edge_inputs, 3)
            ebar = F_.edge_mean(e, n_valid)
        out = F_.graph_conv(x, ebar, adj, self.weights_edge, self.weights_node, self.bias)
        return out if batched else out.squeeze(0)


class GraphConvolution(_GcnBase):
    """CAGGC convolution (GCGCN_glove.py:52-80): ``layer_num`` densely connected GraphConv layers over one
    adjacency, dropout 0.2 on the emitted copies, residual, Linear(D, D)."""

    def __init__(self, layer_num: int, input_dim: int, output_dim: int, bias: bool = False):
        super().__init__()
        self._setup(layer_num, 1, input_dim, output_dim, bias)

    def forward(self, node_feat: Tensor, edge_feat: Tensor, adj_matrix: Tensor,
                n_valid: Optional[Tensor] = None, ride_edge: Optional[Tensor] = None,
                out_dropout: float = 0.0, rowblk: Optional[Tensor] = None) -> Tensor:
        x, batched = _batched(node_feat, 2)
        adj, _ = _batched(adj_matrix, 2)
        ebar = self._edge_mean(edge_feat, n_valid)
        out = self._stack(x, ebar, adj.unsqueeze(1), n_valid, ride_edge, out_dropout, rowblk)
        return out if batched else out.squeeze(0)


class MultiGraphConvolution(_GcnBase):
    """MAGGC convolution (GCGCN_glove.py:82-120): ``head_num`` independent dense stacks, one adjacency per
    head, concatenated heads -> Linear(H*D, D)."""

    def __init__(self, layer_num: int, head_num: int, input_dim: int, output_dim: int, bias: bool = False):
        super().__init__()
        self._setup(layer_num, head_num, input_dim, output_dim, bias)

    @staticmethod
    def _stack_heads(adj_list: Union[Tensor, Sequence[Tensor]], batched: bool) -> Tensor:
        if isinstance(adj_list, torch.Tensor):
            return adj_list if adj_list.dim() == 4 else adj_list.unsqueeze(0)
        H, a0 = len(adj_list), adj_list[0]
        base, step = a0._base, a0.shape[-1] * a0.shape[-2] * a0.element_size()
        # the list MultiHeadAttention returns is H views of one [B,H,N,N] buffer: use it without a copy
        if (base is not None and base.dim() == 4 and base.shape[1] == H and base.is_contiguous()
                and all(a._base is base and a.data_ptr() == base.data_ptr() + h * step
                        for h, a in enumerate(adj_list))):
            return base
        stacked = torch.stack(list(adj_list), dim=1 if batched else 0)
        return stacked if batched else stacked.unsqueeze(0)

    def forward_with_attention(self, attention: "MultiHeadAttention", node_feat: Tensor, edge_feat: Tensor,
                               n_valid: Optional[Tensor] = None, ride_edge: Optional[Tensor] = None, out_dropout: float = 0.0,
                               rowblk: Optional[Tensor] = None) -> Tensor:
        """``self(node_feat, edge_feat, attention(node_feat, edge_feat))`` -- the whole MAGGC hop (glove:336-337) -- as one fused
        call where the shape allows it (functional.maggc_fusable): the attention's launches ride inside the convolution's.  Same
        results as the two separate module calls (same kernels' bodies, same dropout draws); two launches fewer per step."""
        x, batched = _batched(node_feat, 2)
        if not (F_.maggc_fusable(x, self.head_num) and attention.head_num == self.head_num and attention.dim == self.dim
                and isinstance(attention, MultiHeadAttention)):
            al, xa = attention(node_feat, edge_feat, n_valid=n_valid, return_input_alias=True, rowblk=rowblk)
            return self(xa, edge_feat, al, n_valid=n_valid, ride_edge=ride_edge, out_dropout=out_dropout, rowblk=rowblk)
        ebar = self._edge_mean(edge_feat, n_valid)
        e_next = None
        if ride_edge is not None and not isinstance(ride_edge, F_.CompactEdges):
            e_next, _ = _batched(ride_edge, 3)
        r = F_.maggc_hop(x, ebar, attention.flat, self.flat, self.layer_num, self.head_num, n_valid, attention.p, self.p,
                         self.training, e_next=e_next, out_dropout=out_dropout, rowblk=rowblk)
        if e_next is not None:
            out, ebar_next = r
            F_.park_edge_mean(ride_edge, n_valid, ebar_next)
        else:
            out = r
        return out if batched else out.squeeze(0)

    def forward(self, node_feat: Tensor, edge_feat: Tensor, adj_matrix_list: Union[Tensor, Sequence[Tensor]],
                n_valid: Optional[Tensor] = None, ride_edge: Optional[Tensor] = None,
                out_dropout: float = 0.0, rowblk: Optional[Tensor] = None) -> Tensor:
        x, batched = _batched(node_feat, 2)
        adj = self._stack_heads(adj_matrix_list, batched)
        ebar = self._edge_mean(edge_feat, n_valid)
        out = self._stack(x, ebar, adj, n_valid, ride_edge, out_dropout, rowblk)
        return out if batched else out.squeeze(0)


# ======================================================================================================
class EdgeFeatureProducer(_FlatBlock):
    """The edge-feature producer of ONE hop: what GCGCN_glove.forward does between glove:300 and
    :330 -- ``word_attention[i]`` on the head and tail distance embeddings, ``linear_word_att[i]``,
    ``sentence_attention[i]`` on the head and tail node embeddings, ``linear_sentence_att[i]`` -- as one call that
    returns ``context_sent_att`` without ever building an ``[N, N, S, T, hidden]`` tensor.

    ``state_dict()`` keys are the model's attribute paths without the hop index (``word_attention.attention_sent.weight``,
    ``linear_word_att.weight``, ``sentence_attention.attention_pos.bias``, ``linear_sentence_att.weight``, ...);
    :meth:`load_model_hop` takes a GCGCN_glove checkpoint and a hop number.  Reference behaviours kept on purpose: a
    sentence slot counts only if token 0 belongs to it (``sent_att_padding_matrix = ~sen_matrix[:,:,:,0:1]``, glove:305),
    the sentence sum is divided by the number of PADDED slots + 1e-10 (glove:205, 212), the attention weights are
    ``relu`` of the scores (glove:211).
    """

    def __init__(self, hidden_size: int = 128, dis_size: int = 20):
        super().__init__()
        self.hidden, self.dis_size = hidden_size, dis_size
        self.flat = nn.Parameter(torch.zeros(P_.producer_layout(hidden_size, dis_size)[-1]))     # (alignment gaps stay zero)
        with torch.no_grad():
            P_.pack_producer(P_.init_producer(hidden_size, dis_size), hidden_size, dis_size, self.flat)

    def _ref_shapes(self):
        return P_.producer_shapes(self.hidden, self.dis_size)

    def _ref_tensors(self):
        return P_.unpack_producer(self.flat.detach(), self.hidden, self.dis_size)

    def _load_ref(self, sd):
        P_.pack_producer(sd, self.hidden, self.dis_size, self.flat)

    def named_grads(self):
        return P_.unpack_producer(self.flat.grad, self.hidden, self.dis_size) if self.flat.grad is not None else {}

    def load_model_hop(self, model_state_dict: Dict[str, Tensor], hop: int, strict: bool = True):
        """Load ``word_attention.{hop}.*``, ``linear_word_att.{hop}.*``, ``sentence_attention.{hop}.*`` and
        ``linear_sentence_att.{hop}.*`` of a GCGCN_glove / GCGCN_bert checkpoint."""
        sd = {}
        for k in self._ref_shapes():
            head, rest = k.split(".", 1)
            src = f"{head}.{hop}.{rest}"
            if src in model_state_dict:
                sd[k] = model_state_dict[src]
        return self.load_state_dict(sd, strict=strict)

    def forward(self, context_output: Tensor, sen_matrix: Tensor, pos_matrix_h: Tensor, pos_matrix_t: Tensor,
                node_feat: Tensor, dis_embed_weight: Tensor, n_valid: Optional[Tensor] = None,
                max_live_slots: Optional[int] = None, max_live_pairs: Optional[int] = None,
                check_capacity: Optional[bool] = None, compact: bool = False):
        """context_output ``[T,H]`` / ``[1,T,H]`` (the reference's shapes, glove:292) or ``[B,T,H]``; sen_matrix / pos_matrix_*
        ``[N,N,S,T]`` or ``[B,N,N,S,T]``; node_feat ``[N,H]`` / ``[B,N,H]``; dis_embed_weight = ``model.dis_embed.weight``.
        Returns ``context_sent_att``: ``[N,N,H]`` or ``[B,N,N,H]``.

        ``max_live_slots`` / ``max_live_pairs`` (capacities given up front: no host synchronisation, capturable): if they are
        too small nothing is computed, every real pair of the result is NaN and ``self.last_counts[2]`` (a device tensor) is 1;
        ``check_capacity`` (default: ``self.check_capacity``, off) reads that flag back and raises ``ProducerCapacityError``.

        ``compact=True``: returns a :class:`gcgcn_amd.functional.CompactEdges` handle instead of the tensor -- the rows of the
        pairs with a live sentence slot, the pair index and the bias every other pair equals; the graph blocks take it wherever
        they take ``edge_feat`` and the ``[N,N,hidden]`` tensor (and its gradient) is never written."""
        batched = sen_matrix.dim() == 5
        tok = context_output
        if tok.dim() == 2:
            tok = tok.unsqueeze(0)
        if not batched:
            sen_matrix, pos_matrix_h, pos_matrix_t = (t.unsqueeze(0) for t in (sen_matrix, pos_matrix_h, pos_matrix_t))
            node_feat = node_feat.unsqueeze(0)
        chk = self.check_capacity if check_capacity is None else check_capacity
        e, self.last_counts = F_.edge_features(tok, sen_matrix, pos_matrix_h, pos_matrix_t, node_feat, dis_embed_weight, self.flat,
                                               n_valid, max_live_slots, max_live_pairs, check_capacity=chk, return_counts=True,
                                               compact=compact)
        if compact:
            o = P_.producer_layout(self.hidden, self.dis_size)[15]          # linear_sentence_att.bias inside flat
            nv = None if n_valid is None else n_valid.to(device=tok.device, dtype=torch.int32)
            return F_.CompactEdges(e[0], e[1], self.flat[o:o + self.hidden], nv, batched)
        return e if batched else e.squeeze(0)

    check_capacity = False      # read the over-capacity flag back after every call with caller-given capacities (one sync)
    last_counts = None          # device int32[4] of the last call: {live slots, live pairs, over capacity, 0}


# ======================================================================================================
class ClassifierHead(_FlatBlock):
    """The pairwise relation classifier (GCGCN_glove.py:271-276 construction, :306-307 and :344-358
    forward): ``dense_layer`` on ``cat(node_feats, type embedding, relative-position embedding)`` for the head and the
    tail entity of every pair, ``tanh``, then ``bili_layer_01(eh, et) + classification_layer_01(cat(eh, et))``.

    ``state_dict()`` keys equal the model's (``dense_layer.*``, ``bili_layer_01.*``, ``classification_layer_01.*``).  The
    two embedding tables stay with the model (``ner_emb`` is also used by the encoder, ``dis_embed`` by the edge-feature
    producer) and are passed to ``forward``.  The hidden width of the pair features is 128 as in the reference (glove:234).
    """

    def __init__(self, hidden_size: int = 128, graph_hop: int = 2, entity_type_size: int = 20, dis_size: int = 20,
                 relation_num: int = 97, dis_plus: int = 10):
        super().__init__()
        self.hidden, self.nf, self.pt, self.pr, self.R, self.dis_plus = hidden_size, graph_hop + 1, entity_type_size, dis_size, \
            relation_num, dis_plus
        self.flat = nn.Parameter(torch.zeros(P_.head_layout(*self._dims())[-1]))
        with torch.no_grad():
            P_.pack_head(P_.init_head(*self._dims()), *self._dims(), self.flat)

    def _dims(self):
        return self.hidden, self.nf, self.pt, self.pr, self.R

    def _ref_shapes(self):
        s = P_.head_shapes(*self._dims())
        return {k: s[k] for k in P_.HEAD_STATE_ORDER}

    def _ref_tensors(self):
        return P_.unpack_head(self.flat.detach(), *self._dims())

    def _load_ref(self, sd):
        P_.pack_head(sd, *self._dims(), self.flat)

    def named_grads(self):
        return P_.unpack_head(self.flat.grad, *self._dims()) if self.flat.grad is not None else {}

    def forward(self, node_feats: Sequence[Tensor], node_type: Tensor, node_relative_pos: Tensor, ner_emb_weight: Tensor,
                dis_embed_weight: Tensor, n_valid: Optional[Tensor] = None) -> Tensor:
        """node_feats: the model's list (``graph_hop + 1`` tensors ``[N,H]`` or ``[B,N,H]``, glove:311/338); node_type
        ``Long[N]``; node_relative_pos ``Long[N,N]`` in ``-dis_plus..dis_plus``.  Returns ``relation_before_softmax_01``:
        ``[N,N,R]`` or ``[B,N,N,R]``."""
        if len(node_feats) != self.nf:
            raise ValueError(f"node_feats: expected {self.nf} tensors (graph_hop + 1), got {len(node_feats)}")
        batched = node_feats[0].dim() == 3
        if not batched:
            node_feats = [f.unsqueeze(0) for f in node_feats]
            node_type, node_relative_pos = node_type.unsqueeze(0), node_relative_pos.unsqueeze(0)
        out = F_.classifier_head(node_feats, node_type, node_relative_pos, ner_emb_weight, dis_embed_weight, self.flat, self.R,
                                 n_valid, self.dis_plus)
        return out if batched else out.squeeze(0)


# ======================================================================================================
class GraphHops(nn.Module):
    """The model's hop loop restricted to the graph blocks (GCGCN_glove.py:254-262 construction,
    :329-341 forward): hop 0 = GATAttention + GraphConvolution (CAGGC), hop i >= 1 = MultiHeadAttention +
    MultiGraphConvolution (MAGGC), ``x <- dropout(alpha * new + (1 - alpha) * x)``.

    Sub-module names equal the model's attribute names, so ``GraphHops.state_dict()`` is exactly the
    hot-path slice of a ``GCGCN_glove`` checkpoint (keys ``get_weighted_adj_matrix.*``,
    ``get_adj_matrix.{i}.*``, ``graphcnn.{i}.*``).
    """

    def __init__(self, hidden_size: int = 128, layer_num: int = 2, head_num: int = 8, graph_hop: int = 2,
                 alpha: float = 1.0, dropout: float = 0.2, apply_mask: bool = False):
        super().__init__()
        self.graph_hop, self.alpha, self.p = graph_hop, float(alpha), float(dropout)
        self.get_weighted_adj_matrix = GATAttention(hidden_size, hidden_size, apply_mask=apply_mask)
        self.get_adj_matrix = nn.ModuleList([MultiHeadAttention(head_num, hidden_size) for _ in range(graph_hop - 1)])
        self.graphcnn = nn.ModuleList()
        for i in range(graph_hop):
            if i == 0:
                self.graphcnn.append(GraphConvolution(layer_num, hidden_size, hidden_size))
            else:
                self.graphcnn.append(MultiGraphConvolution(layer_num, head_num, hidden_size, hidden_size))

    def forward(self, node_feat: Tensor, edge_feats: Sequence[Tensor], adj_matrix: Optional[Tensor] = None,
                n_valid: Optional[Tensor] = None) -> List[Tensor]:
        """edge_feats[i] is the edge tensor of hop i (``context_sent_att``).  Returns
        ``[x_0, x_1, ..., x_hops]``; the model's ``node_feats`` list (glove:338) is the first ``hops``."""
        feats = [node_feat]
        x = node_feat
        with F_.rng_scope(node_feat.device, 3 * self.graph_hop, enabled=self.training and node_feat.is_cuda):
            return self._hops(x, feats, edge_feats, adj_matrix, n_valid)

    def _hops(self, x, feats, edge_feats, adj_matrix, n_valid):
        def ride(i):
            # hop i + 1 needs its edge tensor only as mean_j E: let that HBM-bound pass ride in hop i's chain launch
            on = self.ride_edge_mean and x.is_cuda and i + 1 < self.graph_hop
            return edge_feats[i + 1] if on else None

        # alpha == 1 (the reference's setting): x <- dropout(new) is applied in the convolution's last epilogue
        fused_out = self.alpha == 1.0 and x.is_cuda
        odrop = self.p if (fused_out and self.training) else 0.0
        # ragged batch: the list of entity-row blocks that exist, built once (on the device) for every block of the loop
        rb = None
        if n_valid is not None and x.is_cuda and x.dim() == 3:
            rb = F_.row_blocks(n_valid, x.shape[0], x.shape[1])

        for i in range(self.graph_hop):
            e = edge_feats[i]
            if i < 1:
                # glove:330 builds mask = eq(adj_matrix, 0) and glove:163-164 then discards it; the mask is
                # not even materialised here (adj_matrix is accepted for signature compatibility only) unless the
                # paper-faithful opt-in asks for it
                mask = None
                if self.get_weighted_adj_matrix.apply_mask and adj_matrix is not None:
                    mask = torch.eq(adj_matrix, 0)                                                          # glove:330
                a, xa = self.get_weighted_adj_matrix(x, e, mask, n_valid=n_valid, return_input_alias=True)  # glove:332
                new = self.graphcnn[i](xa, e, a, n_valid=n_valid, ride_edge=ride(i), out_dropout=odrop, rowblk=rb)  # glove:333
            else:
                if self.fuse_maggc:      # glove:336-337 as one fused call pair (falls back to the two module calls by itself)
                    new = self.graphcnn[i].forward_with_attention(self.get_adj_matrix[i - 1], x, e, n_valid=n_valid,
                                                                  ride_edge=ride(i), out_dropout=odrop, rowblk=rb)
                else:
                    al, xa = self.get_adj_matrix[i - 1](x, e, n_valid=n_valid, return_input_alias=True, rowblk=rb)  # glove:336
                    new = self.graphcnn[i](xa, e, al, n_valid=n_valid, ride_edge=ride(i), out_dropout=odrop, rowblk=rb)  # glove:337
            if fused_out:
                x = new                                                                      # glove:339 + :341, in the block
            else:
                x = self.alpha * new + (1 - self.alpha) * x                                  # glove:339
                x = F_.dropout(x, self.p, self.training)                                     # glove:341
            feats.append(x)
        return feats

    # The next hop's edge mean rides as passenger workgroups of this hop's chain launch (chain.hip): on.  (Streaming it on a
    # side stream, or issuing it before GATAttention, were measured and dropped: DESIGN.md, dropped experiments.)
    ride_edge_mean = True
    # a MAGGC hop's attention rides inside its convolution's launches (functional.MaggcFn): on
    fuse_maggc = True


# ======================================================================================================
class GraphModelTail(nn.Module):
    """Everything GCGCN_glove.forward does AFTER the token encoder (glove:293-360) on the HIP kernels: entity pooling is
    left to the caller (``node_feat = node_pos @ context_output``, glove:297-298, a plain matmul), then per hop the
    edge-feature producer (f1), the CAGGC / MAGGC block (the hot path) and the hop glue, and finally the classifier head
    (f3).  ``state_dict()`` keys are the model's own (``word_attention.{i}.*``, ``linear_word_att.{i}.*``,
    ``sentence_attention.{i}.*``, ``linear_sentence_att.{i}.*``, ``get_weighted_adj_matrix.*``, ``get_adj_matrix.{i}.*``,
    ``graphcnn.{i}.*``, ``dense_layer.*``, ``bili_layer_01.*``, ``classification_layer_01.*``): the matching slice of a
    GCGCN_glove checkpoint loads with ``strict=True``.  The embeddings ``dis_embed`` / ``ner_emb`` stay with the encoder side
    of the model and are passed in.

    Reference behaviour kept: the model's ``node_feats`` list records the PRE-update features (glove:338), so the last hop's
    output never reaches the classifier -- it is still computed, as in the reference.
    """

    def __init__(self, hidden_size: int = 128, layer_num: int = 2, head_num: int = 8, graph_hop: int = 2, alpha: float = 1.0,
                 dis_size: int = 20, entity_type_size: int = 20, relation_num: int = 97, dis_plus: int = 10, dropout: float = 0.2):
        super().__init__()
        self.graph_hop, self.alpha, self.p = graph_hop, float(alpha), float(dropout)
        self.producers = nn.ModuleList([EdgeFeatureProducer(hidden_size, dis_size) for _ in range(graph_hop)])
        self.get_weighted_adj_matrix = GATAttention(hidden_size, hidden_size)
        self.get_adj_matrix = nn.ModuleList([MultiHeadAttention(head_num, hidden_size) for _ in range(graph_hop - 1)])
        self.graphcnn = nn.ModuleList([GraphConvolution(layer_num, hidden_size, hidden_size) if i == 0 else
                                       MultiGraphConvolution(layer_num, head_num, hidden_size, hidden_size) for i in range(graph_hop)])
        self.head = ClassifierHead(hidden_size, graph_hop, entity_type_size, dis_size, relation_num, dis_plus)
        self._install_key_hooks()

    # ---- the model's key names: producers.{i}.<mod>.<rest> <-> <mod>.{i}.<rest>, head.<k> <-> <k> ---------------------------
    # Done with nn.Module's prefix-aware hooks, so the mapping also holds when the tail is a SUBMODULE of a larger model (the
    # documented integration: dis_embed / ner_emb / the encoder live in the parent): parent.state_dict() then carries
    # ``<prefix>word_attention.0.*`` and parent.load_state_dict() accepts them.
    _PROD_MODS = tuple(sorted(set(k.split(".", 1)[0] for k in P_.producer_shapes(1, 1))))
    _HEAD_MODS = tuple(sorted(set(k.split(".", 1)[0] for k in P_.HEAD_STATE_ORDER)))

    @staticmethod
    def _to_model_key(k: str) -> str:
        p = k.split(".")
        if p[0] == "producers":
            return ".".join([p[2], p[1]] + p[3:])
        if p[0] == "head":
            return ".".join(p[1:])
        return k

    @classmethod
    def _to_module_key(cls, k: str) -> str:
        p = k.split(".")
        if p[0] in cls._PROD_MODS and len(p) > 2 and p[1].isdigit():
            return ".".join(["producers", p[1], p[0]] + p[2:])
        if p[0] in cls._HEAD_MODS:
            return "head." + k
        return k

    @staticmethod
    def _rename_keys(sd, prefix: str, fn):
        items = list(sd.items())            # rebuilt in place: order kept, the dict's _metadata attribute survives clear()
        sd.clear()
        for k, v in items:
            sd[prefix + fn(k[len(prefix):]) if k.startswith(prefix) else k] = v

    @staticmethod
    def _state_dict_hook(module, state_dict, prefix, local_metadata):
        module._rename_keys(state_dict, prefix, module._to_model_key)

    def _load_pre_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        self._rename_keys(state_dict, prefix, self._to_module_key)

    def _install_key_hooks(self):
        self._register_state_dict_hook(self._state_dict_hook)
        self._register_load_state_dict_pre_hook(self._load_pre_hook)

    # Opt-in: do not compute the last hop at all.  The model's ``node_feats`` list records the PRE-update features (glove:338),
    # so the last hop's output -- its edge-feature producer, attention and convolution -- never reaches the classifier and none
    # of its parameters ever gets a gradient.  Logits and gradients are identical either way; the default (False)
    # runs it like the reference does.
    skip_dead_hop = False
    # Opt-in: the producers hand the graph blocks their compact rows (CompactEdges) instead of writing context_sent_att[N,N,hidden]
    # per hop; same logits and gradients (tests/test_tail_gpu.py), no E / dE tensors.  Default: the dense tensor, the reference's
    # own call pattern.
    compact_edges = False

    def forward(self, context_output: Tensor, node_feat: Tensor, adj_matrix: Optional[Tensor], sen_matrix: Tensor,
                pos_matrix_h: Tensor, pos_matrix_t: Tensor, node_type: Tensor, node_relative_pos: Tensor,
                dis_embed_weight: Tensor, ner_emb_weight: Tensor, n_valid: Optional[Tensor] = None,
                max_live_slots: Optional[int] = None, max_live_pairs: Optional[int] = None) -> Tensor:
        """Shapes as in the reference (one document: context_output ``[T,H]`` or ``[1,T,H]``, node_feat ``[N,H]``, ...) or with a
        leading batch axis everywhere.  Returns ``relation_before_softmax_01``.  ``max_live_slots`` / ``max_live_pairs``: capacities
        of the edge-feature producers given up front (no host synchronisation; see EdgeFeatureProducer.forward)."""
        x = node_feat
        feats = [x]
        mask = None
        hops = self.graph_hop - 1 if (self.skip_dead_hop and self.graph_hop > 0) else self.graph_hop
        # alpha == 1 (the reference's setting): x <- dropout(new) is applied in the convolution's last epilogue (as GraphHops does)
        fused_out = self.alpha == 1.0 and x.is_cuda
        odrop = self.p if (fused_out and self.training) else 0.0
        rb = F_.row_blocks(n_valid, x.shape[0], x.shape[1]) if (n_valid is not None and x.is_cuda and x.dim() == 3) else None
        with F_.rng_scope(x.device, 3 * max(hops, 1), enabled=self.training and x.is_cuda):
            for i in range(hops):
                e = self.producers[i](context_output, sen_matrix, pos_matrix_h, pos_matrix_t, x, dis_embed_weight, n_valid=n_valid,
                                      max_live_slots=max_live_slots, max_live_pairs=max_live_pairs,
                                      compact=self.compact_edges and not self.get_weighted_adj_matrix.apply_mask)
                if i < 1:
                    if self.get_weighted_adj_matrix.apply_mask and adj_matrix is not None:
                        mask = torch.eq(adj_matrix, 0)                                          # glove:330
                    # (A, alias of x): the convolution's d(node_feat) is routed through the attention's own dX kernel
                    a, xa = self.get_weighted_adj_matrix(x, e, mask, n_valid=n_valid, return_input_alias=True)   # glove:332
                    new = self.graphcnn[i](xa, e, a, n_valid=n_valid, out_dropout=odrop, rowblk=rb)       # glove:333
                else:                                                                          # glove:336-337, fused where possible
                    new = self.graphcnn[i].forward_with_attention(self.get_adj_matrix[i - 1], x, e, n_valid=n_valid, out_dropout=odrop,
                                                                  rowblk=rb)
                feats.append(x)                                                                # glove:338 (pre-update)
                if fused_out:
                    x = new                                                                    # glove:339 + :341, inside the block
                else:
                    x = self.alpha * new + (1 - self.alpha) * x                                # glove:339
                    x = F_.dropout(x, self.p, self.training)                                   # glove:341
        if hops < self.graph_hop:
            feats.append(x)                     # the skipped hop would have recorded its input: the feature list stays complete
        return self.head(feats, node_type, node_relative_pos, ner_emb_weight, dis_embed_weight, n_valid=n_valid)
