"""Model-level drop-ins: the classes ``config/Config.py`` instantiates (``model_pattern(config=self)``, Config.py:294) and
calls with the ten tensors of ``Config.from_list_to_tensor`` (Config.py:354), same constructor, ``forward`` signature and
``state_dict`` keys as the reference's

    GCGCN_glove                           /root/reference/models/GCGCN_glove.py:215-360
    GraphCNN_multihead_bert_gate_cls      /root/reference/models/GraphCNN_multihead_bert_gate_cls.py:219-351

so that ``con.train(gcgcn_amd.models.GCGCN_glove, save_name)`` runs the reference's trainer unchanged and a reference
checkpoint loads with ``strict=True``.

What runs where.  Everything from the entity pooling on (glove:293-360: edge-feature producers, CAGGC / MAGGC blocks, hop glue,
classifier head) is :class:`gcgcn_amd.GraphModelTail`, i.e. the HIP kernels.  The token encoder in front of it -- embeddings,
``EncoderLSTM`` (glove:377-428), ``linear_re`` + tanh, or BERT in the second model -- is plain PyTorch on purpose: it is not on
the path this library accelerates and is restated here only so that the model is complete.

Extension: every ``forward`` also accepts a leading batch axis on all ten tensors (what ``gcgcn_amd.data.collate`` returns)
plus ``n_valid[B]``; the reference's one-document call (``document[T]``, ``sen_matrix[N,N,S,T]`` ...) behaves as the reference's.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import nn

from .modules import GraphModelTail

Tensor = torch.Tensor


class LockedDropout(nn.Module):
    """One mask per (batch entry, feature), shared by all time steps (glove:363-375); identity in eval mode."""

    def __init__(self, dropout: float):
        super().__init__()
        self.dropout = float(dropout)

    def forward(self, x: Tensor) -> Tensor:
        if not self.training or self.dropout <= 0.0:
            return x
        keep = 1.0 - self.dropout
        m = torch.empty(x.size(0), 1, x.size(2), dtype=x.dtype, device=x.device).bernoulli_(keep) / keep
        return m.expand_as(x) * x


class EncoderLSTM(nn.Module):
    """The reference's BiLSTM token encoder (glove:377-428): ``nlayers`` ``nn.LSTM``s with learned initial states, locked dropout
    in front of every layer, layer outputs concatenated.  Parameter names as in the reference: ``rnns.{i}.*``, ``init_hidden.{i}``,
    ``init_c.{i}``.  Plain PyTorch (MIOpen's LSTM on ROCm)."""

    def __init__(self, input_size, num_units, nlayers, concat, bidir, dropout, return_last):
        super().__init__()
        self.rnns = nn.ModuleList(nn.LSTM(input_size if i == 0 else (num_units * 2 if bidir else num_units), num_units, 1,
                                          bidirectional=bidir, batch_first=True) for i in range(nlayers))
        nd = 2 if bidir else 1
        self.init_hidden = nn.ParameterList([nn.Parameter(torch.zeros(nd, 1, num_units)) for _ in range(nlayers)])
        self.init_c = nn.ParameterList([nn.Parameter(torch.zeros(nd, 1, num_units)) for _ in range(nlayers)])
        self.dropout = LockedDropout(dropout)
        self.concat, self.nlayers, self.return_last = concat, nlayers, return_last

    def forward(self, input: Tensor, input_lengths=None) -> Tensor:
        bsz = input.size(0)
        output, outputs = input, []
        for i in range(self.nlayers):
            h0 = self.init_hidden[i].expand(-1, bsz, -1).contiguous()
            c0 = self.init_c[i].expand(-1, bsz, -1).contiguous()
            output, _ = self.rnns[i](self.dropout(output), (h0, c0))
            outputs.append(output)
        return torch.cat(outputs, dim=2) if self.concat else outputs[-1]


class _GraphRelationModel(GraphModelTail):
    """What the two models share: construction of the post-encoder part, the reference's key order in ``state_dict()``, and
    ``forward`` from the encoder's token states on."""

    HIDDEN = 128                       # hard-coded in both reference models (glove:234, bert:237)
    _KEY_ORDER = ()                    # top-level attribute names in the reference's registration order

    def __init__(self, config, layer_num: int, head_num: int):
        super().__init__(hidden_size=self.HIDDEN, layer_num=layer_num, head_num=head_num, graph_hop=config.graph_hop,
                         alpha=config.alpha, dis_size=config.dis_size, entity_type_size=config.entity_type_size,
                         relation_num=config.relation_num, dis_plus=config.dis_plus, dropout=0.2)
        self.config = config
        self.layerNum, self.headNum = layer_num, head_num
        self._register_state_dict_hook(self._reference_order_hook)

    # keys come out grouped the way the reference registers its attributes (the key SET is what load_state_dict needs; the
    # order only matters to code that zips two state dicts)
    @staticmethod
    def _reference_order_hook(module, state_dict, prefix, local_metadata):
        rank = {name: i for i, name in enumerate(module._KEY_ORDER)}
        items = list(state_dict.items())
        mine = [(k, v) for k, v in items if k.startswith(prefix)]
        keyed = sorted(range(len(mine)), key=lambda i: (rank.get(mine[i][0][len(prefix):].split(".", 1)[0], len(rank)), i))
        state_dict.clear()
        for k, v in items:
            if not k.startswith(prefix):
                state_dict[k] = v
        for i in keyed:
            state_dict[mine[i][0]] = mine[i][1]

    def _graph_forward(self, context_output, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_pos, node_type,
                       node_relative_pos, n_valid, batched, **caps):
        # entity pooling (glove:293-298): node_feat[n] = sum_t node_pos[n, t] * context_output[t]
        node_feat = torch.bmm(node_pos, context_output) if batched else node_pos @ context_output.squeeze(0)
        return GraphModelTail.forward(self, context_output, node_feat, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_type,
                                      node_relative_pos, self.dis_embed.weight, self.ner_emb.weight, n_valid=n_valid, **caps)


class GCGCN_glove(_GraphRelationModel):
    """``GCGCN_glove(config)`` (glove:215-360).  ``config`` carries what the reference's does: ``data_word_vec`` (numpy
    ``[V, 100]``), ``entity_type_size``, ``coref_size``, ``max_length``, ``keep_prob``, ``graph_hop``, ``dis_size``, ``dis_num``,
    ``dis_plus``, ``relation_num``, ``alpha``."""

    _KEY_ORDER = ("word_emb", "ner_emb", "entity_embed", "rnn", "get_weighted_adj_matrix", "get_adj_matrix", "graphcnn", "linear_re",
                  "word_attention", "sentence_attention", "linear_word_att", "linear_sentence_att", "dense_layer", "bili_layer_01",
                  "classification_layer_01", "dis_embed")

    def __init__(self, config):
        super().__init__(config, layer_num=2, head_num=8)                                   # glove:250-251
        vec = np.asarray(config.data_word_vec, dtype=np.float32)
        self.word_emb = nn.Embedding(vec.shape[0], vec.shape[1])                             # glove:220-223
        self.word_emb.weight.data.copy_(torch.from_numpy(vec))
        self.word_emb.weight.requires_grad = True
        input_size = vec.shape[1] + config.entity_type_size + config.coref_size
        self.ner_emb = nn.Embedding(7, config.entity_type_size, padding_idx=0)               # glove:241
        self.entity_embed = nn.Embedding(config.max_length, config.coref_size, padding_idx=0)   # glove:246
        self.rnn = EncoderLSTM(input_size, self.HIDDEN, 1, True, True, 1 - config.keep_prob, False)   # glove:248
        self.linear_re = nn.Linear(self.HIDDEN * 2, self.HIDDEN)                             # glove:264
        self.dis_embed = nn.Embedding(config.dis_num, config.dis_size)                       # glove:279

    def encode(self, document: Tensor, document_ner: Tensor, document_pos: Tensor) -> Tensor:
        """Token states ``context_output`` ``[B,T,128]`` (glove:282-292)."""
        doc = torch.cat([self.word_emb(document), self.entity_embed(document_pos), self.ner_emb(document_ner)], dim=-1)
        return torch.tanh(self.linear_re(self.rnn(doc, doc.size(1))))

    def forward(self, document, document_ner, document_pos, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_pos, node_type,
                node_relative_pos, n_valid: Optional[Tensor] = None, **caps):
        batched = document.dim() == 2
        if not batched:
            document, document_ner, document_pos = (t.unsqueeze(0) for t in (document, document_ner, document_pos))
        ctx = self.encode(document, document_ner, document_pos)
        return self._graph_forward(ctx, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_pos, node_type, node_relative_pos,
                                   n_valid, batched, **caps)


class GraphCNN_multihead_bert_gate_cls(_GraphRelationModel):
    """The BERT variant (bert:219-351): token states from ``BertModel`` instead of the BiLSTM, four sub-layers and four heads,
    and ``linear_cls`` of the [CLS] state added to every pair's logits (bert:345-346).

    ``bert``: the encoder module, called as the reference calls it -- ``bert(document[B,T], output_all_encoded_layers=False)`` ->
    ``(states[B,T,768], pooled)``.  Default: ``pytorch_pretrained_bert.BertModel.from_pretrained('./bert/bert-base-uncased')``
    exactly as bert:228; that package / those weights are third-party and are not part of this library, so without them the
    constructor raises ImportError unless an encoder is passed in."""

    WORD_VEC_SIZE = 768
    _KEY_ORDER = ("bert", "ner_emb", "entity_embed", "get_weighted_adj_matrix", "get_adj_matrix", "graphcnn", "linear_re",
                  "word_attention", "sentence_attention", "linear_word_att", "linear_sentence_att", "dense_layer", "bili_layer_01",
                  "classification_layer_01", "linear_cls", "dis_embed")

    def __init__(self, config, bert: Optional[nn.Module] = None):
        super().__init__(config, layer_num=4, head_num=4)                                   # bert:247-248
        if bert is None:
            try:
                from pytorch_pretrained_bert import BertModel
            except ImportError as e:
                raise ImportError("GraphCNN_multihead_bert_gate_cls needs an encoder: install pytorch_pretrained_bert with "
                                  "./bert/bert-base-uncased (bert:228), or pass bert=<module>") from e
            bert = BertModel.from_pretrained("./bert/bert-base-uncased")
        self.bert = bert
        input_size = self.WORD_VEC_SIZE + config.entity_type_size + config.coref_size
        self.ner_emb = nn.Embedding(7, config.entity_type_size, padding_idx=0)
        self.entity_embed = nn.Embedding(config.max_length, config.coref_size, padding_idx=0)
        self.linear_re = nn.Linear(input_size, self.HIDDEN)                                  # bert:259
        self.linear_cls = nn.Linear(self.WORD_VEC_SIZE, config.relation_num)                 # bert:270
        self.dis_embed = nn.Embedding(config.dis_num, config.dis_size)

    def forward(self, document, document_ner, document_pos, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_pos, node_type,
                node_relative_pos, n_valid: Optional[Tensor] = None, **caps):
        batched = document.dim() == 2
        if not batched:
            document, document_ner, document_pos = (t.unsqueeze(0) for t in (document, document_ner, document_pos))
        doc, _ = self.bert(document, output_all_encoded_layers=False)                        # bert:275
        cls_feat = doc[:, 0, :]                                                              # bert:277
        doc = torch.cat([doc, self.entity_embed(document_pos), self.ner_emb(document_ner)], dim=-1)
        ctx = torch.tanh(self.linear_re(doc))                                                # bert:283
        logits = self._graph_forward(ctx, adj_matrix, sen_matrix, pos_matrix_h, pos_matrix_t, node_pos, node_type,
                                     node_relative_pos, n_valid, batched, **caps)
        cls = self.linear_cls(cls_feat)                                                      # bert:345
        if not batched:
            return logits + cls[0]
        add = cls[:, None, None, :]
        if n_valid is not None:                  # padding pairs of a ragged batch stay zero
            ok = torch.arange(logits.shape[1], device=logits.device)[None, :] < n_valid[:, None]
            add = add * (ok[:, :, None] & ok[:, None, :]).unsqueeze(-1).to(add.dtype)
        return logits + add
