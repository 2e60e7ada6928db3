"""Layer base class: builds attentions / FFNs / LayerNorms from cfg lists.

Mirrors the construction contract and state-dict naming (``attentions.N``, ``ffns.N``,
``norms.N``) of the reference's bevformer/modules/custom_base_transformer_layer.py:38-163,
including its defaults (``batch_first=True``, FFN ``embed_dims=768``) and the deprecated
``feedforward_channels`` / ``ffn_dropout`` / ``ffn_num_fcs`` kwargs that vocc.py still uses
(:88-99)."""
import copy
import warnings

from ..registry import (TRANSFORMER_LAYER, ConfigDict, build_attention,
                        build_feedforward_network)
from .bricks import BaseModule, ModuleList, build_norm_layer

_OPS = ('self_attn', 'norm', 'ffn', 'cross_attn')


@TRANSFORMER_LAYER.register_module(force=True)
class MyCustomBaseTransformerLayer(BaseModule):

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=768, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=True,
                 **kwargs):
        ffn_cfgs = copy.deepcopy(ffn_cfgs)      # the reference mutates its shared default dict
        for ori_name, new_name in (('feedforward_channels', 'feedforward_channels'),
                                   ('ffn_dropout', 'ffn_drop'), ('ffn_num_fcs', 'num_fcs')):
            if ori_name in kwargs:
                warnings.warn(f'The arguments `{ori_name}` in BaseTransformerLayer has been '
                              f'deprecated, now you should set `{new_name}` and other FFN related '
                              f'arguments to a dict named `ffn_cfgs`. ')
                if isinstance(ffn_cfgs, dict):
                    ffn_cfgs[new_name] = kwargs[ori_name]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        assert operation_order is not None and set(operation_order) <= set(_OPS), \
            f'The operation_order of {self.__class__.__name__} should only contain {_OPS}'
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs), \
                f'The length of attn_cfg {num_attn} is not consistent with the number of ' \
                f'attention {len(attn_cfgs)} in operation_order {operation_order}.'
            attn_cfgs = [copy.deepcopy(c) for c in attn_cfgs]
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        index = 0
        for operation_name in operation_order:
            if operation_name in ('self_attn', 'cross_attn'):
                if 'batch_first' in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]['batch_first']
                else:
                    attn_cfgs[index]['batch_first'] = self.batch_first
                attention = build_attention(attn_cfgs[index])
                attention.operation_name = operation_name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [ConfigDict(copy.deepcopy(ffn_cfgs)) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for ffn_index in range(num_ffns):
            if 'embed_dims' not in ffn_cfgs[ffn_index]:
                ffn_cfgs[ffn_index]['embed_dims'] = self.embed_dims
            else:
                assert ffn_cfgs[ffn_index]['embed_dims'] == self.embed_dims, \
                    'ffn embed_dims %s != layer embed_dims %s' % (
                        ffn_cfgs[ffn_index]['embed_dims'], self.embed_dims)
            self.ffns.append(build_feedforward_network(ffn_cfgs[ffn_index]))
        self.norms = ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        """Generic op sequencing (reference :165-260).  In a post-norm layer every branch (attention, FFN) is followed by a
        LayerNorm: a branch that can hand over ``PendingResidual(out, identity, p)`` leaves its residual add and dropout to
        that LayerNorm, which then is ONE fused pass each way on the GPU (``bricks.residual_layer_norm`` -> ``ver_add_ln_*``)
        instead of dropout + add + LayerNorm + the bf16 cast of the next Linear."""
        from .bricks import FFN, PendingResidual, residual_layer_norm
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        order = list(self.operation_order)
        for pos, layer in enumerate(order):
            defer = (not self.pre_norm) and pos + 1 < len(order) and order[pos + 1] == 'norm'
            if layer == 'self_attn':
                attn = self.attentions[attn_index]
                extra = dict(defer_residual=True) if (defer and getattr(attn, 'can_defer_residual', False)) else {}
                query = attn(
                    query, query, query, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **extra, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                if isinstance(query, PendingResidual):
                    query = residual_layer_norm(query, self.norms[norm_index])
                else:
                    query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                attn = self.attentions[attn_index]
                extra = dict(defer_residual=True) if (defer and getattr(attn, 'can_defer_residual', False)) else {}
                query = attn(
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **extra, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                ffn = self.ffns[ffn_index]
                if defer and isinstance(ffn, FFN):
                    query = ffn(query, identity if self.pre_norm else None, defer_residual=True)
                else:
                    query = ffn(query, identity if self.pre_norm else None)
                ffn_index += 1
        if isinstance(query, PendingResidual):
            query = query.materialize()
        return query
