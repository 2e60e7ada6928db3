"""Detection decoder over the lifted volume ("next" row 1 of SURVEY.md section 8f).

Registry names / kwargs / parameter names follow the reference's
bevformer/modules/voxel_decoder.py:53-337 and the mmcv / mmdet layers its config composes
(vocc.py:138-166): ``VoxelDetectionTransformerDecoder``, ``VoxelCustomMSDeformableAttention``,
``DetrTransformerDecoderLayer``, ``MultiheadAttention``.

Dense projections / self-attention are hipBLASLt GEMMs (MFMA) through torch; the 3-D
(trilinear) deformable sampling runs on the hand-written HIP kernels ``ver_msda3d_forward`` /
``ver_msda3d_backward`` (same lane organisation as the 2-D drop-in op).  100 queries x 8 heads x
4 points per viewpoint: launch-latency territory, not a roofline kernel."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import ATTENTION, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, USING_MMCV
from .bricks import (BaseModule, PendingResidual, TransformerLayerSequence, const_tensor, constant_init, host_values,
                     xavier_init)
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer


def inverse_sigmoid(x, eps=1e-5):
    """voxel_decoder.py:35-50."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def voxel_multi_scale_deformable_attn(value, value_spatial_shapes, sampling_locations,
                                      attention_weights):
    """3-D (trilinear, zero padded, align_corners=False) deformable sampling
    (voxel_temporal_self_attention.py:275-335).  value [bs,Nk,heads,hd]; shapes [[D,H,W]];
    loc [bs,Nq,heads,L,P,3] (x,y,z); weights [bs,Nq,heads,L,P] -> [bs,Nq,heads*hd]."""
    bs, _, heads, hd = value.shape
    _, nq, _, nl, npt, _ = sampling_locations.shape
    sizes = [int(d) * int(h) * int(w) for d, h, w in value_spatial_shapes]
    per_level = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (d, h, w) in enumerate(value_spatial_shapes):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(bs * heads, hd, int(d), int(h), int(w))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1).unsqueeze(1)
        s = F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False)
        sampled.append(s.reshape(bs * heads, hd, nq, npt))
    aw = attention_weights.transpose(1, 2).reshape(bs * heads, 1, nq, nl * npt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, heads * hd, nq)
    return out.transpose(1, 2).contiguous()


@ATTENTION.register_module(force=True)
class VoxelCustomMSDeformableAttention(BaseModule):
    """voxel_decoder.py:135-337 (sequence-first, with output_proj + dropout + identity)."""


    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 3)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin(), thetas.cos() + thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 3)
        grid = grid.repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    can_defer_residual = True        # forward(..., defer_residual=True) returns bricks.PendingResidual

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag='decoder', defer_residual=False, **kwargs):
        if 'residual' in kwargs and identity is None:        # deprecated_api_warning alias
            identity = kwargs.pop('residual')
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        shapes = [[int(v) for v in row] for row in host_values(spatial_shapes)]
        assert sum(d * h * w for d, h, w in shapes) == num_value
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        offsets = self.sampling_offsets(query).view(bs, num_query, self.num_heads, self.num_levels,
                                                    self.num_points, 3)
        weights = self.attention_weights(query).view(bs, num_query, self.num_heads,
                                                     self.num_levels * self.num_points).softmax(-1)
        weights = weights.view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
        if reference_points.shape[-1] != 3:
            raise ValueError(f'Last dim of reference_points must be 3, '
                             f'but get {reference_points.shape[-1]} instead.')
        normalizer = const_tensor([[float(s[2]), float(s[1]), float(s[0])] for s in shapes], offsets.device, offsets.dtype)
        loc = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        if not value.is_cuda:
            raise RuntimeError('VoxelCustomMSDeformableAttention runs only on the GPU (HIP kernels); '
                               'there is no CPU fallback in this package')
        from ..hipops import voxel_msda
        if level_start_index is None:
            sizes = spatial_shapes.prod(1)
            level_start_index = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
        output = voxel_msda(value, spatial_shapes, level_start_index, loc, weights)
        output = self.output_proj(output.to(query.dtype))
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        if defer_residual:                  # the LayerNorm that follows adds the residual (bricks.residual_layer_norm)
            return PendingResidual(output.contiguous(), identity, self.dropout.p if self.dropout.training else 0.0)
        return self.dropout(output) + identity


class MultiheadAttention(BaseModule):
    """mmcv 1.4.0 wrapper over nn.MultiheadAttention: q += query_pos,
    k += key_pos (= query_pos when shapes match), identity + dropout(out)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        if 'dropout' in kwargs:
            attn_drop = kwargs['dropout']
            dropout_layer = dict(type='Dropout', drop_prob=kwargs.pop('dropout'))
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer['drop_prob']) if dropout_layer else nn.Identity()

    can_defer_residual = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, defer_residual=False, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        if defer_residual and (not self.training or self.proj_drop.p == 0.0):
            drop = self.dropout_layer
            p = drop.p if (isinstance(drop, nn.Dropout) and drop.training) else 0.0
            return PendingResidual(out.contiguous(), identity, p)
        return identity + self.dropout_layer(self.proj_drop(out))


class DetrTransformerDecoderLayer(MyCustomBaseTransformerLayer):
    """mmdet DetrTransformerDecoderLayer = mmcv BaseTransformerLayer with ``batch_first=False``
    and the 6-op order.  FFN defaults follow mmcv (embed_dims 256), the
    deprecated kwargs override as in the in-tree copy."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        kwargs.setdefault('batch_first', False)
        kwargs.setdefault('ffn_cfgs', dict(type='FFN', embed_dims=256, feedforward_channels=1024,
                                           num_fcs=2, ffn_drop=0.,
                                           act_cfg=dict(type='ReLU', inplace=True)))
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == {'self_attn', 'norm', 'cross_attn', 'ffn'}


if not USING_MMCV:
    ATTENTION.register_module(module=MultiheadAttention, force=True)
    TRANSFORMER_LAYER.register_module(module=DetrTransformerDecoderLayer, force=True)


@TRANSFORMER_LAYER_SEQUENCE.register_module(force=True)
class VoxelDetectionTransformerDecoder(TransformerLayerSequence):
    """voxel_decoder.py:53-132: iterative reference-point refinement over the layers."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.fp16_enabled = False

    def forward(self, query, *args, reference_points=None, reg_branches=None, key_padding_mask=None,
                **kwargs):
        output = query
        intermediate, intermediate_reference_points = [], []
        # reg_branches[lid](state of layer lid), in graph: the head evaluates the same branch on the same state again
        # (head:590 after voxel_decoder.py:118) and may take these instead (``take_branch_outputs``)
        self._branch_outputs = [] if (reg_branches is not None and self.return_intermediate) else None
        for lid, layer in enumerate(self.layers):
            reference_points_input = reference_points[..., :3].unsqueeze(2)
            output = layer(output, *args, reference_points=reference_points_input,
                           key_padding_mask=key_padding_mask, **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branches[lid](output)
                if self._branch_outputs is not None:
                    self._branch_outputs.append(tmp)
                assert reference_points.shape[-1] == 3
                # voxel_decoder.py:118-126 fills (x, y) and z of a zero tensor in two assignments and detaches the result:
                # the same three columns in one expression, outside the autograd graph
                with torch.no_grad():
                    new_ref = torch.cat([tmp[..., :2], tmp[..., 4:5]], -1) + inverse_sigmoid(reference_points)
                    reference_points = new_ref.sigmoid()
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points

    def take_branch_outputs(self):
        """The per-layer ``reg_branches`` outputs [bs,Nq,code] of the last ``forward`` (None without box refinement);
        handed out once, so that no graph stays alive on the module."""
        out, self._branch_outputs = getattr(self, '_branch_outputs', None), None
        return out
