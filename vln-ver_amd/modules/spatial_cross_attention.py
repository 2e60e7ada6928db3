"""SpatialCrossAttention / MSDeformableAttention3D on the MI355X kernels.

Same registry names, constructor kwargs, forward signatures and parameter names as the
reference's bevformer/modules/spatial_cross_attention.py:31-402; the arithmetic is
restructured for the GPU:

* the reference re-batches the visible queries of every camera into a zero-padded
  ``[bs, 6, max_len, C]`` tensor behind six ``nonzero()`` host syncs (:139-154), runs the three
  input projections on the padded rows, and scatter-adds back (:166-173).  Because
  ``sampling_offsets`` / ``attention_weights`` are linear maps of the *query row only*, the
  re-batched rows of different cameras are the same vectors: we project every voxel query once
  ([Nq,C] instead of [6*max_len,C]) and hand offsets/logits plus the device-side hit table to
  one fused kernel (``ver_sca_forward``) that produces the camera-averaged ``slots`` directly;
* bs > 1 is supported (the reference hard-wires batch element 0's indices, :140): every
  viewpoint carries its own hit lists, so results per viewpoint equal a bs=1 reference run.
"""
import math
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hipops
from ..registry import ATTENTION, build_attention
from .bricks import BaseModule, PendingResidual, const_tensor, constant_init, lowp_view, tall_linear, xavier_init


# VER_SCA_HEAD_MAJOR=1: value_proj writes the head-major layout (contiguous gather tiles).  OFF by default: measured at 192
# viewpoints per launch the gather gains 14 us forward + 6 us backward per layer, but the batched N = 96 GEMM that produces
# the layout costs 1.07 ms against 0.31 ms for the plain [rows, 768] GEMM (a permuting copy after the plain GEMM: + 0.19 ms)
# -- the step as a whole is faster with the reference's layout.
_HEAD_MAJOR = os.environ.get('VER_SCA_HEAD_MAJOR', '0') == '1'


@ATTENTION.register_module(force=True)
class SpatialCrossAttention(BaseModule):
    """Multi-view gather driver (reference :32-176).

    Args mirror the reference: embed_dims, num_cams, pc_range, dropout, init_cfg,
    batch_first, deformable_attention (cfg dict of an ``MSDeformableAttention3D``).
    """

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False,
                 deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=256,
                                           num_levels=4),
                 **kwargs):
        super().__init__(init_cfg)
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag='encoder', hit_table=None,
                map_hw=None, defer_residual=False, value_lowp=None, **kwargs):
        """query [bs,Nq,C]; key/value [Ncam,Nk,bs,C]; reference_points_cam [Ncam,bs,Nq,D,2];
        bev_mask [Ncam,bs,Nq,D] -> [bs,Nq,C].

        ``hit_table`` (hipops.HitTable) and ``map_hw`` are what our encoder passes so that the
        projection / list building is done once for all layers; when absent they are derived
        here from ``reference_points_cam`` / ``bev_mask`` / ``spatial_shapes``.
        """
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        bs, num_query, _ = query.shape
        att = self.deformable_attention
        if att.num_levels != 1:
            raise NotImplementedError('fused SCA kernel is built for one feature level '
                                      '(vocc.py:60 _num_levels_=1), got %d' % att.num_levels)
        if hit_table is None:
            hit_table = hipops.hits_from_mask(reference_points_cam, bev_mask)
        if map_hw is None:
            hw = spatial_shapes.reshape(-1, 2)[0].tolist()     # one host read; the encoder passes map_hw
            map_hw = (int(hw[0]), int(hw[1]))
        num_cams, nk, vbs, c = value.shape
        assert num_cams == self.num_cams and vbs == bs and nk == map_hw[0] * map_hw[1]
        # [Ncam,Nk,bs,C] -> [bs,Ncam,Nk,C]; a no-copy view when the caller built it that way
        # ``value_lowp``: the encoder's one bf16 cast of ``value`` ([bs,Ncam,Nk,C]) for all its layers, only meaningful
        # under the bf16 autocast it was made for
        # the gather's output buffer: its zero fill (rows of voxels not seen by exactly one camera) depends on the hit
        # table only and runs on a side stream under the three projections below
        prepared = hipops.sca_prepare_slots(hit_table, c)
        use_lowp = (value_lowp is not None and torch.is_autocast_enabled('cuda')
                    and torch.get_autocast_dtype('cuda') == value_lowp.dtype)
        v_in = value_lowp if use_lowp else value.permute(2, 0, 1, 3)
        hd = c // att.num_heads
        # head-major value (bf16 path at the vocc.py shape): value_proj as one batched GEMM over the heads writes
        # [heads, bs, Ncam, Nk, hd], in which a (camera, head) tile of the gather is one contiguous block of HBM
        head_major = (_HEAD_MAJOR and v_in.is_cuda and use_lowp
                      and hipops.sca_head_major_supported(v_in.dtype, hd, att.num_points, map_hw[0], map_hw[1]))
        if head_major:
            v = hipops.head_major_linear(v_in.reshape(-1, c), att.value_proj.weight, att.value_proj.bias, att.num_heads)
            v = v.view(att.num_heads, bs, num_cams, nk, hd)
        else:
            v = tall_linear(att.value_proj, v_in)
            v = v.reshape(bs, num_cams, nk, att.num_heads, hd)
        # sampling_offsets and attention_weights read the same rows: one GEMM [.., C] x [C, 128 + 64] (and one cast of
        # the query under autocast) instead of two narrow ones; the parameters stay the reference's two Linears
        n_off = att.sampling_offsets.out_features
        both = tall_linear(None, lowp_view(query) if query_pos is None else query,
                           torch.cat([att.sampling_offsets.weight, att.attention_weights.weight], 0),
                           torch.cat([att.sampling_offsets.bias, att.attention_weights.bias], 0))
        offsets = both[..., :n_off].reshape(bs, num_query, att.num_heads, att.num_points, 2)
        logits = both[..., n_off:].reshape(bs, num_query, att.num_heads, att.num_points)
        # under bf16 autocast the gather hands output_proj the bf16 rows it would cast to anyway (and gets bf16 gradients back)
        lowp_out = use_lowp and v.dtype == torch.bfloat16
        slots = hipops.sca_gather(v, offsets, logits, hit_table, map_hw[0], map_hw[1], prepared, head_major, lowp_out)
        slots = tall_linear(self.output_proj, slots if lowp_out else slots.to(query.dtype))
        if defer_residual:                  # the caller's LayerNorm adds the residual (residual_layer_norm)
            return PendingResidual(slots, inp_residual, self.dropout.p if self.dropout.training else 0.0)
        return self.dropout(slots) + inp_residual


@ATTENTION.register_module(force=True)
class MSDeformableAttention3D(BaseModule):
    """Deformable sampling of the per-camera maps (reference :180-402).  Holds the three
    projections (``sampling_offsets``, ``attention_weights``, ``value_proj``); no output
    projection and no residual of its own (``output_proj = None``, :223)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        dim_per_head = embed_dims // num_heads
        if not (isinstance(dim_per_head, int) and dim_per_head > 0):
            raise ValueError('invalid dim_per_head %r' % (dim_per_head,))
        self.norm_cfg = norm_cfg
        self.batch_first = batch_first
        self.output_proj = None
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        """Zero offset weights, ring-pattern bias, zero attention weights, xavier value_proj
        (reference :255-273)."""
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2)
        grid = grid.repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        """Stand-alone use with the reference's signature: query [bs,Nq,C], value [bs,Nk,C],
        reference_points [bs,Nq,D,2] -> [bs,Nq,C], through the mmcv-shaped op
        (``ver_msda_forward``).  SpatialCrossAttention does not call this; it uses the three
        projections with the fused kernel."""
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        if not isinstance(spatial_shapes, torch.Tensor):
            spatial_shapes = const_tensor([list(map(int, r)) for r in spatial_shapes], query.device)
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(
            bs, num_query, self.num_heads, self.num_levels * self.num_points).softmax(-1)
        attention_weights = attention_weights.view(bs, num_query, self.num_heads, self.num_levels,
                                                   self.num_points)
        if reference_points.shape[-1] != 2:
            raise ValueError(f'Last dim of reference_points must be 2, '
                             f'but get {reference_points.shape[-1]} instead.')
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        num_z = reference_points.shape[2]
        if self.num_points % num_z != 0:
            raise ValueError('num_points %d not divisible by %d Z-anchors' % (self.num_points, num_z))
        off = sampling_offsets / normalizer[None, None, None, :, None, :].to(sampling_offsets.dtype)
        off = off.view(bs, num_query, self.num_heads, self.num_levels, self.num_points // num_z,
                       num_z, 2)
        loc = reference_points[:, :, None, None, None, :, :] + off
        loc = loc.view(bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        if level_start_index is None:
            sizes = spatial_shapes[:, 0] * spatial_shapes[:, 1]
            level_start_index = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
        if not value.is_cuda:
            raise RuntimeError('MSDeformableAttention3D runs only on the GPU (HIP kernels); '
                               'there is no CPU fallback in this package')
        output = hipops.MultiScaleDeformableAttnFunction_fp32.apply(
            value, spatial_shapes, level_start_index, loc, attention_weights, self.im2col_step)
        output = output.to(query.dtype)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return output


def _warn_non_pow2(dim_per_head):
    # kept for parity of user-visible behaviour (reference :235-240); the HIP kernels have no
    # power-of-two preference, head_dim 96 is the tuned case.
    if dim_per_head & (dim_per_head - 1):
        warnings.warn('head dim %d is not a power of 2 (fine for the gfx950 kernels)' % dim_per_head)
