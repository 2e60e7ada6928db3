"""VoxelPerceptionTransformer: entry of the lifting path (``get_voxel_features``) and, when a
decoder is configured, the detection half of ``forward``.

Reference: bevformer/modules/voxel_transformer.py:24-301 (same registry name, kwargs,
parameter names ``level_embeds`` [4,C] / ``cams_embeds`` [6,C] / ``reference_points``).
The reference hard-wires ``reshape(6, 1, 14, 14, 768)`` (:146); here camera count, batch, map
size and width come from the tensor, and the features are laid out ``[bs, Ncam, Nk, C]``
contiguous so the value projection and the gather kernel read them without a transpose."""
import math

import torch
import torch.nn as nn

from ..registry import TRANSFORMER, build_transformer_layer_sequence
from .bricks import BaseModule, const_tensor, xavier_init
from .spatial_cross_attention import MSDeformableAttention3D


class _EmbedCast(torch.autograd.Function):
    """feats [Ncam,bs,Nk,C] fp32 + embed [Ncam,C] -> bf16 [bs,Ncam,Nk,C] contiguous (the sum formed in fp32, rounded once)."""

    @staticmethod
    def forward(ctx, feats, embed):
        out = torch.empty((feats.shape[1], feats.shape[0]) + tuple(feats.shape[2:]), dtype=torch.bfloat16, device=feats.device)
        torch.add(feats.permute(1, 0, 2, 3), embed[None, :, None, :], out=out)
        return out

    @staticmethod
    def backward(ctx, g):
        d_feats = g.permute(1, 0, 2, 3).float() if ctx.needs_input_grad[0] else None
        return d_feats, g.sum(dim=(0, 2), dtype=torch.float32)


@TRANSFORMER.register_module(force=True)
class VoxelPerceptionTransformer(BaseModule):

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 decoder=None, embed_dims=256, rotate_prev_bev=True, use_shift=True,
                 use_can_bus=True, can_bus_norm=True, use_cams_embeds=True,
                 rotate_center=[100, 100], decoder_on_bev=False, voxel_2_bev_type='mlp', bev_z=1,
                 **kwargs):
        super().__init__(**kwargs)
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = build_transformer_layer_sequence(decoder) if decoder is not None else None
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.rotate_prev_bev = rotate_prev_bev
        self.use_shift = use_shift
        self.use_can_bus = use_can_bus
        self.can_bus_norm = can_bus_norm
        self.use_cams_embeds = use_cams_embeds
        self.decoder_on_bev = decoder_on_bev
        self.voxel_2_bev_type = voxel_2_bev_type
        self.bev_z = bev_z
        self.two_stage_num_proposals = two_stage_num_proposals
        self.rotate_center = rotate_center
        self.init_layers()

    def init_layers(self):
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(self.num_cams, self.embed_dims))
        if self.decoder is not None:
            self.reference_points = nn.Linear(self.embed_dims, 3)
        if self.decoder is not None and self.decoder_on_bev and self.voxel_2_bev_type == 'mlp':
            mid = self.embed_dims * self.bev_z
            self.voxel2bev = nn.Sequential(
                nn.Linear(mid, mid), nn.LayerNorm(mid), nn.ReLU(inplace=True),
                nn.Linear(mid, self.embed_dims), nn.LayerNorm(self.embed_dims), nn.ReLU(inplace=True))

    def init_weights(self):
        """xavier-uniform on every >1-D parameter, then the deformable-attention re-inits and
        N(0,1) embeddings (reference :99-116)."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            # only these three classes are re-initialised by the reference (:107-114); the decoder's
            # VoxelCustomMSDeformableAttention is a plain BaseModule there and keeps its xavier weights
            if isinstance(m, MSDeformableAttention3D):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)
        if self.decoder is not None:
            xavier_init(self.reference_points, distribution='uniform', bias=0.)

    def get_voxel_features(self, mlvl_feats, bev_queries, bev_z, bev_h, bev_w,
                           grid_length=[0.512, 0.512], bev_pos=None, prev_bev=None, **kwargs):
        """mlvl_feats [Ncam, bs, Nk, C] (the detector's (6,1,196,768)); bev_queries [Nq, C]
        -> bev_embed [bs, Nq, C]   (reference :119-185)."""
        num_cam, bs, nk, c = mlvl_feats.shape
        map_h = int(math.isqrt(nk))
        if map_h * map_h != nk:
            raise ValueError('feature maps must be square token grids, got %d tokens' % nk)
        # [Nq,bs,C] as the reference builds it (:150), laid out sample-major: the encoder's [bs,Nq,C] view of it is then
        # contiguous (its first residual and first projection read it without a strided copy)
        bev_queries = bev_queries.unsqueeze(0).repeat(bs, 1, 1).permute(1, 0, 2)
        if bev_pos is not None:
            bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        shift = bev_queries.new_zeros(1, 3)
        feat = mlvl_feats.permute(1, 0, 2, 3)                          # [bs,Ncam,Nk,C]
        embed = self.level_embeds[0].to(feat.dtype)
        if self.use_cams_embeds:
            embed = embed[None, :] + self.cams_embeds.to(feat.dtype)   # [Ncam,C]
        else:
            embed = embed[None, :].expand(num_cam, -1)
        lowp = None
        if (feat.is_cuda and feat.dtype == torch.float32 and torch.is_autocast_enabled('cuda')
                and torch.get_autocast_dtype('cuda') == torch.bfloat16):
            # bf16 autocast: every consumer of the feature maps (the three value projections) reads them in bf16 -- add the
            # embeddings, bring the maps into [bs,Ncam,Nk,C] order and round to bf16 in ONE pass (1.0 GB of traffic at 192
            # viewpoints instead of 4.1 for add + contiguous + cast; the embeddings' gradient is one fp32 reduction of the
            # bf16 gradient instead of a widening copy + a reduction)
            lowp = _EmbedCast.apply(mlvl_feats, embed)
            feat = lowp
        else:
            feat = (feat + embed[None, :, None, :]).contiguous()
        if lowp is not None:
            kwargs['value_lowp'] = lowp
        spatial_shapes = const_tensor([[map_h, map_h]], feat.device)
        level_start_index = spatial_shapes.new_zeros((1,))
        feat_flatten = feat.permute(1, 2, 0, 3)                        # view: [Ncam,Nk,bs,C]
        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_z=bev_z, bev_h=bev_h,
                            bev_w=bev_w, bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, prev_bev=prev_bev, shift=shift,
                            map_hw=(map_h, map_h), **kwargs)

    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_z, bev_h, bev_w,
                grid_length=[0.512, 0.512], bev_pos=None, reg_branches=None, cls_branches=None,
                prev_bev=None, **kwargs):
        """Encoder + detection decoder (reference :188-301) ->
        (voxel_embed [Nq,bs,C], inter_states, init_reference_out, inter_references_out)."""
        if self.decoder is None:
            raise RuntimeError('forward() needs a decoder; use get_voxel_features() for lifting only')
        voxel_embed = self.get_voxel_features(mlvl_feats, bev_queries, bev_z, bev_h, bev_w,
                                              grid_length=grid_length, bev_pos=bev_pos,
                                              prev_bev=prev_bev, **kwargs)
        return self.decode(voxel_embed, object_query_embed, bev_z, bev_h, bev_w, reg_branches=reg_branches,
                           cls_branches=cls_branches, **kwargs)

    def decode(self, voxel_embed, object_query_embed, bev_z, bev_h, bev_w, reg_branches=None, cls_branches=None,
               **kwargs):
        """The detection half of ``forward`` (reference :230-301) on an encoder output ``voxel_embed`` [bs,Nq,C]: a
        caller that wants the decoder on its own HIP stream (the head's training step) runs the two halves itself."""
        if self.decoder is None:
            raise RuntimeError('decode() needs a decoder')
        bs = voxel_embed.shape[0]
        query_pos, query = torch.split(object_query_embed, self.embed_dims, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        init_reference_out = reference_points
        query = query.permute(1, 0, 2)
        query_pos = query_pos.permute(1, 0, 2)
        voxel_embed = voxel_embed.permute(1, 0, 2)
        if self.decoder_on_bev:
            raise NotImplementedError('decoder_on_bev=True is not used by vocc.py (:109)')
        for k in ('world2pixel', 'origin', 'hit_table', 'map_hw', 'img_metas'):
            kwargs.pop(k, None)
        inter_states, inter_references = self.decoder(
            query=query, key=None, value=voxel_embed, query_pos=query_pos,
            reference_points=reference_points, reg_branches=reg_branches, cls_branches=cls_branches,
            spatial_shapes=const_tensor([[bev_z, bev_h, bev_w]], query.device),
            level_start_index=const_tensor([0], query.device), **kwargs)
        return voxel_embed, inter_states, init_reference_out, inter_references
