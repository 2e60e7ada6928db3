"""Learned 3-D positional encoding ``row + col + z`` -> [bs, C, Z, H, W]
(reference: bevformer/modules/voxel_positional_embedding.py:10-79; same parameter names
``row_embed / col_embed / z_embed``).  Computed at head:306-308 but not consumed by the vocc
encoder (no ``self_attn`` op) -- kept for API / checkpoint compatibility."""
import torch
import torch.nn as nn

from ..registry import POSITIONAL_ENCODING
from .bricks import BaseModule


@POSITIONAL_ENCODING.register_module(force=True)
class VoxelLearnedPositionalEncoding(BaseModule):

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, z_num_embed=16,
                 init_cfg=dict(type='Uniform', layer='Embedding')):
        super().__init__(init_cfg)
        self.num_feats = num_feats
        width = num_feats * 2
        self.row_embed = nn.Embedding(row_num_embed, width)
        self.col_embed = nn.Embedding(col_num_embed, width)
        self.z_embed = nn.Embedding(z_num_embed, width)
        self.row_num_embed = row_num_embed
        self.col_num_embed = col_num_embed
        self.z_num_embed = z_num_embed

    def init_weights(self):
        for emb in (self.row_embed, self.col_embed, self.z_embed):     # init_cfg Uniform(0,1)
            nn.init.uniform_(emb.weight, 0, 1)
        self._is_init = True

    def forward(self, mask):
        """mask [bs, d, h, w] (values unused) -> pos [bs, 2*num_feats, d, h, w]."""
        d, h, w = mask.shape[-3:]
        dev = self.col_embed.weight.device
        x = self.col_embed(torch.arange(w, device=dev))
        y = self.row_embed(torch.arange(h, device=dev))
        z = self.z_embed(torch.arange(d, device=dev))
        pos = x[None, None, :, :] + y[None, :, None, :] + z[:, None, None, :]
        # (the reference repeats the encoding over the batch, :76-79; the same values as a stride-0 view -- the vocc.py encoder
        #  never reads it, and a 192-viewpoint copy of it is 0.53 GB written per step)
        return pos.permute(3, 0, 1, 2).unsqueeze(0).expand(mask.shape[0], -1, -1, -1, -1)

    def __repr__(self):
        return (f'{self.__class__.__name__}(num_feats={self.num_feats}, '
                f'row_num_embed={self.row_num_embed}, col_num_embed={self.col_num_embed})')
