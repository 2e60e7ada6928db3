from .bricks import FFN, BaseModule, ModuleList, Sequential, TransformerLayerSequence  # noqa: F401
from .multi_scale_deformable_attn_function import (MultiScaleDeformableAttnFunction_fp16,  # noqa: F401
                                                   MultiScaleDeformableAttnFunction_fp32)
from .spatial_cross_attention import MSDeformableAttention3D, SpatialCrossAttention  # noqa: F401
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer  # noqa: F401
from .voxel_encoder import VoxelFormerEncoder, VoxelFormerLayer  # noqa: F401
from .voxel_positional_embedding import VoxelLearnedPositionalEncoding  # noqa: F401
from .voxel_transformer import VoxelPerceptionTransformer  # noqa: F401
from .voxel_decoder import (DetrTransformerDecoderLayer, MultiheadAttention,  # noqa: F401
                            VoxelCustomMSDeformableAttention, VoxelDetectionTransformerDecoder)
