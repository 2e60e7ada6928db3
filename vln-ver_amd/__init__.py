"""MI355X-native 2D->3D volumetric lifting path for VLN-VER.

Importing the package registers the plugin classes (``SpatialCrossAttention``,
``MSDeformableAttention3D``, ``VoxelFormerEncoder``, ``VoxelFormerLayer``,
``VoxelPerceptionTransformer``, ``VoxelLearnedPositionalEncoding``,
``VoxelFormerOccupancyHead``, the ``VoxelFormer`` detector) under the reference's names -- into mmcv's registries when mmcv
is installed, into ``registry.py``'s otherwise."""
from . import registry  # noqa: F401
from . import modules  # noqa: F401
from . import dense_heads  # noqa: F401
from . import detectors  # noqa: F401
from .registry import (ATTENTION, DETECTORS, HEADS, POSITIONAL_ENCODING, TRANSFORMER,  # noqa: F401
                       TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_detector, build_from_cfg)

__all__ = ['registry', 'modules', 'build_from_cfg']
