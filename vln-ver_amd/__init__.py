"""MI355X-native 2D->3D volumetric lifting path for VLN-VER.

Importing the package registers the plugin classes (``SpatialCrossAttention``,
``MSDeformableAttention3D``, ``VoxelFormerEncoder``, ``VoxelFormerLayer``,
``VoxelPerceptionTransformer``, ``VoxelLearnedPositionalEncoding``,
``VoxelFormerOccupancyHead``, the ``VoxelFormer`` detector) under the reference's names -- into mmcv's registries when mmcv
is installed, into ``registry.py``'s otherwise."""
import os as _os

# ROCm 7.2: hipGraph replay with "AQL packet capture" (the runtime's default fast path for kernel nodes) does not keep the order
# between a MEMSET node and the kernel node behind it -- from the second replay on, kernels that accumulate into a buffer
# zeroed by hipMemsetAsync (ours, and PyTorch's own multi-block reductions, whose semaphores are zeroed that way) meet stale
# bytes: garbage / NaN gradients in a replayed training step (scratch/r06/graph_nan.py, docs/HISTORY.md R6).  The flag is read
# when the HIP runtime starts, so it is set here, at package import, before anything of ours touches the GPU (bench.py and
# tests/conftest.py set it too, in front of their first torch.cuda call); our own launchers zero with a kernel in any case
# (csrc/ver_common.h: ver_zero_async).  Costs ~1 % of a replayed one-viewpoint step.
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

from . import registry  # noqa: F401,E402
from . import modules  # noqa: F401,E402
from . import dense_heads  # noqa: F401,E402
from . import detectors  # noqa: F401,E402
from .registry import (ATTENTION, DETECTORS, HEADS, POSITIONAL_ENCODING, TRANSFORMER,  # noqa: F401,E402
                       TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_detector, build_from_cfg)

__all__ = ['registry', 'modules', 'build_from_cfg']
