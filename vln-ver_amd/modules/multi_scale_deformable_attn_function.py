"""Autograd wrappers of the core sampling op -- same names, call signature and gradient
contract as the reference's bevformer/modules/multi_scale_deformable_attn_function.py:15-163,
but bound to libver_hip.so (ver_msda_forward / ver_msda_backward) instead of mmcv's `_ext`."""
from ..hipops import (MultiScaleDeformableAttnFunction_fp16,  # noqa: F401
                      MultiScaleDeformableAttnFunction_fp32)
