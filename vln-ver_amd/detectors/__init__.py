from .voxelformer import VoxelFormer, bbox3d2result  # noqa: F401
