from .voxelformer_occupancy_head import VoxelFormerOccupancyHead  # noqa: F401
