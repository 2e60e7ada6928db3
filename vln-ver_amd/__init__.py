"""MI355X-native 2D->3D volumetric lifting path for VLN-VER."""
