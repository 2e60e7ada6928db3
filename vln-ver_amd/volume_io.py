"""On-disk formats either side of the lifting path ("next" row 3 of SURVEY.md 8f).

* input feature store (detectors/voxelformer.py:317-325): key ``<scan>_<vp>_i1_<deg>`` ->
  ``(1,197,768)`` ViT tokens, CLS at index 0 (dropped);
* exported volumes (dense_heads/voxelformer_occupancy_head.py:627-638, ``getbev``): key
  ``<scan>_<vp>`` -> float64, gzip, ``bev_embed.view(1,768,Z,H,W).squeeze()`` (the reference's raw
  reinterpretation of the query-major buffer) -- what the downstream VLN agent consumes.

The reference uses HDF5 through h5py.  When h5py is importable the same layout is read/written;
this image has no h5py, so a directory of ``<key>.npy`` files with identical keys, dtypes and
shapes is the portable fallback (chosen by file extension: ``*.hdf5``/``*.h5`` vs a directory)."""
import os

import numpy as np


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def _is_hdf5(path):
    return path.endswith(('.hdf5', '.h5'))


class VolumeWriter:
    def __init__(self, path):
        self.path = path
        if _is_hdf5(path) and _h5py() is None:
            raise RuntimeError('h5py is not installed: give a directory path to write <key>.npy files')
        if not _is_hdf5(path):
            os.makedirs(path, exist_ok=True)

    def write(self, key, voxel_embed, bev_zhw, embed_dims):
        """voxel_embed: one sample's [Nq, C] encoder output (torch tensor or array)."""
        z, h, w = bev_zhw
        arr = voxel_embed.detach().float().cpu().numpy() if hasattr(voxel_embed, 'detach') else np.asarray(voxel_embed)
        vol = np.ascontiguousarray(arr).reshape(embed_dims, z, h, w).astype(np.float64)   # raw view, head:634
        if _is_hdf5(self.path):
            h5 = _h5py()
            with h5.File(self.path, 'a' if os.path.exists(self.path) else 'w') as f:
                if key in f:
                    del f[key]
                f.create_dataset(key, vol.shape, dtype='float', compression='gzip')[...] = vol
        else:
            np.save(os.path.join(self.path, key + '.npy'), vol)
        return vol


def read_volume(path, key):
    if _is_hdf5(path):
        h5 = _h5py()
        if h5 is None:
            raise RuntimeError('h5py is not installed')
        with h5.File(path, 'r') as f:
            return f[key][...]
    return np.load(os.path.join(path, key + '.npy'))


class FeatureStore:
    """Six views of a viewpoint -> ``(6, 1, 196, 768) f32`` (CLS dropped), cached like the
    reference's ``self.vitfeat`` dict."""

    def __init__(self, path, elevation='i1', num_cams=6):
        self.path, self.elevation, self.num_cams = path, elevation, num_cams
        self._cache = {}

    def _get(self, key):
        if key not in self._cache:
            if _is_hdf5(self.path):
                h5 = _h5py()
                if h5 is None:
                    raise RuntimeError('h5py is not installed')
                with h5.File(self.path, 'r') as f:
                    self._cache[key] = f[key][...].astype(np.float32)
            else:
                self._cache[key] = np.load(os.path.join(self.path, key + '.npy')).astype(np.float32)
        return self._cache[key]

    def viewpoint(self, sample_idx):
        views = [self._get('%s_%s_%d' % (sample_idx, self.elevation, deg))[:, 1:, :] for deg in range(self.num_cams)]
        return np.stack(views)                                   # (6, 1, 196, 768)
