"""Camera parameters of a viewpoint, in the reference's on-disk formats.

The reference re-opens ``path to/camera_parameters/world2pixel/<scan>.json`` and
``path to/scanvp2cord.pkl`` on EVERY forward (voxel_encoder.py:121-135).  This store reads each
file once, caches the parsed tables and hands out fp32 arrays; the root directory stands for
the reference's literal ``'path to'`` placeholder (env ``VER_CAMERA_ROOT``)."""
import json
import os
import pickle

import numpy as np

_DEFAULT_ROOT = 'path to'


class CameraStore:
    def __init__(self, root=None, num_cams=6, elevation='i1'):
        self.root = root or os.environ.get('VER_CAMERA_ROOT', _DEFAULT_ROOT)
        self.num_cams = num_cams
        self.elevation = elevation
        self._scans = {}
        self._cords = None

    def _scan(self, scan):
        if scan not in self._scans:
            path = os.path.join(self.root, 'camera_parameters', 'world2pixel', scan + '.json')
            with open(path, 'r') as f:
                self._scans[scan] = json.load(f)
        return self._scans[scan]

    def _origins(self):
        if self._cords is None:
            with open(os.path.join(self.root, 'scanvp2cord.pkl'), 'rb') as f:
                self._cords = pickle.load(f)
        return self._cords

    def lookup(self, sample_idx):
        """'<scan>_<vp>' -> (world2pixel f32[num_cams,4,4], origin f32[3]); KeyError /
        FileNotFoundError propagate exactly as in the reference."""
        scan, vp = sample_idx.split('_')
        table = self._scan(scan)
        mats = [table['%s_%s_%d' % (vp, self.elevation, deg)] for deg in range(self.num_cams)]
        origin = self._origins()[scan + '_' + vp]
        return np.asarray(mats, dtype=np.float32), np.asarray(origin, dtype=np.float32)

    def batch(self, img_metas):
        """list of per-sample meta dicts (or 1-element lists of them) -> stacked arrays.
        A meta may carry 'world2pixel' and 'origin' directly, bypassing the files."""
        w2p, org = [], []
        for meta in img_metas:
            if isinstance(meta, (list, tuple)):
                meta = meta[0]
            if 'world2pixel' in meta and 'origin' in meta:
                w2p.append(np.asarray(meta['world2pixel'], dtype=np.float32))
                org.append(np.asarray(meta['origin'], dtype=np.float32))
            else:
                m, o = self.lookup(meta['sample_idx'])
                w2p.append(m)
                org.append(o)
        return np.stack(w2p), np.stack(org)
