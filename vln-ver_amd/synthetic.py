"""Synthetic inputs for the 2D->3D lifting path.

Everything here is deterministic from integer seeds through
``numpy.random.default_rng`` so that the golden-vector generator (which runs
next to the reference, in the build container only) and the parity tests / the
bench (which run anywhere) see bit-identical inputs without shipping them.

Reference input contract being imitated:
  * features  ``(6, 1, 196, 768) f32`` per viewpoint -- what
    ``VoxelFormer.get_image_feature`` hands to the head
    (bevformer/detectors/voxelformer.py:317-325, CLS token dropped);
  * ``world2pixel/<scan>.json``: dict ``'<vp>_i1_<c>' -> 4x4`` nested list and
    ``scanvp2cord.pkl``: dict ``'<scan>_<vp>' -> [ox, oy, oz]``
    (bevformer/modules/voxel_encoder.py:121-135).
"""
import json
import math
import os
import pickle

import numpy as np

NUM_CAMS = 6
FEAT_HW = 14
FEAT_TOKENS = FEAT_HW * FEAT_HW
IMG_W = 1280.0   # voxel_encoder.py:179
IMG_H = 1024.0   # voxel_encoder.py:180
DEFAULT_ORIGIN = (1.25, -0.5, 1.4)
VOCC_PC_RANGE = (-6.0, -6.0, -1.5, 6.0, 6.0, 2.0)   # vocc.py:9


def camera_rig(origin=DEFAULT_ORIGIN, num_cams=NUM_CAMS, fx=1075.0, fy=1075.0,
               cx=640.0, cy=512.0):
    """Six pinhole cameras at ``origin``, headings 0,60,...,300 degrees.

    Returns ``world2pixel`` as float64 ``[num_cams, 4, 4]`` = K4 . [R | -R o]
    with camera x right, y down, z forward and world z up.
    """
    o = np.asarray(origin, dtype=np.float64)
    k4 = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                  dtype=np.float64)
    mats = []
    for c in range(num_cams):
        th = math.radians(360.0 / num_cams * c)
        fwd = np.array([math.sin(th), math.cos(th), 0.0])
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        rot = np.stack([right, -up, fwd])
        ext = np.eye(4)
        ext[:3, :3] = rot
        ext[:3, 3] = -rot @ o
        mats.append(k4 @ ext)
    return np.stack(mats)


def viewpoint_origins(batch, seed=1, base=DEFAULT_ORIGIN):
    """Origin per viewpoint: ``base`` for element 0, then U(-2,2) m jitter in x,y."""
    rng = np.random.default_rng(seed)
    out = np.tile(np.asarray(base, dtype=np.float64), (batch, 1))
    if batch > 1:
        jit = rng.uniform(-2.0, 2.0, size=(batch, 2))
        jit[0] = 0.0
        out[:, :2] += jit
    return out


def camera_batch(batch, seed=1):
    """``(world2pixel f32[B,6,4,4], origin f32[B,3])`` for ``batch`` viewpoints."""
    org = viewpoint_origins(batch, seed)
    w2p = np.stack([camera_rig(o) for o in org])
    return w2p.astype(np.float32), org.astype(np.float32)


def vit_features(batch, seed=0, channels=768, tokens=FEAT_TOKENS, num_cams=NUM_CAMS):
    """N(0,1) features ``f32[B, num_cams, tokens, channels]``."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, num_cams, tokens, channels)).astype(np.float32)


def write_camera_files(root, scan, vps, w2p, origins):
    """Write the rig in the two on-disk formats ``point_sampling`` reads.

    ``root`` takes the place of the reference's literal ``'path to'`` prefix.
    """
    os.makedirs(os.path.join(root, 'camera_parameters', 'world2pixel'), exist_ok=True)
    table, cords = {}, {}
    for b, vp in enumerate(vps):
        for c in range(w2p.shape[1]):
            table['%s_i1_%d' % (vp, c)] = np.asarray(w2p[b, c], dtype=np.float64).tolist()
        cords['%s_%s' % (scan, vp)] = [float(v) for v in origins[b]]
    with open(os.path.join(root, 'camera_parameters', 'world2pixel', scan + '.json'), 'w') as f:
        json.dump(table, f)
    pkl = os.path.join(root, 'scanvp2cord.pkl')
    old = {}
    if os.path.exists(pkl):
        with open(pkl, 'rb') as f:
            old = pickle.load(f)
    old.update(cords)
    with open(pkl, 'wb') as f:
        pickle.dump(old, f)


def _std_for(key, shape):
    if key.endswith('sampling_offsets.bias'):
        return 2.0          # pixels; mixes in-map and out-of-map samples
    if 'embed' in key:
        return 1.0          # nn.Embedding / cams_embeds / level_embeds are N(0,1)
    if len(shape) >= 2:
        fan = float(np.prod(shape)) / float(shape[0])
        return 1.0 / math.sqrt(fan)
    return 0.1


def seeded_state(named_shapes, seed):
    """Deterministic parameter values for a state-dict.

    ``named_shapes``: iterable of ``(key, shape)``. Keys are visited in sorted
    order; each draws ``standard_normal(shape)`` from one PCG64 stream. 1-D
    keys that look like LayerNorm / norm scales are centred on 1.
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in sorted((k, tuple(s)) for k, s in named_shapes):
        val = rng.standard_normal(shape) * _std_for(key, shape)
        if len(shape) == 1 and key.endswith('.weight'):
            val = 1.0 + val
        out[key] = val.astype(np.float32)
    return out


def load_seeded(module, seed):
    """Fill ``module``'s parameters/buffers in place from :func:`seeded_state`."""
    import torch
    sd = module.state_dict()
    vals = seeded_state([(k, v.shape) for k, v in sd.items() if v.dtype.is_floating_point], seed)
    with torch.no_grad():
        for k, v in sd.items():
            if k in vals:
                v.copy_(torch.from_numpy(vals[k]))
    return module


def detection_gt(seed=31, num_gt=5, num_classes=17):
    """Synthetic ground truth for the detection losses: boxes [G,9] = (cx,cy,cz,w,l,h,yaw,vx,vy)
    inside the vocc.py range, labels in [0, num_classes)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform([-5, -5, -1.2], [5, 5, 1.7], (num_gt, 3))
    d = rng.uniform(0.3, 2.0, (num_gt, 3))
    yaw = rng.uniform(-3.1, 3.1, (num_gt, 1))
    boxes = np.concatenate([c, d, yaw, np.zeros((num_gt, 2))], 1).astype(np.float32)
    labels = rng.integers(0, num_classes, num_gt).astype(np.int64)
    return boxes, labels
