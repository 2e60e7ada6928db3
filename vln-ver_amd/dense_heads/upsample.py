"""Coarse-to-fine upsampling of the voxel volume: the three ``ConvTranspose3d(768,768,(3,5,5),
stride=(1,2,2), padding=(2,4,4), dilation=(2,2,2), output_padding=(0,1,1))`` layers of the
reference head (dense_heads/voxelformer_occupancy_head.py:251-258, applied at :560), computed
on the *even lattice*.

With stride 2 AND dilation 2 in H/W the output index is ``o = 2*(i + k - 2)``: only even output
rows/cols ever receive data, every odd one equals the bias exactly (verified
bit-exactly against the reference).  So layer l+1 sees an input that is its predecessor's bias
vector on 3/4 of the positions.  Writing E_l for the data lattice of layer l's input:

    E_{l+1}[z,m,n] = b_l + sum_{a,b,c} K_l[a,b,c]^T X_l[z-2+2a, m-2+b, n-2+c]
    X_l[y,x]       = E_l[y/2,x/2] if y,x both even else b_{l-1}          (l >= 1)

* data taps: per output parity class (m%2, n%2) only the taps with (m%2+b), (n%2+c) even hit the
  data lattice -> four small correlations over E_l (3x3, 3x2, 2x3, 2x2 taps x 3 in z) = im2col +
  one GEMM each.  Useful MACs: 79.6 + 79.6 + 318.5 GFLOP instead of 79.6 + 318.5 + 1274 (3.5x);
* constant taps: sum of (K_l[tap]^T b_{l-1}) over the in-bounds non-data taps -- a [positions,75]
  0/1 pattern matrix (fixed by the geometry) times a [75,768] matrix.

Everything is plain differentiable torch (GEMMs go to hipBLASLt/MFMA on the GPU), so autograd
provides the backward.  ``full_volume`` scatters the last lattice into the dense
``[B,C,Z,8H,8W]`` tensor the reference's raw ``.view`` expects.
"""
import os

import numpy as np
import torch

from . import _cpu_algebra
from ._cpu_algebra import (ZS_PLAIN, ZS_PLANAR, ZS_PLANAR_SPLIT, ZS_SPLIT, plain_to_planar, plain_to_planar_zs,  # noqa: F401
                           plain_to_zs, planar_to_plain, planar_zs_to_plain, zs_to_plain)

KERNEL = (3, 5, 5)
GEOM = dict(stride=(1, 2, 2), padding=(2, 4, 4), dilation=(2, 2, 2), output_padding=(0, 1, 1))
_HIP_DTYPES = (torch.float32, torch.bfloat16)


class _HipAlgebra:
    """The data-movement operations of the lattice algebra on the HIP kernels (GPU tensors; fp32 / bf16 -- other dtypes of a
    GPU tensor, e.g. an fp64 check on the device, go through the torch formulation).  Same signatures as ``_cpu_algebra``."""

    @staticmethod
    def corr_weight(weight, dtype):
        if weight.dtype == torch.float32 and dtype in _HIP_DTYPES:
            from ..hipops import convt_weight_taps
            return convt_weight_taps(weight, dtype)                       # one LDS-tiled transpose each way
        return _cpu_algebra.corr_weight(weight, dtype)

    @staticmethod
    def im2col(e, taps):
        from ..hipops import lattice_im2col
        return lattice_im2col(e.contiguous(), taps)

    @staticmethod
    def gather27(e, planar, a_mat, ci, hc, wc, taps, offs):
        from ..hipops import lattice_gather
        lattice_gather(e.contiguous(), a_mat, taps, offs, (hc, wc), 1 if planar else 0)

    @staticmethod
    def scatter27(d_a, planar, shape, ci, hc, wc, taps, offs):
        from ..hipops import lattice_scatter
        return lattice_scatter(d_a, d_a.new_empty(shape), taps, offs, (hc, wc), 1 if planar else 0)

    @staticmethod
    def gather_z4(e, layout, a_mat, taps, offs, ci, hc, wc, const=None):
        """``const`` = (table, column offsets): the kernel also writes the constant-pattern blocks of the four parity
        classes.  -> True when it did."""
        from ..hipops import lattice_gather
        if const is not None and e.dtype in _HIP_DTYPES:
            lattice_gather(e.contiguous(), a_mat, taps, offs, (hc, wc), layout, row_z=2, const_rows=const[0], const_offset=const[1])
            return True
        lattice_gather(e.contiguous(), a_mat, taps, offs, (hc, wc), layout, row_z=2)
        return False

    @staticmethod
    def scatter_z4(d_a, layout, shape, taps, offs, ci, hc, wc):
        from ..hipops import lattice_scatter
        return lattice_scatter(d_a, d_a.new_empty(shape), taps, offs, (hc, wc), layout, row_z=2)

    @staticmethod
    def channels_last(x0, dt):
        b, c, z, h, w = x0.shape
        pos = z * h * w
        if dt in _HIP_DTYPES:
            vx = 4 if dt == torch.bfloat16 else 2
            for wd in range(min(pos, 120), 0, -1):        # rows of the flattened positions: 8-byte multiples, tile <= 64 KB
                if pos % wd == 0 and wd % vx == 0:
                    return _ChannelsLast.apply(x0, dt, (pos // wd, wd))
        return _cpu_algebra.channels_last(x0, dt)


def _algebra(t):
    """THE device switch of this module: the HIP kernels for a GPU tensor, the torch formulation (``_cpu_algebra``, what the
    CPU suite checks in fp64) for a CPU tensor.  Nothing here is chosen by "extension missing": hipops raises."""
    return _HipAlgebra if t.is_cuda else _cpu_algebra


def _on_hip(t):
    return _algebra(t) is _HipAlgebra


def is_reference_geometry(conv):
    return (tuple(conv.kernel_size) == KERNEL and tuple(conv.stride) == GEOM['stride'] and
            tuple(conv.padding) == GEOM['padding'] and tuple(conv.dilation) == GEOM['dilation'] and
            tuple(conv.output_padding) == GEOM['output_padding'] and conv.groups == 1)


def _corr_weight(weight, dtype):
    """ConvTranspose weight [Ci,Co,3,5,5] -> correlation taps as ONE contiguous [75, Ci, Co]
    tensor in the compute dtype, tap index = (a*5+b)*5+c, K[a,b,c] = Wt[:, :, 2-a, 4-b, 4-c]."""
    return _algebra(weight).corr_weight(weight, dtype)


_IDX_CACHE = {}


def _tap_index(tap_ids, device):
    key = (tuple(tap_ids), str(device))
    if key not in _IDX_CACHE:
        _IDX_CACHE[key] = torch.tensor(tap_ids, dtype=torch.long, device=device)
    return _IDX_CACHE[key]


def _im2col(e, taps):
    """e [B,Z,H,W,C] channels-last; taps: list of (dz,dy,dx) offsets (zero outside the lattice)
    -> [B*Z*H*W, len(taps)*C] (ver_lattice_im2col / ver_lattice_col2im on the GPU)."""
    return _algebra(e).im2col(e, taps)


_PATTERN_CACHE = {}


def _constant_pattern(z, h_in, w_in, device, dtype):
    """[Z*h_in*w_in, 75] 0/1: tap (a,b,c) of output position (z,m,n) lands in-bounds on a
    NON-data position of a full-resolution input of size (Z, h_in, w_in) whose data lattice is the
    even rows/cols."""
    key = (z, h_in, w_in, str(device), dtype)
    if key not in _PATTERN_CACHE:
        zz, mm, nn = np.meshgrid(np.arange(z), np.arange(h_in), np.arange(w_in), indexing='ij')
        pat = np.zeros((z, h_in, w_in, 3, 5, 5), dtype=np.float32)
        for a in range(3):
            iz = zz - 2 + 2 * a
            for b in range(5):
                iy = mm - 2 + b
                for c in range(5):
                    ix = nn - 2 + c
                    inb = (iz >= 0) & (iz < z) & (iy >= 0) & (iy < h_in) & (ix >= 0) & (ix < w_in)
                    data = (iy % 2 == 0) & (ix % 2 == 0)
                    pat[:, :, :, a, b, c] = inb & ~data
        _PATTERN_CACHE[key] = torch.from_numpy(pat.reshape(z * h_in * w_in, 75)).to(device=device,
                                                                                   dtype=dtype)
    return _PATTERN_CACHE[key]


def _bias_through_taps(prev_bias, k):
    """prev_bias [Ci] @ k [75,Ci,Co] -> [75,Co] (the bias-valued odd positions of the input seen through every tap).  As a
    batched [1 x Ci] x [Ci x Co] product: ``torch.matmul`` of a vector with a 3-D tensor first makes a TRANSPOSED contiguous
    copy of all 75 tap blocks (88 MB at 768 channels, 190 us -- a fifth of the weight-side time of a one-viewpoint step)."""
    t, ci, _ = k.shape
    return torch.bmm(prev_bias.view(1, 1, ci).expand(t, 1, ci).contiguous(), k).squeeze(1)


def _layer0(e, k, bias):
    """All 75 taps hit data.  e [B,Z,H,W,C] channels-last -> E_1 [B,Z,H,W,Co]."""
    b, z, h, w, c = e.shape
    taps = [(2 * a - 2, bb - 2, cc - 2) for a in range(3) for bb in range(5) for cc in range(5)]
    a_mat = _im2col(e, taps)
    out = torch.addmm(bias, a_mat, k.reshape(75 * c, -1))
    return out.view(b, z, h, w, -1)


# ------------------------------------------------------------------------------------------------
# Parity-class layer (layers 1 and 2 of the stack) as ONE autograd Function.
#
# All four output classes (pm, pn) read taps of the same 3x3x3 neighbourhood (dz in {-2,0,2}, dy, dx
# in {-1,0,1}) of the input lattice: class pn=1 only dx in {0,1}, pm=1 only dy in {0,1}.  So ONE
# 27-tap matrix A [B*Z*H*W, Kt] serves the four classes (27 instead of 75 tap blocks are written /
# read back).  The (dy, dx) pairs are ordered in four groups
#     G1 = (dy-, dx-) | G2 = (dy+, dx-) | G3 = (dy+, dx+) | G4 = (dy-, dx+)      (- : -1, + : {0, 1})
# so that EVERY class is one contiguous column range: (0,0) = G1..G4, (1,0) = G2 G3, (1,1) = G3,
# (0,1) = G3 G4.  Four 80-wide constant blocks P_class = [75 0/1 pattern columns | 1 | 0 0 0 0] sit
# between the groups, with weight rows (K[tap]^T prev_bias | bias): the bias-valued odd positions of
# the input and the layer bias ride in the same GEMM.  Layout of a row (C = channels):
#     [P00 | G1 | P10 | G2 | P11 | G3 | G4 | P01]
#   (0,0): P00..G4 (P10, P11 meet zero weight rows)      (1,0): P10 G2 P11 G3 (P11 -> zero rows)
#   (1,1): P11 G3                                          (0,1): G3 G4 P01
# Every class GEMM writes its own contiguous output plane: the result is PLANAR [4,B,Z,H,W,Co]
# (plane 2pm+pn = positions (2y+pm, 2x+pn) of the (2H, 2W) lattice) and is consumed as such by the
# next layer's gather kernel and by occ_proj -- the lattice is never interleaved.
# width of a constant block: 75 pattern columns + the ones column, padded to 96 so that the lo | hi pair of a Z = 4 layer is 192
# = 3 x 64 columns and every segment of the K axis starts on a multiple of 64 -- what the implicit operand loaders want
# (ver_gemm_nn_segments: whole 32-column phases; ver_wgrad_tn_segments: a wave's 64-column piece inside ONE segment)
_PW = 96
_CLASSES = _cpu_algebra.CLASSES


def _group_of(dyi, dxi):
    yp, xp = dyi > 0, dxi > 0
    return 0 if (not yp and not xp) else 1 if (yp and not xp) else 2 if (yp and xp) else 3


def _block_order():
    """27 blocks (dxi, dyi, dzi) in column order: groups G1..G4, inside a group (dx, dy, dz)."""
    blocks = [(dxi, dyi, dzi) for dxi in range(3) for dyi in range(3) for dzi in range(3)]
    return sorted(blocks, key=lambda b: (_group_of(b[1], b[0]), b))


_ORDER = _block_order()
_GROUP_START = [next(i for i, b in enumerate(_ORDER) if _group_of(b[1], b[0]) == g) for g in range(4)] + [27]
# number of constant blocks in front of block position i: P00 before G1, P10 before G2, P11 before G3
_CONST_BEFORE = [1 + (i >= _GROUP_START[1]) + (i >= _GROUP_START[2]) for i in range(27)]


def _block_offset(t, c):
    """column offset of the t-th block (position in _ORDER)."""
    return _PW * _CONST_BEFORE[t] + t * c


def _const_offset(pm, pn, c):
    g = _GROUP_START
    return {(0, 0): 0, (1, 0): _PW + g[1] * c, (1, 1): 2 * _PW + g[2] * c, (0, 1): 3 * _PW + 27 * c}[(pm, pn)]


def _tap27():
    """(dz, dy, dx) of the t-th block."""
    return [(2 * dzi - 2, dyi - 1, dxi - 1) for dxi, dyi, dzi in _ORDER]


def _class_tap_id(pm, pn, t):
    """id in the 75-tap correlation kernel of the t-th block for class (pm, pn), or None."""
    dxi, dyi, dzi = _ORDER[t]
    bb, cc = 2 * dyi - pm, 2 * dxi - pn
    if 0 <= bb < 5 and 0 <= cc < 5:
        return (dzi * 5 + bb) * 5 + cc
    return None


_LAYER_PLAN = {}


def _layer_plan(ci, device):
    """Per class: list of (col_start, col_end, row index into the stacked weight rows
    [75*ci data | 4*80 own-constant | 3*80 dummy zero])."""
    key = (ci, str(device))
    if key in _LAYER_PLAN:
        return _LAYER_PLAN[key]
    kt = 27 * ci + 4 * _PW
    n_data = 75 * ci
    dummy = [n_data + 4 * _PW]                      # next free dummy row (mutable)

    def seg_rows(pm, pn, seg):
        kind, val = seg
        if kind == 'b':
            tid = _class_tap_id(pm, pn, val)
            assert tid is not None
            return np.arange(tid * ci, (tid + 1) * ci)
        if val == (pm, pn):                          # own constant block
            p = _CLASSES.index(val)
            return np.arange(n_data + p * _PW, n_data + (p + 1) * _PW)
        r = np.arange(dummy[0], dummy[0] + _PW)      # foreign constant block: zero rows
        dummy[0] += _PW
        return r
    b = lambda lo, hi: [('b', t) for t in range(lo, hi)]
    g = _GROUP_START
    layout = {
        (0, 0): [[('c', (0, 0))] + b(g[0], g[1]) + [('c', (1, 0))] + b(g[1], g[2]) + [('c', (1, 1))] + b(g[2], g[4])],
        (1, 0): [[('c', (1, 0))] + b(g[1], g[2]) + [('c', (1, 1))] + b(g[2], g[3])],
        (1, 1): [[('c', (1, 1))] + b(g[2], g[3])],
        (0, 1): [b(g[2], g[4]) + [('c', (0, 1))]],
    }
    plan = {}
    for (pm, pn), ranges in layout.items():
        out = []
        for segs in ranges:
            first = segs[0]
            c0 = _block_offset(first[1], ci) if first[0] == 'b' else _const_offset(*first[1], ci)
            rows = np.concatenate([seg_rows(pm, pn, sg) for sg in segs])
            out.append((c0, c0 + len(rows), torch.from_numpy(rows.astype(np.int64)).to(device)))
        plan[(pm, pn)] = out
    total_rows = dummy[0]
    assert total_rows == n_data + 7 * _PW
    _LAYER_PLAN[key] = (plan, kt, total_rows)
    return _LAYER_PLAN[key]


_CLASS_PATTERN = {}


def _class_patterns(z, h, w, device, dtype):
    """[4][Z*H*W, 80]: constant-block columns of class p for an input lattice (Z,H,W)."""
    key = (z, h, w, str(device), dtype)
    if key not in _CLASS_PATTERN:
        full = _constant_pattern(z, 2 * h, 2 * w, device, dtype).view(z, 2 * h, 2 * w, 75)
        pats = []
        for pm, pn in _CLASSES:
            p = full.new_zeros(z, h, w, _PW)
            p[..., :75] = full[:, pm::2, pn::2]
            p[..., 75] = 1
            pats.append(p.view(z * h * w, _PW))
        _CLASS_PATTERN[key] = pats
    return _CLASS_PATTERN[key]


def _gather27(e, planar, a_mat, ci, hc, wc):
    _algebra(e).gather27(e, planar, a_mat, ci, hc, wc, _tap27(), [_block_offset(t, ci) for t in range(27)])


def _scatter27(d_a, planar, shape, ci, hc, wc):
    """adjoint of _gather27: gradient of the source lattice (plain or planar like the source)."""
    return _algebra(d_a).scatter27(d_a, planar, shape, ci, hc, wc, _tap27(), [_block_offset(t, ci) for t in range(27)])


class _LatticeLayer(torch.autograd.Function):

    @staticmethod
    def forward(ctx, e, k, bias, prev_bias, planar):
        """e: input lattice, plain [B,Z,H,W,C] or planar [4,B,Z,H/2,W/2,C]; k [75,Ci,Co] correlation
        taps; -> planar output [4,B,Z,H,W,Co]."""
        if planar:
            _, b, z, hh, wh, ci = e.shape
            hc, wc = 2 * hh, 2 * wh
        else:
            b, z, hc, wc, ci = e.shape
        co = k.shape[-1]
        dt = e.dtype
        plan, kt, total_rows = _layer_plan(ci, e.device)
        m = b * z * hc * wc
        a_mat = e.new_empty(m, kt)
        _gather27(e, planar, a_mat, ci, hc, wc)
        pats = _class_patterns(z, hc, wc, e.device, dt)
        a3 = a_mat.view(b, z * hc * wc, kt)
        for (pm, pn), pat in zip(_CLASSES, pats):
            o = _const_offset(pm, pn, ci)
            a3[:, :, o:o + _PW] = pat
        # stacked weight rows: 75 tap blocks | per class (K^T prev_bias | bias | 0) | zero rows
        v = _bias_through_taps(prev_bias.to(dt), k)                               # [75, Co]
        vaug = torch.cat([v, bias.to(dt)[None], v.new_zeros(_PW - 76, co)])
        rows = torch.cat([k.reshape(75 * ci, co), vaug, vaug, vaug, vaug, v.new_zeros(3 * _PW, co)])
        assert rows.shape[0] == total_rows
        out = e.new_empty(4, m, co)
        for p, cls in enumerate(_CLASSES):
            for i, (c0, c1, ridx) in enumerate(plan[cls]):
                w = rows.index_select(0, ridx)
                if i == 0:
                    torch.mm(a_mat[:, c0:c1], w, out=out[p])
                else:
                    torch.addmm(out[p], a_mat[:, c0:c1], w, out=out[p])
        ctx.save_for_backward(a_mat, rows, k, prev_bias)
        ctx.geom = (planar, tuple(e.shape), b, z, hc, wc, ci, co)
        return out.view(4, b, z, hc, wc, co)

    @staticmethod
    def backward(ctx, grad_out):
        a_mat, rows, k, prev_bias = ctx.saved_tensors
        planar, e_shape, b, z, hc, wc, ci, co = ctx.geom
        plan, kt, total_rows = _layer_plan(ci, a_mat.device)
        dt = a_mat.dtype
        m = a_mat.shape[0]
        g = grad_out.contiguous().view(4, m, co)
        d_a = a_mat.new_empty(m, kt)
        d_a[:, kt - _PW:] = 0                                   # P01 is outside class (0,0)'s range
        d_rows = rows.new_empty(total_rows, co)
        first = True
        for p, cls in enumerate(_CLASSES):
            for c0, c1, ridx in plan[cls]:
                w = rows.index_select(0, ridx)
                if first:                                       # class (0,0): initialises every tap block
                    torch.mm(g[p], w.t(), out=d_a[:, c0:c1])
                    first = False
                else:
                    torch.addmm(d_a[:, c0:c1], g[p], w.t(), out=d_a[:, c0:c1])
                d_rows.index_copy_(0, ridx, rows_tn(a_mat[:, c0:c1], g[p]))
        d_e = _scatter27(d_a, planar, e_shape, ci, hc, wc)
        n_data = 75 * ci
        d_k = d_rows[:n_data].view(75, ci, co)
        d_vaug = d_rows[n_data:n_data + 4 * _PW].view(4, _PW, co).sum(0, dtype=torch.float32 if dt != torch.float64
                                                                      else torch.float64)
        d_v = d_vaug[:75].to(dt)
        d_bias = d_vaug[75]
        # v = prev_bias @ k
        pb = prev_bias.to(dt)
        d_k = torch.addcmul(d_k, pb[None, :, None], d_v[:, None, :])
        d_prev = torch.bmm(k, d_v.unsqueeze(2)).sum(0).squeeze(1)
        return d_e, d_k, d_bias.to(prev_bias.dtype), d_prev.to(prev_bias.dtype), None


def _layer_lattice(e, k, bias, prev_bias, planar):
    """e = data lattice (plain [B,Z,H,W,C] or planar) of a full input that equals ``prev_bias`` off
    the lattice -> planar output lattice [4,B,Z,H,W,Co] (H, W = combined size of the input)."""
    return _LatticeLayer.apply(e, k, bias, prev_bias, planar)



# ------------------------------------------------------------------------------------------------
# Z = 4 (vocc.py: bev_z = 4).  The z taps of the stack are dz in {-2, 0, +2}: with four z-layers
# every output layer has exactly TWO in-range taps -- z = 0,1 read the input layers (z, z+2), z = 2,3
# read (z-2, z) -- and both halves read the SAME pair (zl, zl+2), zl = z & 1.  So the tap matrix needs
# only the rows (b, zl, y, x) (half of them) and 2 instead of 3 z blocks per (dy, dx); the two output
# halves come out of ONE GEMM side by side, A [B*2*H*W, K] x [W_lo | W_hi] [K, 2*Co] with
# W_lo = K[dz = 0], K[dz = +2] and W_hi = K[dz = -2], K[dz = 0].  A third fewer FLOPs in all three
# layers, a third of the tap-matrix traffic, N = 1536 instead of 768.  Lattices are kept Z-SPLIT,
# [B, 2 (zl), H, W, 2 (zh), C] (z = 2*zh + zl): exactly the GEMM output [rows, 2*Co].
_PW2 = 2 * _PW                                                         # lo | hi constant blocks


def lattice_to_plain(e):
    """lattice as ``upsample_lattice`` returns it (plain, planar or planar z-split) -> channels-last
    [B,Z,H,W,C]"""
    if e.dim() == 7:
        return planar_zs_to_plain(e)
    if e.dim() == 6:
        return planar_to_plain(e)
    return e


_to_plain, _from_plain = _cpu_algebra.to_plain, _cpu_algebra.from_plain


_CONST_ROWS4 = {}


def _const_rows_z4(ci, hc, wc, device, dtype):
    """The constant-pattern blocks of one viewpoint's rows of a Z = 4 lattice layer, as the gather kernel copies them:
    [2*hc*wc, 4 classes, 2*_PW] = per class [P_lo | P_hi], and their column offsets."""
    key = (ci, hc, wc, str(device), dtype)
    if key not in _CONST_ROWS4:
        pats = _class_patterns(4, hc, wc, device, dtype)
        blocks = []
        for pat in pats:
            halves = pat.view(2, 2 * hc * wc, _PW)                  # output z = zl (lower), zl + 2 (upper)
            blocks.append(torch.cat([halves[0], halves[1]], 1))
        table = torch.stack(blocks, 1).contiguous()                 # [2hw, 4, 2*_PW]
        _CONST_ROWS4[key] = (table, [_const_offset4(pm, pn, ci) for pm, pn in _CLASSES])
    return _CONST_ROWS4[key]


def _gather_z4(e, layout, a_mat, taps, offs, ci, hc, wc, with_const=False):
    """rows (b, zl, y, x); tap (dz in {0,2}, dy, dx) reads input layer zl + dz.  ``with_const``: ask for the constant-pattern
    blocks of the four parity classes in the same pass.  -> True when they were written (else the caller fills them)."""
    const = _const_rows_z4(ci, hc, wc, e.device, e.dtype) if with_const else None
    return _algebra(e).gather_z4(e, layout, a_mat, taps, offs, ci, hc, wc, const)


def _scatter_z4(d_a, layout, shape, taps, offs, ci, hc, wc):
    return _algebra(d_a).scatter_z4(d_a, layout, shape, taps, offs, ci, hc, wc)


_L0Z4 = {}


def _layer0_z4_plan(ci, device):
    """blocks (bb, cc, j) of the 5x5x2 neighbourhood; row indices into k.reshape(75*ci, co) of the taps
    feeding the lower (a = 1 + j) and the upper (a = j) output half."""
    key = (ci, str(device))
    if key not in _L0Z4:
        taps, lo, hi = [], [], []
        for bb in range(5):
            for cc in range(5):
                for j in range(2):
                    taps.append((2 * j, bb - 2, cc - 2))
                    lo.append(np.arange(ci) + (((1 + j) * 5 + bb) * 5 + cc) * ci)
                    hi.append(np.arange(ci) + ((j * 5 + bb) * 5 + cc) * ci)
        t = lambda a: torch.from_numpy(np.concatenate(a).astype(np.int64)).to(device)
        _L0Z4[key] = (taps, [i * ci for i in range(50)], t(lo), t(hi))
    return _L0Z4[key]


_BLOCK_OFFSETS = {}


def _block_offsets(kind, ci, co, device):
    """int64 [75, 2] for ``ver_convt_weight_backward_blocks``: element offsets, inside the [rows, 2 Co] weight-gradient
    buffer of a z-split layer, of the [Ci x Co] block that holds tap t's "lower half" / "upper half" gradient (-1: none).
    kind 'l0': layer 0 (50 blocks (bb, cc, j) in a row); 'lat': the class-stacked buffer of ``_LatticeLayerZ4``."""
    key = (kind, ci, co, str(device))
    if key not in _BLOCK_OFFSETS:
        off = np.full((75, 2), -1, dtype=np.int64)
        ld = 2 * co

        def put(t, half, row):
            assert off[t, half] == -1
            off[t, half] = row * ld + half * co
        if kind == 'l0':
            i = 0
            for bb in range(5):
                for cc in range(5):
                    for j in range(2):
                        put(((1 + j) * 5 + bb) * 5 + cc, 0, i * ci)
                        put((j * 5 + bb) * 5 + cc, 1, i * ci)
                        i += 1
        else:
            for (pm, pn), (roff, segs) in _class_rows_z4(ci).items():
                for kind_, val, r0 in segs:
                    if kind_ == 'b':
                        dxi, dyi, j = _ORDER4[val]
                        bb, cc = 2 * dyi - pm, 2 * dxi - pn
                        put(((1 + j) * 5 + bb) * 5 + cc, 0, roff + r0)
                        put((j * 5 + bb) * 5 + cc, 1, roff + r0)
        a = np.arange(75) // 25
        assert ((off[:, 0] >= 0) == (a >= 1)).all() and ((off[:, 1] >= 0) == (a <= 1)).all()
        _BLOCK_OFFSETS[key] = torch.from_numpy(off).to(device)
    return _BLOCK_OFFSETS[key]


# Implicit tap matrix (round 6): from `_OWN_GEMM_MIN_ROWS` rows on, the forward and weight-gradient GEMMs of a bf16 lattice layer read
# their A operand straight from the lattice (ver_gemm_nn_segments / ver_wgrad_tn_segments: a tap block of a row is the
# contiguous channel vector of a neighbouring cell, fetched by the kernels' LDS-DMA) -- the tap matrix (10 GB for layer 3 at 192
# viewpoints) is neither written nor kept for the backward pass; only d(input) still goes through an explicit matrix.  Below,
# the skinny / library paths on the explicit matrix are faster.  VER_IMPLICIT_TAPS=0: explicit everywhere.
_IMPLICIT_TAPS = os.environ.get('VER_IMPLICIT_TAPS', '1') in ('1', '2', '3')
# ... and d(input) as ONE gather-form product per input half over the four class planes of the output gradient
# (ver_gemm_nn_planes: d_e[cell] = sum over (class, tap, output half) of g_class[cell - tap] W^T): no explicit d(tap matrix)
# (10 GB transient at layer 3), no ver_lattice_scatter, fp32 sums over all classes and taps rounded once.
# VER_IMPLICIT_TAPS=3 (A/B runs): d(input) through the library and the explicit matrix.
_IMPLICIT_DGRAD = os.environ.get('VER_IMPLICIT_TAPS', '1') == '1'
# VER_IMPLICIT_TAPS=2 (A/B runs): implicit forward, but the backward pass writes the tap matrix after all and takes the explicit
# weight-gradient kernel (ver_wgrad_tn)
_IMPLICIT_WGRAD = os.environ.get('VER_IMPLICIT_TAPS', '1') != '2'


def _implicit_taps(e, layout, rows, hw, ci, raw):
    """True when a layer with source lattice ``e`` (layout 0 / 2 / 3), ``rows`` = B * 2 * H * W and H * W = ``hw`` takes the
    implicit-operand kernels."""
    if not (_IMPLICIT_TAPS and _OWN_GEMM and raw is not None and _on_hip(e) and e.dtype == torch.bfloat16 and rows >= _OWN_GEMM_MIN_ROWS):
        return False
    return (e.is_contiguous() and e.numel() * 2 < 2 ** 31 - 1 and ci % 64 == 0 and ci >= 64 and 16 <= 2 * hw < 65536
            and _PW2 % 64 == 0)


_DGRAD_PLAN = {}


def _dgrad_plan(kind, ci, device):
    """d(input) of a Z = 4 layer as gather-form products: per input half j the blocks that read it -- (first row of the
    block in the layer's stacked weight matrix) as an index tensor, and the (dz, dy, dx) taps / source planes of
    ``hipops.gemm_nn_taps`` on the output gradient: block (class p, dy, dx, j) of the forward contributes
    g_p[cell - (dy, dx)][half h] W_block[:, h]^T for both output halves h (K order: block, h, co).
    kind 'l0': layer 1 (one plane, 25 (bb, cc) blocks per j); 'lat': the class-stacked layers (4 planes)."""
    key = (kind, ci, str(device))
    if key not in _DGRAD_PLAN:
        per_j = ([], [])
        if kind == 'l0':
            i = 0
            for bb in range(5):
                for cc in range(5):
                    for j in range(2):
                        per_j[j].append((0, i * ci, bb - 2, cc - 2))
                        i += 1
        else:
            for p, cls in enumerate(_CLASSES):
                roff, segs = _class_rows_z4(ci)[cls]
                for kind_, val, r0 in segs:
                    if kind_ == 'b':
                        dxi, dyi, j = _ORDER4[val]
                        per_j[j].append((p, roff + r0, dyi - 1, dxi - 1))
        # both input halves read the SAME (class, dy, dx, h) sequence of the output gradient: one operand, weights side by side
        assert [q[2:] for q in per_j[0]] == [q[2:] for q in per_j[1]] and [q[0] for q in per_j[0]] == [q[0] for q in per_j[1]]
        rows = torch.from_numpy(np.stack([np.stack([np.arange(r0, r0 + ci) for _, r0, _, _ in per_j[j]]) for j in range(2)], 1)
                                .reshape(-1).astype(np.int64)).to(device)                  # (block, j, ci)
        taps = [(2 * h, -dy, -dx) for _, _, dy, dx in per_j[0] for h in range(2)]
        planes = [p for p, _, _, _ in per_j[0] for _ in range(2)]
        _DGRAD_PLAN[key] = (rows, taps, planes, len(per_j[0]))
    return _DGRAD_PLAN[key]


def _dgrad_implicit(kind, g_planes, weights, b, hc, wc, ci, co):
    """d(input) lattice, z-split [B,2,hc,wc,2,Ci], from the output gradient ``g_planes`` bf16 [planes, B*2*hc*wc, 2 Co] and the
    layer's stacked weight matrix ``weights`` [rows, 2 Co] (``_dgrad_plan``): ONE product [M, blocks * 2 Co] x [., 2 Ci]."""
    from ..hipops import gemm_nn_taps
    m = b * 2 * hc * wc
    lat = g_planes.view(g_planes.shape[0], b, 2, hc, wc, 2, co)
    rows, taps, planes, nb = _dgrad_plan(kind, ci, g_planes.device)
    # W [(block, h, co), (j, ci)] = S[row of block (.., j) + ci, h Co + co]: the two input halves' blocks transposed, side by side
    wcat = weights.index_select(0, rows).view(nb, 2 * ci, 2 * co).transpose(1, 2).reshape(nb * 2 * co, 2 * ci)
    d_e = gemm_nn_taps(lat if kind != 'l0' else lat[0], ZS_SPLIT, (hc, wc), taps, wcat, planes=planes if kind != 'l0' else None,
                       timer_class='head_gemm_dgrad')
    return d_e.view(b, 2, hc, wc, 2, ci)


class _Layer0Z4(torch.autograd.Function):
    """First layer for Z = 4: every tap hits data.  x plain [B,4,H,W,Ci] -> z-split [B,2,H,W,2,Co]."""

    @staticmethod
    def forward(ctx, x, k, bias, raw=None):
        """``raw``: the fp32 ConvTranspose3d weight [Ci,Co,3,5,5] instead of its taps ``k`` (GPU training steps): the
        backward then returns the weight's gradient straight from the GEMM's (``ver_convt_weight_backward_blocks``)."""
        b, z, h, w, ci = x.shape
        ctx.raw = raw is not None
        co = raw.shape[1] if raw is not None else k.shape[-1]
        taps, offs, lo, hi = _layer0_z4_plan(ci, x.device)
        ctx.implicit = _implicit_taps(x, ZS_PLAIN, b * 2 * h * w, h * w, ci, raw)
        if ctx.implicit:
            from ..hipops import convt_weight_forward_blocks, gemm_nn_taps
            x = x.contiguous()
            wmat = convt_weight_forward_blocks(raw, _block_offsets('l0', ci, co, x.device), x.new_empty(50 * ci, 2 * co), ci, co)
            out = gemm_nn_taps(x, ZS_PLAIN, (h, w), taps, wmat, bias=torch.cat([bias, bias]).to(x.dtype).float())
            ctx.save_for_backward(x, wmat)
            ctx.geom = (tuple(x.shape), ci, co, h, w)
            return out.view(b, 2, h, w, 2, co)
        a_mat = x.new_empty(b * 2 * h * w, 50 * ci)
        _gather_z4(x, ZS_PLAIN, a_mat, taps, offs, ci, h, w)
        if raw is not None:
            # [W_lo | W_hi] written by ONE kernel from the fp32 parameter (no tap tensor, no row gathers)
            from ..hipops import convt_weight_forward_blocks
            wmat = convt_weight_forward_blocks(raw, _block_offsets('l0', ci, co, x.device), x.new_empty(50 * ci, 2 * co), ci, co)
        else:
            rows = k.reshape(75 * ci, co)
            wmat = torch.cat([rows.index_select(0, lo), rows.index_select(0, hi)], 1)    # [50 ci, 2 co]
        out = mm_fwd(a_mat, wmat, bias=torch.cat([bias, bias]))
        ctx.save_for_backward(a_mat, wmat)
        ctx.geom = (tuple(x.shape), ci, co, h, w)
        return out.view(b, 2, h, w, 2, co)

    @staticmethod
    def backward(ctx, grad_out):
        a_mat, wmat = ctx.saved_tensors
        shape, ci, co, h, w = ctx.geom
        taps, offs, lo, hi = _layer0_z4_plan(ci, a_mat.device)
        g = grad_out.contiguous().view(-1, 2 * co)
        if ctx.implicit and _IMPLICIT_DGRAD:
            d_x = zs_to_plain(_dgrad_implicit('l0', g[None], wmat, shape[0], h, w, ci, co))
        else:
            with gemm_timed('head_gemm_dgrad', g.shape[0], g.shape[1], wmat.shape[0]):
                d_a = torch.mm(g, wmat.t())
            d_x = _scatter_z4(d_a, ZS_PLAIN, shape, taps, offs, ci, h, w)
            del d_a
        if ctx.implicit and _IMPLICIT_WGRAD:                                               # (a_mat is the input lattice x here)
            from ..hipops import wgrad_tn_segments
            d_w = wgrad_tn_segments(a_mat, ZS_PLAIN, (h, w), taps, g)
        elif ctx.implicit:
            full = a_mat.new_empty(g.shape[0], 50 * ci)
            _gather_z4(a_mat, ZS_PLAIN, full, taps, offs, ci, h, w)
            d_w = rows_tn(full, g)
            del full
        else:
            d_w = rows_tn(a_mat, g)                                                        # [50 ci, 2 co]
        acc = torch.float64 if g.dtype == torch.float64 else torch.float32
        d_b = g.sum(0, dtype=acc)
        if ctx.raw:
            from ..hipops import convt_weight_backward_blocks
            d_raw = convt_weight_backward_blocks(d_w, _block_offsets('l0', ci, co, d_w.device), None, None, ci, co)
            return d_x, None, (d_b[:co] + d_b[co:]).to(g.dtype), d_raw
        d_lo = d_w.new_zeros(75 * ci, co)
        d_hi = d_w.new_zeros(75 * ci, co)
        d_lo.index_copy_(0, lo, d_w[:, :co])
        d_hi.index_copy_(0, hi, d_w[:, co:])
        return d_x, (d_lo + d_hi).view(75, ci, co), (d_b[:co] + d_b[co:]).to(g.dtype), None


def _block_order_z4():
    """18 blocks (dxi, dyi, j) in column order: groups G1..G4 over (dy, dx), then (dx, dy, j)."""
    blocks = [(dxi, dyi, j) for dxi in range(3) for dyi in range(3) for j in range(2)]
    return sorted(blocks, key=lambda b: (_group_of(b[1], b[0]), b))


_ORDER4 = _block_order_z4()
_GROUP_START4 = [next(i for i, b in enumerate(_ORDER4) if _group_of(b[1], b[0]) == g) for g in range(4)] + [18]
_CONST_BEFORE4 = [1 + (i >= _GROUP_START4[1]) + (i >= _GROUP_START4[2]) for i in range(18)]


def _block_offset4(t, c):
    return _PW2 * _CONST_BEFORE4[t] + t * c


def _const_offset4(pm, pn, c):
    g = _GROUP_START4
    return {(0, 0): 0, (1, 0): _PW2 + g[1] * c, (1, 1): 2 * _PW2 + g[2] * c, (0, 1): 3 * _PW2 + 18 * c}[(pm, pn)]


_LAYER_PLAN4 = {}


def _class_layout_z4():
    b = lambda lo, hi: [('b', t) for t in range(lo, hi)]
    g = _GROUP_START4
    return {
        (0, 0): [('c', (0, 0))] + b(g[0], g[1]) + [('c', (1, 0))] + b(g[1], g[2]) + [('c', (1, 1))] + b(g[2], g[4]),
        (1, 0): [('c', (1, 0))] + b(g[1], g[2]) + [('c', (1, 1))] + b(g[2], g[3]),
        (1, 1): [('c', (1, 1))] + b(g[2], g[3]),
        (0, 1): b(g[2], g[4]) + [('c', (0, 1))],
    }


def _class_rows_z4(ci):
    """{class: (first row of the class in the class-stacked [sum K_c, 2 Co] buffer (classes in _CLASSES order),
    [(kind, val, first row inside the class)])}."""
    layout = _class_layout_z4()
    out, roff = {}, 0
    for cls in _CLASSES:
        segs, r = [], 0
        for kind, val in layout[cls]:
            segs.append((kind, val, r))
            r += ci if kind == 'b' else _PW2
        out[cls] = (roff, segs)
        roff += r
    return out


def _class_segments_z4(cls, ci):
    """The K axis of class ``cls`` as the segments ``hipops.gemm_nn_taps`` takes, in column order of the tap matrix: a tap
    (dz, dy, dx) per block, ('c', p) for the constant-pattern block of class index p."""
    segs = []
    for kind, val in _class_layout_z4()[cls]:
        if kind == 'b':
            dxi, dyi, j = _ORDER4[val]
            segs.append((2 * j, dyi - 1, dxi - 1))
        else:
            segs.append(('c', _CLASSES.index(val)))
    return segs


_AUG_ROWS = {}


def _aug_rows_z4(ci, device):
    """Rows of the class-stacked buffer viewed as [2 sum K_c, Co] (row 2r + half) that hold the gradient of a class's own
    constant block: [4 classes x (lower, upper)] x 80."""
    key = (ci, str(device))
    if key not in _AUG_ROWS:
        idx = []
        for cls, (roff, segs) in _class_rows_z4(ci).items():
            r0 = next(r for kind, val, r in segs if kind == 'c' and val == cls)
            idx.append(2 * (roff + r0 + np.arange(_PW)))                    # [P_lo] rows, lower-half columns
            idx.append(2 * (roff + r0 + _PW + np.arange(_PW)) + 1)          # [P_hi] rows, upper-half columns
        _AUG_ROWS[key] = torch.from_numpy(np.concatenate(idx).astype(np.int64)).to(device)
    return _AUG_ROWS[key]


_STACK_TABLES = {}


def _stack_tables_z4(ci, device):
    """Index tables of the class-stacked weight matrix S [sum K_c, 2 Co] of a Z = 4 lattice layer (rows: ``_class_rows_z4``),
    viewed as S2 [2 sum K_c, Co] (row 2 r + half) where rows are addressed:
    ``block_rows`` int64 [50]: first row of every tap block; ``tap_slot`` int64 [75]: 2 * block + half of ONE slot that
    holds tap t (a tap with a = 1 sits in two: the lower half of j = 0 and the upper half of j = 1; the first is taken);
    ``const_rows`` / ``const_src``: the S2 rows of all constant blocks and, for each, the row of [vaug (80) | zero row] it
    holds (own block, matching half: K^T b_prev | bias | 0; everything else zero)."""
    key = (ci, str(device))
    if key not in _STACK_TABLES:
        block_rows, tap_slot = [], np.full(75, -1, dtype=np.int64)
        const_rows, const_src = [], []
        for cls, (roff, segs) in _class_rows_z4(ci).items():
            pm, pn = cls
            for kind, val, r0 in segs:
                if kind == 'b':
                    dxi, dyi, j = _ORDER4[val]
                    bb, cc = 2 * dyi - pm, 2 * dxi - pn
                    i = len(block_rows)
                    block_rows.append(roff + r0)
                    for half, a in ((0, 1 + j), (1, j)):
                        t = (a * 5 + bb) * 5 + cc
                        if tap_slot[t] < 0:
                            tap_slot[t] = 2 * i + half
                else:
                    own = val == cls
                    for r in range(_PW2):                           # rows [P_lo (80) | P_hi (80)] of the block
                        for half in range(2):
                            const_rows.append(2 * (roff + r0 + r) + half)
                            live = own and ((r < _PW and half == 0) or (r >= _PW and half == 1))
                            const_src.append(r % _PW if live else _PW)
        assert (tap_slot >= 0).all() and len(block_rows) == 50
        t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64)).to(device)
        _STACK_TABLES[key] = (t(block_rows), t(tap_slot), t(const_rows), t(const_src))
    return _STACK_TABLES[key]


def _layer_plan_z4(ci, device):
    """Per class (c0, c1, lo, hi): one column range and the rows of the stacked weight matrix
    [75*ci taps | 4 x 80 (K^T b_prev | bias | 0) | zero rows] feeding the lower / upper output half."""
    key = (ci, str(device))
    if key in _LAYER_PLAN4:
        return _LAYER_PLAN4[key]
    kt = 18 * ci + 4 * _PW2
    n_data = 75 * ci
    zero_base = n_data + 4 * _PW
    dummy = {'lo': zero_base, 'hi': zero_base}

    def zeros(which, n):
        r = np.arange(dummy[which], dummy[which] + n)
        dummy[which] += n
        return r

    def seg(pm, pn, sg):
        kind, val = sg
        if kind == 'b':
            dxi, dyi, j = _ORDER4[val]
            bb, cc = 2 * dyi - pm, 2 * dxi - pn
            assert 0 <= bb < 5 and 0 <= cc < 5
            lo = np.arange(ci) + (((1 + j) * 5 + bb) * 5 + cc) * ci
            hi = np.arange(ci) + ((j * 5 + bb) * 5 + cc) * ci
            return lo, hi
        if val == (pm, pn):                          # own constant block: [P_lo | P_hi]
            p = _CLASSES.index(val)
            own = np.arange(n_data + p * _PW, n_data + (p + 1) * _PW)
            return (np.concatenate([own, zeros('lo', _PW)]), np.concatenate([zeros('hi', _PW), own]))
        return zeros('lo', _PW2), zeros('hi', _PW2)    # foreign constant block
    layout = _class_layout_z4()
    plan = {}
    for (pm, pn), segs in layout.items():
        first = segs[0]
        c0 = _block_offset4(first[1], ci) if first[0] == 'b' else _const_offset4(*first[1], ci)
        parts = [seg(pm, pn, sg) for sg in segs]
        lo = np.concatenate([p_[0] for p_ in parts])
        hi = np.concatenate([p_[1] for p_ in parts])
        t = lambda a: torch.from_numpy(a.astype(np.int64)).to(device)
        # (lo0, hi0, lo1, hi1, ...): ONE gather of the stacked weight rows gives [K, 2, Co] = [W_lo | W_hi] row by row
        lohi = np.stack([lo, hi], 1).reshape(-1)
        plan[(pm, pn)] = (c0, c0 + len(lo), t(lo), t(hi), t(lohi))
    total_rows = max(dummy.values())
    taps = [(2 * j, dyi - 1, dxi - 1) for dxi, dyi, j in _ORDER4]
    offs = [_block_offset4(t, ci) for t in range(18)]
    _LAYER_PLAN4[key] = (plan, kt, total_rows, taps, offs)
    return _LAYER_PLAN4[key]


class _LatticeLayerZ4(torch.autograd.Function):

    @staticmethod
    def forward(ctx, e, k, bias, prev_bias, planar, raw=None):
        """e: z-split [B,2,H,W,2,C] or planar z-split [4,B,2,H/2,W/2,2,C]; k [75,Ci,Co]
        -> planar z-split output [4,B,2,H,W,2,Co] (H, W = combined size of the input).
        ``raw``: the fp32 ConvTranspose3d weight [Ci,Co,3,5,5] instead of ``k`` (GPU training steps): the four class
        weight gradients are then written into one stacked buffer and turned into the weight's gradient by ONE kernel."""
        ctx.raw = raw is not None
        if planar:
            _, b, _, hh, wh, _, ci = e.shape
            hc, wc = 2 * hh, 2 * wh
        else:
            b, _, hc, wc, _, ci = e.shape
        layout = ZS_PLANAR_SPLIT if planar else ZS_SPLIT
        co = raw.shape[1] if raw is not None else k.shape[-1]
        dt = e.dtype
        plan, kt, total_rows, taps, offs = _layer_plan_z4(ci, e.device)
        m = b * 2 * hc * wc
        ctx.implicit = _implicit_taps(e, layout, m, hc * wc, ci, raw)
        a_mat = None if ctx.implicit else e.new_empty(m, kt)
        if not ctx.implicit and not _gather_z4(e, layout, a_mat, taps, offs, ci, hc, wc, with_const=True):
            pats = _class_patterns(4, hc, wc, e.device, dt)
            a3 = a_mat.view(b, 2 * hc * wc, kt)
            for (pm, pn), pat in zip(_CLASSES, pats):
                o = _const_offset4(pm, pn, ci)
                halves = pat.view(2, 2 * hc * wc, _PW)              # output z = zl (lower), zl + 2 (upper)
                a3[:, :, o:o + _PW] = halves[0]
                a3[:, :, o + _PW:o + _PW2] = halves[1]
        out = e.new_empty(4, m, 2 * co)
        if raw is not None:
            # Weight side of the step, which does not shrink with the batch (config.latency): the four class matrices
            # [W_lo | W_hi] are ONE stacked buffer S [sum K_c, 2 Co] written straight from the fp32 parameter
            # (ver_convt_weight_forward_blocks); v = b_prev^T K[t] for all taps is one pass over S
            # (ver_blocks_vec_forward); the constant rows (K^T b_prev | bias | 0) go in by one indexed copy.  No tap
            # tensor, no 88-MB concatenation, no row gather per class.
            from ..hipops import blocks_vec_forward, convt_weight_forward_blocks
            class_rows = _class_rows_z4(ci)
            block_rows, tap_slot, const_rows, const_src = _stack_tables_z4(ci, e.device)
            stack = e.new_empty(sum(plan[cls][1] - plan[cls][0] for cls in _CLASSES), 2 * co)
            convt_weight_forward_blocks(raw, _block_offsets('lat', ci, co, e.device), stack, ci, co)
            v = blocks_vec_forward(stack, block_rows, ci, prev_bias).view(-1, co).index_select(0, tap_slot)     # [75, Co] fp32
            vaug = torch.cat([v.to(dt), bias.to(dt)[None], v.new_zeros(_PW - 75, co, dtype=dt)])                # + one zero row
            stack.view(-1, co).index_copy_(0, const_rows, vaug.index_select(0, const_src))
            if ctx.implicit:
                from ..hipops import gemm_nn_taps
                e = e.contiguous()
                table, _ = _const_rows_z4(ci, hc, wc, e.device, dt)
            for p, cls in enumerate(_CLASSES):
                c0, c1 = plan[cls][:2]
                r0 = class_rows[cls][0]
                if ctx.implicit:
                    gemm_nn_taps(e, layout, (hc, wc), _class_segments_z4(cls, ci), stack[r0:r0 + c1 - c0], const_rows=table, out=out[p])
                else:
                    mm_fwd(a_mat[:, c0:c1], stack[r0:r0 + c1 - c0], out=out[p])
            ctx.save_for_backward(e if ctx.implicit else a_mat, stack, prev_bias)
            ctx.geom = (layout, tuple(e.shape), b, hc, wc, ci, co, total_rows)
            return out.view(4, b, 2, hc, wc, 2, co)
        v = _bias_through_taps(prev_bias.to(dt), k)                               # [75, Co]
        vaug = torch.cat([v, bias.to(dt)[None], v.new_zeros(_PW - 76, co)])
        rows = torch.cat([k.reshape(75 * ci, co), vaug, vaug, vaug, vaug,
                          v.new_zeros(total_rows - 75 * ci - 4 * _PW, co)])
        ws = []
        for p, cls in enumerate(_CLASSES):
            c0, c1, lo, hi, lohi = plan[cls]
            w = rows.index_select(0, lohi).view(c1 - c0, 2 * co)          # [W_lo | W_hi], one gather
            mm_fwd(a_mat[:, c0:c1], w, out=out[p])
            ws.append(w)
        ctx.save_for_backward(a_mat, k, prev_bias, *ws)
        ctx.geom = (layout, tuple(e.shape), b, hc, wc, ci, co, total_rows)
        return out.view(4, b, 2, hc, wc, 2, co)

    @staticmethod
    def backward(ctx, grad_out):
        a_mat, k, prev_bias = ctx.saved_tensors[:3]
        layout, e_shape, b, hc, wc, ci, co, total_rows = ctx.geom
        plan, kt, total_rows, taps, offs = _layer_plan_z4(ci, a_mat.device)
        if ctx.raw:                                             # (k is the class-stacked weight matrix S here)
            cr = _class_rows_z4(ci)
            ws = [k[cr[cls][0]:cr[cls][0] + plan[cls][1] - plan[cls][0]] for cls in _CLASSES]
        else:
            ws = ctx.saved_tensors[3:]
        dt = a_mat.dtype
        implicit = getattr(ctx, 'implicit', False)              # (a_mat is then the source lattice e, not its tap matrix)
        m = b * 2 * hc * wc
        g = grad_out.contiguous().view(4, m, 2 * co)
        if implicit and not _IMPLICIT_WGRAD:                    # (A/B mode: the tap matrix after all, for the explicit kernel)
            lattice, a_mat, implicit = a_mat, a_mat.new_empty(m, kt), False
            _gather_z4(lattice, layout, a_mat, taps, offs, ci, hc, wc, with_const=True)
        dgrad_implicit = implicit and _IMPLICIT_DGRAD
        if not dgrad_implicit:
            d_a = a_mat.new_empty(m, kt)
            d_a[:, kt - _PW2:] = 0                              # P01 is outside class (0,0)'s range
        fused = ctx.raw
        if fused:
            class_rows = _class_rows_z4(ci)
            stacked = a_mat.new_empty(sum(plan[cls][1] - plan[cls][0] for cls in _CLASSES), 2 * co)
        else:
            d_lo = a_mat.new_zeros(total_rows, co)
            d_hi = a_mat.new_zeros(total_rows, co)
        for p, cls in enumerate(_CLASSES):
            c0, c1, lo, hi, _ = plan[cls]
            w = ws[p]
            # (library GEMMs: a one-pass kernel of our own over all four classes was built and measured 7 % slower,
            #  scratch/experiments/k_dgrad_nt.hip.inc)
            if not dgrad_implicit:
                with gemm_timed('head_gemm_dgrad', m, 2 * co, c1 - c0):
                    if p == 0:                                  # class (0,0): initialises every tap block
                        torch.mm(g[p], w.t(), out=d_a[:, c0:c1])
                    else:
                        torch.addmm(d_a[:, c0:c1], g[p], w.t(), out=d_a[:, c0:c1])
            if fused:
                r0 = class_rows[cls][0]
                if implicit:
                    from ..hipops import wgrad_tn_segments
                    wgrad_tn_segments(a_mat, layout, (hc, wc), _class_segments_z4(cls, ci), g[p], out=stacked[r0:r0 + c1 - c0],
                                      const_rows=_const_rows_z4(ci, hc, wc, a_mat.device, dt)[0])
                else:
                    rows_tn(a_mat[:, c0:c1], g[p], out=stacked[r0:r0 + c1 - c0])
                continue
            d_w = rows_tn(a_mat[:, c0:c1], g[p])
            d_lo.index_copy_(0, lo, d_w[:, :co])
            d_hi.index_copy_(0, hi, d_w[:, co:])
        if dgrad_implicit:
            d_e = _dgrad_implicit('lat', g, k, b, hc, wc, ci, co)                  # z-split [B,2,hc,wc,2,Ci] (k: stacked weights)
            if layout == ZS_PLANAR_SPLIT:                                           # -> the planar form of the source lattice
                d_e = d_e.view(b, 2, hc // 2, 2, wc // 2, 2, 2, ci).permute(3, 5, 0, 1, 2, 4, 6, 7).reshape(e_shape)
        else:
            d_e = _scatter_z4(d_a, layout, e_shape, taps, offs, ci, hc, wc)
            del d_a
        acc = torch.float64 if dt == torch.float64 else torch.float32
        if fused:
            from ..hipops import convt_weight_backward_blocks
            # own constant blocks of the four classes -> d(v | bias); every tap's two half gradients + prev_bias (x) d(v)
            # -> the gradient of the ConvTranspose3d weight, in one pass over the stacked buffer
            d_vaug = stacked.view(-1, co).index_select(0, _aug_rows_z4(ci, stacked.device)).view(8, _PW, co).sum(0, dtype=acc)
            d_v = d_vaug[:75].to(dt)
            pb = prev_bias.to(dt)
            d_raw = convt_weight_backward_blocks(stacked, _block_offsets('lat', ci, co, stacked.device), pb, d_v, ci, co)
            # d(b_prev) = sum_t K[t] d_v[t]: d_v scattered to ONE slot per tap of the stacked weights, one pass over them
            from ..hipops import blocks_vec_backward
            block_rows, tap_slot, _, _ = _stack_tables_z4(ci, stacked.device)
            dv2 = d_vaug.new_zeros(2 * block_rows.numel(), co).index_copy_(0, tap_slot, d_vaug[:75])
            d_prev = blocks_vec_backward(k, block_rows, ci, dv2.view(block_rows.numel(), 2 * co))
            return d_e, None, d_vaug[75].to(prev_bias.dtype), d_prev.to(prev_bias.dtype), None, d_raw
        d_rows = d_lo + d_hi
        n_data = 75 * ci
        d_k = d_rows[:n_data].view(75, ci, co)
        d_vaug = d_rows[n_data:n_data + 4 * _PW].view(4, _PW, co).sum(0, dtype=acc)
        d_v = d_vaug[:75].to(dt)
        d_bias = d_vaug[75]
        pb = prev_bias.to(dt)
        d_k = torch.addcmul(d_k, pb[None, :, None], d_v[:, None, :])
        d_prev = torch.bmm(k, d_v.unsqueeze(2)).sum(0).squeeze(1)
        return d_e, d_k, d_bias.to(prev_bias.dtype), d_prev.to(prev_bias.dtype), None, None


def gemm_timed(name, m, k, n):
    """HIP-event bracket around one library GEMM of the head ([m, k] x [k, n]) for bench.py's dense-part roofline entry
    (hipops.timed: free unless a KernelTimer is set)."""
    from ..hipops import timed
    return timed(name, 2.0 * m * k * n)


# Forward GEMMs of the lattice layers on ver_gemm_nn (csrc/ver_gemm.hip) from this many rows on (and K >= 2048); below, and
# for every other dtype / device, the library.  Measured on the 192-viewpoint shapes (scratch/r05/gemm_bench.py): layer 3
# 1 266-1 292 -> 1 312-1 338 TFLOP/s, layers 1 / 2 1 121-1 210 -> 1 310; occ_proj (K = 832) and the 8-viewpoint shapes are
# faster in the library and stay there.  VER_OWN_GEMM=0: library everywhere.
_OWN_GEMM = os.environ.get('VER_OWN_GEMM', '1') == '1'
# (round 6: 14 000 instead of 49 152 -- with the implicit operands layer 3 of an eight-viewpoint step (14 400 rows) and layers 1-2 of a
#  64-viewpoint one gain 0.5-1.6 %; 3 000 loses 9 % at eight viewpoints: too few tiles)
_OWN_GEMM_MIN_ROWS = int(os.environ.get('VER_OWN_GEMM_MIN_ROWS', '14000'))
# ... and the skinny products of the small-batch steps (config.latency: 450 / 1 800 rows at one viewpoint per step) cut into K
# slices: the library's best recorded solution runs (450 x 38 400) x (38 400 x 1 536) in 144 us on 12 workgroups, the weight
# matrix alone streams in 15
_OWN_GEMM_SKINNY_ROWS = 2048


def mm_fwd(a, w, out=None, bias=None):
    """``a @ w (+ bias)`` for the forward GEMM of a layer: a [M, K] (may be a column range of the tap matrix), w [K, N]
    row-major, optional bias [N]; ``out`` [M, N] is written when given."""
    m, k = a.shape
    n = w.shape[1]
    if _OWN_GEMM and _on_hip(a) and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and k >= 2048 and \
            (m >= _OWN_GEMM_MIN_ROWS or m <= _OWN_GEMM_SKINNY_ROWS):
        from ..hipops import gemm_nn, gemm_nn_splits, gemm_nn_supported
        if gemm_nn_supported(a, w) and (out is None or (out.stride(-1) == 1 and out.dtype == torch.bfloat16)) and \
                (m >= _OWN_GEMM_MIN_ROWS or gemm_nn_splits(m, k, n) > 1):
            # (the library adds the bias as a bf16 vector: the same rounded values here)
            return gemm_nn(a, w, None if bias is None else bias.to(a.dtype).float(), out)
    with gemm_timed('head_gemm_fwd', m, k, n):
        if bias is None:
            return torch.mm(a, w, out=out) if out is not None else torch.mm(a, w)
        b = bias.to(a.dtype)
        return torch.addmm(b, a, w, out=out) if out is not None else torch.addmm(b, a, w)


def rows_tn(a, g, out_dtype=None, out=None):
    """a^T g for tall operands (a [M,K] may be a column range of a wider matrix, g [M,N]): the weight gradient of a GEMM
    layer, rows on the contraction axis.  bf16 GPU operands run on ``ver_wgrad_tn`` (csrc/ver_wgrad.hip: both
    operands streamed row-major into LDS, fragments through transposing LDS reads, fp32 partial sums over row chunks
    added up in fp32: 1.3 PFLOP/s where the library's T x N class reaches 1.05); everything else
    (fp32 / fp64, CPU tensors of the algebra tests) is a plain matmul."""
    if _on_hip(a) and a.dtype == torch.bfloat16 and g.dtype == torch.bfloat16:
        from ..hipops import wgrad_tn, wgrad_tn_supported
        if g.stride(-1) != 1 or g.stride(0) % 8:
            g = g.contiguous()
        if wgrad_tn_supported(a, g):
            return wgrad_tn(a, g, out_dtype=out_dtype if out is None else out.dtype, out=out)
    if out is not None:                                  # (a row range of a stacked gradient buffer)
        return torch.mm(a.t(), g, out=out) if out.dtype == a.dtype else out.copy_(torch.mm(a.t(), g))
    res = torch.mm(a.t(), g)
    return res if out_dtype is None else res.to(out_dtype)


def _compute_dtype(x):
    """bf16 under ``torch.autocast`` (im2col, GEMM operands and lattices all in bf16, fp32
    accumulation inside the GEMM), else the input's dtype."""
    if _on_hip(x) and torch.is_autocast_enabled('cuda'):
        return torch.get_autocast_dtype('cuda')
    return x.dtype


class _ChannelsLast(torch.autograd.Function):
    """x0 [B,C,Z,H,W] (the head's raw view of the encoder output, head:558) -> the compute dtype, channels-last
    [B,Z,H,W,C]; one cast + one LDS-tiled transpose each way (``ver_lattice_transpose`` over the flattened Z*H*W
    positions) instead of a cast of the permuted view and an element-wise strided copy."""

    @staticmethod
    def forward(ctx, x0, dt, split):
        from ..hipops import lattice_transpose
        b, c, z, h, w = x0.shape
        cf = x0.to(dt).contiguous().view(b, c * z * h * w)
        e = torch.empty(b, z, h, w, c, dtype=dt, device=x0.device)
        lattice_transpose(e.view(b, 1, split[0], split[1], c), cf, split, 0, False)
        ctx.split, ctx.in_dtype = split, x0.dtype
        return e

    @staticmethod
    def backward(ctx, g):
        from ..hipops import lattice_transpose
        b, z, h, w, c = g.shape
        g = g.contiguous()
        cf = torch.empty(b, c * z * h * w, dtype=g.dtype, device=g.device)
        lattice_transpose(g.view(b, 1, ctx.split[0], ctx.split[1], c), cf, ctx.split, 0, True)
        return cf.view(b, c, z, h, w).to(ctx.in_dtype), None, None


def _channels_last(x0, dt):
    return _algebra(x0).channels_last(x0, dt)


def upsample_lattice(x0, weights, biases):
    """x0 [B,C,Z,H,W] -> (E_3, last bias).  E_3 holds the even positions of the reference's dense
    output ``up_sample(x0)`` [B,C,Z,8H,8W], as a PLANAR lattice [4,B,Z,2H,2W,C] or, for Z = 4, planar
    z-split [4,B,2,2H,2W,2,C] (``lattice_to_plain`` gives the channels-last [B,Z,4H,4W,C] lattice)."""
    dt = _compute_dtype(x0)
    e = _channels_last(x0, dt)
    bs = [b.to(dt) for b in biases]
    if e.shape[1] == 4 and _on_hip(e) and dt in _HIP_DTYPES and all(_on_hip(w) and w.dtype == torch.float32 for w in weights):
        # GPU: the layers take the ConvTranspose3d weights themselves (taps made inside, weight gradient made from the class
        # GEMMs' gradients by one kernel: the weight-side work of a step does not shrink with the batch)
        e = _Layer0Z4.apply(e, None, bs[0], weights[0])
        e = _LatticeLayerZ4.apply(e, None, bs[1], bs[0], False, weights[1])
        e = _LatticeLayerZ4.apply(e, None, bs[2], bs[1], True, weights[2])
        return e, bs[2]
    ks = [_corr_weight(w, dt) for w in weights]
    if e.shape[1] == 4:                                   # bev_z = 4: z-split path (a third fewer FLOPs)
        e = _Layer0Z4.apply(e, ks[0], bs[0])
        e = _LatticeLayerZ4.apply(e, ks[1], bs[1], bs[0], False)
        e = _LatticeLayerZ4.apply(e, ks[2], bs[2], bs[1], True)
        return e, bs[2]
    e = _layer0(e, ks[0], bs[0])
    e = _layer_lattice(e, ks[1], bs[1], bs[0], planar=False)
    e = _layer_lattice(e, ks[2], bs[2], bs[1], planar=True)
    return e, bs[2]


def full_volume(e, bias):
    """Even lattice (plain [B,Z,H,W,C], planar [4,B,Z,H/2,W/2,C] or planar z-split
    [4,B,2,H/2,W/2,2,C]) + bias -> dense [B,C,Z,2H,2W] (odd rows/cols = bias)."""
    if e.dim() == 7:
        e = planar_zs_to_plain(e)
    elif e.dim() == 6:
        e = planar_to_plain(e)
    b, z, h, w, c = e.shape
    y = bias.view(1, c, 1, 1, 1).expand(b, c, z, 2 * h, 2 * w).contiguous()
    y[:, :, :, ::2, ::2] = e.permute(0, 4, 1, 2, 3)
    return y


def upsample_dense(x0, weights, biases):
    """Drop-in value of ``nn.Sequential(ConvTranspose3d x3)(x0)`` for the reference geometry."""
    e, b = upsample_lattice(x0, weights, biases)
    return full_volume(e, b)
