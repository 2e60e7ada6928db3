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
import numpy as np
import torch
import torch.nn.functional as F

KERNEL = (3, 5, 5)
GEOM = dict(stride=(1, 2, 2), padding=(2, 4, 4), dilation=(2, 2, 2), output_padding=(0, 1, 1))


def is_reference_geometry(conv):
    return (tuple(conv.kernel_size) == KERNEL and tuple(conv.stride) == GEOM['stride'] and
            tuple(conv.padding) == GEOM['padding'] and tuple(conv.dilation) == GEOM['dilation'] and
            tuple(conv.output_padding) == GEOM['output_padding'] and conv.groups == 1)


def _corr_weight(weight, dtype):
    """ConvTranspose weight [Ci,Co,3,5,5] -> correlation taps as ONE contiguous [75, Ci, Co]
    tensor in the compute dtype, tap index = (a*5+b)*5+c, K[a,b,c] = Wt[:, :, 2-a, 4-b, 4-c]
    (one cast + one gather kernel; the per-class tap subsets below are index_select's of it)."""
    ci, co = weight.shape[:2]
    return weight.to(dtype).flip(2, 3, 4).permute(2, 3, 4, 0, 1).reshape(75, ci, co)


_IDX_CACHE = {}


def _tap_index(tap_ids, device):
    key = (tuple(tap_ids), str(device))
    if key not in _IDX_CACHE:
        _IDX_CACHE[key] = torch.tensor(tap_ids, dtype=torch.long, device=device)
    return _IDX_CACHE[key]


def _im2col(e, taps):
    """e [B,Z,H,W,C] channels-last; taps: list of (dz,dy,dx) offsets (zero outside the lattice)
    -> [B*Z*H*W, len(taps)*C].  On the GPU one HIP kernel each way (ver_lattice_im2col /
    ver_lattice_col2im); the slice-and-cat form below only serves CPU tensors in the algebra tests."""
    if e.is_cuda:
        from ..hipops import lattice_im2col
        return lattice_im2col(e.contiguous(), taps)
    b, z, h, w, c = e.shape
    pz = max(abs(t[0]) for t in taps)
    py = max(abs(t[1]) for t in taps)
    px = max(abs(t[2]) for t in taps)
    e_pad = F.pad(e, (0, 0, px, px, py, py, pz, pz))
    cols = [e_pad[:, pz + dz:pz + dz + z, py + dy:py + dy + h, px + dx:px + dx + w, :] for dz, dy, dx in taps]
    a = torch.cat(cols, dim=-1)
    return a.reshape(-1, a.shape[-1])


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


def _layer0(e, k, bias):
    """All 75 taps hit data.  e [B,Z,H,W,C] channels-last -> E_1 [B,Z,H,W,Co]."""
    b, z, h, w, c = e.shape
    taps = [(2 * a - 2, bb - 2, cc - 2) for a in range(3) for bb in range(5) for cc in range(5)]
    a_mat = _im2col(e, taps)
    out = torch.addmm(bias, a_mat, k.reshape(75 * c, -1))
    return out.view(b, z, h, w, -1)


def _layer_lattice(e, k, bias, prev_bias):
    """e = data lattice [B,Z,H,W,C] of a full input of size (Z,2H,2W) that equals ``prev_bias``
    off the lattice -> output lattice [B,Z,2H,2W,Co]."""
    b, z, h, w, c = e.shape
    co = k.shape[-1]
    # constant taps: pattern [Z*2H*2W, 75] @ (K[tap]^T prev_bias) [75, Co], bias folded in
    # (a [75]-batch of 1-row GEMMs; a single fp32 gemv over the raw weight measured slower end to end
    # because its backward materialises a dense fp32 outer product per layer)
    v = torch.matmul(prev_bias, k)                                       # [75, Co]
    const = torch.addmm(bias, _constant_pattern(z, 2 * h, 2 * w, e.device, e.dtype), v)   # [.., Co]
    const = const.view(z, 2 * h, 2 * w, co)
    out = e.new_empty(b, z, 2 * h, 2 * w, co)
    for pm in (0, 1):
        bs_ = [bb for bb in range(5) if (pm + bb) % 2 == 0]
        for pn in (0, 1):
            cs_ = [cc for cc in range(5) if (pn + cc) % 2 == 0]
            # lattice row of tap b for output row m = 2m'+pm:  m' - 1 + (pm+b)/2
            taps = [(2 * a - 2, (pm + bb) // 2 - 1, (pn + cc) // 2 - 1)
                    for a in range(3) for bb in bs_ for cc in cs_]
            ids = [(a * 5 + bb) * 5 + cc for a in range(3) for bb in bs_ for cc in cs_]
            ksub = k.index_select(0, _tap_index(ids, k.device))
            a_mat = _im2col(e, taps)
            res = (a_mat @ ksub.reshape(-1, co)).view(b, z, h, w, co)
            out[:, :, pm::2, pn::2, :] = res + const[None, :, pm::2, pn::2, :]
    return out


def _compute_dtype(x):
    """bf16 under ``torch.autocast`` (im2col, GEMM operands and lattices all in bf16, fp32
    accumulation inside the GEMM), else the input's dtype."""
    if x.is_cuda and torch.is_autocast_enabled('cuda'):
        return torch.get_autocast_dtype('cuda')
    return x.dtype


def upsample_lattice(x0, weights, biases):
    """x0 [B,C,Z,H,W] -> (E_3 channels-last [B,Z,4H,4W,C], last bias).  E_3 holds the even
    positions of the reference's dense output ``up_sample(x0)`` [B,C,Z,8H,8W]."""
    dt = _compute_dtype(x0)
    e = x0.permute(0, 2, 3, 4, 1).to(dt)
    ks = [_corr_weight(w, dt) for w in weights]
    bs = [b.to(dt) for b in biases]
    e = _layer0(e, ks[0], bs[0])
    e = _layer_lattice(e, ks[1], bs[1], bs[0])
    e = _layer_lattice(e, ks[2], bs[2], bs[1])
    return e, bs[2]


def full_volume(e, bias):
    """Even lattice [B,Z,H,W,C] + bias -> dense [B,C,Z,2H,2W] (odd rows/cols = bias)."""
    b, z, h, w, c = e.shape
    y = bias.view(1, c, 1, 1, 1).expand(b, c, z, 2 * h, 2 * w).contiguous()
    y[:, :, :, ::2, ::2] = e.permute(0, 4, 1, 2, 3)
    return y


def upsample_dense(x0, weights, biases):
    """Drop-in value of ``nn.Sequential(ConvTranspose3d x3)(x0)`` for the reference geometry."""
    e, b = upsample_lattice(x0, weights, biases)
    return full_volume(e, b)
