"""The lattice algebra of ``upsample.py`` / ``occ_proj_lattice.py`` as plain torch ops on any device and dtype: what the CPU
suite checks in fp64 against ``conv_transpose3d`` (tests/test_head_cpu.py, test_modules_cpu.py).  The product path never
comes here with a GPU tensor of a dtype the HIP kernels take: ``upsample._algebra(tensor)`` is the ONE place that chooses
between this module and the HIP kernels (by the tensor's device, never by "extension missing" -- without libver_hip.so a GPU
tensor raises)."""
import torch
import torch.nn.functional as F

CLASSES = ((0, 0), (0, 1), (1, 0), (1, 1))
ZS_PLAIN, ZS_PLANAR, ZS_SPLIT, ZS_PLANAR_SPLIT = 0, 1, 2, 3          # = hipops lattice layouts


# ---- lattice layouts
def planar_to_plain(e):
    """[4,B,Z,H,W,C] (plane 2pm+pn = positions (2y+pm, 2x+pn)) -> [B,Z,2H,2W,C]."""
    _, b, z, h, w, c = e.shape
    out = e.new_empty(b, z, 2 * h, 2 * w, c)
    for p, (pm, pn) in enumerate(CLASSES):
        out[:, :, pm::2, pn::2] = e[p]
    return out


def plain_to_planar(e):
    return torch.stack([e[:, :, pm::2, pn::2] for pm, pn in CLASSES])


def zs_to_plain(e):
    """[B,2,H,W,2,C] -> [B,4,H,W,C]"""
    b, _, h, w, _, c = e.shape
    return e.permute(0, 4, 1, 2, 3, 5).reshape(b, 4, h, w, c)


def plain_to_zs(e):
    b, z, h, w, c = e.shape
    assert z == 4
    return e.reshape(b, 2, 2, h, w, c).permute(0, 2, 3, 4, 1, 5).contiguous()


def planar_zs_to_plain(e):
    """[4,B,2,H,W,2,C] -> [B,4,2H,2W,C]"""
    _, b, _, h, w, _, c = e.shape
    out = e.new_empty(b, 4, 2 * h, 2 * w, c)
    for p, (pm, pn) in enumerate(CLASSES):
        out[:, :, pm::2, pn::2] = zs_to_plain(e[p])
    return out


def plain_to_planar_zs(e):
    return torch.stack([plain_to_zs(e[:, :, pm::2, pn::2]) for pm, pn in CLASSES])


def to_plain(e, layout):
    return {ZS_PLAIN: lambda t: t, ZS_PLANAR: planar_to_plain, ZS_SPLIT: zs_to_plain,
            ZS_PLANAR_SPLIT: planar_zs_to_plain}[layout](e)


def from_plain(e, layout):
    return {ZS_PLAIN: lambda t: t.contiguous(), ZS_PLANAR: lambda t: plain_to_planar(t).contiguous(),
            ZS_SPLIT: plain_to_zs, ZS_PLANAR_SPLIT: lambda t: plain_to_planar_zs(t).contiguous()}[layout](e)


# ---- the operations (same signatures as upsample._HipAlgebra)
def corr_weight(weight, dtype):
    """ConvTranspose weight [Ci,Co,3,5,5] -> correlation taps [75, Ci, Co], K[a,b,c] = Wt[:, :, 2-a, 4-b, 4-c]."""
    ci, co = weight.shape[:2]
    return weight.to(dtype).flip(2, 3, 4).permute(2, 3, 4, 0, 1).reshape(75, ci, co)


def im2col(e, taps):
    b, z, h, w, c = e.shape
    pz = max(abs(t[0]) for t in taps)
    py = max(abs(t[1]) for t in taps)
    px = max(abs(t[2]) for t in taps)
    e_pad = F.pad(e, (0, 0, px, px, py, py, pz, pz))
    cols = [e_pad[:, pz + dz:pz + dz + z, py + dy:py + dy + h, px + dx:px + dx + w, :] for dz, dy, dx in taps]
    a = torch.cat(cols, dim=-1)
    return a.reshape(-1, a.shape[-1])


def gather27(e, planar, a_mat, ci, hc, wc, taps, offs):
    src = planar_to_plain(e) if planar else e
    b, z = src.shape[:2]
    pad = F.pad(src, (0, 0, 1, 1, 1, 1, 2, 2))
    view = a_mat.view(b, z, hc, wc, -1)
    for (dz, dy, dx), o in zip(taps, offs):
        view[..., o:o + ci] = pad[:, 2 + dz:2 + dz + z, 1 + dy:1 + dy + hc, 1 + dx:1 + dx + wc]


def scatter27(d_a, planar, shape, ci, hc, wc, taps, offs):
    b, z = (shape[1], shape[2]) if planar else (shape[0], shape[1])
    pad = d_a.new_zeros(b, z + 4, hc + 2, wc + 2, ci)
    view = d_a.view(b, z, hc, wc, -1)
    for (dz, dy, dx), o in zip(taps, offs):
        pad[:, 2 + dz:2 + dz + z, 1 + dy:1 + dy + hc, 1 + dx:1 + dx + wc] += view[..., o:o + ci]
    plain = pad[:, 2:2 + z, 1:1 + hc, 1:1 + wc]
    return plain_to_planar(plain).contiguous() if planar else plain.contiguous()


def gather_z4(e, layout, a_mat, taps, offs, ci, hc, wc, const=None):
    """rows (b, zl, y, x); tap (dz in {0,2}, dy, dx) reads input layer zl + dz.  (``const``: the constant-pattern blocks the
    HIP kernel writes in the same pass; here the caller fills them.)"""
    src = to_plain(e, layout)
    b = src.shape[0]
    py = max(abs(t[1]) for t in taps)
    px = max(abs(t[2]) for t in taps)
    pad = F.pad(src, (0, 0, px, px, py, py))
    view = a_mat.view(b, 2, hc, wc, -1)
    for (dz, dy, dx), o in zip(taps, offs):
        view[..., o:o + ci] = pad[:, dz:dz + 2, py + dy:py + dy + hc, px + dx:px + dx + wc]
    return False                                              # the constant blocks are still to be written


def scatter_z4(d_a, layout, shape, taps, offs, ci, hc, wc):
    b = d_a.shape[0] // (2 * hc * wc)
    py = max(abs(t[1]) for t in taps)
    px = max(abs(t[2]) for t in taps)
    pad = d_a.new_zeros(b, 4, hc + 2 * py, wc + 2 * px, ci)
    view = d_a.view(b, 2, hc, wc, -1)
    for (dz, dy, dx), o in zip(taps, offs):
        pad[:, dz:dz + 2, py + dy:py + dy + hc, px + dx:px + dx + wc] += view[..., o:o + ci]
    return from_plain(pad[:, :, py:py + hc, px:px + wc], layout)


def channels_last(x0, dt):
    """[B,C,Z,H,W] -> the compute dtype, channels-last [B,Z,H,W,C]."""
    return x0.permute(0, 2, 3, 4, 1).to(dt).contiguous()
