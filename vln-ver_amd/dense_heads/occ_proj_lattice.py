"""``occ_proj`` on the even lattice.

After the coarse-to-fine upsample the reference re-interprets the dense volume
``Y [bs, C, Z, 8H, 8W]`` (contiguous) as ``[bs, Z, 8H, 8W, C]`` *without* permuting
(dense_heads/voxelformer_occupancy_head.py:564), permutes/flattens it to rows of ``Z*C``
features (:570) and applies ``occ_proj = Linear(Z*C, 35*128)`` (:571).  Three quarters of ``Y``
are the last ConvTranspose3d's bias (odd rows / columns), so every 3072-wide
input row is ~3/4 constants whose positions depend only on the row's position:

    out[row] = bias + sum_{data cols j} W[:, j] * E[perm(row, j)]          (gathered GEMM, K/4)
                    + sum_k  b_up[chan(row, k)] * (sum_{non-data cols of token k} W[:, j])

Rows are grouped by their data-column pattern (5 patterns at the vocc.py sizes: the pattern
depends on ``b % 5`` only); per group ONE GEMM ``[bs*rows, K_g] x [K_g, 4480]`` with
``K_g = data cols + Z + 1`` replaces the ``[.., 3072] x [3072, 4480]`` slice of the dense
product: the Z+1 extra columns of the gathered operand carry ``b_up[chan(row,k)]`` and a 1, the
matching weight rows the summed constant columns and the bias, so the constant part costs no
separate pass.  99 instead of 396 GFLOP per viewpoint and the 4x larger dense volume is never
materialised.

One autograd Function (``_OccProjLattice``) with a hand-written backward: every GEMM writes
straight into its slice of a preallocated GROUP-MAJOR buffer (row of (group g, sample b, member i)
= bs*offset_g + b*n_g + i), so there is no concatenation, no broadcast add and no gradient
accumulation over slices.  ``occ_branches`` is row-wise and runs in that order; ``rows_to_voxels``
brings the 16-wide logits into the reference's voxel order.  The index tables are built once per
geometry by brute force from the definition of the two raw views, so they hold for any (C,Z,H,W).
"""
import os

import numpy as np
import torch

from .upsample import gemm_timed, rows_tn

_PLAN_CACHE = {}
_OWN_DGRAD = os.environ.get('VER_OCC_PROJ_OWN_DGRAD', '1') == '1'


class _Plan:
    pass


def _build_plan(C, Z, Hf, Wf):
    """Index tables for a dense volume [C, Z, Hf, Wf] whose data sits on even (y, x)."""
    assert Hf % 2 == 0 and Wf % 2 == 0
    Hl, Wl = Hf // 2, Wf // 2
    rows = Hf * Wf                                   # (a, b) positions of the reinterpreted tensor
    feat = Z * C
    k = np.arange(Z, dtype=np.int32)[None, :, None]
    ab = np.arange(rows, dtype=np.int32)[:, None, None]
    cp = np.arange(C, dtype=np.int32)[None, None, :]
    f = (k * rows + ab) * C + cp                     # flat index into Y for feature (k, c') of row (a,b)
    c, rem = np.divmod(f, Z * Hf * Wf)
    z, rem = np.divmod(rem, Hf * Wf)
    y, x = np.divmod(rem, Wf)
    data = (y % 2 == 0) & (x % 2 == 0)               # [rows, Z, C]
    lat = ((c * Z + z) * Hl + y // 2) * Wl + x // 2  # flat index into the lattice [C, Z, Hl, Wl]
    data2 = data.reshape(rows, feat)
    lat2 = lat.reshape(rows, feat)
    # channel of each (row, token): constant within a token iff tokens do not straddle channels
    cmin, cmax = c.min(axis=2), c.max(axis=2)
    if not np.array_equal(cmin, cmax):
        return None                                   # geometry without whole tokens per channel
    chan = cmin                                       # [rows, Z]
    # group rows by data pattern
    packed = np.packbits(data2, axis=1)
    _, first, inverse = np.unique(packed, axis=0, return_index=True, return_inverse=True)
    inverse = inverse.reshape(-1)
    plan = _Plan()
    plan.groups = []
    plan.rows, plan.feat, plan.lattice_size, plan.C, plan.Z = rows, feat, C * Z * Hl * Wl, C, Z
    used = np.zeros(plan.lattice_size, dtype=np.int64)
    for gi, r0 in enumerate(first):
        members = np.nonzero(inverse == gi)[0]
        cols = np.nonzero(data2[r0])[0]
        ncols = np.nonzero(~data2[r0])[0]
        g = _Plan()
        g.rows = members
        g.cols = cols
        g.gather = lat2[np.ix_(members, cols)]        # [n_rows, n_cols] lattice indices
        np.add.at(used, g.gather.reshape(-1), 1)
        # per token k: which non-data columns belong to it (for the constant term)
        g.ncols_by_token = [ncols[(ncols // C) == kk] for kk in range(Z)]
        g.chan = chan[members]                        # [n_rows, Z]
        plan.groups.append(g)
    # every lattice element feeds exactly one (row, column): the data part of the operand is a permutation
    plan.is_permutation = bool((used == 1).all())
    return plan


def _device_plan(plan, device):
    L, C, Z = plan.lattice_size, plan.C, plan.Z
    dev = _Plan()
    dev.rows, dev.feat, dev.lattice_size, dev.C, dev.Z = plan.rows, plan.feat, L, C, Z
    dev.groups = []
    off = 0

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).to(device)
    for g in plan.groups:
        d = _Plan()
        d.n_rows, d.n_cols = g.gather.shape
        # operand width: a multiple of 64 (the library's macro-tile depth).  Measured on the 552 960-row GEMMs of the bench
        # (scratch/r04/occproj_shapes.py, tuned solutions): K = 824 -> 832: 4.32 -> 3.49 ms forward, 728 -> 768: 3.93 -> 3.26,
        # 776 -> 832: 4.12 -> 3.49 -- with K % 64 != 0 the library falls back to its stream-K kernel at 0.37 of the MFMA peak
        d.k_aug = (d.n_cols + Z + 1 + 63) // 64 * 64
        # operand columns: data | b_up[chan(row,k)] k<Z | 1 | zero padding, as indices into
        # lat_aug = [lattice (L), up_bias (C), 1, 0]
        idx = np.full((d.n_rows, d.k_aug), L + C + 1, dtype=np.int64)
        idx[:, :d.n_cols] = g.gather
        idx[:, d.n_cols:d.n_cols + Z] = L + g.chan
        idx[:, d.n_cols + Z] = L + C
        d.gather_aug = t(idx.reshape(-1))
        d.scatter = t(g.gather.reshape(-1))           # lattice position of every data element
        # run structure of the data columns (R equal runs of contiguous lattice elements per row -- one per token
        # of the raw view): the HIP run copies (ver_run_gather / ver_run_scatter) replace index_select / index_copy_
        d.run_len = 0
        brk = np.nonzero(np.diff(g.gather[0]) != 1)[0] + 1 if d.n_cols and d.n_rows else np.zeros(0, dtype=np.int64)
        starts = np.concatenate([[0], brk])
        lens = np.diff(np.concatenate([starts, [d.n_cols]]))
        if d.n_cols and d.n_rows and len(set(lens.tolist())) == 1 and lens[0] % 4 == 0 and d.n_cols % 4 == 0:
            rl = int(lens[0])
            ok = bool((g.gather.reshape(d.n_rows, -1, rl) == g.gather[:, ::rl][:, :, None] + np.arange(rl)).all())
            if ok and bool((g.gather[:, ::rl] % 4 == 0).all()):
                d.run_len = rl
                d.run_start = torch.from_numpy(np.ascontiguousarray(g.gather[:, ::rl]).astype(np.int32)).to(device)
                d.aug_idx = torch.from_numpy(np.ascontiguousarray(idx[:, d.n_cols:]).astype(np.int32)).to(device)
        d.aug_tail = t(idx[:, d.n_cols:].reshape(-1) - L)          # the same columns as indices into [up_bias (C), 1, 0]
        d.cols = t(g.cols)
        d.ncols_by_token = [t(n) for n in g.ncols_by_token]
        d.chan = t(g.chan.reshape(-1))
        d.members = g.rows
        d.offset = off
        off += d.n_rows
        dev.groups.append(d)
    dev._row_index = {}
    dev.row_map = _periodic_row_map(plan, dev)
    return dev


def _periodic_row_map(plan, dev):
    """The run structure as ``ver_lattice_rows`` wants it, or None: every row of every group is R runs of one length, the
    k-th run of row r of group g starts at flat lattice index k * (L / R) + r * period + offset_g, and the groups' runs tile
    a period in order (vocc.py: 5 groups, runs of 204 | 204 | 192 | 180 | 180 = period 960, 2 880 rows per group)."""
    L = plan.lattice_size
    gs = list(zip(plan.groups, dev.groups))
    if not gs or any(not d.run_len for _, d in gs) or L >= 2 ** 31:
        return None
    R = {d.n_cols // d.run_len for _, d in gs}
    n_rows = {d.n_rows for _, d in gs}
    if len(R) != 1 or len(n_rows) != 1:
        return None
    R, n_rows = R.pop(), n_rows.pop()
    period = sum(d.run_len for _, d in gs)
    if R > 8 or L % R or L // R != n_rows * period or len(gs) > 8:
        return None
    quarter = L // R
    segs = []
    for gi, (g, d) in enumerate(gs):
        off = int(g.gather[0, 0])
        want = (np.arange(R)[None, :, None] * quarter + np.arange(n_rows)[:, None, None] * period + off
                + np.arange(d.run_len)[None, None, :]).reshape(n_rows, R * d.run_len)
        if off >= period or off % 4 or d.run_len % 4 or not np.array_equal(g.gather, want):
            return None
        segs.append((off, d.run_len, gi))
    segs.sort()
    pos = 0
    for off, rl, _ in segs:
        if off != pos:
            return None
        pos += rl
    return dict(quarter=quarter, period=period, n_rows=n_rows, seg_off=[s_[0] for s_ in segs], seg_len=[s_[1] for s_ in segs],
                seg_group=[s_[2] for s_ in segs])


def _row_map_for(plan, bs):
    """(row map with the element offsets of a ``bs``-sample operand buffer, [(first element, elements) per group], total)."""
    spans, total = [], 0
    for d in plan.groups:
        n = bs * d.n_rows * d.k_aug
        spans.append((total, n))
        total += n
    rm = dict(plan.row_map)
    rm['seg_base'] = [spans[gi][0] for gi in rm['seg_group']]
    rm['seg_pitch'] = [plan.groups[gi].k_aug for gi in rm['seg_group']]
    rm['seg_rows'] = [plan.groups[gi].n_rows for gi in rm['seg_group']]
    return rm, spans, total


def get_plan(C, Z, Hf, Wf, device):
    key = (C, Z, Hf, Wf, str(device))
    if key not in _PLAN_CACHE:
        host_key = (C, Z, Hf, Wf, 'host')
        if host_key not in _PLAN_CACHE:
            _PLAN_CACHE[host_key] = _build_plan(C, Z, Hf, Wf)
        plan = _PLAN_CACHE[host_key]
        _PLAN_CACHE[key] = None if (plan is None or not plan.is_permutation) else _device_plan(plan, device)
    return _PLAN_CACHE[key]


def _row_index(plan, bs, device):
    """[bs * rows]: buffer row of (sample b, position q = a*Wf + b') and its inverse."""
    if bs not in plan._row_index:
        fwd = np.empty((bs, plan.rows), dtype=np.int64)
        for g in plan.groups:
            fwd[:, g.members] = bs * g.offset + np.arange(bs)[:, None] * g.n_rows + np.arange(g.n_rows)[None, :]
        fwd = fwd.reshape(-1)
        inv = np.empty_like(fwd)
        inv[fwd] = np.arange(fwd.size)
        plan._row_index[bs] = (torch.from_numpy(fwd).to(device), torch.from_numpy(inv).to(device))
    return plan._row_index[bs]


def _token_matrix(plan, like):
    """[Z*C, groups*Z] 0/1 (counts): column (g, k) marks the non-data columns of token k in group g (``ncols_by_token``);
    built once per plan and dtype."""
    cache = plan.__dict__.setdefault('_token_matrices', {})
    key = (like.dtype, str(like.device))
    if key not in cache:
        Z = plan.Z
        m = torch.zeros(plan.Z * plan.C, len(plan.groups) * Z, dtype=like.dtype, device=like.device)
        for gi, g in enumerate(plan.groups):
            for k, n in enumerate(g.ncols_by_token):
                m[:, gi * Z + k].index_add_(0, n.to(like.device), torch.ones(n.numel(), dtype=like.dtype, device=like.device))
        cache[key] = m
    return cache[key]


class _OccProjLattice(torch.autograd.Function):

    @staticmethod
    def forward(ctx, e, up_bias, weight, bias, plan):
        # lattice layouts of the upsample: 0 plain [bs,Z,Hl,Wl,C], 1 planar [4,bs,Z,Hl/2,Wl/2,C]
        # (plane 2pm+pn = (2y+pm, 2x+pn)), 3 planar z-split [4,bs,2,Hl/2,Wl/2,2,C] (Z = 4)
        layout = {5: 0, 6: 1, 7: 3}[e.dim()]
        if layout == 3:
            _, bs, _, hh, wh, _, C = e.shape
            Z, Hl, Wl = 4, 2 * hh, 2 * wh
        elif layout == 1:
            _, bs, Z, hh, wh, C = e.shape
            Hl, Wl = 2 * hh, 2 * wh
        else:
            bs, Z, Hl, Wl, C = e.shape
        dt = e.dtype
        out_dim = weight.shape[0]
        L = plan.lattice_size
        use_hip = e.is_cuda and dt in (torch.float32, torch.bfloat16)
        # bf16 on the GPU with the periodic run structure: the lattice goes straight into the operand rows of all groups
        # (ver_lattice_rows) -- the channel-first copy of it (1.06 GB at 192 viewpoints) is neither written nor read
        fused = use_hip and dt == torch.bfloat16 and plan.row_map is not None and os.environ.get('VER_LATTICE_ROWS', '1') == '1'
        if fused:
            from ..hipops import lattice_rows
            row_map, spans, total = _row_map_for(plan, bs)
            rows_buf = torch.empty(total, dtype=dt, device=e.device)
            lattice_rows(e.contiguous(), rows_buf, row_map, (Hl, Wl), layout, True)
            tail = torch.cat([up_bias.to(dt), torch.ones(1, dtype=dt, device=e.device), torch.zeros(1, dtype=dt, device=e.device)])
        else:
            lat = torch.empty(bs, (L + C + 2 + 7) // 8 * 8, dtype=dt, device=e.device)   # (rows 16-byte aligned)
            lat5 = lat[:, :L].view(bs, C, Z, Hl, Wl)                            # channel-first lattice
            if use_hip:
                from ..hipops import lattice_transpose
                lattice_transpose(e.contiguous(), lat, (Hl, Wl), layout, True)
            else:
                from .upsample import _to_plain
                lat5.copy_(_to_plain(e, layout).permute(0, 4, 1, 2, 3))
            lat[:, L:L + C] = up_bias.to(dt)
            lat[:, L + C] = 1
            lat[:, L + C + 1] = 0
        w = weight.to(dt)
        out = torch.empty(bs * plan.rows, out_dim, dtype=dt, device=e.device)
        operands, weights = [], []
        # summed constant columns of every (group, token): weight @ (0/1 membership matrix) -- one small product in the
        # master weight's precision instead of an index_select + sum per token and group (20 pairs at Z = 4)
        with torch.autocast(weight.device.type, enabled=False):
            s_all = weight @ _token_matrix(plan, weight)                                   # [out, groups * Z]
        for gi, g in enumerate(plan.groups):
            if fused:
                a = rows_buf[spans[gi][0]:spans[gi][0] + spans[gi][1]].view(bs * g.n_rows, g.k_aug)
                # the augmentation columns (b_up of the row's channels | 1 | 0 ...) are the same for every sample
                a.view(bs, g.n_rows, g.k_aug)[:, :, g.n_cols:] = tail.index_select(0, g.aug_tail).view(g.n_rows, g.k_aug - g.n_cols)
            elif use_hip and g.run_len:
                from ..hipops import run_gather
                a = torch.empty(bs * g.n_rows, g.k_aug, dtype=dt, device=e.device)
                run_gather(lat, g.run_start, g.aug_idx, a, g.n_rows, g.run_len)
            else:
                a = lat.index_select(1, g.gather_aug).view(bs * g.n_rows, g.k_aug)
            # W_aug^T [out, k_aug] = data columns | summed constant columns per token | bias | 0
            s = s_all[:, gi * Z:(gi + 1) * Z]                                              # [out, Z] fp32
            pad = g.k_aug - g.n_cols - Z - 1
            wa = torch.cat([w.index_select(1, g.cols), s.to(dt), bias.to(dt)[:, None],
                            w.new_zeros(out_dim, pad)], 1)
            with gemm_timed('head_gemm_fwd', a.shape[0], a.shape[1], wa.shape[0]):
                torch.mm(a, wa.t(), out=out[bs * g.offset: bs * (g.offset + g.n_rows)])
            operands.append(a)
            weights.append(wa)
        ctx.plan, ctx.shape, ctx.layout, ctx.e_shape, ctx.fused = plan, (bs, Z, Hl, Wl, C), layout, tuple(e.shape), fused
        ctx.save_for_backward(weight, *operands, *weights)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        bs, Z, Hl, Wl, C = ctx.shape
        weight = ctx.saved_tensors[0]
        ng = len(plan.groups)
        operands, weights = ctx.saved_tensors[1:1 + ng], ctx.saved_tensors[1 + ng:]
        grad_out = grad_out.contiguous()
        dt = grad_out.dtype
        acc = torch.float64 if dt == torch.float64 else torch.float32       # parameter gradients
        fused = ctx.fused
        if fused:
            row_map, spans, total = _row_map_for(plan, bs)
            d_rows = torch.empty(total, dtype=dt, device=grad_out.device)
        else:
            d_lat = torch.empty(bs, plan.lattice_size, dtype=dt, device=grad_out.device)
        d_weight = torch.zeros(weight.shape, dtype=acc, device=weight.device)
        d_bias = torch.zeros(weight.shape[0], dtype=acc, device=weight.device)
        d_up = torch.zeros(C, dtype=acc, device=weight.device)
        d_tokens = []
        for gi, (g, a, wa) in enumerate(zip(plan.groups, operands, weights)):
            go = grad_out[bs * g.offset: bs * (g.offset + g.n_rows)]
            # d(operand): data columns go back to their lattice positions (each written exactly once)
            if fused:
                d_all = d_rows[spans[gi][0]:spans[gi][0] + spans[gi][1]].view(bs * g.n_rows, g.k_aug)
                # operand widths that are not whole 256-column tiles (832 at the vocc.py sizes) go to ver_gemm_nn: the
                # library's solution for [552 960, 4480] x [4480, 832] runs at 0.92 PFLOP/s, ours at 1.04 (4.49 -> 3.97 ms,
                # three groups per step; scratch/r06/occproj_gemm_bench.py -- at 768 columns, and in the forward, they tie)
                from .. import hipops
                if _OWN_DGRAD and g.k_aug % 256 and go.shape[0] >= 14000 and hipops.gemm_nn_supported(go, wa):
                    hipops.gemm_nn(go, wa, out=d_all, timer_class='head_gemm_dgrad')
                else:
                    with gemm_timed('head_gemm_dgrad', go.shape[0], go.shape[1], wa.shape[1]):
                        torch.mm(go, wa, out=d_all)
                d_const = d_all[:, g.n_cols:g.n_cols + Z]
            elif d_lat.is_cuda and g.run_len and dt in (torch.float32, torch.bfloat16):
                # the Z constant columns sit right behind the data columns of W_aug: ONE GEMM returns both (their own
                # [rows, Z] product read all of `go` again for four output columns: 0.85 ms per group at 192 viewpoints,
                # profiles/r04_gemm_ledger.csv); the run copies take the row pitch of the wider buffer as it is
                from ..hipops import run_scatter
                with gemm_timed('head_gemm_dgrad', go.shape[0], go.shape[1], wa.shape[1]):
                    d_all = torch.mm(go, wa)                                         # [bs*n_rows, k_aug] (pads: zero columns)
                run_scatter(d_all, d_lat, g.run_start, g.n_rows, g.run_len)
                d_const = d_all[:, g.n_cols:g.n_cols + Z]
            else:
                d_data = torch.mm(go, wa[:, :g.n_cols])                              # [bs*n_rows, n_cols]
                d_lat.index_copy_(1, g.scatter, d_data.view(bs, g.n_rows * g.n_cols))
                d_const = torch.mm(go, wa[:, g.n_cols:g.n_cols + Z].contiguous())     # [bs*n_rows, Z]
            d_up.index_add_(0, g.chan, d_const.reshape(bs, -1).sum(0, dtype=acc))
            # d(W_aug^T) = go^T a
            d_wa = rows_tn(go, a, out_dtype=acc)                                     # [out, k_aug], fp32 sums
            d_weight.index_add_(1, g.cols, d_wa[:, :g.n_cols])
            d_tokens.append(d_wa[:, g.n_cols:g.n_cols + Z])
            d_bias += d_wa[:, g.n_cols + Z]
        # every constant column of a token receives that token's gradient: the adjoint of `weight @ membership`
        d_weight.addmm_(torch.cat(d_tokens, 1), _token_matrix(plan, d_weight).t())
        if fused:
            from ..hipops import lattice_rows
            d_e = d_rows.new_empty(ctx.e_shape)
            lattice_rows(d_e, d_rows, row_map, (Hl, Wl), ctx.layout, False)
            return d_e, d_up, d_weight, d_bias, None
        d5 = d_lat.view(bs, C, Z, Hl, Wl)
        if d_lat.is_cuda and dt in (torch.float32, torch.bfloat16):
            from ..hipops import lattice_transpose
            d_e = d_lat.new_empty(ctx.e_shape)
            lattice_transpose(d_e, d_lat, (Hl, Wl), ctx.layout, False)
        else:
            from .upsample import _from_plain
            d_e = _from_plain(d5.permute(0, 2, 3, 4, 1), ctx.layout)
        return d_e, d_up, d_weight, d_bias, None


class _SelectRows(torch.autograd.Function):
    """y = x.index_select(0, perm) for a permutation; backward = index_select with its inverse."""

    @staticmethod
    def forward(ctx, x, perm, inv_perm):
        ctx.save_for_backward(inv_perm)
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, g):
        inv_perm, = ctx.saved_tensors
        return g.index_select(0, inv_perm), None, None


def occ_proj_from_lattice(e, up_bias, weight, bias):
    """e: even lattice of the upsample output, channels-last [bs, Z, Hl, Wl, C], planar
    [4, bs, Z, Hl/2, Wl/2, C] or planar z-split [4, bs, 2, Hl/2, Wl/2, 2, C]; up_bias: bias of
    the last ConvTranspose3d [C]; weight [out, Z*C], bias [out] of ``occ_proj``.
    Returns ``(rows [bs*Hf*Wf, out] in group-major order, plan)`` -- ``rows_to_voxels`` maps
    anything computed row-wise from it to the (a, b) order of
    ``occ_proj(Y.view(bs,Z,Hf,Wf,C).permute(0,2,3,1,4).flatten(3))`` -- or None when the geometry has
    no whole-token structure (the caller then takes the dense path)."""
    if e.dim() == 7:                                     # planar z-split lattice (Z = 4)
        _, bs, _, hh, wh, _, C = e.shape
        Z, Hl, Wl = 4, 2 * hh, 2 * wh
    elif e.dim() == 6:                                   # planar lattice
        _, bs, Z, hh, wh, C = e.shape
        Hl, Wl = 2 * hh, 2 * wh
    else:
        bs, Z, Hl, Wl, C = e.shape
    plan = get_plan(C, Z, 2 * Hl, 2 * Wl, e.device)
    if plan is None:
        return None
    return _OccProjLattice.apply(e, up_bias, weight, bias, plan), plan


def rows_to_voxels(x, plan, bs):
    """x [bs*Hf*Wf, ...] in the group-major row order -> [bs, Hf*Wf, ...] in (a, b) order."""
    fwd, inv = _row_index(plan, bs, x.device)
    return _SelectRows.apply(x, fwd, inv).view(bs, plan.rows, *x.shape[1:])


def voxels_to_rows(x, plan, bs):
    """Inverse of ``rows_to_voxels`` for tensors that need no gradient (targets): x [bs, Hf*Wf, ...] in (a, b) order ->
    [bs*Hf*Wf, ...] in the group-major row order of the GEMM buffer."""
    _, inv = _row_index(plan, bs, x.device)
    return x.reshape(bs * plan.rows, *x.shape[2:]).index_select(0, inv)
