"""Parity of the HIP kernels (through the C ABI, via ctypes) against the CPU oracle and the
golden vectors.  Needs an MI355X: run with ``-m gpu``.

Tolerances: forward 1e-4 absolute (north_star, fp32); gradients |a-b| <= 1e-4 + 1e-5*|b|
(values reach O(100)); visibility masks / index lists bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
from util import close, golden, maxdiff, oracle, oracle_slots, pkg

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda'


def _fn():
    return pkg('hipops').MultiScaleDeformableAttnFunction_fp32


# ------------------------------------------------------------------------------- a6: mmcv boundary
@pytest.mark.parametrize('name', list(cases.MSDA_CASES))
def test_msda_forward_backward_vs_golden_and_oracle(name):
    o = oracle()
    g = golden('msda_core_' + name)
    c = cases.msda_inputs(**cases.MSDA_CASES[name])
    value = T(c['value']).to(DEV).requires_grad_(True)
    loc = T(c['loc']).to(DEV).requires_grad_(True)
    w = T(c['w']).to(DEV).requires_grad_(True)
    out = _fn().apply(value, T(c['shapes']).to(DEV), T(c['level_start']).to(DEV), loc, w, 64)
    out.backward(T(c['grad_out']).to(DEV))
    vc = T(c['value']).requires_grad_(True)
    lc = T(c['loc']).requires_grad_(True)
    wc = T(c['w']).requires_grad_(True)
    ref = o.msda_core(vc, c['shapes'].tolist(), lc, wc)
    ref.backward(T(c['grad_out']))
    assert maxdiff(out.detach().cpu(), ref.detach()) < 2e-5
    assert close(value.grad.cpu(), vc.grad)
    assert close(loc.grad.cpu(), lc.grad)
    assert close(w.grad.cpu(), wc.grad)
    so, sv, sl = (5, 3, 5) if name == 'vocc' else (1, 1, 1)
    assert maxdiff(out.detach().cpu()[:, ::so], g['out']) < 2e-5
    assert close(value.grad.cpu()[:, ::sv], g['grad_value'])
    assert close(loc.grad.cpu()[:, ::sl], g['grad_loc'])
    assert close(w.grad.cpu()[:, ::sl], g['grad_w'])


def test_msda_edge_cases():
    """empty query set; every sample outside the map; NaN locations; one-pixel map."""
    fn = _fn()
    shapes = torch.tensor([[3, 4]], device=DEV)
    lsi = torch.tensor([0], device=DEV)
    value = torch.randn(2, 12, 2, 16, device=DEV)
    out = fn.apply(value, shapes, lsi, torch.zeros(2, 0, 2, 1, 4, 2, device=DEV),
                   torch.zeros(2, 0, 2, 1, 4, device=DEV), 64)
    assert out.shape == (2, 0, 32)
    w = torch.full((2, 5, 2, 1, 4), 0.25, device=DEV)
    far = torch.full((2, 5, 2, 1, 4, 2), 7.5, device=DEV)
    assert float(fn.apply(value, shapes, lsi, far, w, 64).abs().max()) == 0.0
    assert float(fn.apply(value, shapes, lsi, -far, w, 64).abs().max()) == 0.0
    nan = torch.full((2, 5, 2, 1, 4, 2), float('nan'), device=DEV)
    assert float(fn.apply(value, shapes, lsi, nan, w, 64).abs().max()) == 0.0
    # constant map: inside the map the bilinear weights sum to 1 -> output = const * sum(w)
    const = torch.full((1, 12, 2, 16), 3.0, device=DEV)
    mid = torch.rand(1, 7, 2, 1, 4, 2, device=DEV) * 0.5 + 0.25
    w1 = torch.rand(1, 7, 2, 1, 4, device=DEV)
    got = fn.apply(const, shapes, lsi, mid, w1, 64).view(1, 7, 2, 16)
    want = 3.0 * w1.sum(-1).sum(-1)
    assert maxdiff(got.cpu(), want[..., None].expand(-1, -1, -1, 16).cpu()) < 1e-5
    one = torch.tensor([[1, 1]], device=DEV)
    v1 = torch.randn(1, 1, 1, 8, device=DEV)
    centre = torch.full((1, 1, 1, 1, 4, 2), 0.5, device=DEV)
    got = fn.apply(v1, one, lsi, centre, torch.full((1, 1, 1, 1, 4), 0.25, device=DEV), 64)
    assert maxdiff(got.cpu().view(-1), v1.cpu().view(-1)) < 1e-6


def test_msda_linearity_at_full_size():
    """Size-independent property on the vocc 50x50x16 shape (L = 6423 rows / camera):
    the op is linear in value and in the attention weights."""
    fn = _fn()
    gen = torch.Generator(device='cpu').manual_seed(3)
    B, Nq = 6, 6423
    shapes = torch.tensor([[14, 14]], device=DEV)
    lsi = torch.tensor([0], device=DEV)
    v1 = torch.randn(B, 196, 8, 96, generator=gen).to(DEV)
    v2 = torch.randn(B, 196, 8, 96, generator=gen).to(DEV)
    loc = (torch.rand(B, Nq, 8, 1, 8, 2, generator=gen) * 1.4 - 0.2).to(DEV)
    w = torch.rand(B, Nq, 8, 1, 8, generator=gen).to(DEV)
    a = fn.apply(v1, shapes, lsi, loc, w, 64)
    b = fn.apply(v2, shapes, lsi, loc, w, 64)
    ab = fn.apply(v1 * 2.0 + v2, shapes, lsi, loc, w, 64)
    assert float((ab - (2.0 * a + b)).abs().max()) < 1e-4
    half = fn.apply(v1, shapes, lsi, loc, w * 0.5, 64)
    assert float((half - 0.5 * a).abs().max()) < 1e-5


# ------------------------------------------------------------------------------- a2 + a3: projection
@pytest.mark.parametrize('gname', list(cases.GRIDS))
def test_projection_visibility_and_lists(gname):
    hip = pkg('hipops')
    syn = pkg('synthetic')
    o = oracle()
    g = golden('point_sampling')
    z, h, w = cases.GRIDS[gname]
    nq = z * h * w
    w2p, org = syn.camera_batch(2, seed=1)
    hit = hip.project_points(T(w2p).to(DEV), T(org).to(DEV), cases.PC_RANGE, z, h, w)
    torch.cuda.synchronize()
    vis = hit.vis.cpu().numpy()
    ref3d = o.reference_points_3d(z, h, w)
    for b in range(2):
        key = '%s_b%d_' % (gname, b)
        want = np.unpackbits(g[key + 'mask'], axis=1)[:, :nq].astype(bool)        # reference's mask
        got = ((vis[b][None, :] >> np.arange(6)[:, None]) & 1).astype(bool)
        assert np.array_equal(got, want), 'visibility differs from the reference in %d voxels' % (
            (got != want).sum())
        uv_o, mask_o = o.point_sampling(ref3d, T(w2p[b]), T(org[b]), cases.PC_RANGE)
        assert np.array_equal(got, mask_o.numpy())
        # visible voxels: tight.  Elsewhere |uv| reaches 1e6 (depth clamped to 1e-5) and voxels
        # near a camera plane lose digits to cancellation in q_z, so only a loose check.
        uv_h = hit.uv[b, :, :, 0].cpu()
        m = T(want)
        assert close(uv_h[m], uv_o[m], atol=1e-6, rtol=1e-6)
        rel = ((uv_h - uv_o).abs() / (1e-5 + uv_o.abs()))
        assert float((rel > 1e-4).float().mean()) < 0.01 and bool(torch.isfinite(uv_h).all())
        step = 16 if gname == 'c2' else 1
        gm = m[:, ::step]
        assert close(uv_h[:, ::step][gm], T(g[key + 'uv'])[gm], atol=1e-6, rtol=1e-6)
        cnt = hit.vis_cnt[b].cpu().tolist()
        assert cnt == g[key + 'hits'].tolist()
        for c in range(6):
            lst = hit.vis_list[b, c, :cnt[c]].cpu().numpy()
            assert np.array_equal(lst, np.nonzero(want[c])[0])          # = reference's indexes[c]
        zc = int(hit.zero_cnt[b])
        assert np.array_equal(hit.zero_list[b, :zc].cpu().numpy(), np.nonzero(want.sum(0) != 1)[0])
    assert hit.mask().shape == (6, 2, nq, 1)


def test_projection_analytic_identity_camera():
    """Known answer: world2pixel = K with no rotation: p=(x,y,z) -> u = (fx*x/z + cx)/1280."""
    hip = pkg('hipops')
    k = np.eye(4, dtype=np.float32)
    k[0, 0] = k[1, 1] = 100.0
    k[0, 2], k[1, 2] = 640.0, 512.0
    w2p = np.tile(k, (1, 6, 1, 1))
    org = np.zeros((1, 3), dtype=np.float32)
    rng = (-1.0, -1.0, 0.0, 1.0, 1.0, 2.0)
    hit = hip.project_points(T(w2p).to(DEV), T(org).to(DEV), rng, 2, 2, 2)
    uv = hit.uv[0, 0, :, 0].cpu().numpy()
    n = 0
    for kz in range(2):
        for j in range(2):
            for i in range(2):
                x, y, zz = -0.5 + i, -0.5 + j, 0.5 + kz
                assert abs(uv[n, 0] - (100 * x / zz + 640) / 1280) < 1e-6
                assert abs(uv[n, 1] - (100 * y / zz + 512) / 1024) < 1e-6
                n += 1
    assert hit.vis.cpu().tolist() == [[63] * 8]


def test_hits_from_reference_layout_mask():
    hip = pkg('hipops')
    g = golden('sca_small')
    mask = T(g['mask'])                    # [6,1,64,1]
    hit = hip.hits_from_mask(T(g['uv']).to(DEV), mask.to(DEV))
    m = mask[:, 0, :, 0].numpy()
    cnt = hit.vis_cnt[0].cpu().tolist()
    assert cnt == m.sum(1).tolist() and cnt[3] == 0
    for c in range(6):
        assert np.array_equal(hit.vis_list[0, c, :cnt[c]].cpu().numpy(), np.nonzero(m[c])[0])
    assert maxdiff(hit.uv[0].cpu(), T(g['uv'])[:, 0]) == 0.0
    # D = 2 anchors: a voxel is visible if any anchor is
    mask2 = torch.zeros(6, 1, 10, 2, dtype=torch.bool)
    mask2[1, 0, 3, 1] = True
    mask2[4, 0, 3, 0] = True
    hit2 = hip.hits_from_mask(torch.zeros(6, 1, 10, 2, 2, device=DEV), mask2.to(DEV))
    assert int(hit2.vis[0, 3]) == (1 << 1) | (1 << 4) and int(hit2.vis[0].sum()) == 18


# ------------------------------------------------------------------------------- a4 + a5 + a6: fused gather
def _random_sca_case(seed, B, grid, heads, hd, P, map_hw=(14, 14), D=1):
    syn = pkg('synthetic')
    hip = pkg('hipops')
    rng = np.random.default_rng(seed)
    z, h, w = grid
    nq = z * h * w
    nk = map_hw[0] * map_hw[1]
    w2p, org = syn.camera_batch(B, seed=1)
    hit = hip.project_points(T(w2p).to(DEV), T(org).to(DEV), cases.PC_RANGE, z, h, w)
    value = rng.standard_normal((B, 6, nk, heads, hd)).astype(np.float32)
    offsets = (rng.standard_normal((B, nq, heads, P, 2)) * 3.0).astype(np.float32)
    logits = rng.standard_normal((B, nq, heads, P)).astype(np.float32)
    gslots = rng.standard_normal((B, nq, heads * hd)).astype(np.float32)
    return hit, value, offsets, logits, gslots


@pytest.mark.parametrize('heads,hd,P,grid,mhw', [
    (8, 96, 8, (4, 15, 15), (14, 14)), (4, 8, 8, (2, 6, 5), (14, 14)), (2, 16, 4, (2, 6, 5), (14, 14)),
    (2, 32, 8, (3, 7, 7), (14, 14)), (2, 64, 4, (2, 9, 8), (14, 14)), (1, 128, 8, (2, 6, 5), (9, 11))])
def test_sca_gather_forward_backward_vs_oracle(heads, hd, P, grid, mhw):
    hip = pkg('hipops')
    o = oracle()
    B = 3
    hit, value, offsets, logits, gslots = _random_sca_case(7, B, grid, heads, hd, P, map_hw=mhw)
    v = T(value).to(DEV).requires_grad_(True)
    of = T(offsets).to(DEV).requires_grad_(True)
    lg = T(logits).to(DEV).requires_grad_(True)
    slots = hip.sca_gather(v, of, lg, hit, mhw[0], mhw[1])
    slots.backward(T(gslots).to(DEV))
    mask = hit.mask()[:, :, :, 0].permute(1, 0, 2).cpu()
    vc, oc, lc = (T(value).requires_grad_(True), T(offsets).requires_grad_(True),
                  T(logits).requires_grad_(True))
    ref = oracle_slots(o, vc, oc, lc, hit.uv.cpu(), mask, mhw)
    ref.backward(T(gslots))
    assert maxdiff(slots.detach().cpu(), ref.detach()) < 2e-5
    assert close(v.grad.cpu(), vc.grad)
    assert close(of.grad.cpu(), oc.grad)
    assert close(lg.grad.cpu(), lc.grad)
    # unseen voxels produce exact zeros, forward is run-to-run deterministic
    unseen = ~mask.any(1)
    assert float(slots.detach().cpu()[unseen].abs().max()) == 0.0 if unseen.any() else True
    again = hip.sca_gather(v.detach(), of.detach(), lg.detach(), hit, mhw[0], mhw[1])
    assert torch.equal(again, slots.detach())


@pytest.mark.parametrize('heads,hd,P,grid', [(8, 96, 8, (4, 15, 15)), (2, 32, 4, (2, 6, 5)), (4, 8, 8, (2, 6, 5)),
                                             (4, 64, 8, (2, 6, 5)), (2, 32, 8, (3, 7, 6))])
def test_sca_gather_bf16_value(heads, hd, P, grid):
    """value stored as bf16 (what value_proj emits under bf16 autocast).  Against the oracle evaluated on the SAME
    bf16-rounded values: the generic kernels compute in fp32 and keep the fp32 tolerance; the corner-slot kernel
    (8 points, head_dim % 32 == 0) accumulates the <= 8 points of a (voxel, head, corner) in packed fp16 by default
    (VER_SCA_FWD_MATH=2) -- inside the north star's 1e-2 for bf16 arithmetic; its exact modes
    (0 / 1) are held to 2e-5 by test_sca_gather_launch_modes."""
    hip = pkg('hipops')
    o = oracle()
    hit, value, offsets, logits, gslots = _random_sca_case(17, 2, grid, heads, hd, P)
    vb = T(value).to(DEV).to(torch.bfloat16).requires_grad_(True)
    of = T(offsets).to(DEV).requires_grad_(True)
    lg = T(logits).to(DEV).requires_grad_(True)
    slots = hip.sca_gather(vb, of, lg, hit, 14, 14)
    assert slots.dtype == torch.float32
    slots.backward(T(gslots).to(DEV))
    assert vb.grad.dtype == torch.bfloat16
    mask = hit.mask()[:, :, :, 0].permute(1, 0, 2).cpu()
    vc = vb.detach().float().cpu().requires_grad_(True)
    oc, lc = T(offsets).requires_grad_(True), T(logits).requires_grad_(True)
    ref = oracle_slots(o, vc, oc, lc, hit.uv.cpu(), mask, (14, 14))
    ref.backward(T(gslots))
    if P == 8 and hd % 32 == 0:
        from util import rel_l2
        assert maxdiff(slots.detach().cpu(), ref.detach()) < 1e-2
        assert rel_l2(slots.detach().cpu(), ref.detach()) < 2e-3
    else:
        assert maxdiff(slots.detach().cpu(), ref.detach()) < 2e-5
    assert close(of.grad.cpu(), oc.grad)
    assert close(lg.grad.cpu(), lc.grad)
    # d(value) is rounded to bf16 ONCE, at the end (accumulation in fp32 / exact integers): one bf16 ulp (2^-8 relative)
    # per element plus a small fraction of the largest element, and a round-to-nearest relative L2 (2^-9 / sqrt(3) = 1.1e-3)
    from util import rel_l2
    gv, gr = vb.grad.float().cpu(), vc.grad
    assert close(gv, gr, atol=1e-3 * float(gr.abs().max()), rtol=2.0 ** -8)
    assert rel_l2(gv, gr) < 2e-3


def test_sca_gather_bf16_value_range_contract():
    """include/ver_ops.h, ``ver_sca_forward`` on VER_BF16 value at the vocc.py shape: the tile is converted to fp16 in LDS,
    so |value| > 65504 SATURATES (round toward zero: no inf, no NaN from finite inputs) and tiny magnitudes truncate
    toward zero.  With every value at +-1e6 the output is the saturated constant wherever a sample falls inside the map
    (a convex combination of +-65504), never inf / NaN; values of 1e-6 (fp16 subnormals, grid 6e-8) come out within a few
    grid steps; and magnitudes up to 6e4 are still exact to fp16 accumulation accuracy."""
    hip = pkg('hipops')
    o = oracle()
    hit, value, offsets, logits, _ = _random_sca_case(29, 2, (4, 15, 15), 8, 96, 8)
    of, lg = T(offsets).to(DEV), T(logits).to(DEV)
    mask = hit.mask()[:, :, :, 0].permute(1, 0, 2).cpu()

    def ref_of(v):
        return oracle_slots(o, v.float().cpu(), T(offsets), T(logits), hit.uv.cpu(), mask, (14, 14))

    big = torch.full(value.shape, 1e6, dtype=torch.bfloat16, device=DEV)
    big[:, :, ::2] = -1e6
    out = hip.sca_gather(big, of, lg, hit, 14, 14)
    assert torch.isfinite(out).all()
    sat = torch.clamp(big.float(), -65504.0, 65504.0)
    want = ref_of(sat)
    assert float((out.cpu() - want).abs().max()) <= 2e-3 * 65504.0          # fp16 accumulation of saturated terms
    assert float(out.abs().max()) <= 65504.0 * 1.001
    tiny = torch.full(value.shape, 1e-6, dtype=torch.bfloat16, device=DEV)
    out = hip.sca_gather(tiny, of, lg, hit, 14, 14)
    assert float((out.cpu() - ref_of(tiny)).abs().max()) <= 5e-7
    large = (T(value).to(DEV) * 2.0e4).clamp(-6.0e4, 6.0e4).to(torch.bfloat16)
    out = hip.sca_gather(large, of, lg, hit, 14, 14)
    want = ref_of(large)
    from util import rel_l2
    assert torch.isfinite(out).all() and rel_l2(out.cpu(), want) < 2e-3


def test_sca_gather_head_major_value_equals_reference_layout():
    """VER_SCA_VALUE_HEAD_MAJOR (include/ver_ops.h): the same bf16 values laid out [heads, B, Ncam, Nk, hd] -- a
    (camera, head) tile is one contiguous block, staged row-major into LDS -- give the SAME slots bit for bit (the
    arithmetic per sample is unchanged), the same d(offsets) / d(logits), and d(value) comes back as the permuted view of
    a reference-layout buffer with the same numbers.  ``head_major_linear`` (value_proj as one batched GEMM over the
    heads) equals the plain Linear, forward and gradients."""
    hip = pkg('hipops')
    heads, hd, P = 8, 96, 8
    assert hip.sca_head_major_supported(torch.bfloat16, hd, P, 14, 14)
    assert not hip.sca_head_major_supported(torch.float32, hd, P, 14, 14)
    assert not hip.sca_head_major_supported(torch.bfloat16, hd, 4, 14, 14)
    hit, value, offsets, logits, gslots = _random_sca_case(31, 3, (4, 15, 15), heads, hd, P)
    gs = T(gslots).to(DEV)
    res = {}
    for hm in (False, True):
        vb = T(value).to(DEV).to(torch.bfloat16)
        if hm:
            vb = vb.permute(3, 0, 1, 2, 4).contiguous()
        vb.requires_grad_(True)
        of = T(offsets).to(DEV).requires_grad_(True)
        lg = T(logits).to(DEV).requires_grad_(True)
        slots = hip.sca_gather(vb, of, lg, hit, 14, 14, None, hm)
        slots.backward(gs)
        gv = vb.grad
        if hm:
            assert gv.shape == vb.shape
            gv = gv.permute(1, 2, 3, 0, 4)
        res[hm] = (slots.detach().cpu(), of.grad.cpu(), lg.grad.cpu(), gv.float().cpu())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    # the projection that produces the layout
    gen = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(3 * 6 * 196, 768, generator=gen).to(DEV).bfloat16().requires_grad_(True)
    lin = torch.nn.Linear(768, 768).to(DEV)
    g = torch.randn(heads, x.shape[0], hd, generator=gen).to(DEV).bfloat16()
    out = hip.head_major_linear(x, lin.weight, lin.bias, heads)
    out.backward(g)
    got = (out.detach().float().cpu(), x.grad.float().cpu(), lin.weight.grad.float().cpu(), lin.bias.grad.float().cpu())
    x2 = x.detach().clone().requires_grad_(True)
    lin.weight.grad = lin.bias.grad = None
    ref = torch.nn.functional.linear(x2, lin.weight.bfloat16(), lin.bias.bfloat16()).view(-1, heads, hd).permute(1, 0, 2)
    ref.backward(g)
    from util import rel_l2
    assert rel_l2(got[0], ref.detach().float().cpu()) < 2e-3
    assert rel_l2(got[1], x2.grad.float().cpu()) < 5e-3
    assert rel_l2(got[2], lin.weight.grad.float().cpu()) < 5e-3
    assert rel_l2(got[3], lin.bias.grad.float().cpu()) < 5e-3


def test_sca_backward_grad_value_dtype_contract():
    """C ABI: ver_sca_backward_grad_dtype names the cheapest d(value) dtype (bf16 on the matrix-core path), and
    VER_F32 is accepted for the same problem: both buffers hold the same gradient up to bf16 rounding, and the
    d(offsets) / d(logits) of the two calls are bitwise equal."""
    hip = pkg('hipops')
    lib = hip.lib()
    heads, hd, P = 8, 96, 8
    assert lib.ver_sca_backward_grad_dtype(1, hd, P, 14, 14) == 1          # bf16 tiles, 8 points: matrix cores
    assert lib.ver_sca_backward_grad_dtype(0, hd, P, 14, 14) == 0          # fp32 tiles
    assert lib.ver_sca_backward_grad_dtype(1, hd, 4, 14, 14) == 0          # 4 points
    assert lib.ver_sca_backward_grad_dtype(1, 8, P, 14, 14) == 0           # head_dim 8
    hit, value, offsets, logits, gslots = _random_sca_case(23, 2, (4, 15, 15), heads, hd, P)
    vb = T(value).to(DEV).to(torch.bfloat16)
    of, lg, gs = T(offsets).to(DEV), T(logits).to(DEV), T(gslots).to(DEV)
    B, ncam = vb.shape[0], vb.shape[1]
    p = hip._p
    res = {}
    for gdt, dt in ((1, torch.bfloat16), (0, torch.float32)):
        gv = torch.full(vb.shape, float('nan'), dtype=dt, device=DEV)      # every element must be written
        go, gl = torch.empty_like(of), torch.empty_like(lg)
        rc = lib.ver_sca_backward(p(vb), 1, p(of), p(lg), p(hit.uv), p(hit.vis), p(hit.vis_list), p(hit.vis_cnt),
                                  p(hit.fwd_list), p(hit.fwd_cnt), p(gs), p(gv), gdt, p(go), p(gl), B, ncam, hit.Nq,
                                  hit.D, heads, hd, P, 14, 14, 0, hip._stream())
        assert rc == 0, lib.ver_last_error()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(gv.float()).all())
        res[gdt] = (gv.float().cpu(), go.cpu(), gl.cpu())
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert close(res[1][0], res[0][0], atol=2e-2, rtol=1e-2)               # bf16 rounding of the fp32 result
    assert torch.equal(res[0][0].bfloat16(), res[1][0].bfloat16())          # ... and exactly that rounding


def test_forward_work_lists_partition_the_visible_voxels():
    """fwd_list / fwd_cnt (the forward kernel's work order): per camera the voxels only it sees from the front, the
    voxels it shares with other cameras from the back; together exactly vis_list."""
    hip = pkg('hipops')
    rng = np.random.default_rng(3)
    B, nq = 2, 77
    mask = rng.uniform(size=(6, B, nq, 1)) < 0.3
    mask[:, :, :3] = True                                             # voxels seen by every camera but the blind one
    mask[4] = False                                                   # a camera that sees nothing
    hit = hip.hits_from_mask(torch.zeros(6, B, nq, 1, 2, device=DEV), T(mask).to(DEV))
    vis = hit.vis.cpu().numpy()
    for b in range(B):
        for c in range(6):
            n_single, n_multi = hit.fwd_cnt[b, c].cpu().tolist()
            assert n_single + n_multi == int(hit.vis_cnt[b, c])
            lst = hit.fwd_list[b, c].cpu().numpy()
            single, multi = lst[:n_single], lst[nq - n_multi:][::-1]
            seen = np.nonzero((vis[b] >> c) & 1)[0]
            pop = np.array([bin(int(v)).count('1') for v in vis[b]])
            assert np.array_equal(single, seen[pop[seen] == 1])       # ascending
            assert np.array_equal(multi, seen[pop[seen] > 1])
    assert hit.fwd_cnt[:, 4].abs().sum().item() == 0


def test_sca_gather_samples_outside_the_map():
    """The forward compacts the samples whose footprint misses the map: the reference's initial ring of offsets (40 %
    outside), offsets that throw EVERY sample out (visible voxels must come out as exact zeros), and a single live
    sample per voxel."""
    hip = pkg('hipops')
    o = oracle()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sca_modes_helper as H
    _, hit, value, offsets, logits = H.case(6, 2, (4, 15, 15), 8, 96, True)
    mask = hit.mask()[:, :, :, 0].permute(1, 0, 2).cpu()
    for vt in (value, value.bfloat16()):
        tol = 2e-5 if vt.dtype == torch.float32 else 1e-2         # bf16 tiles: packed fp16 accumulation (bf16 bound)
        got = hip.sca_gather(vt, offsets, logits, hit, 14, 14)
        ref = oracle_slots(o, vt.float().cpu(), offsets.cpu(), logits.cpu(), hit.uv.cpu(), mask, (14, 14))
        assert maxdiff(got.cpu(), ref) < tol
        far = hip.sca_gather(vt, offsets + 40.0, logits, hit, 14, 14)
        assert float(far.abs().max()) == 0.0
        one = offsets + 40.0
        one[:, :, :, 3] = offsets[:, :, :, 3] * 0.25                   # only point 3 can land inside
        got1 = hip.sca_gather(vt, one, logits, hit, 14, 14)
        ref1 = oracle_slots(o, vt.float().cpu(), one.cpu(), logits.cpu(), hit.uv.cpu(), mask, (14, 14))
        assert maxdiff(got1.cpu(), ref1) < tol


def test_sca_gather_launch_modes(tmp_path):
    """The other launch shapes of ver_sca_forward -- persistent workgroups with loader waves, several units per
    workgroup, the generic 16-lane kernel -- give the default shape's result (the environment is read once per
    process: each mode runs in its own interpreter)."""
    import subprocess
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sca_modes_helper.py')
    modes = {
        'default': {'VER_SCA_FWD_MATH': '0'},         # bf16 tile unpacked to fp32 per use: the exact reference mode
        'f16_acc': {},                                # the shipped default: packed fp16 accumulation on bf16 tiles
        'odd_threads': {'VER_SCA_FWD_MATH': '0', 'VER_SCA_CS_THREADS_BF16': '320', 'VER_SCA_CS_THREADS_F32': '384'},
        'generic': {'VER_SCA_FWD_CS': '0'},
        'loaders': {'VER_SCA_FWD_MATH': '0', 'VER_SCA_CS_NLOAD': '1', 'VER_SCA_CS_THREADS_BF16': '1024', 'VER_SCA_CS_THREADS_F32': '1024',
                    'VER_SCA_CS_HSPLIT': '2', 'VER_SCA_CS_UNITS_PER_WG': '0'},
        'loaders2': {'VER_SCA_FWD_MATH': '2', 'VER_SCA_CS_NLOAD': '2', 'VER_SCA_CS_THREADS_BF16': '512', 'VER_SCA_CS_THREADS_F32': '1024',
                     'VER_SCA_CS_HSPLIT': '1', 'VER_SCA_CS_UNITS_PER_WG': '3'},
        'multi_unit': {'VER_SCA_FWD_MATH': '0', 'VER_SCA_CS_HSPLIT': '4', 'VER_SCA_CS_UNITS_PER_WG': '5', 'VER_SCA_CS_THREADS_BF16': '512'},
        # the shipped kernel is the compile-time specialisation of the default launch shape: this is its general form
        'general_form': {'VER_SCA_CS_ONE': '0'},
    }
    res = {}
    for name, env in modes.items():
        out = str(tmp_path / (name + '.npz'))
        subprocess.run([sys.executable, helper, out], check=True, env=dict(os.environ, **env), timeout=600)
        res[name] = np.load(out)
    for name in modes:
        for key in res['default'].files:
            d = float(np.abs(res[name][key] - res['default'][key]).max())
            if name == 'general_form':
                # same arithmetic as the specialised kernel: equal up to the order of the float atomics on shared voxels
                dd = float(np.abs(res[name][key] - res['f16_acc'][key]).max())
                assert dd < 2e-5, (name, key, dd)
            elif name in ('f16_acc', 'loaders2') and key.endswith('_bf16'):
                # packed fp16 accumulation: inside the bf16 bound of the north star, and only on bf16 tiles
                rel = float(np.linalg.norm(res[name][key] - res['default'][key]) / np.linalg.norm(res['default'][key]))
                assert d < 1e-2 and rel < 2e-3, (name, key, d, rel)
            else:
                assert d < 2e-5, (name, key, d)


def test_sca_gather_multi_camera_and_anchors():
    """Hand-made masks: voxels seen by all 6 cameras, by none, a camera that sees nothing, and
    D=2 Z-anchors (which the reference's MSDA3D supports although vocc produces D=1)."""
    hip = pkg('hipops')
    o = oracle()
    rng = np.random.default_rng(9)
    B, nq, heads, hd, P, D = 2, 50, 4, 8, 8, 2
    mask = rng.uniform(size=(6, B, nq, D)) < 0.25
    mask[:, :, :4] = False
    mask[:, :, 4:8] = True
    mask[2] = False
    uv = rng.uniform(-0.1, 1.1, (6, B, nq, D, 2)).astype(np.float32)
    hit = hip.hits_from_mask(T(uv).to(DEV), T(mask).to(DEV))
    value = rng.standard_normal((B, 6, 49, heads, hd)).astype(np.float32)
    offsets = (rng.standard_normal((B, nq, heads, P, 2)) * 2.0).astype(np.float32)
    logits = rng.standard_normal((B, nq, heads, P)).astype(np.float32)
    gs = rng.standard_normal((B, nq, heads * hd)).astype(np.float32)
    v = T(value).to(DEV).requires_grad_(True)
    of = T(offsets).to(DEV).requires_grad_(True)
    lg = T(logits).to(DEV).requires_grad_(True)
    slots = hip.sca_gather(v, of, lg, hit, 7, 7)
    slots.backward(T(gs).to(DEV))
    vc, oc, lc = (T(value).requires_grad_(True), T(offsets).requires_grad_(True),
                  T(logits).requires_grad_(True))
    m = T(mask).any(-1).permute(1, 0, 2)
    ref = oracle_slots(o, vc, oc, lc, T(uv).permute(1, 0, 2, 3, 4), m, (7, 7))
    ref.backward(T(gs))
    assert maxdiff(slots.detach().cpu(), ref.detach()) < 2e-5
    assert close(v.grad.cpu(), vc.grad)
    assert close(of.grad.cpu(), oc.grad)
    assert close(lg.grad.cpu(), lc.grad)
    assert float(v.grad[:, 2].abs().max()) == 0.0          # blind camera gets a zero gradient tile


def test_sca_gather_full_size_properties():
    """50x50x16 grid (Nq = 40 000, chunked lists, atomic tile flush in backward):
    linearity in value, and slots of a constant map = constant * (in-map weight mass)."""
    hip = pkg('hipops')
    hit, value, offsets, logits, gslots = _random_sca_case(11, 1, (16, 50, 50), 8, 96, 8)
    v = T(value).to(DEV)
    of = T(offsets).to(DEV)
    lg = T(logits).to(DEV)
    a = hip.sca_gather(v, of, lg, hit, 14, 14)
    b = hip.sca_gather(v * -0.5, of, lg, hit, 14, 14)
    assert float((b + 0.5 * a).abs().max()) < 1e-5
    ones = hip.sca_gather(torch.ones_like(v), of, lg, hit, 14, 14)
    assert float(ones.max()) <= 1.0 + 1e-5 and float(ones.min()) >= -1e-6
    seen = (hit.vis[0] != 0)
    assert int(seen.sum()) == 40000 - int((hit.vis[0] == 0).sum())
    assert float(a[0][~seen].abs().max()) == 0.0
    # backward at this size: d(sum slots * g)/d value is linear in g; compare two g's
    vg = v.clone().requires_grad_(True)
    g1 = T(gslots).to(DEV)
    s = hip.sca_gather(vg, of, lg, hit, 14, 14)
    (gv1,) = torch.autograd.grad(s, vg, g1, retain_graph=True)
    (gv2,) = torch.autograd.grad(s, vg, 2.0 * g1)
    assert close(gv2.cpu(), 2.0 * gv1.cpu(), atol=2e-4, rtol=1e-4)
    # <g, J v> == <J^T g, v>  (adjoint identity, checks backward against forward at full size)
    lhs = float((s.detach().double() * g1.double()).sum())
    rhs = float((gv1.double() * v.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_errors_are_loud():
    hip = pkg('hipops')
    hit, value, offsets, logits, _ = _random_sca_case(1, 1, (2, 6, 5), 2, 8, 8)
    bad = torch.zeros(1, 6, 196, 2, 24, device=DEV)          # head_dim 24 is not built
    with pytest.raises(RuntimeError, match='head_dim 24'):
        hip.sca_gather(bad, T(offsets).to(DEV), T(logits).to(DEV), hit, 14, 14)
    with pytest.raises(RuntimeError, match='GPU'):
        hip.sca_gather(T(value), T(offsets), T(logits), hit, 14, 14)


# ------------------------------------------------------------------------------- a10: lattice im2col
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lattice_im2col_and_adjoint(dtype):
    """HIP im2col / col2im vs the torch slice-and-cat form (bit-exact: pure data movement;
    col2im sums <= 27 bf16 terms in fp32)."""
    hip = pkg('hipops')
    up = pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(5)
    e = torch.randn(2, 4, 5, 7, 64, generator=gen).to(dtype)
    taps = [(2 * a - 2, b - 1, c - 1) for a in range(3) for b in range(3) for c in range(2)]
    want = up._im2col(e, taps)                      # CPU reference form
    ed = e.to(DEV).requires_grad_(True)
    got = hip.lattice_im2col(ed, taps)
    assert got.dtype == dtype and torch.equal(got.cpu(), want)
    g = torch.randn(want.shape, generator=gen).to(dtype)
    got.backward(g.to(DEV))
    ec = e.float().requires_grad_(True)
    up._im2col(ec, taps).backward(g.float())
    if dtype == torch.float32:
        assert float((ed.grad.cpu() - ec.grad).abs().max()) <= 1e-5
    else:                                        # result rounded to bf16 (8 mantissa bits)
        assert close(ed.grad.float().cpu(), ec.grad, atol=1e-2, rtol=1e-2)
    assert ed.grad.dtype == dtype


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('planar', [False, True])
def test_lattice_gather_scatter_with_offsets(dtype, planar):
    """ver_lattice_gather / _scatter (27 taps at per-tap column offsets, plain or planar source) vs
    the slice-based torch form used on CPU; columns between the tap blocks stay untouched."""
    ups = pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(21)
    b, z, hc, wc, c = 2, 3, 6, 4, 16
    kt = 27 * c + 4 * ups._PW
    src = torch.randn(b, z, hc, wc, c, generator=gen).to(dtype)
    e_cpu = ups.plain_to_planar(src).contiguous() if planar else src
    a_cpu = torch.full((b * z * hc * wc, kt), 7.0, dtype=dtype)
    ups._gather27(e_cpu, planar, a_cpu, c, hc, wc)
    a_gpu = torch.full((b * z * hc * wc, kt), 7.0, dtype=dtype, device=DEV)
    ups._gather27(e_cpu.to(DEV), planar, a_gpu, c, hc, wc)
    assert torch.equal(a_gpu.cpu(), a_cpu)                                  # pure data movement: bit-exact
    assert float((a_cpu == 7.0).sum()) >= b * z * hc * wc * 4 * ups._PW     # constant blocks untouched
    d_a = torch.randn(b * z * hc * wc, kt, generator=gen).to(dtype)
    want = ups._scatter27(d_a.double(), planar, tuple(e_cpu.shape), c, hc, wc)
    got = ups._scatter27(d_a.to(DEV), planar, tuple(e_cpu.shape), c, hc, wc)
    assert got.shape == e_cpu.shape and got.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert close(got.float().cpu(), want, atol=tol, rtol=tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', [0, 2, 3])
def test_lattice_gather_scatter_z_split(dtype, layout):
    """The Z = 4 form: rows (b, z & 1, y, x), two z taps (dz in {0, 2}); source plain (first layer),
    z-split or planar z-split (what the layers' GEMMs leave behind)."""
    ups = pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(23 + layout)
    b, hc, wc, c = 2, 6, 4, 16
    plain = torch.randn(b, 4, hc, wc, c, generator=gen).to(dtype)
    e_cpu = ups._from_plain(plain, layout)
    if layout == 0:
        taps, offs, _, _ = ups._layer0_z4_plan(c, 'cpu')
        kt = 50 * c
    else:
        _, kt, _, taps, offs = ups._layer_plan_z4(c, 'cpu')
    m = b * 2 * hc * wc
    a_cpu = torch.full((m, kt), 7.0, dtype=dtype)
    ups._gather_z4(e_cpu, layout, a_cpu, taps, offs, c, hc, wc)
    a_gpu = torch.full((m, kt), 7.0, dtype=dtype, device=DEV)
    ups._gather_z4(e_cpu.to(DEV), layout, a_gpu, taps, offs, c, hc, wc)
    assert torch.equal(a_gpu.cpu(), a_cpu)
    d_a = torch.randn(m, kt, generator=gen).to(dtype)
    want = ups._scatter_z4(d_a.double(), layout, tuple(e_cpu.shape), taps, offs, c, hc, wc)
    got = ups._scatter_z4(d_a.to(DEV), layout, tuple(e_cpu.shape), taps, offs, c, hc, wc)
    assert got.shape == e_cpu.shape and got.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert close(got.float().cpu(), want, atol=tol, rtol=tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_convt_weight_taps_and_lattice_transpose(dtype):
    """Layout kernels: ConvTranspose3d weight -> flipped correlation taps (and its adjoint), lattice
    channels-last (plain / planar) <-> channel-first rows.  Pure data movement: bit-exact."""
    hip, ups = pkg('hipops'), pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(22)
    w = torch.randn(24, 40, 3, 5, 5, generator=gen)
    want = w.to(dtype).flip(2, 3, 4).permute(2, 3, 4, 0, 1).reshape(75, 24, 40)
    wd = w.to(DEV).requires_grad_(True)
    got = hip.convt_weight_taps(wd, dtype)
    assert torch.equal(got.detach().cpu(), want)
    g = torch.randn(75, 24, 40, generator=gen).to(dtype)
    got.backward(g.to(DEV))
    want_g = g.float().view(3, 5, 5, 24, 40).permute(3, 4, 0, 1, 2).flip(2, 3, 4)
    assert torch.equal(wd.grad.cpu(), want_g)
    # C not a multiple of the 128-channel tile; (W = 10, odd row stride): element-wise kernel,
    # (W = 12, stride a multiple of 8): the 16-byte / 8-byte vector kernel
    for (b, z, h, wl, c, tail) in ((2, 4, 6, 10, 200, 5), (2, 4, 6, 12, 200, 8)):
        plain = torch.randn(b, z, h, wl, c, generator=gen).to(dtype)
        L = c * z * h * wl
        for layout in (0, 1, 2, 3):                        # plain, planar, z-split, planar z-split
            src = ups._from_plain(plain, layout)
            cf = torch.full((b, L + tail), 3.0, dtype=dtype, device=DEV)
            hip.lattice_transpose(src.to(DEV), cf, (h, wl), layout, True)
            assert torch.equal(cf[:, :L].cpu().view(b, c, z, h, wl), plain.permute(0, 4, 1, 2, 3))
            assert bool((cf[:, L:] == 3.0).all())
            back = torch.empty_like(src, device=DEV)
            hip.lattice_transpose(back, cf, (h, wl), layout, False)
            assert torch.equal(back.cpu(), src)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('kind,ci,co', [('l0', 24, 40), ('lat', 24, 40), ('lat', 5, 41), ('lat', 3, 130)])
def test_convt_weight_gradient_from_class_blocks(kind, ci, co, dtype):
    """ver_convt_weight_backward_blocks against the sequence of torch ops it replaces in the z-split layers' backward
    (dense_heads/upsample.py): per class two indexed copies of the half gradients into zero-filled [75 Ci, Co] buffers,
    their sum, + prev_bias (x) d(v), the adjoint of the tap flip.  fp32 sums of at most three terms: exact up to the
    rounding of the bf16 intermediates the torch sequence keeps (compared in fp64 with those intermediates left out)."""
    hip, up = pkg('hipops'), pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(31)
    # (Co = 40, 130: the two-pairs-per-lane bf16 kernel with ragged last workgroups; Co = 41: the element-wise one)
    if kind == 'l0':
        _, _, lo, hi = up._layer0_z4_plan(ci, 'cpu')
        rows = 50 * ci
        pieces = [(0, rows, lo, hi)]
        pb = dv = None
    else:
        plan, kt, total_rows, _, _ = up._layer_plan_z4(ci, 'cpu')
        class_rows = up._class_rows_z4(ci)
        pieces = [(class_rows[cls][0], plan[cls][1] - plan[cls][0], plan[cls][2], plan[cls][3]) for cls in up._CLASSES]
        rows = sum(p[1] for p in pieces)
        pb = torch.randn(ci, generator=gen).to(dtype)
        dv = torch.randn(75, co, generator=gen).to(dtype)
    stacked = torch.randn(rows, 2 * co, generator=gen).to(dtype)
    total = 75 * ci + 8 * up._PW2                    # data rows + every constant / dummy row the plans may name
    d_lo = torch.zeros(total, co, dtype=torch.float64)
    d_hi = torch.zeros(total, co, dtype=torch.float64)
    for r0, n, lo, hi in pieces:
        d_lo.index_copy_(0, lo, stacked[r0:r0 + n, :co].double())
        d_hi.index_copy_(0, hi, stacked[r0:r0 + n, co:].double())
    d_k = (d_lo + d_hi)[:75 * ci].view(75, ci, co)
    if pb is not None:
        d_k = d_k + pb.double()[None, :, None] * dv.double()[:, None, :]
    want = d_k.view(3, 5, 5, ci, co).permute(3, 4, 0, 1, 2).flip(2, 3, 4)
    got = hip.convt_weight_backward_blocks(stacked.to(DEV), up._block_offsets(kind, ci, co, DEV),
                                           None if pb is None else pb.to(DEV), None if dv is None else dv.to(DEV), ci, co)
    assert got.dtype == torch.float32 and tuple(got.shape) == (ci, co, 3, 5, 5)
    assert close(got.cpu(), want, atol=1e-5, rtol=1e-6)
    if kind == 'lat':                                # the rows of the classes' own constant blocks, as the backward reads them
        aug = stacked.view(-1, co).index_select(0, up._aug_rows_z4(ci, 'cpu')).view(8, up._PW, co).double().sum(0)
        n_data = 75 * ci
        want_aug = (d_lo + d_hi)[n_data:n_data + 4 * up._PW].view(4, up._PW, co).sum(0)
        assert close(aug, want_aug, atol=1e-9, rtol=1e-9)


@pytest.mark.parametrize('layout', [0, 1, 2, 3])
def test_lattice_rows_equals_transpose_plus_run_copies(layout):
    """ver_lattice_rows (bf16 lattice <-> the operand rows of all occ_proj pattern groups in one pass) against the two
    passes it replaces, ver_lattice_transpose + ver_run_gather / ver_run_scatter, on the vocc.py geometry (768 channels,
    4 x 60 x 60 lattice, 5 groups, period 960): pure data movement, bit-exact both ways, rows of other samples and the
    augmentation columns untouched."""
    hip, ups, opl = pkg('hipops'), pkg('dense_heads.upsample'), pkg('dense_heads.occ_proj_lattice')
    C, Z, Hl, Wl, bs = 768, 4, 60, 60, 2
    plan = opl.get_plan(C, Z, 2 * Hl, 2 * Wl, torch.device(DEV))
    assert plan is not None and plan.row_map is not None and plan.row_map['period'] == 960
    gen = torch.Generator(device='cpu').manual_seed(41)
    plain = torch.randn(bs, Z, Hl, Wl, C, generator=gen).bfloat16()
    src = ups._from_plain(plain, layout).contiguous().to(DEV)
    L = C * Z * Hl * Wl
    row_map, spans, total = opl._row_map_for(plan, bs)
    # forward: lattice -> rows
    lat = torch.empty(bs, (L + C + 2 + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
    hip.lattice_transpose(src, lat, (Hl, Wl), layout, True)
    lat[:, L:] = 0
    buf = torch.full((total,), 7.0, dtype=torch.bfloat16, device=DEV)
    hip.lattice_rows(src, buf, row_map, (Hl, Wl), layout, True)
    for g, (b0, n) in zip(plan.groups, spans):
        want = torch.empty(bs * g.n_rows, g.k_aug, dtype=torch.bfloat16, device=DEV)
        hip.run_gather(lat, g.run_start, g.aug_idx, want, g.n_rows, g.run_len)
        got = buf[b0:b0 + n].view(bs * g.n_rows, g.k_aug)
        assert torch.equal(got[:, :g.n_cols], want[:, :g.n_cols])
        assert bool((got[:, g.n_cols:] == 7.0).all())                     # augmentation columns are not this kernel's
    # backward: rows -> lattice
    rows = torch.randn(total, generator=gen).bfloat16().to(DEV)
    d_lat = torch.empty(bs, L, dtype=torch.bfloat16, device=DEV)
    for g, (b0, n) in zip(plan.groups, spans):
        hip.run_scatter(rows[b0:b0 + n].view(bs * g.n_rows, g.k_aug), d_lat, g.run_start, g.n_rows, g.run_len)
    want_e = torch.empty_like(src)
    hip.lattice_transpose(want_e, d_lat, (Hl, Wl), layout, False)
    got_e = torch.empty_like(src)
    hip.lattice_rows(got_e, rows, row_map, (Hl, Wl), layout, False)
    assert torch.equal(got_e, want_e)


def test_occ_proj_lattice_fused_rows_path_is_bit_identical(monkeypatch):
    """``occ_proj_from_lattice`` in bf16 with the one-pass lattice <-> rows kernel (default) and with the channel-first
    copy + run copies (VER_LATTICE_ROWS=0): the same operands reach the same GEMMs -- outputs and every gradient equal."""
    opl = pkg('dense_heads.occ_proj_lattice')
    gen = torch.Generator(device='cpu').manual_seed(43)
    C, Z, hh, wh = 768, 4, 30, 30
    e0 = (torch.randn(4, 1, 2, hh, wh, 2, C, generator=gen) * 0.5).bfloat16().to(DEV)
    up_bias = torch.randn(C, generator=gen).to(DEV)
    weight = (torch.randn(4480, Z * C, generator=gen) * 0.02).to(DEV)
    bias = torch.randn(4480, generator=gen).to(DEV)
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('VER_LATTICE_ROWS', mode)
        e, ub, w, b = (t.clone().requires_grad_(True) for t in (e0, up_bias, weight, bias))
        rows, plan = opl.occ_proj_from_lattice(e, ub, w, b)
        g = torch.randn(rows.shape, generator=torch.Generator(device='cpu').manual_seed(5)).bfloat16().to(DEV)
        rows.backward(g)
        res[mode] = (rows.detach(), e.grad, ub.grad, w.grad, b.grad)
    assert plan.row_map is not None
    for a, b in zip(res['0'], res['1']):
        assert torch.equal(a, b)


def test_occ_proj_lattice_input_gradient_on_ver_gemm_nn_equals_the_library(monkeypatch):
    """d(input) of ``occ_proj``'s 832-column groups runs on ``ver_gemm_nn`` from 14 000 rows per group on (round 6:
    dense_heads/occ_proj_lattice.py, ``_OWN_DGRAD``); five viewpoints = 14 400 rows per group.  Same operands, fp32 accumulation
    either way: the lattice gradient agrees with the library's product to bf16 rounding (rel. L2 < 2e-3, most elements identical),
    everything that does not pass through that product is bit-identical, and the kernel did run for three of the five groups."""
    opl, hip = pkg('dense_heads.occ_proj_lattice'), pkg('hipops')
    from util import rel_l2
    gen = torch.Generator(device='cpu').manual_seed(44)
    C, Z, hh, wh, bs = 768, 4, 30, 30, 5
    e0 = (torch.randn(4, bs, 2, hh, wh, 2, C, generator=gen) * 0.5).bfloat16().to(DEV)
    up_bias = torch.randn(C, generator=gen).to(DEV)
    weight = (torch.randn(4480, Z * C, generator=gen) * 0.02).to(DEV)
    bias = torch.randn(4480, generator=gen).to(DEV)
    calls = []
    real = hip.gemm_nn
    monkeypatch.setattr(hip, 'gemm_nn', lambda *a, **k: (calls.append((tuple(a[0].shape), tuple(a[1].shape), k.get('timer_class'))), real(*a, **k))[1])
    res = {}
    for own in (False, True):
        monkeypatch.setattr(opl, '_OWN_DGRAD', own)
        e, ub, w, b = (t.clone().requires_grad_(True) for t in (e0, up_bias, weight, bias))
        rows, plan = opl.occ_proj_from_lattice(e, ub, w, b)
        g = torch.randn(rows.shape, generator=torch.Generator(device='cpu').manual_seed(5)).bfloat16().to(DEV)
        n_before = len(calls)
        rows.backward(g)
        res[own] = (rows.detach(), e.grad, ub.grad, w.grad, b.grad, len(calls) - n_before)
    assert res[False][5] == 0 and res[True][5] == 3, (res[False][5], res[True][5])
    assert all(c == ((bs * 2880, 4480), (4480, 832), 'head_gemm_dgrad') for c in calls), calls
    assert torch.equal(res[False][0], res[True][0])                       # forward untouched
    assert torch.equal(res[False][3], res[True][3]) and torch.equal(res[False][4], res[True][4])      # d(weight), d(bias)
    r = rel_l2(res[True][1].float(), res[False][1].float())
    same = float((res[True][1] == res[False][1]).float().mean())
    assert r < 2e-3 and same > 0.9, (r, same)
    assert rel_l2(res[True][2], res[False][2]) < 1e-3                      # d(up_bias): column sums of the same product


def test_upsample_on_gpu_matches_conv_transpose():
    up = pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(6)
    x = torch.randn(2, 16, 4, 5, 6, generator=gen).to(DEV).requires_grad_(True)
    ws = [(torch.randn(16, 16, 3, 5, 5, generator=gen) * 0.05).to(DEV).requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(16, generator=gen).to(DEV).requires_grad_(True) for _ in range(3)]
    y = up.upsample_dense(x, ws, bs)
    xc = x.detach().cpu().double().requires_grad_(True)
    wc = [w.detach().cpu().double().requires_grad_(True) for w in ws]
    bc = [b.detach().cpu().double().requires_grad_(True) for b in bs]
    r = xc
    for w, b in zip(wc, bc):
        r = torch.nn.functional.conv_transpose3d(r, w, b, **up.GEOM)
    assert close(y.detach().cpu(), r.detach(), atol=1e-4, rtol=1e-4)
    g = torch.randn(r.shape, generator=gen)
    y.backward(g.to(DEV))
    r.backward(g.double())
    assert close(x.grad.cpu(), xc.grad, atol=1e-3, rtol=1e-3)
    for a, b in zip(ws + bs, wc + bc):
        assert close(a.grad.cpu(), b.grad, atol=1e-3, rtol=1e-3)


# ------------------------------------------------------------------------------- a10: occupancy MLP
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layer_norm_relu_fused(dtype):
    """ver_ln_relu_* vs relu(F.layer_norm(x)) in fp64 (head:241-248)."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(8)
    n = 50021                                       # not a multiple of the 16 rows per block
    x = (torch.randn(n, 128, generator=gen) * 2 + 0.3).to(dtype)
    gamma = torch.randn(128, generator=gen) * 0.5 + 1.0
    beta = torch.randn(128, generator=gen) * 0.2
    gy = torch.randn(n, 128, generator=gen).to(dtype)
    xd = x.to(DEV).requires_grad_(True)
    gd = gamma.to(DEV).requires_grad_(True)
    bd = beta.to(DEV).requires_grad_(True)
    y = hip.layer_norm_relu(xd, gd, bd, 1e-5)
    assert y.dtype == dtype and y.shape == x.shape
    y.backward(gy.to(DEV))
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    yr = torch.relu(torch.nn.functional.layer_norm(xr, (128,), gr, br, 1e-5))
    yr.backward(gy.double())
    if dtype == torch.float32:
        assert close(y.detach().cpu(), yr.detach(), atol=1e-5, rtol=1e-5)
        assert close(xd.grad.cpu(), xr.grad, atol=1e-4, rtol=1e-4)
    else:
        assert close(y.detach().float().cpu(), yr.detach(), atol=2e-2, rtol=1e-2)
        assert close(xd.grad.float().cpu(), xr.grad, atol=3e-2, rtol=2e-2)
    assert close(gd.grad.cpu(), gr.grad, atol=2e-2 * (1 if dtype == torch.float32 else 20), rtol=2e-3)
    assert close(bd.grad.cpu(), br.grad, atol=2e-2 * (1 if dtype == torch.float32 else 20), rtol=2e-3)
    assert hip.layer_norm_relu(torch.zeros(0, 128, device=DEV), gd, bd).shape == (0, 128)


def test_row_linear_split_k_gradients():
    """row_linear (batched split-K weight/bias gradient) vs F.linear in fp64."""
    rl = pkg('dense_heads.row_linear')
    gen = torch.Generator(device='cpu').manual_seed(9)
    n = 3 * rl._CHUNK + 517                        # three full chunks + a tail
    x = torch.randn(n, 128, generator=gen)
    w = torch.randn(16, 128, generator=gen) * 0.1
    b = torch.randn(16, generator=gen)
    gy = torch.randn(n, 16, generator=gen)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = rl.row_linear(xd, wd, bd)
    y.backward(gy.to(DEV))
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(gy.double())
    assert close(y.detach().cpu(), yr.detach(), atol=1e-4, rtol=1e-4)
    assert close(xd.grad.cpu(), xr.grad, atol=1e-4, rtol=1e-4)
    assert close(wd.grad.cpu(), wr.grad, atol=2e-3, rtol=1e-4)
    assert close(bd.grad.cpu(), br.grad, atol=2e-3, rtol=1e-4)
    # bf16 rows (autocast): gradients within bf16 rounding of the fp64 ones
    xd2, wd2, bd2 = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y2 = rl.row_linear(xd2, wd2, bd2)
    assert y2.dtype == torch.bfloat16
    y2.backward(gy.to(DEV).bfloat16())
    assert pkg('hipops') is not None and torch.isfinite(wd2.grad).all()
    rel = (wd2.grad.cpu().double() - wr.grad).norm() / wr.grad.norm()
    assert rel < 1e-2, rel
    assert (bd2.grad.cpu().double() - br.grad).norm() / br.grad.norm() < 1e-2


def _occ_mlp_reference(x, p, round_hidden=True):
    """occ_branches in fp64 on bf16-rounded operands: the oracle's restatement (head:241-248) when the
    hidden activations stay exact; with ``round_hidden`` they are rounded to bf16 between the layers as
    the fused kernel (and the layer-by-layer bf16 autocast path) hands them on."""
    F = torch.nn.functional
    if not round_hidden:
        keys = dict(zip(('0.weight', '0.bias', '1.weight', '1.bias', '3.weight', '3.bias', '4.weight', '4.bias',
                         '6.weight', '6.bias'), ('w1', 'b1', 'g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')))
        return oracle().occ_branches({k: p[v] for k, v in keys.items()}, '', x)
    r = lambda t: t.bfloat16().double()
    h = r(torch.relu(F.layer_norm(F.linear(x, p['w1'], p['b1']), (128,), p['g1'], p['be1'], 1e-5)))
    h = r(torch.relu(F.layer_norm(F.linear(h, p['w2'], p['b2']), (128,), p['g2'], p['be2'], 1e-5)))
    return F.linear(h, p['w3'], p['b3'])


def _occ_mlp_params(gen):
    p = dict(w1=torch.randn(128, 128, generator=gen) * 0.12, b1=torch.randn(128, generator=gen) * 0.3,
             g1=torch.randn(128, generator=gen) * 0.3 + 1.0, be1=torch.randn(128, generator=gen) * 0.3,
             w2=torch.randn(128, 128, generator=gen) * 0.12, b2=torch.randn(128, generator=gen) * 0.3,
             g2=torch.randn(128, generator=gen) * 0.3 + 1.0, be2=torch.randn(128, generator=gen) * 0.3,
             w3=torch.randn(16, 128, generator=gen) * 0.12, b3=torch.randn(16, generator=gen) * 0.3)
    return p


def test_occ_mlp_fused_forward():
    """ver_occ_mlp_forward (MFMA chain in registers) vs occ_branches evaluated in fp64 (head:241-248)."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(11)
    p = _occ_mlp_params(gen)
    n = 64 * 37 + 21                                 # ragged: not a multiple of the 64-row wave block
    x = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
    image = hip.occ_mlp_pack(p['w1'].to(DEV), p['w2'].to(DEV), p['w3'].to(DEV))
    vec = hip.occ_mlp_vectors(*(p[k].to(DEV) for k in ('b1', 'g1', 'be1', 'b2', 'g2', 'be2', 'b3')))
    got = hip.occ_mlp_forward(x.to(DEV), image, vec)
    assert got.shape == (n, 16) and got.dtype == torch.bfloat16
    pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()) for k, v in p.items()}
    ref = _occ_mlp_reference(x.double(), pr)
    err = (got.float().cpu().double() - ref).abs()
    assert float(err.max()) <= 1e-2 * float(ref.abs().max()) + 2e-2, float(err.max())
    exact = _occ_mlp_reference(x.double(), pr, round_hidden=False)            # the oracle, no hidden rounding
    assert float((got.float().cpu().double() - exact).norm() / exact.norm()) < 1e-2
    assert float((got.float().cpu().double() - ref).norm() / ref.norm()) < 5e-3
    assert hip.occ_mlp_forward(torch.zeros(0, 128, device=DEV, dtype=torch.bfloat16), image, vec).shape == (0, 16)


def test_occ_mlp_fused_backward():
    """ver_occ_mlp_backward (re-computed chain + dgrad chain in registers, row-reduced weight
    gradients) vs autograd through the fp64 reference on the same bf16-rounded operands."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(12)
    p = _occ_mlp_params(gen)
    n = 8000 * 2 + 16 * 31 + 5
    x = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    keys = ('w1', 'b1', 'g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    pd = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
    xd = x.to(DEV).requires_grad_(True)
    out = hip.occ_mlp(xd, *(pd[k] for k in keys))
    out.backward(gy.to(DEV))
    pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()).requires_grad_(True) for k, v in p.items()}
    xr = x.double().requires_grad_(True)
    ref = _occ_mlp_reference(xr, pr, round_hidden=False)
    ref.backward(gy.double())

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    assert rel(out.detach().float(), ref.detach()) < 5e-3
    assert xd.grad.dtype == torch.bfloat16
    assert rel(xd.grad.float(), xr.grad) < 6e-2, rel(xd.grad.float(), xr.grad)
    for k in keys:
        assert pd[k].grad.shape == p[k].shape
        assert rel(pd[k].grad, pr[k].grad) < 8e-2, (k, rel(pd[k].grad, pr[k].grad))


@pytest.mark.parametrize('fused', [True, False])
def test_occ_mlp_fused_with_folded_first_linear(fused, monkeypatch):
    """first_linear = 0 (the first Linear folded into the producer of x): the kernels start at the first LayerNorm.
    ``fused``: the wave-specialised N-split backward kernel that accumulates d(W2) itself (ver_occ_mlp_backward_fused, the
    default) / the row-split kernel + host GEMM over its side tensors.
    x := bf16(Linear1(x0)); forward and every gradient (d x = gradient w.r.t. that output) vs autograd through the
    fp64 chain LayerNorm -> ReLU -> Linear2 -> LayerNorm -> ReLU -> Linear3 on the same x."""
    hip = pkg('hipops')
    monkeypatch.setattr(hip, '_OCC_MLP_BWD_FUSED', fused)
    gen = torch.Generator(device='cpu').manual_seed(13)
    p = _occ_mlp_params(gen)
    n = 8000 + 16 * 7 + 3
    x0 = (torch.randn(n, 128, generator=gen) * 1.5)
    a1 = (x0 @ p['w1'].t() + p['b1']).bfloat16()
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    pd = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
    xd = a1.to(DEV).requires_grad_(True)
    out = hip.occ_mlp(xd, None, None, *(pd[k] for k in keys))
    out.backward(gy.to(DEV))
    pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()).requires_grad_(True) for k, v in p.items()}
    xr = a1.double().requires_grad_(True)
    F = torch.nn.functional
    h = F.relu(F.layer_norm(xr, (128,), pr['g1'], pr['be1'], 1e-5))
    h = F.relu(F.layer_norm(h @ pr['w2'].t() + pr['b2'], (128,), pr['g2'], pr['be2'], 1e-5))
    ref = h @ pr['w3'].t() + pr['b3']
    ref.backward(gy.double())

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    assert rel(out.detach().float(), ref.detach()) < 5e-3
    assert xd.grad.dtype == torch.bfloat16 and xd.grad.shape == a1.shape
    assert rel(xd.grad.float(), xr.grad) < 6e-2, rel(xd.grad.float(), xr.grad)
    for k in keys:
        assert pd[k].grad.shape == p[k].shape
        assert rel(pd[k].grad, pr[k].grad) < 8e-2, (k, rel(pd[k].grad, pr[k].grad))
    assert hip.occ_mlp_forward(torch.zeros(0, 128, device=DEV, dtype=torch.bfloat16),
                               hip.occ_mlp_pack(p['w1'].to(DEV), p['w2'].to(DEV), p['w3'].to(DEV)),
                               hip.occ_mlp_vectors(*(p[k].to(DEV) for k in ('b1', 'g1', 'be1', 'b2', 'g2', 'be2', 'b3'))),
                               first_linear=False).shape == (0, 16)


def test_occ_mlp_focal_loss_fused_equals_the_two_ops():
    """``OccMLPFocalLossFunction`` (the occupancy term of a training step as ONE autograd Function: MLP forward, focal
    forward that leaves the unscaled gradient in the logits buffer, MLP backward reading it with the scalar factor) against
    ``occ_mlp`` followed by ``sigmoid_focal_loss_sum``: the same loss bit for bit (the same two forward kernels), the same
    gradients up to one extra bf16 rounding of d(logits) -- including a non-trivial upstream factor and a bad label's NaN."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(5)
    p = _occ_mlp_params(gen)
    n = 64 * 400 + 9
    a1 = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
    tgt = torch.randint(0, 17, (n,), generator=gen)
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    res = {}
    for fused in (True, False):
        pd = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
        xd = a1.to(DEV).requires_grad_(True)
        if fused:
            s = hip.occ_mlp_focal_loss_sum(xd, *(pd[k] for k in keys), tgt.to(DEV))
        else:
            s = hip.sigmoid_focal_loss_sum(hip.occ_mlp(xd, None, None, *(pd[k] for k in keys)), tgt.to(DEV))
        (s * 0.37 / 1234.0).backward()
        res[fused] = (float(s), xd.grad.float().cpu(), {k: v.grad.float().cpu() for k, v in pd.items()})
    assert res[True][0] == res[False][0]

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    assert rel(res[True][1], res[False][1]) < 1e-2
    for k in keys:
        assert rel(res[True][2][k], res[False][2][k]) < 1e-2, (k, rel(res[True][2][k], res[False][2][k]))
    flag = hip.LabelRangeFlag.of(torch.device(DEV))
    flag.reset()
    try:
        bad = tgt.clone()
        bad[77] = 99
        assert torch.isnan(hip.occ_mlp_focal_loss_sum(a1.to(DEV), *(p[k].to(DEV) for k in keys), bad.to(DEV)))
        with pytest.raises(RuntimeError, match='outside'):
            flag.poll(sync=True)
    finally:
        flag.reset()


def test_occ_mlp_centered_equals_plain_and_the_fp64_chain():
    """VER_OCC_MLP_CENTERED (include/ver_ops.h): with the hidden Linears centred over their output axis
    (W <- W - mean_o W, b <- b - mean b) every LayerNorm input has zero row mean and the kernels skip the mean pass.
    LayerNorm is invariant to a per-row constant, so this is the SAME function of the parameters: forward and every
    gradient (autograd maps the gradients of the centred parameters back through the projection) against the fp64 chain
    on the UNcentred parameters, and against the uncentred fused kernels within the bf16 noise of either."""
    hip = pkg('hipops')
    F = torch.nn.functional
    gen = torch.Generator(device='cpu').manual_seed(21)
    p = _occ_mlp_params(gen)
    n = 64 * 300 + 17
    x0 = torch.randn(n, 128, generator=gen) * 1.5
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')

    def center(w, b):
        return w - w.mean(0, keepdim=True), b - b.mean()

    res = {}
    for centered in (False, True):
        pd = {k: p[k].to(DEV).requires_grad_(True) for k in ('w1', 'b1') + keys}
        w1, b1 = (center(pd['w1'], pd['b1']) if centered else (pd['w1'], pd['b1']))
        w2, b2 = (center(pd['w2'], pd['b2']) if centered else (pd['w2'], pd['b2']))
        a1 = (x0.to(DEV) @ w1.t() + b1).bfloat16()                  # the producer of x (occ_proj with the folded Linear 1)
        a1.retain_grad()
        out = hip.occ_mlp(a1, None, None, pd['g1'], pd['be1'], w2, b2, pd['g2'], pd['be2'], pd['w3'], pd['b3'],
                          centered=centered)
        out.backward(gy.to(DEV))
        if centered:
            assert float(a1.float().mean(1).abs().max()) < 0.05      # rows ARE centred (bf16 rounding of the producer aside)
        res[centered] = (out.detach().float().cpu(), {k: v.grad.float().cpu() for k, v in pd.items()})
    pr = {k: v.double().requires_grad_(True) for k, v in p.items()}
    a = (x0.double() @ pr['w1'].t() + pr['b1'])
    h = F.relu(F.layer_norm(a, (128,), pr['g1'], pr['be1'], 1e-5))
    h = F.relu(F.layer_norm(h @ pr['w2'].t() + pr['b2'], (128,), pr['g2'], pr['be2'], 1e-5))
    ref = h @ pr['w3'].t() + pr['b3']
    ref.backward(gy.double())

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    for centered in (False, True):
        out, grads = res[centered]
        assert rel(out, ref.detach()) < 1e-2, (centered, rel(out, ref.detach()))
        for k in ('w1', 'b1') + keys:
            assert rel(grads[k], pr[k].grad) < 9e-2, (centered, k, rel(grads[k], pr[k].grad))
    assert rel(res[True][0], res[False][0]) < 1e-2
    # the projection leaves no gradient along the all-ones direction of the centred parameters
    assert float(res[True][1]['w2'].sum(0).abs().max()) < 1e-3 * float(res[True][1]['w2'].abs().max()) * 128


@pytest.mark.parametrize('n', [1, 65, 64 * 3 + 5, 64 * 256 + 1, 64 * 1027 + 33])
def test_occ_mlp_backward_with_saved_statistics_equals_the_recomputing_kernel(n, monkeypatch):
    """ver_occ_mlp_forward_stats / ver_occ_mlp_backward_fused_stats (round 5): on centred rows the forward kernel saves
    1/std of both LayerNorms per row and the wave-specialised backward reads them back instead of recomputing the
    statistics (its two LayerNorm-forward steps become elementwise).  The saved values are what the recomputation yields
    (to fp32 rounding for LayerNorm 1; for LayerNorm 2 up to the bf16 rounding of a2 between the kernel's two views: a
    relative 1e-4 of 1/std -- enough to move ~2 % of the normalised values by one bf16 ulp and, through those, to flip a
    ReLU gate in ~1e-4 of the elements, each of which changes its gradient element entirely: sqrt(1e-4) = 1 % relative
    L2, the same mechanism that puts ANY two bf16 evaluations of this chain 3-5 % apart,
    test_occ_mlp_backward_kernels_on_ragged_sizes), so d(x) and every parameter gradient must agree with the recomputing
    form inside that distance; the statistics themselves are checked against torch."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(300 + n % 89)
    p = _occ_mlp_params(gen)
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    x0 = torch.randn(n, 128, generator=gen) * 1.5
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    from util import rel_l2

    def center(w, b):
        return w - w.mean(0, keepdim=True), b - b.mean()
    res = {}
    for saved in (True, False):
        monkeypatch.setattr(hip, '_OCC_MLP_SAVE_RSTD', saved)
        pd = {k: p[k].to(DEV).requires_grad_(True) for k in ('w1', 'b1') + keys}
        w1, b1 = center(pd['w1'], pd['b1'])
        w2, b2 = center(pd['w2'], pd['b2'])
        a1 = (x0.to(DEV) @ w1.t() + b1).bfloat16().detach().requires_grad_(True)
        out = hip.occ_mlp(a1, None, None, pd['g1'], pd['be1'], w2, b2, pd['g2'], pd['be2'], pd['w3'], pd['b3'], centered=True)
        out.backward(gy.to(DEV))
        res[saved] = (out.detach().float().cpu(), a1.grad.float().cpu(), {k: pd[k].grad.float().cpu() for k in keys})
    assert torch.equal(res[True][0], res[False][0])                  # the forward arithmetic is untouched
    assert rel_l2(res[True][1], res[False][1]) < 3e-2, rel_l2(res[True][1], res[False][1])
    for k in keys:
        assert rel_l2(res[True][2][k], res[False][2][k]) < 3e-2, (k, rel_l2(res[True][2][k], res[False][2][k]))
    # the statistics the forward kernel wrote: LayerNorm 1 on the loaded rows (no mean pass on centred rows)
    pd = {k: p[k].to(DEV) for k in ('w1', 'b1') + keys}
    w1, b1 = center(pd['w1'], pd['b1'])
    w2, b2 = center(pd['w2'], pd['b2'])
    a1 = (x0.to(DEV) @ w1.t() + b1).bfloat16()
    image = hip.occ_mlp_pack(w2, w2, pd['w3'])
    vec = hip.occ_mlp_vectors(torch.zeros(128, device=DEV), pd['g1'], pd['be1'], b2, pd['g2'], pd['be2'], pd['b3'])
    _, rstd = hip.occ_mlp_forward(a1, image, vec, 1e-5, first_linear=False, centered=True, want_rstd=True)
    want1 = torch.rsqrt(a1.float().pow(2).mean(1) + 1e-5)
    assert rstd.shape == (n, 2) and float(((rstd[:, 0] - want1) / want1).abs().max()) < 1e-5
    assert bool(torch.isfinite(rstd).all()) and float(rstd[:, 1].min()) > 0


@pytest.mark.parametrize('n', [1, 63, 64, 65, 128, 191, 64 * 256 + 1, 64 * 513 + 7])
def test_occ_mlp_backward_kernels_on_ragged_sizes(n, monkeypatch):
    """Both backward kernels of the folded MLP on row counts around the block / pipeline edges of the wave-specialised
    kernel (64-row blocks, two in flight per workgroup, 256 persistent workgroups: one row, one block, one block + a row,
    an odd number of blocks, more blocks than workgroups, more than two rounds): d(x) and every parameter gradient
    against autograd through the fp64 chain, and the two kernels within the same distance of each other (what separates
    any two bf16 evaluations of this chain is a handful of ReLU gates that flip with the rounding of the pre-activations:
    3-5 % relative L2 at these sizes, for either kernel)."""
    hip = pkg('hipops')
    F = torch.nn.functional
    gen = torch.Generator(device='cpu').manual_seed(100 + n % 97)
    p = _occ_mlp_params(gen)
    a1 = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()).requires_grad_(True) for k, v in p.items()}
    xr = a1.double().requires_grad_(True)
    h = F.relu(F.layer_norm(xr, (128,), pr['g1'], pr['be1'], 1e-5))
    h = F.relu(F.layer_norm(h @ pr['w2'].t() + pr['b2'], (128,), pr['g2'], pr['be2'], 1e-5))
    ((h @ pr['w3'].t() + pr['b3']) * gy.double()).sum().backward()
    from util import rel_l2
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(hip, '_OCC_MLP_BWD_FUSED', fused)
        pd = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
        xd = a1.to(DEV).requires_grad_(True)
        hip.occ_mlp(xd, None, None, *(pd[k] for k in keys)).backward(gy.to(DEV))
        res[fused] = dict(x=xd.grad.float().cpu(), **{k: pd[k].grad.float().cpu() for k in keys})
        assert torch.isfinite(res[fused]['x']).all()
        assert rel_l2(res[fused]['x'], xr.grad) < 7e-2, (fused, rel_l2(res[fused]['x'], xr.grad))
        for k in keys:
            assert rel_l2(res[fused][k], pr[k].grad) < 8e-2, (fused, k, rel_l2(res[fused][k], pr[k].grad))
    for k in ('x',) + keys:
        assert rel_l2(res[True][k], res[False][k]) < 8e-2, (k, rel_l2(res[True][k], res[False][k]))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_run_gather_scatter_equal_index_select(dtype):
    """ver_run_gather / ver_run_scatter (the gathered occ_proj operand as run copies) vs torch.index_select /
    index_copy_ through the plan's index tables, bit for bit, on the vocc.py geometry."""
    hip = pkg('hipops')
    opl = pkg('dense_heads.occ_proj_lattice')
    plan = opl.get_plan(768, 4, 120, 120, torch.device(DEV))
    L, C, bs = plan.lattice_size, plan.C, 2
    gen = torch.Generator(device='cpu').manual_seed(3)
    lat = torch.randn(bs, (L + C + 2 + 7) // 8 * 8, generator=gen).to(dtype).to(DEV)
    for g in plan.groups[:3]:
        assert g.run_len > 0
        want = lat.index_select(1, g.gather_aug).view(bs * g.n_rows, g.k_aug)
        got = torch.full_like(want, float('nan'))
        hip.run_gather(lat, g.run_start, g.aug_idx, got, g.n_rows, g.run_len)
        assert torch.equal(got, want)
        d_data = torch.randn(bs * g.n_rows, g.n_cols, generator=gen).to(dtype).to(DEV)
        ref = torch.zeros(bs, L, dtype=dtype, device=DEV)
        ref.index_copy_(1, g.scatter, d_data.view(bs, -1))
        out = torch.zeros(bs, L, dtype=dtype, device=DEV)
        hip.run_scatter(d_data, out, g.run_start, g.n_rows, g.run_len)
        assert torch.equal(out, ref)
    with pytest.raises(RuntimeError):                      # run length not a multiple of 8 bytes
        hip.run_gather(lat, plan.groups[0].run_start, plan.groups[0].aug_idx,
                       torch.empty(bs * plan.groups[0].n_rows, plan.groups[0].k_aug, dtype=dtype, device=DEV),
                       plan.groups[0].n_rows, 181)


# ------------------------------------------------------------------------------- next row 2: occupancy loss
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('gamma,alpha', [(2.0, 0.25), (1.5, 0.4)])
def test_focal_loss_fused(dtype, gamma, alpha):
    """ver_focal_loss_* vs the oracle's restatement of mmdet's sigmoid focal loss (head:977-989),
    evaluated in fp64 on the same (dtype-rounded) logits."""
    hip = pkg('hipops')
    orc = oracle()
    gen = torch.Generator(device='cpu').manual_seed(10)
    n, c = 70001, 16
    logits = (torch.randn(n, c, generator=gen) * 4).to(dtype)
    logits[0, :4] = torch.tensor([60.0, -60.0, 0.0, 20.0]).to(dtype)      # saturated sigmoid both ways
    target = torch.randint(0, c + 1, (n,), generator=gen)
    target[0] = 0
    avg = float((target < c).sum())
    ld = logits.to(DEV).requires_grad_(True)
    s = hip.sigmoid_focal_loss_sum(ld, target.to(DEV), gamma, alpha)
    (s / avg * 0.7).backward()
    lr = logits.double().requires_grad_(True)
    ref = orc.focal_loss(lr, target, gamma=gamma, alpha=alpha, avg_factor=avg, loss_weight=0.7)
    ref.backward()
    assert abs(float(s) / avg * 0.7 - float(ref)) <= 1e-5 * abs(float(ref))
    assert ld.grad.dtype == dtype
    if dtype == torch.float32:
        assert close(ld.grad.cpu(), lr.grad, atol=1e-9, rtol=1e-4)
    else:
        assert close(ld.grad.float().cpu(), lr.grad, atol=1e-9, rtol=1e-2)
    # the registered loss routes large GPU inputs through the kernel and gives the same value
    loss_mod = pkg('dense_heads.losses').FocalLoss(gamma=gamma, alpha=alpha, loss_weight=0.7)
    got = loss_mod(logits.to(DEV), target.to(DEV), avg_factor=avg)
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))
    # empty input
    z = hip.sigmoid_focal_loss_sum(torch.zeros(0, 16, device=DEV, dtype=dtype),
                                   torch.zeros(0, dtype=torch.long, device=DEV))
    assert float(z) == 0.0


# ------------------------------------------------------------------------------- next row 1: decoder op
@pytest.mark.parametrize('name', list(cases.MSDA3D_CASES))
def test_voxel_msda_forward_backward(name):
    """3-D (trilinear) sampling op of the detection decoder: HIP vs the reference's in-tree
    function (golden) and vs the oracle."""
    hip = pkg('hipops')
    o = oracle()
    g = golden('msda3d_core_' + name)
    c = cases.msda3d_inputs(**cases.MSDA3D_CASES[name])
    value = T(c['value']).to(DEV).requires_grad_(True)
    loc = T(c['loc']).to(DEV).requires_grad_(True)
    w = T(c['w']).to(DEV).requires_grad_(True)
    out = hip.voxel_msda(value, T(c['shapes']).to(DEV), T(c['level_start']).to(DEV), loc, w)
    out.backward(T(c['grad_out']).to(DEV))
    assert maxdiff(out.detach().cpu(), g['out']) < 2e-5
    assert close(value.grad.cpu()[:, ::3], g['grad_value'])
    assert close(loc.grad.cpu(), g['grad_loc'])
    assert close(w.grad.cpu(), g['grad_w'])
    vc, lc, wc = T(c['value']).requires_grad_(True), T(c['loc']).requires_grad_(True), T(c['w']).requires_grad_(True)
    ref = o.voxel_msda_core(vc, c['shapes'].tolist(), lc, wc)
    ref.backward(T(c['grad_out']))
    assert maxdiff(out.detach().cpu(), ref.detach()) < 2e-5
    assert close(value.grad.cpu(), vc.grad)
    far = torch.full_like(loc, 3.0)
    assert float(hip.voxel_msda(value.detach(), T(c['shapes']).to(DEV), T(c['level_start']).to(DEV), far,
                                w.detach()).abs().max()) == 0.0


# ------------------------------------------------------------------------------- a7: residual + dropout + LayerNorm
@pytest.mark.parametrize('adtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p_drop,c', [(0.0, 768), (0.1, 768), (0.1, 256), (0.0, 1024)])
def test_add_dropout_layer_norm_fused(adtype, p_drop, c):
    """ver_add_ln_*: y = LayerNorm(residual + dropout(a)) (the tail of both branches of a VoxelFormerLayer,
    voxel_encoder.py:344-464) vs torch.  With dropout the kept set is read off d(a) (zero exactly where an element was
    dropped), which also checks that the backward pass recomputes the forward's decisions."""
    hip = pkg('hipops')
    F = torch.nn.functional
    gen = torch.Generator(device='cpu').manual_seed(31)
    n = 777
    a = torch.randn(n, c, generator=gen).to(adtype)
    res = torch.randn(n, c, generator=gen)
    gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.1
    gy = torch.randn(n, c, generator=gen)
    gy16 = (torch.randn(n, c, generator=gen) * 0.5).bfloat16()
    ad, rd = a.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    y, y16 = hip.add_dropout_layer_norm(ad, rd, gd, bd, p_drop, 1e-5, want_bf16=True)
    assert y.dtype == torch.float32 and y16.dtype == torch.bfloat16
    assert torch.equal(y16, y.detach().bfloat16())
    torch.autograd.backward([y, y16], [gy.to(DEV), gy16.to(DEV)])
    keep = (ad.grad != 0).float().cpu() if p_drop > 0 else torch.ones(n, c)
    if p_drop > 0:
        assert abs(float(keep.mean()) - (1 - p_drop)) < 5e-3
    ar, rr = a.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    x = rr + ar * keep.double() / (1 - p_drop)
    ref = F.layer_norm(x, (c,), gr, br, 1e-5)
    ref.backward(gy.double() + gy16.double())
    assert maxdiff(y.detach().cpu(), ref.detach().float()) < 2e-5
    tol = 2e-2 if adtype == torch.bfloat16 else 1e-4
    assert close(ad.grad.float().cpu(), ar.grad.float(), atol=tol, rtol=tol)
    assert close(rd.grad.cpu(), rr.grad.float(), atol=1e-4, rtol=1e-4)
    assert close(gd.grad.cpu(), gr.grad.float(), atol=2e-3, rtol=1e-4)
    assert close(bd.grad.cpu(), br.grad.float(), atol=2e-3, rtol=1e-4)
    # no bf16 copy requested / empty input
    y2, none = hip.add_dropout_layer_norm(ad.detach(), rd.detach(), gd.detach(), bd.detach(), 0.0, 1e-5, want_bf16=False)
    assert none is None and y2.shape == (n, c)
    e, _ = hip.add_dropout_layer_norm(torch.zeros(0, c, device=DEV), torch.zeros(0, c, device=DEV), gd.detach(), bd.detach())
    assert e.shape == (0, c)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_relu_dropout_fused(dtype, p_drop):
    """ver_relu_dropout_*: y = dropout(relu(x)) (hidden activation of the layer's FFN); the kept set is read off y,
    the backward pass must reproduce d(x) = y > 0 ? d(y) / (1 - p) : 0."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(33)
    x = torch.randn(1000, 1536, generator=gen).to(dtype)
    gy = torch.randn(1000, 1536, generator=gen).to(dtype)
    xd = x.to(DEV).requires_grad_(True)
    y = hip.relu_dropout(xd, p_drop)
    assert y.dtype == dtype
    y.backward(gy.to(DEV))
    yc, pos = y.detach().float().cpu(), x.float() > 0
    kept = (yc != 0)
    assert not bool((kept & ~pos).any())                                  # nothing appears where relu(x) = 0
    if p_drop > 0:
        assert abs(float(kept[pos].float().mean()) - (1 - p_drop)) < 5e-3
    else:
        assert bool((kept == pos).all())
    want = torch.where(kept, x.float() / (1 - p_drop), torch.zeros(())).to(dtype).float()
    assert torch.equal(yc, want) if p_drop == 0 else close(yc, want, atol=1e-6, rtol=1e-2 if dtype == torch.bfloat16 else 1e-6)
    want_g = torch.where(kept, gy.float() / (1 - p_drop), torch.zeros(()))
    assert close(xd.grad.float().cpu(), want_g, atol=1e-6, rtol=1e-2 if dtype == torch.bfloat16 else 1e-6)
    assert hip.relu_dropout(torch.zeros(0, 8, device=DEV), 0.1).shape == (0, 8)


def test_focal_loss_label_range_is_loud_without_a_sync_per_step():
    """A label outside [0, C] raises in F.one_hot (and in the reference).  The fused path: (a) the first call of a
    FocalLoss module checks the range on the host and raises with the range in the message; (b) afterwards no host
    check runs (no device->host synchronisation inside a training step): the kernel itself answers a bad label
    with a NaN loss instead of silently counting the row as background AND raises a sticky device flag, whose
    asynchronous host mirror makes a later fused call -- or ``FocalLoss.check_labels()`` -- raise."""
    hip = pkg('hipops')
    losses = pkg('dense_heads.losses')
    gen = torch.Generator(device='cpu').manual_seed(3)
    logits = torch.randn(5000, 16, generator=gen).to(DEV)
    good = torch.randint(0, 17, (5000,), generator=gen).to(DEV)
    flag = hip.LabelRangeFlag.of(logits.device)
    try:
        for bad_label in (17, -1, 2 ** 32 + 3):           # (2^32 + 3 would alias class 3 if the kernel truncated to 32 bits)
            flag.reset()
            bad = good.clone()
            bad[1234] = bad_label
            mod = losses.FocalLoss(loss_weight=1.0)
            with pytest.raises(RuntimeError, match='target labels must be in'):
                mod(logits, bad, avg_factor=100.0)
            mod = losses.FocalLoss(loss_weight=1.0)
            assert torch.isfinite(mod(logits, good, avg_factor=100.0))        # first call: host check passes
            losses.FocalLoss.check_labels()                                   # nothing bad so far
            assert torch.isnan(mod(logits, bad, avg_factor=100.0))            # later calls: the kernel's NaN ...
            torch.cuda.synchronize()
            with pytest.raises(RuntimeError, match='outside'):                # ... and the next fused call is loud
                mod(logits, good, avg_factor=100.0)
            with pytest.raises(RuntimeError, match='outside'):
                losses.FocalLoss.check_labels()
            flag.reset()
            assert torch.isnan(hip.sigmoid_focal_loss_sum(logits.bfloat16(), bad))
            with pytest.raises(RuntimeError, match='outside'):
                losses.FocalLoss.check_labels()
        flag.reset()
        assert torch.isfinite(hip.sigmoid_focal_loss_sum(logits, good))
        losses.FocalLoss.check_labels()
    finally:
        flag.reset()


def test_sca_gather_three_camera_overlap_determinism_bound():
    """The determinism contract of ``ver_sca_forward`` (include/ver_ops.h): rows seen by <= 2 cameras are bitwise
    reproducible; rows seen by >= 3 cameras are accumulated with fp32 atomics whose order is not fixed, so they may
    differ run to run -- by no more than the rounding of a 3-6 term fp32 sum.  A wide-angle rig (137 degree cameras
    every 60 degrees) makes 3-camera voxels; 20 runs are compared with the first and with the oracle."""
    hip = pkg('hipops')
    syn = pkg('synthetic')
    o = oracle()
    rng = np.random.default_rng(17)
    B, grid, heads, hd, P = 2, (4, 15, 15), 8, 96, 8
    z, h, w = grid
    nq = z * h * w
    org = syn.viewpoint_origins(B, seed=1)
    w2p = np.stack([syn.camera_rig(og, fx=250.0, fy=250.0) for og in org]).astype(np.float32)
    hit = hip.project_points(T(w2p).to(DEV), T(org.astype(np.float32)).to(DEV), cases.PC_RANGE, z, h, w)
    mask = hit.mask()[:, :, :, 0].permute(1, 0, 2).cpu()           # [B, Ncam, Nq]
    ncam = mask.sum(1)
    assert int(ncam.max()) >= 3 and int((ncam >= 3).sum()) > 50, 'rig does not produce 3-camera voxels'
    value = T(rng.standard_normal((B, 6, 196, heads, hd)).astype(np.float32))
    offsets = T((rng.standard_normal((B, nq, heads, P, 2)) * 3.0).astype(np.float32))
    logits = T(rng.standard_normal((B, nq, heads, P)).astype(np.float32))
    for vdev in (value.to(DEV), value.to(DEV).bfloat16()):
        first = hip.sca_gather(vdev, offsets.to(DEV), logits.to(DEV), hit, 14, 14)
        worst = 0.0
        for _ in range(20):
            again = hip.sca_gather(vdev, offsets.to(DEV), logits.to(DEV), hit, 14, 14)
            d = (again - first).abs()
            assert float(d[ncam.to(DEV) <= 2].max()) == 0.0          # <= 2 cameras: bitwise
            worst = max(worst, float(d.max()))
        scale = float(first.abs().max())
        print('3-camera rows: %d, run-to-run max |diff| %.3e (|slots| max %.2f)' % (int((ncam >= 3).sum()), worst, scale))
        assert worst <= 8 * 2.0 ** -24 * scale                        # a few ulps of the largest partial sum
    ref = oracle_slots(o, value, offsets, logits, hit.uv.cpu(), mask, (14, 14))
    first = hip.sca_gather(value.to(DEV), offsets.to(DEV), logits.to(DEV), hit, 14, 14)
    assert maxdiff(first.cpu(), ref) < 2e-5


# ------------------------------------------------------------------------------------------ head GEMMs (round 5)
@pytest.mark.parametrize('M,Ka,lda,c0,N,ldg,splits', [
    (2048, 256, 256, 0, 256, 256, 1), (4096, 480, 640, 64, 384, 384, 2), (16384, 224, 1024, 128, 260, 264, 8),
    (32768, 1000, 1024, 0, 772, 776, 16), (1800, 300, 304, 0, 132, 136, 0), (14400 + 7, 520, 520, 0, 256, 256, 3),
    (50000, 64, 64, 0, 64, 64, 0), (15, 40, 40, 0, 8, 8, 0)])
def test_wgrad_tn_equals_the_fp32_product(M, Ka, lda, c0, N, ldg, splits):
    """ver_wgrad_tn (csrc/ver_wgrad.hip): dW = A^T G with the ROWS on the contraction axis -- the weight gradient of the
    reference's ConvTranspose3d stack and occ_proj (head:251-258, :560, :571) as dense_heads/upsample.py::rows_tn forms it.
    Operands that are column ranges of wider matrices, ragged tile edges in both output dimensions, row counts that are
    not multiples of the 16-row slab or of the chunk count (the tail reads as zeros through the buffer range), every
    split count; fp32 output against the fp32 product of the same bf16 operands (fp32 accumulation order only), bf16
    output one rounding away."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(M + Ka + N)
    Af = torch.randn(M, lda, generator=gen).to(DEV).to(torch.bfloat16)
    Gf = torch.randn(M, ldg, generator=gen).to(DEV).to(torch.bfloat16)
    A, G = Af[:, c0:c0 + Ka], Gf[:, :N]
    want = A.float().t() @ G.float()
    got = hip.wgrad_tn(A, G, out_dtype=torch.float32, splits=splits)
    assert got.shape == (Ka, N)
    assert float((got - want).abs().max()) < 3e-5 * float(want.abs().max()) * max(1.0, (M / 4096) ** 0.5)
    got16 = hip.wgrad_tn(A, G, splits=splits)
    assert got16.dtype == torch.bfloat16
    from util import rel_l2
    assert rel_l2(got16.float(), want) < 3e-3
    # transpose-detecting: A and G are different random matrices, Ka != N in most cases; and the lattice layers' helper
    up = pkg('dense_heads.upsample')
    assert rel_l2(up.rows_tn(A, G).float(), want) < 3e-3
    assert rel_l2(up.rows_tn(G, A, out_dtype=torch.float32), want.t()) < 1e-4


def test_wgrad_tn_row_chunk_beyond_2_gib():
    """One row chunk whose byte range exceeds 2^31 (40 000 rows of a 64-KiB row pitch, one chunk): the running slab offset
    of the LDS-DMA descriptors is an UNSIGNED 32-bit byte count (the launcher admits chunks up to 4 GiB)."""
    hip = pkg('hipops')
    M, lda, Ka, N = 40000, 32768, 256, 256
    torch.manual_seed(5)
    Af = torch.empty(M, lda, device=DEV, dtype=torch.bfloat16)
    Af[:, 512:768] = torch.randn(M, Ka, device=DEV).to(torch.bfloat16)
    G = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    A = Af[:, 512:768]
    want = A.float().t() @ G.float()
    got = hip.wgrad_tn(A, G, out_dtype=torch.float32, splits=1)
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
    # the last rows (offsets above 2^31) carry weight: zero them and the product changes
    G2 = G.clone()
    G2[-4096:] = 0
    assert float((hip.wgrad_tn(A, G2, out_dtype=torch.float32, splits=1) - got).abs().max()) > 1.0


def test_wgrad_tn_refuses_what_it_cannot_take():
    hip = pkg('hipops')
    a = torch.randn(64, 32, device=DEV).to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        hip.wgrad_tn(a.float(), a.float())                       # fp32 operands: the caller's plain matmul
    with pytest.raises(RuntimeError):
        hip.wgrad_tn(a[:, 1:9], a)                                # rows not 16-byte aligned
    z = hip.wgrad_tn(a[:0], a[:0], out_dtype=torch.float32)       # no rows: zeros
    assert z.shape == (32, 32) and float(z.abs().max()) == 0.0


@pytest.mark.parametrize('M,K,lda,c0,N,ldw,bias', [
    (256, 64, 64, 0, 256, 256, False), (512, 128, 192, 64, 256, 256, True), (300, 96, 96, 0, 200, 200, True),
    (1000, 1024, 1088, 32, 772, 776, False), (4096, 2048, 2048, 0, 1536, 1536, False), (77, 320, 320, 0, 40, 40, True),
    (2049, 4096, 4096, 0, 520, 520, True)])
def test_gemm_nn_equals_the_fp32_product(M, K, lda, c0, N, ldw, bias):
    """ver_gemm_nn (csrc/ver_gemm.hip): C = A W (+ bias), the forward GEMM of a lattice layer (head:560 on the even
    lattice).  A as a column range of a wider matrix, partial tiles in M and N, results written into a column range of a
    wider output without touching its neighbours; against the fp32 product of the same bf16 operands rounded once."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(M + K + N)
    Af = torch.randn(M, lda, generator=gen).to(DEV).to(torch.bfloat16)
    Wf = torch.randn(K, ldw, generator=gen).to(DEV).to(torch.bfloat16)
    b = torch.randn(N, generator=gen).to(DEV) if bias else None
    A, W = Af[:, c0:c0 + K], Wf[:, :N]
    want = A.float() @ W.float() + (b if bias else 0)
    got = hip.gemm_nn(A, W, b)
    from util import rel_l2
    assert got.dtype == torch.bfloat16 and rel_l2(got.float(), want) < 3e-3
    assert float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())       # one bf16 rounding
    wide = torch.full((M, N + 24), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nn(A, W, b, out=wide[:, 8:8 + N])
    assert torch.equal(wide[:, 8:8 + N], got) and float(wide[:, :8].min()) == 7.0 and float(wide[:, 8 + N:].min()) == 7.0
    with pytest.raises(RuntimeError):
        hip.gemm_nn(Af[:, :K - 16] if K > 80 else Af[:, :48], Wf[:K - 16 if K > 80 else 48, :N])      # K % 32 != 0


def test_focal_forward_grad_with_byte_labels_is_bit_identical():
    """ver_focal_loss_forward_grad_u8 (labels as bytes) against the int64-label entry on the same logits: the same partial
    sums and the same in-place gradients, and 255 (a wrapped -1) is an invalid label that poisons the sum and raises the
    device-side flag."""
    import ctypes
    hip = pkg('hipops')
    L = hip.lib()
    gen = torch.Generator(device='cpu').manual_seed(57)
    n = 100003
    logits = (torch.randn(n, 16, generator=gen) * 2).bfloat16().to(DEV)
    lab = torch.randint(0, 17, (n,), generator=gen)
    blocks = L.ver_focal_loss_blocks(ctypes.c_long(n), 16)
    res = []
    for labels, entry in ((lab.to(DEV), L.ver_focal_loss_forward_grad), (lab.to(torch.uint8).to(DEV), L.ver_focal_loss_forward_grad_u8)):
        x = logits.clone()
        partial = torch.zeros(blocks, dtype=torch.float32, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        rc = entry(hip._p(x), hip._p(labels), hip._p(partial), hip._p(x), ctypes.c_long(n), 16, ctypes.c_float(2.0),
                   ctypes.c_float(0.25), 1, hip._p(flag), hip._stream())
        assert rc == 0 and int(flag) == 0
        res.append((partial.clone(), x))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    bad = lab.to(torch.uint8).to(DEV)
    bad[17] = 255
    partial = torch.zeros(blocks, dtype=torch.float32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    x = logits.clone()
    assert L.ver_focal_loss_forward_grad_u8(hip._p(x), hip._p(bad), hip._p(partial), hip._p(x), ctypes.c_long(n), 16,
                                             ctypes.c_float(2.0), ctypes.c_float(0.25), 1, hip._p(flag), hip._stream()) == 0
    assert int(flag) == 1 and bool(torch.isnan(partial.sum()))


# ------------------------------------------------------------------------------- the optimizer step of the bench
@pytest.mark.parametrize('max_norm', [0.5, 1e9, 0.0])
def test_clip_adamw_equals_clip_grad_norm_plus_torch_adamw(max_norm):
    """``optim.ClipAdamW`` (ver_clip_adamw_step: global-norm clipping + AdamW in two launches) against
    ``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.AdamW`` over four steps: tensors of awkward sizes (1, 5, a 4-byte
    aligned view, 1 000 003 elements, a matrix), clipping active (0.5), inactive (1e9) and off (0); the returned norm is the
    unclipped gradient norm; parameters without a gradient are left alone."""
    opt_mod = pkg('optim')
    gen = torch.Generator(device='cpu').manual_seed(77)
    shapes = [(1,), (5,), (1000003,), (33, 7), (257,)]
    base = [torch.randn(s, generator=gen) for s in shapes]
    flat = torch.randn(1031, generator=gen)

    def make():
        ps = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
        buf = flat.clone().to(DEV)
        ps.append(torch.nn.Parameter(buf[1:1025]))              # data pointer 4 bytes off a 16-byte boundary
        ps.append(torch.nn.Parameter(torch.ones(9, device=DEV)))   # never receives a gradient
        return ps
    ours, ref = make(), make()
    a = opt_mod.ClipAdamW(ours, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, max_norm=max_norm)
    b = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    for it in range(4):
        grads = [torch.randn(p.shape, generator=gen) * (10.0 if it == 2 else 1.0) for p in ours[:-1]]
        for p, q, g in zip(ours, ref, grads):
            p.grad = g.clone().to(DEV)
            q.grad = g.clone().to(DEV)
        want_norm = torch.linalg.vector_norm(torch.cat([g.reshape(-1) for g in grads]).double())
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(ref[:-1], max_norm)
        b.step()
        got_norm = a.step()
        assert float(got_norm) == pytest.approx(float(want_norm), rel=1e-5)
        for p, g in zip(ours, grads):
            assert torch.equal(p.grad.cpu(), g)                    # gradients are read, not rewritten
        # (a few fp32 roundings apart: torch forms the first moment as a lerp and divides by sqrt(bias correction 2))
        near = lambda x, y: close(x.cpu(), y.cpu(), atol=2e-6 * max(float(y.abs().max()), 1e-30), rtol=2e-6)
        for p, q in zip(ours, ref):
            assert near(p.detach(), q.detach()), it
        for p, q in zip(ours[:-1], ref[:-1]):
            assert near(a.state[p]['exp_avg'], b.state[q]['exp_avg']) and near(a.state[p]['exp_avg_sq'], b.state[q]['exp_avg_sq'])
    assert torch.equal(ours[-1].detach().cpu(), torch.ones(9)) and not a.state[ours[-1]]
    cpu_p = torch.nn.Parameter(torch.ones(3))                      # no CPU path: loud
    cpu_p.grad = torch.ones(3)
    with pytest.raises(TypeError):
        opt_mod.ClipAdamW([cpu_p], max_norm=1.0).step()


def test_clip_adamw_survives_state_reload_groups_late_parameters_and_nan():
    """``optim.ClipAdamW`` against torch (round-5 advisor findings): (a) step, ``state_dict`` -> ``load_state_dict`` (new moment
    tensors, steps as loaded), step again -- the RESTORED moments are the ones that keep moving; (b) a torch AdamW
    checkpoint (one step tensor per parameter) loads; (c) two parameter groups with their own lr / weight decay
    (vocc.py:260-267 paramwise_cfg) under ONE global clip norm; (d) a parameter whose first gradient arrives two steps
    late gets its own bias corrections; (e) ``p.data`` moved to new storage is followed; (f) a NaN gradient poisons every
    parameter, as ``clip_grad_norm_`` + AdamW does, instead of leaving the finite ones unclipped."""
    opt_mod = pkg('optim')
    gen = torch.Generator(device='cpu').manual_seed(78)
    shapes = [(3000,), (17, 5), (40000,)]
    base = [torch.randn(s, generator=gen) for s in shapes]
    near = lambda x, y: close(x.cpu(), y.cpu(), atol=3e-6 * max(float(y.abs().max()), 1e-30), rtol=3e-6)

    def make():
        return [torch.nn.Parameter(b.clone().to(DEV)) for b in base]

    def groups(ps):
        return [dict(params=ps[:2], lr=3e-3, weight_decay=0.05), dict(params=ps[2:], lr=3e-4, weight_decay=0.0)]
    ours, ref = make(), make()
    a = opt_mod.ClipAdamW(groups(ours), betas=(0.9, 0.99), max_norm=0.7)
    b = torch.optim.AdamW(groups(ref), betas=(0.9, 0.99))

    def both(it, late=2):
        for i, (p, q) in enumerate(zip(ours, ref)):
            if i == 1 and it < late:                                # (d): no gradient for the first `late` steps
                p.grad = q.grad = None
                continue
            g = torch.randn(p.shape, generator=gen)
            p.grad, q.grad = g.clone().to(DEV), g.clone().to(DEV)
        torch.nn.utils.clip_grad_norm_([q for q in ref if q.grad is not None], 0.7)
        b.step()
        a.step()
        for p, q in zip(ours, ref):
            assert near(p.detach(), q.detach()), it
    for it in range(4):
        both(it)
    assert a.state[ours[0]]['step'] == 4 and a.state[ours[1]]['step'] == 2
    # (a) reload our own state: new tensors behind the same parameters
    a.load_state_dict(a.state_dict())
    b.load_state_dict(b.state_dict())
    both(4)
    for p, q in zip(ours, ref):
        assert near(a.state[p]['exp_avg'], b.state[q]['exp_avg']) and near(a.state[p]['exp_avg_sq'], b.state[q]['exp_avg_sq'])
    # (b) a torch checkpoint into ours (step tensors) and ours into torch
    import copy
    a.load_state_dict(copy.deepcopy(b.state_dict()))        # (deep copies, like a checkpoint file: torch's loader keeps fp32 GPU
    both(5)                                                  #  tensors as they are, and two optimizers would share their moments)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    both(6)
    assert a.state[ours[0]]['step'] == 7 and a.state[ours[1]]['step'] == 5
    # (e) the parameter's storage moves
    for p, q in zip(ours, ref):
        p.data = p.data.clone()
        q.data = q.data.clone()
    both(7)
    # a scheduler moves lr of one group
    a.param_groups[1]['lr'] = b.param_groups[1]['lr'] = 1e-3
    both(8)
    # (f) one NaN in one gradient
    for p, q in zip(ours, ref):
        g = torch.randn(p.shape, generator=gen)
        p.grad, q.grad = g.clone().to(DEV), g.clone().to(DEV)
    ours[2].grad[5] = float('nan')
    ref[2].grad[5] = float('nan')
    torch.nn.utils.clip_grad_norm_(ref, 0.7)
    b.step()
    norm = a.step()
    assert bool(torch.isnan(norm))
    for p, q in zip(ours, ref):
        assert bool(torch.isnan(q).all()) and bool(torch.isnan(p).all())


@pytest.mark.parametrize('kind,ci,co,dtype', [('lat', 16, 8, torch.float32), ('lat', 768, 768, torch.bfloat16), ('l0', 24, 16, torch.bfloat16)])
def test_class_stacked_weights_straight_from_the_parameter(kind, ci, co, dtype):
    """ver_convt_weight_forward_blocks: the ConvTranspose3d weight (head:251-258) written into the class-stacked
    [W_lo | W_hi] matrices of a Z = 4 lattice layer, against the passes it replaces (flipped taps -> row gathers per class,
    dense_heads/upsample.py) -- pure data movement + one rounding: bit-exact; rows of the constant blocks untouched.
    ver_blocks_vec_forward / _backward: a vector through every tap block of the stacked matrix (v = b_prev^T K[t]) and its
    adjoint, against the batched products over the tap tensor."""
    hip, up = pkg('hipops'), pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(91)
    w = torch.randn(ci, co, 3, 5, 5, generator=gen)
    k = up._corr_weight(w, dtype)                                               # [75, ci, co] (CPU: flip + permute)
    rows = k.reshape(75 * ci, co)
    if kind == 'l0':
        _, _, lo, hi = up._layer0_z4_plan(ci, 'cpu')
        want = torch.cat([rows.index_select(0, lo), rows.index_select(0, hi)], 1)
        got = hip.convt_weight_forward_blocks(w.to(DEV), up._block_offsets('l0', ci, co, torch.device(DEV)),
                                              torch.full((50 * ci, 2 * co), 7.0, dtype=dtype, device=DEV), ci, co)
        assert torch.equal(got.cpu(), want)
        return
    plan, kt, total_rows, _, _ = up._layer_plan_z4(ci, 'cpu')
    big = torch.cat([rows, torch.full((total_rows - 75 * ci, co), 7.0, dtype=dtype)])     # constant rows: a marker value
    want = torch.cat([big.index_select(0, plan[cls][4]).view(-1, 2 * co) for cls in up._CLASSES])
    stack = torch.full(want.shape, 7.0, dtype=dtype, device=DEV)
    hip.convt_weight_forward_blocks(w.to(DEV), up._block_offsets('lat', ci, co, torch.device(DEV)), stack, ci, co)
    assert torch.equal(stack.cpu(), want)
    block_rows, tap_slot, const_rows, const_src = up._stack_tables_z4(ci, torch.device(DEV))
    # every constant row of the stacked matrix is in const_rows, and nothing else is
    marks = (stack.view(-1, co) == 7.0).all(1).nonzero().squeeze(1)
    assert torch.equal(marks.sort().values, const_rows.sort().values)
    x = torch.randn(ci, generator=gen)
    v = hip.blocks_vec_forward(stack, block_rows, ci, x.to(DEV)).view(-1, co).index_select(0, tap_slot)
    want_v = torch.einsum('c,tcd->td', x.double(), k.double())
    assert close(v.cpu(), want_v, atol=1e-4 * float(want_v.abs().max()), rtol=1e-5)
    dv = torch.randn(75, co, generator=gen)
    dv2 = torch.zeros(100, co, device=DEV).index_copy_(0, tap_slot, dv.to(DEV)).view(50, 2 * co)
    dx = hip.blocks_vec_backward(stack, block_rows, ci, dv2)
    want_dx = torch.einsum('tcd,td->c', k.double(), dv.double())
    assert close(dx.cpu(), want_dx, atol=1e-4 * float(want_dx.abs().max()), rtol=1e-5)


@pytest.mark.parametrize('M,K,lda,c0,N,splits', [(450, 38400, 38400, 0, 1536, None), (1800, 9536, 14464, 1696, 1536, None),
                                                  (450, 6304, 14464, 4928, 1536, 5), (77, 2048, 2048, 0, 264, 4), (300, 4096, 4096, 0, 1536, 64)])
def test_gemm_nn_split_over_k_for_skinny_operands(M, K, lda, c0, N, splits):
    """ver_gemm_nn_splitk: the forward product of a lattice layer at the reference's own batch point (vocc.py:222, one viewpoint
    per step: 450 / 1 800 rows) cut into K slices with fp32 partial tiles -- the library's choice of slices and explicit ones
    (uneven last slice, more slices than 512-column pieces), a column range of a wider tap matrix, results into a column
    range of a wider output; against the fp32 product rounded once, and against the one-pass kernel."""
    hip = pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(M + K)
    Af = torch.randn(M, lda, generator=gen).to(DEV).to(torch.bfloat16)
    W = (torch.randn(K, N, generator=gen) * 0.05).to(DEV).to(torch.bfloat16)
    b = torch.randn(N, generator=gen).to(DEV)
    A = Af[:, c0:c0 + K]
    want = A.float() @ W.float() + b
    if splits is None:
        assert hip.gemm_nn_splits(M, K, N) > 1
    wide = torch.full((M, N + 16), 7.0, device=DEV, dtype=torch.bfloat16)
    got = hip.gemm_nn(A, W, b, out=wide[:, 8:8 + N], splits=splits)
    from util import rel_l2
    assert rel_l2(got.float(), want) < 3e-3
    assert float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    assert float(wide[:, :8].min()) == 7.0 and float(wide[:, 8 + N:].min()) == 7.0
    one = hip.gemm_nn(A, W, b, splits=1)
    assert rel_l2(got.float(), one.float()) < 3e-3                       # (fp32 sums in another order, one rounding each)
    assert hip.gemm_nn_splits(345600, 14304, 1536) == 1                   # the tall products stay one pass


@pytest.mark.parametrize('layout,B,hc,wc,C,N', [(2, 3, 6, 4, 64, 256), (3, 2, 6, 8, 96, 200), (0, 2, 5, 3, 64, 128), (3, 5, 30, 30, 768, 1536),
                                                (2, 9, 15, 15, 768, 1536)])
def test_gemm_nn_taps_equals_gather_then_gemm(layout, B, hc, wc, C, N):
    """ver_gemm_nn_taps (implicit operand: the tap matrix of a Z = 4 lattice layer read straight from the lattice) against
    ver_lattice_gather + ver_gemm_nn on the same lattice, taps and weights: the same fragments reach the same MFMAs in the
    same order -- bit-exact; with the position table and the bias: one rounding of the same fp32 sum + table."""
    hip, ups = pkg('hipops'), pkg('dense_heads.upsample')
    gen = torch.Generator(device='cpu').manual_seed(97 + layout + B)
    plain = torch.randn(B, 4, hc, wc, C, generator=gen).bfloat16()
    e = ups._from_plain(plain, layout).contiguous().to(DEV)
    if layout == 0:
        taps = [(2 * j, bb - 2, cc - 2) for bb in range(5) for cc in range(5) for j in range(2)][:18]
    else:
        taps = [(2 * j, dy, dx) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for j in range(2)]
    nt = len(taps)
    offs = [t * C for t in range(nt)]
    m = B * 2 * hc * wc
    a = torch.empty(m, nt * C, dtype=torch.bfloat16, device=DEV)
    hip.lattice_gather(e, a, taps, offs, (hc, wc), layout, row_z=2)
    w = (torch.randn(nt * C, N, generator=gen) * 0.05).bfloat16().to(DEV)
    want = hip.gemm_nn(a, w, splits=1)
    got = hip.gemm_nn_taps(e, layout, (hc, wc), taps, w)
    assert torch.equal(got, want)
    # a subset of the taps (a parity class reads 8-12 of the 18 blocks) through a row range of the weights
    sub = taps[4:12]
    want = hip.gemm_nn(a[:, 4 * C:12 * C], w[4 * C:12 * C], splits=1)
    assert torch.equal(hip.gemm_nn_taps(e, layout, (hc, wc), sub, w[4 * C:12 * C]), want)
    rowpos = torch.randn(2 * hc * wc, N, generator=gen).to(DEV)
    bias = torch.randn(N, generator=gen).to(DEV)
    wide = torch.full((m, N + 16), 7.0, device=DEV, dtype=torch.bfloat16)
    got = hip.gemm_nn_taps(e, layout, (hc, wc), taps, w, rowpos=rowpos, bias=bias, out=wide[:, 8:8 + N])
    ref = a.float() @ w.float() + rowpos.repeat(B, 1) + bias
    from util import rel_l2
    assert rel_l2(got.float(), ref) < 3e-3
    assert float((got.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float(wide[:, :8].min()) == 7.0 and float(wide[:, 8 + N:].min()) == 7.0


@pytest.mark.parametrize('planar,B,hc', [(False, 3, 6), (True, 2, 8), (True, 4, 30)])
def test_gemm_nn_segments_reads_the_class_ranges_of_the_tap_matrix(planar, B, hc):
    """ver_gemm_nn_segments on the class layout of a Z = 4 lattice layer ([P00 | G1 | P10 | G2 | P11 | G3 | G4 | P01]: tap
    blocks with constant-pattern blocks between them) against ver_lattice_gather (with the pattern blocks) + ver_gemm_nn on
    every parity class's column range: the same operands in the same order -- bit-exact."""
    hip, ups = pkg('hipops'), pkg('dense_heads.upsample')
    C, N = 64, 256
    gen = torch.Generator(device='cpu').manual_seed(101 + hc)
    layout = ups.ZS_PLANAR_SPLIT if planar else ups.ZS_SPLIT
    plain = torch.randn(B, 4, hc, hc, C, generator=gen).bfloat16()
    e = ups._from_plain(plain, layout).contiguous().to(DEV)
    plan, kt, _, taps, offs = ups._layer_plan_z4(C, torch.device(DEV))
    m = B * 2 * hc * hc
    a = torch.full((m, kt), 7.0, dtype=torch.bfloat16, device=DEV)
    assert ups._gather_z4(e, layout, a, taps, offs, C, hc, hc, with_const=True)
    table, coffs = ups._const_rows_z4(C, hc, hc, torch.device(DEV), torch.bfloat16)
    for cls in ups._CLASSES:
        c0, c1 = plan[cls][:2]
        segs = ups._class_segments_z4(cls, C)
        w = (torch.randn(c1 - c0, N, generator=gen) * 0.1).bfloat16().to(DEV)
        want = hip.gemm_nn(a[:, c0:c1], w, splits=1)
        got = hip.gemm_nn_taps(e, layout, (hc, hc), segs, w, const_rows=table)
        assert torch.equal(got, want), cls


@pytest.mark.parametrize('planar,B,hc,C,splits', [(False, 3, 6, 64, 0), (True, 2, 8, 128, 3), (True, 3, 30, 768, 0), (False, 5, 15, 768, 0)])
def test_wgrad_tn_segments_equals_gather_then_wgrad(planar, B, hc, C, splits):
    """ver_wgrad_tn_segments (implicit tap matrix as the A of dW = A^T G) against ver_lattice_gather + ver_wgrad_tn on every
    parity class's column range, same chunking: the same slabs reach the same MFMAs -- bit-exact in fp32; and layer 1's form
    (plain source, 50 taps, no pattern blocks)."""
    hip, ups = pkg('hipops'), pkg('dense_heads.upsample')
    N = 256
    gen = torch.Generator(device='cpu').manual_seed(103 + hc)
    layout = ups.ZS_PLANAR_SPLIT if planar else ups.ZS_SPLIT
    plain = torch.randn(B, 4, hc, hc, C, generator=gen).bfloat16()
    e = ups._from_plain(plain, layout).contiguous().to(DEV)
    plan, kt, _, taps, offs = ups._layer_plan_z4(C, torch.device(DEV))
    m = B * 2 * hc * hc
    a = torch.full((m, kt), 7.0, dtype=torch.bfloat16, device=DEV)
    assert ups._gather_z4(e, layout, a, taps, offs, C, hc, hc, with_const=True)
    table, _ = ups._const_rows_z4(C, hc, hc, torch.device(DEV), torch.bfloat16)
    g = torch.randn(m, N, generator=gen).bfloat16().to(DEV)
    for cls in ups._CLASSES:
        c0, c1 = plan[cls][:2]
        segs = ups._class_segments_z4(cls, C)
        s_ = splits or hip.lib().ver_wgrad_tn_splits_ld(__import__('ctypes').c_long(m), c1 - c0, N, __import__('ctypes').c_long(kt))
        want = hip.wgrad_tn(a[:, c0:c1], g, out_dtype=torch.float32, splits=s_)
        got = hip.wgrad_tn_segments(e, layout, (hc, hc), segs, g, out_dtype=torch.float32, const_rows=table, splits=s_)
        assert torch.equal(got, want), cls
        stacked = torch.full((c1 - c0 + 16, N), 7.0, dtype=torch.bfloat16, device=DEV)
        hip.wgrad_tn_segments(e, layout, (hc, hc), segs, g, out=stacked[8:8 + c1 - c0], const_rows=table)
        from util import rel_l2
        assert rel_l2(stacked[8:8 + c1 - c0].float(), want) < 3e-3 and float(stacked[:8].min()) == 7.0 and float(stacked[8 + c1 - c0:].min()) == 7.0
    if C % 64 == 0 and not planar:
        taps0, offs0, _, _ = ups._layer0_z4_plan(C, torch.device(DEV))
        x = plain.contiguous().to(DEV)
        a0 = torch.empty(m, 50 * C, dtype=torch.bfloat16, device=DEV)
        ups._gather_z4(x, 0, a0, taps0, offs0, C, hc, hc)
        want = hip.wgrad_tn(a0, g, out_dtype=torch.float32, splits=2)
        got = hip.wgrad_tn_segments(x, 0, (hc, hc), taps0, g, out_dtype=torch.float32, splits=2)
        assert torch.equal(got, want)


def test_lattice_stack_with_implicit_operands_equals_the_explicit_tap_matrices(monkeypatch):
    """The three lattice layers (head:251-258 on the even lattice) at 110 viewpoints in bf16, where every product takes the
    implicit-operand kernels (ver_gemm_nn_segments, ver_wgrad_tn_segments, ver_gemm_nn_planes for d(input)), against the same
    layers on explicit tap matrices (VER_IMPLICIT_TAPS=0: gather, ver_gemm_nn / library, scatter): the lattice out of the
    forward and every weight / bias gradient bit for bit (same operands, same order), d(input) within bf16 rounding (one
    fp32 sum over all classes and taps instead of bf16 partial sums per class)."""
    ups, hip = pkg('dense_heads.upsample'), pkg('hipops')
    gen = torch.Generator(device='cpu').manual_seed(131)
    B, C = 110, 768
    x0 = (torch.randn(B, C, 4, 15, 15, generator=gen) * 0.5).to(DEV)
    ws = [(torch.randn(C, C, 3, 5, 5, generator=gen) * 0.02).to(DEV) for _ in range(3)]
    bs = [(torch.randn(C, generator=gen) * 0.1).to(DEV) for _ in range(3)]
    res = {}
    calls = []
    real = hip.gemm_nn_taps
    monkeypatch.setattr(hip, 'gemm_nn_taps', lambda *a, **k: (calls.append(k.get('timer_class', 'fwd')), real(*a, **k))[1])
    for mode in ('implicit', 'explicit'):
        monkeypatch.setattr(ups, '_IMPLICIT_TAPS', mode == 'implicit')
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        with torch.autocast('cuda', dtype=torch.bfloat16):
            e, _ = ups.upsample_lattice(x, w, b)
        gg = torch.randn(e.shape, generator=torch.Generator(device='cpu').manual_seed(5)).bfloat16().to(DEV)
        e.backward(gg)
        res[mode] = (e.detach(), x.grad, [t.grad for t in w], [t.grad for t in b])
    assert calls.count('fwd') == 9 and calls.count('head_gemm_dgrad') == 3                  # (all in the implicit pass)
    (e_i, dx_i, dw_i, db_i), (e_e, dx_e, dw_e, db_e) = res['implicit'], res['explicit']
    assert torch.equal(e_i, e_e)
    from util import rel_l2
    assert torch.equal(dw_i[2], dw_e[2]) and torch.equal(db_i[2], db_e[2])      # the last layer: same gradient in, bit for bit
    for a, c in zip(dw_i[:2] + db_i[:2] + [dx_i], dw_e[:2] + db_e[:2] + [dx_e]):
        # layers 1-2 see the d(input) of the layer above: ONE fp32 sum over all classes and taps (implicit) against bf16
        # partial sums per class (explicit) -- two bf16 evaluations apart (measured 5e-3 on the first layer's weight)
        assert rel_l2(a, c) < 1e-2, rel_l2(a, c)


@pytest.mark.parametrize('planar,B,hc', [(False, 3, 6), (True, 2, 8)])
def test_implicit_input_gradient_of_a_lattice_layer_against_fp32(planar, B, hc):
    """d(input) of a class-stacked lattice layer as gather-form products over the four class planes of the output gradient
    (dense_heads/upsample.py::_dgrad_implicit on ver_gemm_nn_planes) against the definition in fp32 -- d(tap matrix) =
    sum over classes of g_class W_class^T, scattered back by the adjoint of the gather: one bf16 rounding of the same sums,
    and no further from them than the explicit bf16 path (which rounds d(tap matrix) after every class)."""
    ups = pkg('dense_heads.upsample')
    ci = co = 64
    gen = torch.Generator(device='cpu').manual_seed(137 + hc)
    layout = ups.ZS_PLANAR_SPLIT if planar else ups.ZS_SPLIT
    plan, kt, _, taps, offs = ups._layer_plan_z4(ci, torch.device(DEV))
    m = B * 2 * hc * hc
    rows = sum(plan[cls][1] - plan[cls][0] for cls in ups._CLASSES)
    stack = (torch.randn(rows, 2 * co, generator=gen) * 0.1).bfloat16().to(DEV)
    g = torch.randn(4, m, 2 * co, generator=gen).bfloat16().to(DEV)
    cr = ups._class_rows_z4(ci)
    e_shape = (4, B, 2, hc // 2, hc // 2, 2, ci) if planar else (B, 2, hc, hc, 2, ci)
    d_a32 = torch.zeros(m, kt, device=DEV)
    d_a16 = torch.zeros(m, kt, device=DEV, dtype=torch.bfloat16)
    for p, cls in enumerate(ups._CLASSES):
        c0, c1 = plan[cls][:2]
        w = stack[cr[cls][0]:cr[cls][0] + c1 - c0]
        d_a32[:, c0:c1] += g[p].float() @ w.float().t()
        d_a16[:, c0:c1] = (d_a16[:, c0:c1].float() + g[p].float() @ w.float().t()).bfloat16()
    want = ups._scatter_z4(d_a32, layout, e_shape, taps, offs, ci, hc, hc)
    explicit = ups._scatter_z4(d_a16, layout, e_shape, taps, offs, ci, hc, hc)
    got = ups._dgrad_implicit('lat', g, stack, B, hc, hc, ci, co)
    if planar:
        got = got.view(B, 2, hc // 2, 2, hc // 2, 2, 2, ci).permute(3, 5, 0, 1, 2, 4, 6, 7).reshape(e_shape)
    from util import rel_l2
    assert got.dtype == torch.bfloat16 and rel_l2(got.float(), want) < 3e-3, rel_l2(got.float(), want)
    assert float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    assert rel_l2(got.float(), want) <= rel_l2(explicit.float(), want) * 1.05
