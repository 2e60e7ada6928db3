"""The regime the headline is measured in (bench.py: 192 viewpoints per step, bf16): everything else under ``-m gpu`` runs
the head on <= 3 viewpoints, where the own GEMM kernels are not even selected (dense_heads/upsample.py: from 14 000 rows on) and
no operand reaches 2^31 elements.  Here

* the head at B = 192 (96 x the two golden viewpoints) against the reference's own vectors (``head_vocc.npz``, reference
  head dense_heads/voxelformer_occupancy_head.py:554-580) and against the same step at B = 2, with ``ver_gemm_nn`` on and off;
* every kernel of the step whose operands pass 2^31 elements / 4 GiB at that batch, at the step's own shapes: sampled blocks
  against the fp32 product (GEMMs) or against the SAME kernel launched on the block alone (row-wise and per-viewpoint kernels:
  bit-exact), so that a 32-bit row x pitch product anywhere in a launcher or an address computation shows.

The operands are 10-50 GB each: the tests free the allocator's cache around themselves.  ``-m gpu``."""
import gc
import warnings

import numpy as np
import pytest
import torch

import cases
from util import golden, maxdiff, pkg, rel_l2

warnings.filterwarnings('ignore')
pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda'
B_BENCH = 192                      # bench.py's viewpoints per GPU and step


@pytest.fixture(autouse=True)
def _free_hbm():
    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


def _head(seed=7):
    torch.backends.cuda.matmul.allow_tf32 = False
    pkg()
    h = pkg('registry').build_head(cases.vocc_head_cfg()).eval()
    code_weights = h.code_weights.detach().clone()
    pkg('synthetic').load_seeded(h, seed)
    h.code_weights.data.copy_(code_weights)
    return h.to(DEV)


def _inputs(reps):
    """``reps`` x the two golden viewpoints (features seed 0, cameras seed 1: what head_vocc.npz was generated from)."""
    syn = pkg('synthetic')
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous()         # [6, 2, 196, 768]
    return (feats.repeat(1, reps, 1, 1).contiguous(), T(w2p).to(DEV).repeat(reps, 1, 1, 1).contiguous(),
            T(org).to(DEV).repeat(reps, 1).contiguous())


def test_head_at_192_viewpoints_matches_the_reference_vectors():
    """bf16 autocast forward of the lifting path at the bench's batch: the logits of ALL 192 viewpoints against the
    reference's fp32 vectors (rel. L2 < 1e-2, max |d| < 2.5e-2: the B = 1 bounds of test_head_gpu.py).  The batch is 96
    copies of two viewpoints, so every pair can also be held against the first one: measured (scratch/r06/copies_diag2.py)
    the fp32 path reproduces the gather output bit for bit in every copy and the layer outputs to 1e-6; under bf16 autocast
    pairs 0..88 are bit-identical and the last few differ -- the library's GEMMs accumulate their tail tiles in another
    order, a handful of bf16 roundings of the value / query projections flip, and three encoder layers + three lattice
    layers carry that to 4e-3 rel. L2 on the logits.  Bound: 3/4 of the pairs bit-identical, the rest within 6e-3."""
    g = golden('head_vocc')
    head = _head()
    feats, w2p, org = _inputs(B_BENCH // 2)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        emb, occ = head.lift(feats, None, world2pixel=w2p, origin=org)
    assert occ.shape == (B_BENCH, 504000, 16)
    want = [T(g['c3_b%d_occ' % b]).to(DEV) for b in range(2)]
    want_bev = [T(g['c3_b%d_bev' % b]).to(DEV) for b in range(2)]
    worst_rl2 = worst_max = 0.0
    for b in range(B_BENCH):
        got = occ[b, ::997].float()
        rl2, mx = rel_l2(got, want[b % 2]), maxdiff(got, want[b % 2])
        worst_rl2, worst_max = max(worst_rl2, rl2), max(worst_max, mx)
        assert rl2 < 1e-2 and mx < 2.5e-2, (b, rl2, mx)
        assert rel_l2(emb[b, ::7].float(), want_bev[b % 2]) < 1e-2, b
    first = occ[:2].float()
    norm = float(first.norm())
    dist = [float((occ[b0:b0 + 2].float() - first).norm()) / norm for b0 in range(2, B_BENCH, 2)]
    print('B = 192 forward vs head_vocc.npz: worst rel. L2 %.2e, worst max |d| %.2e; copies against the first pair: %d of %d '
          'bit-identical, worst rel. L2 %.2e' % (worst_rl2, worst_max, sum(d == 0.0 for d in dist), len(dist), max(dist)))
    assert max(dist) < 6e-3, max(dist)
    assert sum(d == 0.0 for d in dist) >= 3 * len(dist) // 4


def _train_step(head, feats, w2p, org, gt):
    """One forward + backward of bench.py's LiftTrainer (eval mode: dropout off, the training kernels otherwise --
    the fused MLP + focal-loss Function only needs grad mode) -> (loss, {name: fp32 gradient on the host})."""
    for p in head.parameters():
        p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        emb = head(feats, None, only_bev=True, world2pixel=w2p, origin=org)
        loss = head.occupancy_loss_from_volume(emb, gt)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.float().cpu() for k, p in head.named_parameters() if p.grad is not None}
    for p in head.parameters():
        p.grad = None
    return float(loss), grads


def test_training_step_at_192_viewpoints_equals_the_two_viewpoint_step(monkeypatch):
    """The bench's step (bf16 autocast, fused MLP + focal loss, ``ver_gemm_nn_segments`` / ``ver_wgrad_tn_segments`` on 345 600-row implicit operands,
    96.8 M-row MLP and loss launches) on 96 copies of two viewpoints and their labels: the loss is a mean over occupied
    voxels, so loss and EVERY parameter gradient must equal the two-viewpoint step's (which tests/test_head_gpu.py holds to
    the reference's gradient vectors) up to bf16 rounding of 96x larger sums -- and the same step with the library's GEMMs
    in place of the own kernels (VER_OWN_GEMM=0: explicit tap matrices, library forward) within rel. L2 5e-3 (two valid bf16 roundings of the lattices apart)."""
    ups = pkg('dense_heads.upsample')
    head = _head()
    lift = ('transformer.encoder.', 'transformer.level_embeds', 'transformer.cams_embeds', 'voxel_embedding.', 'up_sample.',
            'occ_proj.', 'occ_branches.')
    for k, p in head.named_parameters():
        p.requires_grad_(k.startswith(lift))
    gt2 = T(np.random.default_rng(3).integers(0, 17, size=(2, 504000))).to(DEV)
    f2, w2, o2 = _inputs(1)
    loss2, g2 = _train_step(head, f2, w2, o2, gt2)
    f, w, o = _inputs(B_BENCH // 2)
    gt = gt2.repeat(B_BENCH // 2, 1)
    calls = []
    hip = pkg('hipops')
    real, real_w = hip.gemm_nn_taps, hip.wgrad_tn_segments
    monkeypatch.setattr(hip, 'gemm_nn_taps', lambda *a, **k: (calls.append(('fwd', a[0].shape)), real(*a, **k))[1])
    monkeypatch.setattr(hip, 'wgrad_tn_segments', lambda *a, **k: (calls.append(('wgrad', a[0].shape)), real_w(*a, **k))[1])
    loss, gb = _train_step(head, f, w, o, gt)
    # the implicit-operand kernels did run: 1 + 4 + 4 forward products, one d(input) product per layer and 1 + 4 + 4 weight
    # gradients, layer 3's on the [4, 192, 2, 15, 15, 2, 768] lattice (345 600 rows)
    assert sum(c[0] == 'fwd' for c in calls) == 9 + 3 and sum(c[0] == 'wgrad' for c in calls) == 9, calls
    assert (4, B_BENCH, 2, 15, 15, 2, 768) in [tuple(c[1]) for c in calls]
    monkeypatch.setattr(ups, '_OWN_GEMM', False)
    n_before = len(calls)
    loss_lib, gl = _train_step(head, f, w, o, gt)
    assert len(calls) == n_before
    assert abs(loss - loss2) < 2e-3 * abs(loss2), (loss, loss2)
    assert abs(loss - loss_lib) < 1e-3 * abs(loss_lib), (loss, loss_lib)
    assert set(gb) == set(g2) == set(gl)
    rows = []
    for k in g2:
        n2, nb = float(g2[k].double().norm()), float(gb[k].double().norm())
        rows.append((k, n2, nb, rel_l2(gb[k], g2[k]) if n2 > 1e-9 else 0.0, rel_l2(gb[k], gl[k]) if n2 > 1e-9 else 0.0))
    worst = max((r[3], r[0]) for r in rows)
    worst_lib = max((r[4], r[0]) for r in rows)
    for k, n2, nb, r, rl in sorted(rows, key=lambda r: -r[3])[:5]:
        print('  %-90s |g| %.4e / %.4e  rel. L2 vs B = 2: %.2e, vs library GEMMs: %.2e' % (k, nb, n2, r, rl))
    for k, n2, nb, r, rl in rows:
        assert abs(nb - n2) <= 2e-2 * max(n2, 1e-6), (k, nb, n2)
        # whole gradient tensors, element-wise (rel. L2): two bf16 evaluations of the same sums apart -- the B = 2 step scales
        # d(logits) by 96x, every product rounds elsewhere.  Measured <= 4e-3 on the dense layers; the gather's geometry
        # gradients (sampling offsets, attention logits: differences of neighbouring bf16 values) are the noisiest at 3.4e-2
        assert r < (8e-2 if 'deformable_attention' in k else 3e-2), (k, r)
        # own kernels (implicit operands, d(input) as one fp32 sum over classes and taps) against explicit tap matrices and the
        # library (bf16 partial sums per class): two bf16 evaluations of d(volume) apart -- 2.5e-3 on the dense layers, up to
        # 7e-3 on the gather's geometry gradients downstream of it
        assert rl < (1.5e-2 if 'deformable_attention' in k else 6e-3), (k, rl)
    print('B = 192 step vs B = 2 step: loss %.6f / %.6f, worst gradient rel. L2 %.2e (%s); ver_gemm_nn vs library: loss %.6f, '
          'worst %.2e (%s)' % (loss, loss2, worst[0], worst[1], loss_lib, worst_lib[0], worst_lib[1]))


# ------------------------------------------------------------------------------------------ kernels at the step's shapes
M3, KT3, CO2 = B_BENCH * 2 * 30 * 30, 18 * 768 + 4 * 192, 1536          # layer 3: 345 600 rows, tap matrix pitch 14 592


def _randn_bf16(*shape, scale=1.0, seed=0):
    torch.manual_seed(seed)
    t = torch.empty(*shape, device=DEV, dtype=torch.bfloat16)
    step = max(1, (1 << 28) // max(1, t[0].numel()))
    for r0 in range(0, shape[0], step):                                 # (normal_ in row blocks: no fp32 temporary of the whole)
        t[r0:r0 + step].normal_(0.0, scale)
    return t


def _row_blocks(m, pitch, n=256):
    """Row blocks worth checking: both ends, the middle, and the rows around element offsets 2^31 and 2^32 of a
    [m, pitch] matrix (where a 32-bit row x pitch product wraps)."""
    starts = {0, m // 2, m - n}
    for lim in (1 << 31, 1 << 32):
        r = lim // pitch
        if n <= r < m - n:
            starts.add(r - n // 2)
    return sorted(starts)


@pytest.mark.parametrize('cls', [(0, 0), (1, 0), (0, 1)])
def test_gemm_nn_on_the_ten_gigabyte_tap_matrix(cls):
    """``ver_gemm_nn`` at layer 3's shape (M = 345 600, lda = 14 464: 5.0e9 elements, every row block past row 148 470 beyond
    2^31) on the column ranges of three parity classes as ``_LatticeLayerZ4`` passes them (class (0,0): columns 0..14 304;
    the others start inside the row), sampled 256-row blocks against the fp32 product."""
    plan = pkg('dense_heads.upsample')._layer_plan_z4(768, torch.device(DEV))[0]
    c0, c1 = plan[cls][:2]
    k = c1 - c0
    hip = pkg('hipops')
    assert k % 32 == 0
    a = _randn_bf16(M3, KT3, seed=1)
    w = _randn_bf16(k, CO2, scale=0.05, seed=2)
    bias = torch.randn(CO2, device=DEV)
    out = torch.full((M3, CO2), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.gemm_nn(a[:, c0:c0 + k], w, bias, out=out)
    torch.cuda.synchronize()
    for r0 in _row_blocks(M3, KT3):
        want = a[r0:r0 + 256, c0:c0 + k].float() @ w.float() + bias
        got = out[r0:r0 + 256].float()
        assert rel_l2(got, want) < 3e-3, (r0, rel_l2(got, want))
        assert float((got - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()), r0


def test_wgrad_tn_on_the_ten_gigabyte_tap_matrix():
    """``ver_wgrad_tn`` at layer 3, class (0,0): dW [14 304, 1 536] = A^T G over 345 600 rows of a 14 464-pitch matrix; 256
    sampled output rows (columns of A, incl. the first and last tile) against the fp32 product."""
    hip = pkg('hipops')
    ka = KT3 - 160
    a = _randn_bf16(M3, KT3, seed=3)
    g = _randn_bf16(M3, CO2, seed=4)
    got = hip.wgrad_tn(a[:, :ka], g, out_dtype=torch.float32)
    cols = torch.cat([torch.arange(0, 64), torch.arange(7000, 7064), torch.arange(ka - 128, ka)]).to(DEV)
    want = torch.zeros(len(cols), CO2, device=DEV, dtype=torch.float64)
    for r0 in range(0, M3, 43200):
        want += (a[r0:r0 + 43200].index_select(1, cols).float().t() @ g[r0:r0 + 43200].float()).double()
    err = float((got.index_select(0, cols).double() - want).abs().max())
    assert err < 3e-5 * float(want.abs().max()) * (M3 / 4096) ** 0.5, err


def _mlp_operands(gen):
    p = dict(g1=torch.randn(128, generator=gen) * 0.3 + 1.0, be1=torch.randn(128, generator=gen) * 0.3,
             w2=torch.randn(128, 128, generator=gen) * 0.12, b2=torch.randn(128, generator=gen) * 0.3,
             g2=torch.randn(128, generator=gen) * 0.3 + 1.0, be2=torch.randn(128, generator=gen) * 0.3,
             w3=torch.randn(16, 128, generator=gen) * 0.12, b3=torch.randn(16, generator=gen) * 0.3)
    p['w2'] = p['w2'] - p['w2'].mean(0, keepdim=True)                     # centred over the output axis, as the head passes it
    p['b2'] = p['b2'] - p['b2'].mean()
    return {k: v.to(DEV) for k, v in p.items()}


def test_occ_mlp_and_focal_loss_at_96_8_million_rows():
    """The occupancy term of the 192-viewpoint step as the head runs it (``occ_mlp_focal_loss_sum``: ver_occ_mlp_forward_stats
    -> ver_focal_loss_forward_grad_u8 -> ver_occ_mlp_backward_fused_stats on 96 768 000 rows: x and d(x) 24.8 GB each).  The
    kernels are row-wise: d(x) of sampled 64 K-row blocks (both ends, around 2^31 / 2^32 elements) must equal the SAME
    Function on the block alone bit for bit; loss sum and every parameter gradient against the sum over sixteen 6 M-row
    launches (fp32 accumulation order only)."""
    hip = pkg('hipops')
    n = B_BENCH * 504000
    gen = torch.Generator(device='cpu').manual_seed(71)
    p = _mlp_operands(gen)
    x = _randn_bf16(n, 128, scale=1.5, seed=5)
    torch.manual_seed(6)
    tgt = torch.randint(0, 17, (n,), device=DEV, dtype=torch.uint8)
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')

    def run(xs, ts):
        xs = xs.clone().requires_grad_(True)
        ps = [p[k].clone().requires_grad_(True) for k in keys]
        s = hip.occ_mlp_focal_loss_sum(xs, *ps, ts, 1e-5, 2.0, 0.25, centered=True)
        (s * 0.37).backward()
        return float(s), xs.grad, [q.grad.double() for q in ps]

    s_all, gx_all, pg_all = run(x, tgt)
    blk = 65536
    for r0 in _row_blocks(n, 128, blk):
        _, gx_b, _ = run(x[r0:r0 + blk], tgt[r0:r0 + blk])
        assert torch.equal(gx_all[r0:r0 + blk], gx_b), 'd(x) of rows %d.. differs from the block launched alone' % r0
    del gx_all
    s_sum, pg_sum = 0.0, [torch.zeros_like(q) for q in pg_all]
    step = n // 16
    for r0 in range(0, n, step):
        s_c, _, pg_c = run(x[r0:r0 + step], tgt[r0:r0 + step])
        s_sum += s_c
        pg_sum = [a + b for a, b in zip(pg_sum, pg_c)]
    assert abs(s_all - s_sum) < 1e-4 * abs(s_sum), (s_all, s_sum)
    for k, a, b in zip(keys, pg_all, pg_sum):
        assert float((a - b).norm()) < 2e-3 * float(b.norm()), (k, float((a - b).norm() / b.norm()))


def test_focal_loss_beyond_2_31_logits():
    """``ver_focal_loss_forward_grad_u8`` / ``ver_focal_loss_forward`` on [150 M, 16] bf16 logits (2.4e9 elements, 4.8 GB):
    in-place gradients of sampled blocks against the block launched alone, bit for bit; partial sums add up."""
    import ctypes
    hip = pkg('hipops')
    L = hip.lib()
    n = 150_000_000
    logits = _randn_bf16(n, 16, scale=2.0, seed=8)
    torch.manual_seed(9)
    lab = torch.randint(0, 17, (n,), device=DEV, dtype=torch.uint8)

    def run(x, t):
        x = x.clone()
        m = x.shape[0]
        partial = torch.zeros(L.ver_focal_loss_blocks(ctypes.c_long(m), 16), dtype=torch.float32, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        rc = L.ver_focal_loss_forward_grad_u8(hip._p(x), hip._p(t), hip._p(partial), hip._p(x), ctypes.c_long(m), 16,
                                              ctypes.c_float(2.0), ctypes.c_float(0.25), 1, hip._p(flag), hip._stream())
        assert rc == 0 and int(flag) == 0
        return float(partial.double().sum()), x
    s_all, g_all = run(logits, lab)
    blk = 1 << 20
    for r0 in _row_blocks(n, 16, blk):
        _, g_b = run(logits[r0:r0 + blk], lab[r0:r0 + blk])
        assert torch.equal(g_all[r0:r0 + blk], g_b), r0
    s_sum = sum(run(logits[r0:r0 + n // 10], lab[r0:r0 + n // 10])[0] for r0 in range(0, n, n // 10))
    assert abs(s_all - s_sum) < 1e-5 * abs(s_sum)
    # the int64-label, loss-only entry on the same logits
    want = hip.sigmoid_focal_loss_sum(logits, lab.long())
    assert abs(float(want) - s_all) < 1e-5 * abs(s_all)


@pytest.mark.parametrize('planar', [False, True])
def test_lattice_gather_scatter_at_192_viewpoints(planar):
    """``ver_lattice_gather`` / ``ver_lattice_scatter`` (Z = 4 form with the constant-pattern blocks) on the tap matrices of
    the 192-viewpoint step -- layer 2 (z-split source, 86 400 rows) and layer 3 (planar z-split source, 345 600 rows x 14 464
    columns = 10 GB): every viewpoint's rows depend on that viewpoint's lattice only, so viewpoints 0, 82, 83 (around 2^31
    elements of the layer-3 matrix), 165, 166 (around 2^32) and 191 must equal a one-viewpoint launch bit for bit."""
    ups = pkg('dense_heads.upsample')
    ci = 768
    hc = wc = 30 if planar else 15
    layout = ups.ZS_PLANAR_SPLIT if planar else ups.ZS_SPLIT
    _, kt, _, taps, offs = ups._layer_plan_z4(ci, torch.device(DEV))
    assert kt == KT3
    shape = (4, B_BENCH, 2, hc // 2, wc // 2, 2, ci) if planar else (B_BENCH, 2, hc, wc, 2, ci)
    e = _randn_bf16(*shape, seed=10) if not planar else _randn_bf16(4 * B_BENCH, 2, hc // 2, wc // 2, 2, ci, seed=10).view(shape)
    rows = 2 * hc * wc
    a = torch.full((B_BENCH * rows, kt), 7.0, device=DEV, dtype=torch.bfloat16)
    ups._gather_z4(e, layout, a, taps, offs, ci, hc, wc, with_const=True)
    one = lambda t, b: (t[:, b:b + 1] if planar else t[b:b + 1]).contiguous()
    picks = (0, 82, 83, 165, 166, B_BENCH - 1)
    for b in picks:
        a1 = torch.full((rows, kt), 7.0, device=DEV, dtype=torch.bfloat16)
        ups._gather_z4(one(e, b), layout, a1, taps, offs, ci, hc, wc, with_const=True)
        assert torch.equal(a[b * rows:(b + 1) * rows], a1), 'tap rows of viewpoint %d' % b
    assert not bool((a[:rows] == 7.0).all(0).any())                         # every column is written (taps + pattern blocks)
    del a
    d_a = _randn_bf16(B_BENCH * rows, kt, seed=11)
    d_e = ups._scatter_z4(d_a, layout, shape, taps, offs, ci, hc, wc)
    shape1 = (4, 1) + shape[2:] if planar else (1,) + shape[1:]
    for b in picks:
        d1 = ups._scatter_z4(d_a[b * rows:(b + 1) * rows], layout, shape1, taps, offs, ci, hc, wc)
        assert torch.equal(one(d_e, b), d1), 'lattice gradient of viewpoint %d' % b


def test_lattice_rows_at_192_viewpoints():
    """``ver_lattice_rows`` (the last lattice <-> the operand rows of all occ_proj pattern groups) on the 192-viewpoint
    step's buffers (4.2 GB lattice, rows buffer past 2^31 elements): viewpoints 0, 95, 96, 191 against one-viewpoint
    launches, both directions, bit for bit."""
    hip, ups, opl = pkg('hipops'), pkg('dense_heads.upsample'), pkg('dense_heads.occ_proj_lattice')
    C, Z, Hl, Wl = 768, 4, 60, 60
    layout = ups.ZS_PLANAR_SPLIT
    plan = opl.get_plan(C, Z, 2 * Hl, 2 * Wl, torch.device(DEV))
    assert plan is not None and plan.row_map is not None
    shape = (4, B_BENCH, 2, Hl // 2, Wl // 2, 2, C)
    src = _randn_bf16(4 * B_BENCH, 2, Hl // 2, Wl // 2, 2, C, seed=12).view(shape)
    row_map, spans, total = opl._row_map_for(plan, B_BENCH)
    row_map1, spans1, total1 = opl._row_map_for(plan, 1)
    buf = torch.full((total,), 7.0, dtype=torch.bfloat16, device=DEV)
    hip.lattice_rows(src, buf, row_map, (Hl, Wl), layout, True)
    picks = (0, 95, 96, B_BENCH - 1)
    for b in picks:
        buf1 = torch.full((total1,), 7.0, dtype=torch.bfloat16, device=DEV)
        hip.lattice_rows(src[:, b:b + 1].contiguous(), buf1, row_map1, (Hl, Wl), layout, True)
        for g, (b0, n), (c0, n1) in zip(plan.groups, spans, spans1):
            per = g.n_rows * g.k_aug
            assert n == B_BENCH * per and n1 == per
            assert torch.equal(buf[b0 + b * per:b0 + (b + 1) * per], buf1[c0:c0 + per]), (b, g.n_rows)
    rows = _randn_bf16(total, seed=13)
    got = torch.empty_like(src)
    hip.lattice_rows(got, rows, row_map, (Hl, Wl), layout, False)
    for b in picks:
        rows1 = torch.cat([rows[b0 + b * (n // B_BENCH):b0 + (b + 1) * (n // B_BENCH)] for b0, n in spans])
        got1 = torch.empty_like(src[:, b:b + 1].contiguous())
        hip.lattice_rows(got1, rows1, row_map1, (Hl, Wl), layout, False)
        assert torch.equal(got[:, b:b + 1], got1), b
