"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/ver_ops.h declares; the Python side fails loudly without a GPU."""
import ctypes
import numpy as np
import os
import re

import pytest
import torch

from util import ROOT, pkg


def _declared():
    text = open(os.path.join(ROOT, 'include', 'ver_ops.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ver_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for need in ('ver_msda_forward', 'ver_msda_backward', 'ver_project_points',
                 'ver_hits_from_mask', 'ver_sca_forward', 'ver_sca_backward', 'ver_last_error',
                 'ver_abi_version'):
        assert need in names


def test_library_exports_every_declared_symbol():
    hip = pkg('hipops')
    build = pkg('csrc.build')
    build.build_hip(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(hip.LIB_PATH)
    handle = ctypes.CDLL(hip.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), name
    assert sorted(hip.SYMBOLS) == _declared()
    assert handle.ver_abi_version() == hip.ABI_VERSION


def test_argument_validation_without_gpu():
    """Null pointers / bad sizes are rejected before anything touches the device."""
    hip = pkg('hipops')
    lib = hip.lib()
    rc = lib.ver_msda_forward(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 64, None)
    assert rc == -1 and b'null' in lib.ver_last_error()
    rc = lib.ver_sca_forward(None, 0, None, None, None, None, None, None, None, None, None, 1, 6, 1, 1,
                             8, 96, 8, 14, 14, 0, None)
    assert rc == -1
    rc = lib.ver_sca_zero_rows(None, None, None, 2, 900, 768, None)
    assert rc == -1 and b'null' in lib.ver_last_error()
    assert lib.ver_sca_zero_rows(None, None, None, 0, 900, 768, None) == 0
    assert lib.ver_sca_zero_rows(None, None, None, 2, 900, 770, None) == -2


def test_argument_validation_of_the_head_entry_points():
    """Same for the lattice / MLP / loss entry points: unsupported widths and null pointers come
    back as error codes with a message, nothing is launched."""
    hip = pkg('hipops')
    lib = hip.lib()
    lib.ver_occ_mlp_image_bytes.restype = ctypes.c_long
    assert lib.ver_occ_mlp_image_bytes() == 140 * 1024 and lib.ver_occ_mlp_vector_floats() == 6 * 128 + 16
    buf = (ctypes.c_float * 16)()
    rc = lib.ver_occ_mlp_forward(None, buf, buf, None, ctypes.c_long(4), 64, 16, ctypes.c_float(1e-5), 1, None)
    assert rc == -2 and b'width' in lib.ver_last_error()                  # built for 128 / 16
    rc = lib.ver_occ_mlp_forward(None, buf, buf, None, ctypes.c_long(4), 128, 16, ctypes.c_float(1e-5), 1, None)
    assert rc == -1 and b'null' in lib.ver_last_error()
    assert lib.ver_occ_mlp_forward(None, buf, buf, None, ctypes.c_long(0), 128, 16, ctypes.c_float(1e-5), 1, None) == 0
    rc = lib.ver_focal_loss_forward(None, None, None, ctypes.c_long(8), 10, ctypes.c_float(2), ctypes.c_float(.25), 0, None, None)
    assert rc == -2 and b'multiple of 8' in lib.ver_last_error()
    assert lib.ver_focal_loss_blocks(ctypes.c_long(0), 16) == 1
    taps = (ctypes.c_int * 3)(0, 0, 0)
    offs = (ctypes.c_long * 1)(8)
    rc = lib.ver_lattice_gather(buf, buf, taps, offs, ctypes.c_long(16), 1, 1, 1, 1, 3, 2, 8, 1, 0, None, None, 0, 0, None)
    assert rc == -1 and b'even' in lib.ver_last_error()                   # planar needs even H, W
    rc = lib.ver_lattice_gather(buf, buf, taps, offs, ctypes.c_long(12), 1, 1, 1, 1, 2, 2, 8, 0, 0, None, None, 0, 0, None)
    assert rc == -1 and b'outside the row' in lib.ver_last_error()        # tap block past the row end
    rc = lib.ver_lattice_gather(buf, buf, taps, offs, ctypes.c_long(16), 1, 1, 2, 3, 2, 2, 8, 2, 0, None, None, 0, 0, None)
    assert rc == -2 and b'z-split' in lib.ver_last_error()                # z-split layouts: 4 z-layers
    rc = lib.ver_lattice_transpose(buf, buf, ctypes.c_long(3), 1, 1, 2, 2, 8, 0, 1, 0, None)
    assert rc == -1 and b'stride' in lib.ver_last_error()
    assert lib.ver_convt_weight_forward(None, None, ctypes.c_long(0), 1, None) == 0
    rc = lib.ver_ln_relu_forward(buf, buf, buf, buf, buf, buf, ctypes.c_long(2), 96, ctypes.c_float(1e-5), 0, None)
    assert rc == -2


def test_argument_validation_of_the_gemm_entry_points():
    """ver_wgrad_tn / ver_gemm_nn / the *_stats MLP entries (ABI 24), ver_convt_weight_backward_blocks (ABI 25): sizes, alignment, row pitches, workspace size and the
    split arithmetic are checked on the host before anything is launched; the split chooser and the workspace size are
    pure host functions."""
    hip = pkg('hipops')
    lib = hip.lib()
    lib.ver_wgrad_tn_workspace.restype = ctypes.c_long
    L = ctypes.c_long
    buf = (ctypes.c_float * 64)()
    # the chunk count of the shapes of the 192-viewpoint step (cost model of csrc/ver_wgrad.hip) and its bounds
    assert lib.ver_wgrad_tn_splits(L(345600), 14304, 1536) == 8          # one chunk per XCD
    assert lib.ver_wgrad_tn_splits(L(1800), 14304, 1536) == 1            # one viewpoint: no fp32 partials
    assert lib.ver_wgrad_tn_splits(L(0), 8, 8) == 1
    for m, ka, n in ((345600, 14304, 1536), (552960, 4480, 832), (14400, 14304, 1536), (96768000, 128, 128), (17, 40, 8)):
        s = lib.ver_wgrad_tn_splits(L(m), ka, n)
        assert 1 <= s <= 65536 and m / s <= 45000 + 1, (m, ka, n, s)  # a chunk stays inside a 32-bit buffer offset
        assert lib.ver_wgrad_tn_workspace(L(m), ka, n, 0) == s * ka * n * 4
    assert lib.ver_wgrad_tn_workspace(L(1000), 256, 128, 3) == 3 * 256 * 128 * 4
    args = lambda a, g, out, ws, m=64, ka=32, n=32, lda=32, ldg=32, ldo=32, dt=1, sp=0, fl=0, wsb=1 << 20: (
        a, L(lda), g, L(ldg), L(m), ka, n, out, L(ldo), dt, sp, fl, ws, L(wsb), None)
    rc = lib.ver_wgrad_tn(*args(None, None, buf, buf))
    assert rc == -1 and b'null' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, ka=0))
    assert rc == -1 and b'bad sizes' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, lda=16))
    assert rc == -1 and b'pitch' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, lda=36, ldg=36))
    assert rc == -2 and b'multiples of 8' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, n=30))
    assert rc == -2 and b'multiples of 4' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, dt=7))
    assert rc == -1 and b'out_dtype' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, wsb=16))
    assert rc == -1 and b'workspace' in lib.ver_last_error()
    rc = lib.ver_wgrad_tn(*args(buf, buf, buf, buf, m=1 << 40, lda=40000, ka=40000, sp=1, wsb=1 << 40))
    assert rc == -2 and b'4-GiB' in lib.ver_last_error()                  # one chunk of 2^40 rows
    g = lambda a, w, c, m=64, k=64, n=32, lda=64, ldw=32, ldc=32, fl=0: (a, L(lda), w, L(ldw), None, c, L(ldc), L(m), k, n, fl, None)
    assert lib.ver_gemm_nn(*g(None, None, None, m=0)) == 0
    rc = lib.ver_gemm_nn(*g(None, buf, buf))
    assert rc == -1 and b'null' in lib.ver_last_error()
    rc = lib.ver_gemm_nn(*g(buf, buf, buf, k=48, lda=48))
    assert rc == -2 and b'multiple of 32' in lib.ver_last_error()
    rc = lib.ver_gemm_nn(*g(buf, buf, buf, ldw=16))
    assert rc == -1 and b'pitch' in lib.ver_last_error()
    rc = lib.ver_gemm_nn(*g(buf, buf, buf, fl=1))
    assert rc == -1 and b'flags' in lib.ver_last_error()
    # ver_convt_weight_backward_blocks (ABI 25): sizes, dtype and the prev_bias / grad_v pairing before any launch
    blk = lambda src, off, pb, dv, gw, ci=8, co=8, ld=16, dt=1: (src, off, L(ld), pb, dv, gw, ci, co, dt, None)
    assert lib.ver_convt_weight_backward_blocks(*blk(None, None, None, None, None, ci=0)) == 0
    rc = lib.ver_convt_weight_backward_blocks(*blk(buf, buf, None, None, buf, ld=4))
    assert rc == -1 and b'bad sizes' in lib.ver_last_error()
    rc = lib.ver_convt_weight_backward_blocks(*blk(buf, buf, None, None, buf, dt=5))
    assert rc == -1 and b'dtype' in lib.ver_last_error()
    rc = lib.ver_convt_weight_backward_blocks(*blk(buf, buf, buf, None, buf))
    assert rc == -1 and b'come together' in lib.ver_last_error()
    rc = lib.ver_convt_weight_backward_blocks(*blk(None, buf, None, None, buf))
    assert rc == -1 and b'null' in lib.ver_last_error()
    # ver_lattice_rows (ABI 26): the periodic run tables are checked on the host (order, coverage, alignment, row count)
    I, LL = ctypes.c_int * 2, ctypes.c_long * 2
    rows = lambda cl, buf_, off=(0, 8), ln=(8, 8), base=(0, 4096), pitch=(32, 32), nrows=(4, 4), quarter=64, period=16, B=1, dt=1, C=8, W=4: (
        cl, buf_, L(quarter), period, 2, I(*off), I(*ln), LL(*base), I(*pitch), I(*nrows), B, 2, 4, W, C, 0, 1, dt, None)
    assert lib.ver_lattice_rows(*rows(None, None, B=0)) == 0                 # C*Z*H*W = 8*2*4*4 = 256 = 4 quarters of 64
    rc = lib.ver_lattice_rows(*rows(buf, buf, dt=0))
    assert rc == -2 and b'bf16' in lib.ver_last_error()
    rc = lib.ver_lattice_rows(*rows(buf, buf, off=(0, 12)))
    assert rc == -1 and b'segment 1' in lib.ver_last_error()
    rc = lib.ver_lattice_rows(*rows(buf, buf, ln=(8, 4)))
    assert rc == -1 and b'cover' in lib.ver_last_error()
    rc = lib.ver_lattice_rows(*rows(buf, buf, nrows=(4, 3)))
    assert rc == -1 and b'segment 1' in lib.ver_last_error()
    rc = lib.ver_lattice_rows(*rows(buf, buf, quarter=60))
    assert rc == -1 and b'quarter' in lib.ver_last_error()
    rc = lib.ver_lattice_rows(*rows(None, buf))
    assert rc == -1 and b'null' in lib.ver_last_error()
    # ver_focal_loss_forward_grad_u8 (ABI 28): byte labels hold at most 254 classes
    rc = lib.ver_focal_loss_forward_grad_u8(buf, buf, buf, buf, L(8), 256, ctypes.c_float(2.0), ctypes.c_float(0.25), 0, None, None)
    assert rc == -1 and b'byte labels' in lib.ver_last_error()
    # ver_clip_adamw_step (ABI 27)
    F = ctypes.c_float
    adam = lambda tab, n=1, chunks=1, chunk=1024, step=1, lr=1e-3, b1=0.9: (
        tab, tab, tab, tab, n, chunks, chunk, tab, None, F(1.0), F(lr), F(b1), F(0.999), F(1e-8), F(0.01), L(step), None)
    assert lib.ver_clip_adamw_step(*adam(None, n=0, chunks=0)) == 0
    rc = lib.ver_clip_adamw_step(*adam(buf, chunk=1022))
    assert rc == -1 and b'bad sizes' in lib.ver_last_error()
    rc = lib.ver_clip_adamw_step(*adam(buf, step=0))
    assert rc == -1 and b'step 0' in lib.ver_last_error()
    rc = lib.ver_clip_adamw_step(*adam(buf, b1=1.0))
    assert rc == -1 and b'hyper-parameters' in lib.ver_last_error()
    rc = lib.ver_clip_adamw_step(*adam(None))
    assert rc == -1 and b'null' in lib.ver_last_error()
    # ver_clip_adamw_step_tensors (ABI 29): per-tensor hyper-parameters and device-side update counts
    adam_t = lambda tab, n=1, chunks=1, chunk=1024, hyper=buf, steps=buf: (tab, tab, tab, tab, hyper, steps, n, chunks, chunk, tab, None, F(1.0), None)
    assert lib.ver_clip_adamw_step_tensors(*adam_t(None, n=0, chunks=0)) == 0
    rc = lib.ver_clip_adamw_step_tensors(*adam_t(buf, chunk=1022))
    assert rc == -1 and b'bad sizes' in lib.ver_last_error()
    rc = lib.ver_clip_adamw_step_tensors(*adam_t(buf, steps=None))
    assert rc == -1 and b'null' in lib.ver_last_error()
    # the MLP entries with the saved statistics: the rstd pointer must be 8-byte aligned and N < 2^28
    off = ctypes.cast(ctypes.addressof(buf) + 4, ctypes.c_void_p)
    rc = lib.ver_occ_mlp_forward_stats(buf, buf, buf, buf, off, L(4), 128, 16, ctypes.c_float(1e-5), 2, None)
    assert rc == -2 and b'rstd' in lib.ver_last_error()
    assert lib.ver_occ_mlp_forward_stats(None, buf, buf, None, None, L(0), 128, 16, ctypes.c_float(1e-5), 2, None) == 0
    rc = lib.ver_occ_mlp_backward_fused_stats(None, None, None, buf, buf, None, None, buf, L(4), 128, 16, ctypes.c_float(1e-5), None, 2, None)
    assert rc == -1 and b'null' in lib.ver_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_ops_refuse_cpu_tensors():
    hip = pkg('hipops')
    v = torch.zeros(1, 4, 1, 8)
    with pytest.raises(RuntimeError, match='GPU'):
        hip.MultiScaleDeformableAttnFunction_fp32.apply(
            v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 1, 1, 1, 2),
            torch.zeros(1, 1, 1, 1, 1), 64)
    with pytest.raises(RuntimeError, match='GPU'):
        hip.project_points(torch.zeros(1, 6, 4, 4), torch.zeros(1, 3), [-1, -1, -1, 1, 1, 1], 2, 2, 2)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    hip = pkg('hipops')
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.setattr(hip, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(hip.HipLibraryError, match='no CPU/PyTorch fallback'):
        hip.lib()


def test_host_launchers_under_address_sanitizer():
    """SURVEY section 5 (sanitizers): the HOST side of the C ABI -- argument checks, error formatting, launch-shape
    arithmetic up to the first device call -- built with -fsanitize=address (device code untouched: GPU ASan is not
    available on this pool) and driven through the same calls as the validation tests above, in a child python with
    the ASan runtime preloaded.  Any heap / stack / global overflow or use-after-free in those paths aborts the child."""
    import subprocess
    import sys
    build = pkg('csrc.build')
    rt = build.asan_runtime()
    if rt is None:
        pytest.skip('no libclang_rt.asan in this ROCm install')
    lib = build.build_hip(verbose=False, asan=True)
    env = dict(os.environ, LD_PRELOAD=rt, VER_HIP_LIB=lib,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=66:verify_asan_link_order=0')
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider', os.path.abspath(__file__), '-k',
           'argument_validation or exports_every or missing_library']
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert 'AddressSanitizer' not in proc.stderr and 'AddressSanitizer' not in proc.stdout, proc.stderr[-3000:]
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert '5 passed' in proc.stdout, proc.stdout[-500:]


def test_magic_division_of_the_gather_launcher_is_exact_in_its_range():
    """csrc/ver_sca.hip `cs_magic` / `cs_div`: n // d == mulhi(n, floor(2^32 / d) + 1) for every unit index the launcher lets
    through (n < 2^24 units, wave * pairs < 2^28 for ncons <= 16) -- the bound the launcher's `cs_addr_ok` check relies on."""
    rng = np.random.default_rng(0)
    for d in list(range(2, 65)) + [96, 128, 255, 256]:
        m = (1 << 32) // d + 1
        lim = min(1 << 24, (1 << 32) // d)
        n = np.concatenate([np.arange(0, 4096), rng.integers(0, lim, 20000), np.array([lim - 1, lim - 2])]).astype(np.uint64)
        # multiples of d and their predecessors: where a truncated reciprocal would first go wrong
        k = rng.integers(1, max(2, lim // d), 2000).astype(np.uint64) * np.uint64(d)
        n = np.concatenate([n, k[k < lim], k[k < lim] - np.uint64(1)])
        assert np.array_equal((n * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), d
    for d in range(2, 17):                                  # the pair split: wave * TP < 2^28
        m = (1 << 32) // d + 1
        n = np.concatenate([rng.integers(0, 1 << 28, 50000), np.array([(1 << 28) - 1])]).astype(np.uint64)
        assert np.array_equal((n * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), d
