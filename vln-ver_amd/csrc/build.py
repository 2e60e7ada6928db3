"""Build libver_hip.so (the C-ABI library of include/ver_ops.h) for gfx950, in-tree.

    python vln-ver_amd/csrc/build.py          # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so lands next to the package so that it
travels with the tree; it is never installed into site-packages.
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libver_hip.so')
SOURCES = ['ver_abi.hip', 'ver_msda.hip', 'ver_sca.hip', 'ver_lattice.hip', 'ver_mlp.hip', 'ver_loss.hip', 'ver_occ_mlp.hip', 'ver_layout.hip', 'ver_addln.hip', 'ver_post.hip', 'ver_wgrad.hip', 'ver_gemm.hip', 'ver_optim.hip']
HEADERS = ['ver_common.h', os.path.join('..', '..', 'include', 'ver_ops.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-munsafe-fp-atomics',
         '-Wall', '-Wno-unused-function']


# Per-source device target features.  ver_occ_mlp.hip: no packed-fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) -- hipcc
# forms them from every float4 expression, and beside a matrix-core wave on the same SIMD one of them costs more issue time
# than the two scalar instructions it replaces (k_occ_mlp_bwd_ws 30.2 -> 28.2 ms over 96.8 M rows, k_occ_mlp_fwd 7.2 -> 6.75;
# scratch/r06/ubench/valu_cost.hip).  A function attribute would do it too, but then nothing un-attributed inlines into the
# kernel.  The host pass does not know the feature and says so on stderr: that one line is filtered below.
TARGET_FEATURES = {'ver_occ_mlp.hip': ['-packed-fp32-ops']}
_HOST_PASS_NOISE = 'is not a recognized feature for this target'

OBJ_DIR = os.path.join(HERE, 'build')
CFLAGS = [f for f in FLAGS if f != '-shared']


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale():
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return _newer(LIB, deps)


ASAN_LIB = os.path.join(PKG, 'libver_hip_asan.so')
ASAN_FLAGS = ['-O1', '-g', '-fsanitize=address', '-fno-gpu-sanitize', '-shared-libsan', '-fno-omit-frame-pointer']


def asan_runtime():
    """libclang_rt.asan of the ROCm clang (to LD_PRELOAD into the python that loads the ASan build)."""
    import glob
    hits = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    return hits[-1] if hits else None


def build_hip(force=False, verbose=True, defines=(), lib=None, asan=False):
    """Compile each source to csrc/build/<name>.o (only the stale ones) and link libver_hip.so.
    ``defines`` / ``lib``: experiment builds (scratch/) with extra -D flags into another .so.
    ``asan``: AddressSanitizer build of the HOST side (launchers, argument checks, error paths) into
    libver_hip_asan.so -- device code is compiled as usual (GPU ASan is not available on this pool)."""
    if asan:
        lib = lib or ASAN_LIB
    lib = lib or LIB
    if not force and not defines and lib == LIB and not stale():
        return LIB
    if asan and not force and not _newer(lib, [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]):
        return lib
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tag = ''.join('_' + d.replace('=', '-') for d in defines)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, procs = [], []
    for src in SOURCES:
        path = os.path.join(HERE, src)
        # experiment defines only rebuild the sources that mention them
        mytag = tag if (defines and any(d.split('=')[0] in open(path).read() for d in defines)) else ''
        if asan:
            mytag += '_asan'
        obj = os.path.join(OBJ_DIR, src.replace('.hip', mytag + '.o'))
        objs.append(obj)
        if force or _newer(obj, [path] + hdrs):
            cflags = [f for f in CFLAGS if f != '-O3'] + ASAN_FLAGS if asan else CFLAGS
            feats = [a for f in TARGET_FEATURES.get(src, ()) for a in ('-Xclang', '-target-feature', '-Xclang', f)]
            cmd = [hipcc] + cflags + feats + (['-D' + d for d in defines] if mytag.replace('_asan', '') else []) + ['-c', path, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            err = tempfile.TemporaryFile(mode='w+') if feats else None
            procs.append((cmd, subprocess.Popen(cmd, stderr=err), err))
    for cmd, pr, err in procs:
        rc = pr.wait()
        if err is not None:
            err.seek(0)
            sys.stderr.write(''.join(ln for ln in err if _HOST_PASS_NOISE not in ln))
            err.close()
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + (['-fsanitize=address', '-shared-libsan'] if asan else []) + objs + ['-o', lib]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    defs = tuple(a[2:] for a in sys.argv[1:] if a.startswith('-D'))
    out = [a[6:] for a in sys.argv[1:] if a.startswith('--lib=')]
    build_hip(force='--force' in sys.argv, defines=defs, lib=out[0] if out else None, asan='--asan' in sys.argv)
