"""Build libver_hip.so (the C-ABI library of include/ver_ops.h) for gfx950, in-tree.

    python vln-ver_amd/csrc/build.py          # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so lands next to the package so that it
travels with the tree; it is never installed into site-packages.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libver_hip.so')
SOURCES = ['ver_abi.hip', 'ver_msda.hip', 'ver_sca.hip', 'ver_lattice.hip', 'ver_mlp.hip', 'ver_loss.hip', 'ver_occ_mlp.hip', 'ver_layout.hip', 'ver_addln.hip', 'ver_post.hip', 'ver_wgrad.hip', 'ver_gemm.hip', 'ver_optim.hip']
HEADERS = ['ver_common.h', os.path.join('..', '..', 'include', 'ver_ops.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-munsafe-fp-atomics',
         '-Wall', '-Wno-unused-function']


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
            cmd = [hipcc] + cflags + (['-D' + d for d in defines] if mytag.replace('_asan', '') else []) + ['-c', path, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + (['-fsanitize=address', '-shared-libsan'] if asan else []) + objs + ['-o', lib]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    defs = tuple(a[2:] for a in sys.argv[1:] if a.startswith('-D'))
    out = [a[6:] for a in sys.argv[1:] if a.startswith('--lib=')]
    build_hip(force='--force' in sys.argv, defines=defs, lib=out[0] if out else None, asan='--asan' in sys.argv)
