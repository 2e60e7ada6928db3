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
SOURCES = ['ver_abi.hip', 'ver_msda.hip', 'ver_sca.hip', 'ver_lattice.hip', 'ver_mlp.hip', 'ver_loss.hip', 'ver_occ_mlp.hip', 'ver_layout.hip']
HEADERS = ['ver_common.h', os.path.join('..', '..', 'include', 'ver_ops.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-munsafe-fp-atomics',
         '-Wall', '-Wno-unused-function']


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
