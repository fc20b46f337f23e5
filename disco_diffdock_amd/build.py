"""Build libddk.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m disco_diffdock_amd.build            # incremental
    python -m disco_diffdock_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libddk.so')
SOURCES = ['ddk_capi.hip', 'k_conv.hip', 'k_tp.hip', 'k_graph.hip', 'k_heads.hip', 'k_se3.hip', 'model.hip', 'conf.hip', 'k_conv_h.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-mllvm', '-amdgpu-mfma-vgpr-form', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.sep not in c or os.path.exists(c):
            return c
    return 'hipcc'


def build(force=False, verbose=True):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'ddk.h')]
    newest = max(os.path.getmtime(d) for d in deps)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < newest:
            cmd = [_hipcc()] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
