"""Build libddk.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m disco_diffdock_amd.build            # incremental
    python -m disco_diffdock_amd.build --force
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libddk.so')
SOURCES = ['ddk_capi.hip', 'k_conv.hip', 'k_tp.hip', 'k_graph.hip', 'k_heads.hip', 'k_se3.hip', 'model.hip', 'conf.hip', 'k_conv_x.hip', 'k_conv_x2.hip', 'k_ar.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-mllvm', '-amdgpu-mfma-vgpr-form', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']


# per-file flags.  k_conv_x: the SLP vectoriser packs the fp32 epilogue FMAs into v_pk_fma_f32 with a v_mov shuffle per operand pair (packed
# f32 VALU has no rate advantage on gfx950 and is an anti-lever next to MFMAs, MI355X_MICROARCH.md)
FILE_FLAGS = {'k_conv_x.hip': ['-fno-slp-vectorize'], 'k_conv_x2.hip': ['-fno-slp-vectorize']}
# sources that are #included by another source (besides the headers): part of that object's content stamp
INCLUDED_SOURCES = {'k_conv_x2.hip': ['k_conv_x.hip']}


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.sep not in c or os.path.exists(c):
            return c
    return 'hipcc'


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(path, digest):
    try:
        with open(path + '.stamp') as f:
            return f.read().strip() == digest
    except OSError:
        return False


def stamp_headers():
    """Every file an object's content stamp covers besides its own source: the csrc headers, the generated .inc asm statements
    and the two public ABI headers (a change of ddk_config or an enum must recompile every object)."""
    inc = os.path.normpath(os.path.join(HERE, '..', 'include'))
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h') or f.endswith('.inc')) \
        + [os.path.join(inc, 'ddk.h'), os.path.join(inc, 'ddk_debug.h')]


def build(force=False, verbose=True):
    """Incremental and CONTENT based (a stamp with the hash of the source, every header and the flags sits next to each object):
    file times do not survive the copy to the GPU box, and a rebuild there must only happen when something really changed."""
    headers = stamp_headers()
    objs, todo, relink = [], [], force or not os.path.exists(LIB)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        flags = FLAGS + FILE_FLAGS.get(src, [])
        dg = _digest([s] + [os.path.join(CSRC, f) for f in INCLUDED_SOURCES.get(src, [])] + headers, ' '.join(flags))
        if force or not os.path.exists(o) or not _stamp_ok(o, dg):
            todo.append((o, dg, [_hipcc()] + flags + ['-c', s, '-o', o]))

    def _compile(job):
        o, dg, cmd = job
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(o + '.stamp', 'w') as f:
            f.write(dg)

    if todo:      # the objects are independent: one hipcc per core, the slowest file (k_conv_x.hip) bounds the wall time
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(_compile, todo))
        relink = True
    ldg = _digest([o + '.stamp' for o in objs])
    if relink or not _stamp_ok(LIB, ldg):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(LIB + '.stamp', 'w') as f:
            f.write(ldg)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
