"""Development aid: build libddk variants with extra compiler flags / -D switches for k_conv.hip so that several kernel
experiments can be timed in ONE gpurun call:   python tools/build_variant.py NAME [extra hipcc flags for k_conv.hip ...]
-> disco_diffdock_amd/variants/libddk_NAME.so   (select with DDK_LIB=<path>; git-ignored, travels to the GPU box)."""
import os, subprocess, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
SRC = 'k_conv.hip'
if extra and extra[0].endswith('.hip'):     # optional: which kernel source the extra flags apply to
    SRC, extra = extra[0], extra[1:]
B.build(verbose=False)
vdir = os.path.join(B.HERE, 'variants')
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, f'k_conv_{name}.o')
subprocess.check_call([B._hipcc()] + B.FLAGS + extra + ['-c', os.path.join(B.CSRC, SRC), '-o', obj])
objs = [os.path.join(B.CSRC, s.replace('.hip', '.o')) for s in B.SOURCES if s != SRC] + [obj]
lib = os.path.join(vdir, f'libddk_{name}.so')
subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
print(lib)
