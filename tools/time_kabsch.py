"""Time the device Kabsch (csrc/k_se3.hip: kabsch_block = centroids, fp64 covariance, Horn's 4x4 eigenvector by Jacobi on one lane) in isolation:
40 samples x 40 atoms, flex = a torsion-sized perturbation of rigid - the shape se3_update_kernel runs it on.  Run on the GPU box."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_diffdock_amd.tensor_layers import _shape_context   # noqa: E402

dev = torch.device('cuda', 0)
ctx = _shape_context(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
nb, n = 40, 40
A = (torch.randn(nb, n, 3, device=dev) * 4).contiguous()
Bp = (A + 0.3 * torch.randn(nb, n, 3, device=dev)).contiguous()
R = torch.empty((nb, 3, 3), device=dev)
t = torch.empty((nb, 3), device=dev)


def run():
    ctx._check(ctx.L.ddk_debug_kabsch(ctx.h, nb, n, C.c_void_p(A.data_ptr()), C.c_void_p(Bp.data_ptr()), C.c_void_p(R.data_ptr()), C.c_void_p(t.data_ptr()), st), 'k')


for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record()
torch.cuda.synchronize()
print('debug_kabsch_kernel (40 x 40 atoms): %.2f us per launch' % (e0.elapsed_time(e1) * 1000 / 200))
