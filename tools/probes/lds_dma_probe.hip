#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* src, float* out, int n) {
  extern __shared__ char lds[];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
  const int tid = threadIdx.x;
  if (tid < 200)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 1024 * (tid >> 6)), 16, tid * 16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[tid * 4 + j] = reinterpret_cast<float*>(lds)[tid * 4 + j];
}
int main() {
  const int n = 256 * 4;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(o, 0, n * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, (const char*)d, o, n * 4);
  std::vector<float> r(n); hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 800; ++i) if (r[i] != h[i]) ++bad;
  printf("bad (first 200 threads) %d ; r[800..803] = %g %g %g %g\n", bad, r[800], r[801], r[802], r[803]);
}
