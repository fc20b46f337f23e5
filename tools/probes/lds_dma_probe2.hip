// LDS-DMA probe 2: buffer_load_dwordx4 ... lds issued from inline asm (the compiler does not see an LDS write, so it adds no vmcnt wait in front of later LDS
// reads), destination above 64 KB, exec-masked second chunk, soffset in an SGPR
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const char* src, float* out, int n) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long p = (unsigned long long)src;
  u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((unsigned)p); rs.y = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32)); rs.z = 0x7fffffffu; rs.w = 0x00020000u;
  const unsigned base = 140000u;   // LDS byte address of the destination (dynamic LDS starts at 0 here)
  const unsigned m0a = __builtin_amdgcn_readfirstlane(base + 1024u * wave), m0b = __builtin_amdgcn_readfirstlane(base + 16u * (512 + 46 * wave));
  const unsigned vo0 = 16u * tid, vo1 = 16u * (512 + 46 * wave + lane);
  const int nl = wave == 7 ? 39 : 46;
  const unsigned long long mask_v = (1ull << nl) - 1ull;
  const unsigned long long mask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(mask_v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)mask_v);
  unsigned long long sv;
  for (int j = tid; j < 1024 * 4; j += 512) reinterpret_cast<float*>(lds + base)[j] = -1.0f;
  __syncthreads();
  int soff = __builtin_amdgcn_readfirstlane(n);   // record offset
  asm volatile("s_mov_b32 m0, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
               "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %7\n\ts_mov_b32 m0, %5\n\tbuffer_load_dwordx4 %6, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
               : "=&s"(sv) : "s"(m0a), "v"(vo0), "s"(rs), "s"(soff), "s"(m0b), "v"(vo1), "s"(mask) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = tid; j < 873 * 4; j += 512) out[j] = reinterpret_cast<float*>(lds + base)[j];
  for (int j = 873 * 4 + tid; j < 1024 * 4; j += 512) out[j] = reinterpret_cast<float*>(lds + base)[j];
}
int main() {
  const int rec = 13968, n = 3 * rec / 4;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 16384); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(o, 0, 16384);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 160 * 1024, 0, (const char*)d, o, rec);
  std::vector<float> r(4096); hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < rec / 4; ++i) if (r[i] != h[rec / 4 + i]) ++bad;
  int touched = 0; for (int i = rec / 4; i < 4096; ++i) if (r[i] != -1.0f) ++touched;
  printf("floats behind the record that were written: %d\n", touched);
  printf("record 1 through LDS DMA: %d of %d floats wrong; the float behind the record: %g (must not be %g)\n", bad, rec / 4, r[rec / 4], h[2 * rec / 4]);
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
