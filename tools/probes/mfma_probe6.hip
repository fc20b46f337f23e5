// Micro-probe 6: error-compensated 3 x f16 MFMA (v_mfma_f32_32x32x16_f16, XDL pipe) as a replacement of the fp32 MFMA in the
// GEMM2 tile loop:   w = w_hi + w_lo/2^11,  h = h_hi + h_lo/2^11   (hi = fp16(x), lo = fp16((x - hi) * 2^11))
//                    D = w_hi.h_hi + (w_hi.h_lo + w_lo.h_hi) / 2^11          (the lo.lo term, 2^-22 relative, is dropped)
// (1) numerics against an fp64 reference, next to the fp32 MFMA chain; (2) speed of the 8-wave / LDS-ring loop of probe5.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int KP = 80;            // K = 72 padded to 5 x 16
constexpr float SC = 2048.0f, ISC = 1.0f / 2048.0f;

// ---------------- numerics: one 32x32 tile, W [32][72], H [72][32] ----------------
__global__ void numerics(const float* W, const float* H, float* D32, float* D16) {
  const int lane = threadIdx.x, col = lane & 31, hh = lane >> 5;
  f32x16 acc = {0};
  for (int k2 = 0; k2 < 36; ++k2) acc = MFMA32(W[col * 72 + 2 * k2 + hh], H[(2 * k2 + hh) * 32 + col], acc);   // A[i=l&31][k=l>>5], B[k][j=l&31]
  f32x16 am = {0}, ac = {0};
  for (int s = 0; s < 5; ++s) {
    f16x8 ahi, alo, bhi, blo;
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * s + 8 * hh + i;
      const float a = k < 72 ? W[col * 72 + k] : 0.f, b = k < 72 ? H[k * 32 + col] : 0.f;
      const _Float16 ah = (_Float16)a, bh = (_Float16)b;
      ahi[i] = ah; alo[i] = (_Float16)((a - (float)ah) * SC);
      bhi[i] = bh; blo[i] = (_Float16)((b - (float)bh) * SC);
    }
    am = MFMA16(ahi, bhi, am);
    ac = MFMA16(ahi, blo, ac);
    ac = MFMA16(alo, bhi, ac);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
    D32[row * 32 + col] = acc[r];
    D16[row * 32 + col] = am[r] + ac[r] * ISC;
  }
}

// ---------------- speed: 8-wave workgroup, 2-stage LDS ring of pre-split tile records ----------------
constexpr int REC_H = 5 * 64 * 8;                  // halfs per hi (or lo) fragment set of a tile
constexpr int REC_BYTES = 2 * REC_H * 2 + 128;     // hi + lo + bias[2][16] floats = 10,368 B
template <int NV, int NW>
__global__ __launch_bounds__(64 * NW) void loop16(const float* in, const char* recs, float* out, int passes, int nt) {
  __shared__ __attribute__((aligned(16))) char ring[2][REC_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  f16x8 bhi[5], blo[5];
  for (int s = 0; s < 5; ++s)
    for (int i = 0; i < 8; ++i) { const float v = in[(lane * 5 + s * 8 + i + tid) & 1023]; const _Float16 h = (_Float16)v; bhi[s][i] = h; blo[s][i] = (_Float16)((v - (float)h) * SC); }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float hv[36];
  for (int i = 0; i < 36; ++i) hv[i] = in[(lane + i) & 1023];
  constexpr int N16 = REC_BYTES / 16;               // 648 x 16 B per record
  constexpr int PER = (N16 + 64 * NW - 1) / (64 * NW);
  float4 stg[PER];
  const char* base = recs + (size_t)(blockIdx.x & 3) * nt * REC_BYTES;
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < N16) *reinterpret_cast<float4*>(ring[k] + 16 * q) = *reinterpret_cast<const float4*>(base + (size_t)k * REC_BYTES + 16 * q); }
  __syncthreads();
  f16x8 a0h[5], a0l[5], a1h[5], a1l[5];
  f32x16 B0, B1;
#define FRAGS(AH, AL, BB, st)                                                                    \
  { const char* r_ = ring[st];                                                                   \
    _Pragma("unroll") for (int s = 0; s < 5; ++s) { AH[s] = *reinterpret_cast<const f16x8*>(r_ + (s * 64 + lane) * 16); AL[s] = *reinterpret_cast<const f16x8*>(r_ + REC_H * 2 + (s * 64 + lane) * 16); } \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const float4 v = *reinterpret_cast<const float4*>(r_ + 4 * REC_H + (hh * 16 + 4 * j) * 4); BB[4 * j] = v.x; BB[4 * j + 1] = v.y; BB[4 * j + 2] = v.z; BB[4 * j + 3] = v.w; } }
  FRAGS(a0h, a0l, B0, 0)
  const int total = passes * nt;
#define TILE(I, AH, AL, BC, NH, NL, BN)                                                          \
  {                                                                                              \
    const int t2 = ((I) + 2) % nt;                                                               \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < N16) stg[i] = *reinterpret_cast<const float4*>(base + (size_t)t2 * REC_BYTES + 16 * q); } \
    FRAGS(NH, NL, BN, ((I) + 1) & 1)                                                             \
    f32x16 Dm = BC, Dc = {0};                                                                    \
    _Pragma("unroll") for (int s = 0; s < 5; ++s) { Dm = MFMA16(AH[s], bhi[s], Dm); Dc = MFMA16(AH[s], blo[s], Dc); Dc = MFMA16(AL[s], bhi[s], Dc); } \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r & 7] = fmaf(fmaf(Dc[r], ISC, Dm[r]), hv[r], acc[r & 7]); \
    _Pragma("unroll") for (int k = 0; k < NV; ++k) acc[k & 7] = fmaf(acc[k & 7], hv[k % 36], hv[(k + 5) % 36]); \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < N16) *reinterpret_cast<float4*>(ring[(I) & 1] + 16 * q) = stg[i]; } \
    __syncthreads();                                                                             \
  }
  for (int i = 0; i < total; i += 2) {
    TILE(i, a0h, a0l, B0, a1h, a1l, B1)
    TILE(i + 1, a1h, a1l, B1, a0h, a0l, B0)
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 64 * NW + tid] = s;
}

int main() {
  // ---- numerics
  std::vector<float> W(32 * 72), H(72 * 32);
  srand(1);
  for (auto& v : W) v = ((rand() % 20001) / 10000.0f - 1.0f) * 0.12f;                 // radial-MLP-like weights
  for (auto& v : H) { const float u = (rand() % 20001) / 10000.0f - 1.0f; v = u > 0 ? u * 3.0f : 0.0f; }   // post-ReLU activations
  float *dW, *dH, *d32, *d16;
  (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&dH, H.size() * 4); (void)hipMalloc(&d32, 4096); (void)hipMalloc(&d16, 4096);
  (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(numerics, dim3(1), dim3(64), 0, 0, dW, dH, d32, d16);
  std::vector<float> r32(1024), r16(1024);
  (void)hipMemcpy(r32.data(), d32, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(r16.data(), d16, 4096, hipMemcpyDeviceToHost);
  double e32 = 0, e16 = 0, mx = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int k = 0; k < 72; ++k) ref += (double)W[i * 72 + k] * (double)H[k * 32 + j];
      e32 = fmax(e32, fabs(r32[i * 32 + j] - ref)); e16 = fmax(e16, fabs(r16[i * 32 + j] - ref)); mx = fmax(mx, fabs(ref));
    }
  printf("numerics (max abs err / max |D|):  fp32 MFMA chain %.3e   3 x f16 split %.3e   (|D|max = %.3f)\n", e32 / mx, e16 / mx, mx);
  // ---- speed
  const int nt = 66, passes = 40;
  const size_t rb = (size_t)4 * nt * REC_BYTES;
  char* recs; float *in, *out;
  (void)hipMalloc(&recs, rb); (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 8 * 64 * 4);
  std::vector<_Float16> hr(rb / 2);
  for (auto& v : hr) v = (_Float16)(((rand() % 2001) / 1000.0f - 1.0f) * 0.12f);
  (void)hipMemcpy(recs, hr.data(), rb, hipMemcpyHostToDevice);
  float hin[1024]; for (int i = 0; i < 1024; ++i) hin[i] = (rand() % 1000) / 500.f;
  (void)hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms;
#define RUN(NV)                                                                                                           \
  hipLaunchKernelGGL((loop16<NV, 8>), dim3(256), dim3(512), 0, 0, in, recs, out, 2, nt); (void)hipDeviceSynchronize();       \
  (void)hipEventRecord(a); hipLaunchKernelGGL((loop16<NV, 8>), dim3(256), dim3(512), 0, 0, in, recs, out, passes, nt);       \
  (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);                              \
  printf("3 x f16 loop, 8-wave WG, NV=%3d: %.3f ms  -> %.1f fp32-equivalent TFLOP/s (K=72 useful)\n", NV, ms,                \
         256.0 * 8 * passes * nt * 36 * 4096.0 / ms / 1e9);
  RUN(32) RUN(96)
  return 0;
}
