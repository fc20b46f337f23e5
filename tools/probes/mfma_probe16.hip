// Micro-probe 16 (round 5): probe 15's hand-placed half-burst with TWO waves per SIMD (512-thread workgroups), ONE 32-edge block per wave: a tile = one
// half-burst per wave whose shadows carry the epilogue of the wave's OWN previous tile (accumulators ping-pong), one barrier per tile.  Does the second
// wave of a SIMD cover the first one's VMEM / LDS issue stalls (k_conv_y.hip loses ~170 cycles per tile to them)?  Floor: 2 x 28 x 32 = 1792.
// (probe 15:) ONE wave per SIMD (256-thread workgroups), every wave owns
// TWO 32-edge column blocks a / b.  A tile = two half-bursts: HB_a = the 28 MFMAs of block a (one accumulator chain) with, hand-placed in their shadows,
// the epilogue of block b's PREVIOUS tile (bias, tensor-product FMAs), the fragment reads one K step ahead, half of the ring traffic; HB_b likewise
// with the epilogue of block a's CURRENT tile.  One s_barrier per tile, 4-stage ring (tile t reads stage t & 3, record t+2 is stored during tile t).
// Question: cycles per tile (floor 56 x 32 = 1792) with the real LDS / L2 traffic, the barrier and ~60 fillers per half-burst.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int TILE_BYTES = 13968, LIMB = 4608, BIAS_OFF = 3 * LIMB, DESC_OFF = BIAS_OFF + 128, FS = 100;

#define MF16Z(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(D) : "v"(a), "v"(b))
#define MF16(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "v"(b))
#define MF8(D, a, b) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "v"(b))
#define DSR128(v, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off))
#define DSR64(v, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off))
#define DSR2ST64(v, addr, o0, o1) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "i"(o0), "i"(o1))
#define DSW128(addr, v, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(off) : "memory")
#define BUFLD(v, voff, rsrc, soff) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define VMC(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FMAC(d, x, y) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))

struct HB { f16x8 hh[4], hm[4], hl[4], tmh, thl; f16x4 thi, tmid; };   // B operands of one block

// one half-burst: 28 MFMAs of block X on DX; epilogue of block Y (DY complete): DY += bias * bsc, accY[rq] += f0 . DY[4rq..]
// fa = stage base of X's tile + lane*16, ft = + lane*8; ea = stage base of Y's tile + hh*64 (bias); fy = Y's F row + the tile's offset
// na = stage base for the NEXT half-burst's first K step (+ lane*16); ca/cb: ring chunk registers (stored, then re-requested)
#define HALF_BURST(X, DX, DY, accY, bscY, fa, ft, ea, fy, na, C0, C1, w0addr, w1addr, sst, voff0, voff1, soff, EXTRA)                 \
  {                                                                                                                               \
    f16x8 a1h, a1m, a1l;                                                                                                          \
    f32x4 b0, b1, b2, b3, f0;                                                                                                     \
    f16x8 ahm, alh;                                                                                                               \
    MF16Z(DX, a0h, X.hl[0]);  DSR128(a1h, fa, 1024);            DSR128(b0, ea, BIAS_OFF);                                          \
    MF16(DX, a0l, X.hh[0]);   DSR128(a1m, fa, LIMB + 1024);     DSR128(b1, ea, BIAS_OFF + 16);                                     \
    MF16(DX, a0m, X.hm[0]);   DSR128(a1l, fa, 2 * LIMB + 1024); DSR128(b2, ea, BIAS_OFF + 32);                                     \
    MF16(DX, a0h, X.hm[0]);   DSR128(b3, ea, BIAS_OFF + 48);    DSR128(f0, fy, 0);                                                 \
    MF16(DX, a0m, X.hh[0]);   VMC(2); DSW128(w0addr, C0, 0);                                                                       \
    MF16(DX, a0h, X.hh[0]);   DSW128(w1addr, C1, 0);                                                                               \
    LGKM(4);                                                                                                                      \
    MF16(DX, a1h, X.hl[1]);   DSR128(a0h, fa, 2048);            FMAC(DY[0], b0.x, bscY); FMAC(DY[1], b0.y, bscY);                 \
    MF16(DX, a1l, X.hh[1]);   DSR128(a0m, fa, LIMB + 2048);     FMAC(DY[2], b0.z, bscY); FMAC(DY[3], b0.w, bscY);                 \
    MF16(DX, a1m, X.hm[1]);   DSR128(a0l, fa, 2 * LIMB + 2048); FMAC(DY[4], b1.x, bscY); FMAC(DY[5], b1.y, bscY);                 \
    MF16(DX, a1h, X.hm[1]);   BUFLD(C0, voff0, rsrc, soff);     FMAC(DY[6], b1.z, bscY); FMAC(DY[7], b1.w, bscY);                 \
    LGKM(3);                                                                                                                      \
    MF16(DX, a1m, X.hh[1]);   BUFLD(C1, voff1, rsrc, soff);     FMAC(DY[8], b2.x, bscY); FMAC(DY[9], b2.y, bscY); FMAC(DY[10], b2.z, bscY); \
    MF16(DX, a1h, X.hh[1]);   FMAC(DY[11], b2.w, bscY); FMAC(DY[12], b3.x, bscY); FMAC(DY[13], b3.y, bscY);                       \
    LGKM(0);                                                                                                                      \
    MF16(DX, a0h, X.hl[2]);   DSR128(a1h, fa, 3072);            FMAC(DY[14], b3.z, bscY); FMAC(DY[15], b3.w, bscY);               \
    MF16(DX, a0l, X.hh[2]);   DSR128(a1m, fa, LIMB + 3072);     FMAC(accY[0], f0.x, DY[0]); FMAC(accY[1], f0.x, DY[4]);           \
    MF16(DX, a0m, X.hm[2]);   DSR128(a1l, fa, 2 * LIMB + 3072); FMAC(accY[2], f0.x, DY[8]); FMAC(accY[3], f0.x, DY[12]);          \
    MF16(DX, a0h, X.hm[2]);   FMAC(accY[0], f0.y, DY[1]); FMAC(accY[1], f0.y, DY[5]); FMAC(accY[2], f0.y, DY[9]);                 \
    MF16(DX, a0m, X.hh[2]);   FMAC(accY[3], f0.y, DY[13]); FMAC(accY[0], f0.z, DY[2]); FMAC(accY[1], f0.z, DY[6]);                \
    MF16(DX, a0h, X.hh[2]);   FMAC(accY[2], f0.z, DY[10]); FMAC(accY[3], f0.z, DY[14]); FMAC(accY[0], f0.w, DY[3]);               \
    LGKM(0);                                                                                                                      \
    MF16(DX, a1h, X.hl[3]);   DSR2ST64(ahm, ft, 8, 17);         FMAC(accY[1], f0.w, DY[7]);                                        \
    MF16(DX, a1l, X.hh[3]);   DSR2ST64(alh, ft, 26, 8);         FMAC(accY[2], f0.w, DY[11]);                                       \
    MF16(DX, a1m, X.hm[3]);   DSR128(a0h, na, 0);               FMAC(accY[3], f0.w, DY[15]);                                       \
    MF16(DX, a1h, X.hm[3]);   DSR128(a0m, na, LIMB);                                                                               \
    MF16(DX, a1m, X.hh[3]);   DSR128(a0l, na, 2 * LIMB);                                                                           \
    MF16(DX, a1h, X.hh[3]);   EXTRA                                                                                                \
    LGKM(3);                                                                                                                      \
    MF16(DX, alh, X.thl);                                                                                                          \
    MF16(DX, ahm, X.tmh);                                                                                                          \
    { const f16x4 th_ = __builtin_shufflevector(ahm, ahm, 0, 1, 2, 3), tm_ = __builtin_shufflevector(ahm, ahm, 4, 5, 6, 7);        \
      MF8(DX, tm_, X.tmid);                                                                                                        \
      MF8(DX, th_, X.thi); }                                                                                                       \
  }

template <bool BARRIER, bool RING>
__global__ __launch_bounds__(512) void probe(const float* in, const char* w, float* out, unsigned* cyc, int tiles, int n_rec) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, el = lane & 31, hh = lane >> 5;
  char* ring = lds + 256 * FS * 4;
  HB A;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      A.hh[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023]; A.hm[s][i] = (_Float16)(in[(lane * 3 + s * 8 + i + 5) & 1023] * 0.001f); A.hl[s][i] = (_Float16)(in[(lane * 5 + s * 8 + i + 9) & 1023] * 1e-6f);
    }
  for (int i = 0; i < 8; ++i) { A.tmh[i] = (_Float16)in[(lane + i) & 1023]; A.thl[i] = (_Float16)in[(lane + i + 11) & 1023]; }
  for (int i = 0; i < 4; ++i) { A.thi[i] = (_Float16)in[(lane + i + 3) & 1023]; A.tmid[i] = (_Float16)(in[(lane + i + 17) & 1023] * 0.001f); }
  for (int i = tid; i < (256 * FS * 4 + 4 * TILE_BYTES) / 4; i += 512) reinterpret_cast<float*>(lds)[i] = in[i & 1023] * 0.01f;
  float accA[4];
  for (int k = 0; k < 4; ++k) accA[k] = 0.f;
  f32x16 Da, Db;
  for (int r = 0; r < 16; ++r) { Da[r] = 0.f; Db[r] = 0.f; }
  const float bscA = in[lane & 255];
  __syncthreads();
  const unsigned ring0 = (unsigned)(size_t)(ring - lds);     // LDS byte offset of the ring (dynamic LDS starts at 0 here)
  const unsigned fa_l = ring0 + lane * 16, ft_l = ring0 + lane * 8, ea_l = ring0 + hh * 64;
  const unsigned fyA = (unsigned)((wave * 32 + el) * FS * 4);
  // ring chunks: 873 16-B chunks per record, 256 threads: chunks tid, tid+256, tid+512, min(tid+768, 872)
  const unsigned ck0 = 16u * tid, ck1 = 16u * min(tid + 512, 872);
  i32x4 rsrc;
  { const unsigned long long p = (unsigned long long)w; rsrc[0] = (int)(unsigned)p; rsrc[1] = (int)(unsigned)(p >> 32); rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000; }
  i32x4 c0, c1;
  for (int i = 0; i < 4; ++i) { c0[i] = 0; c1[i] = 0; }
  f16x8 a0h, a0m, a0l;
  DSR128(a0h, fa_l, 0); DSR128(a0m, fa_l, LIMB); DSR128(a0l, fa_l, 2 * LIMB);
  LGKM(0);
  int soff = 0;
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    const unsigned st = (unsigned)((t & 3) * TILE_BYTES), stp = (unsigned)(((t + 3) & 3) * TILE_BYTES), stn = (unsigned)(((t + 1) & 3) * TILE_BYTES),
                   sts = (unsigned)(((t + 2) & 3) * TILE_BYTES);
    const unsigned fa = fa_l + st, ft = ft_l + st, fan = fa_l + stn, eaP = ea_l + stp, eaC = ea_l + st;
    const unsigned fo = (unsigned)((t * 4) & 63) * 4;
    const unsigned wa0 = ring0 + sts + ck0, wa1 = ring0 + sts + ck1;
    if (RING) {
      if (t & 1) { HALF_BURST(A, Db, Da, accA, bscA, fa, ft, eaP, fyA + fo, fan, c0, c1, wa0, wa1, sts, ck0, ck1, soff, ;) }
      else { HALF_BURST(A, Da, Db, accA, bscA, fa, ft, eaP, fyA + fo, fan, c0, c1, wa0, wa1, sts, ck0, ck1, soff, ;) }
    }
    soff += TILE_BYTES;
    if (soff >= n_rec * TILE_BYTES) soff = 0;
    if (BARRIER) { LGKM(0); __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += Da[r] + Db[r];
  for (int k = 0; k < 4; ++k) s += accA[k];
  s += (float)(c0[0] + c1[0]);
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool BARRIER, bool RING>
void run(const char* name, float* in, char* w, float* out, unsigned* cyc, int n_rec) {
  const int tiles = 1180, grid = 256;
  const int ldsb = 256 * FS * 4 + 4 * TILE_BYTES + 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<BARRIER, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((probe<BARRIER, RING>), dim3(grid), dim3(512), ldsb, 0, in, w, out, cyc, 40, n_rec);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<BARRIER, RING>), dim3(grid), dim3(512), ldsb, 0, in, w, out, cyc, tiles, n_rec);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[2048];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0, mx = 0;
  for (int i = 0; i < 2048; ++i) { m += h[i]; if (h[i] > mx) mx = h[i]; }
  m /= 2048;
  printf("%-60s mean %7.1f  max %7.1f cycles per tile of 256 edges (56 MFMAs per SIMD = 1792)%s\n", name, m / tiles, mx / tiles, e == hipSuccess ? "" : " ** ERROR **");
}

int main() {
  float *in, *out; unsigned* cyc; char* w;
  const int n_rec = 4 * 59 + 3;
  (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 4); (void)hipMalloc(&w, (size_t)n_rec * TILE_BYTES + 65536);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMemset(w, 0x11, (size_t)n_rec * TILE_BYTES + 65536);
  run<true, true>("software-pipelined tile loop, barrier per tile", in, w, out, cyc, n_rec);
  run<false, true>("... without the barrier (LDS races: timing only)", in, w, out, cyc, n_rec);
  return 0;
}
