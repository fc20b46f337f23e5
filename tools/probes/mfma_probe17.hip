// Micro-probe 17 (round 5): SPECIALISED WAVES.  512-thread workgroups, two waves per SIMD: waves 0-3 (one per SIMD) are PRODUCERS - they only run the matrix
// pipe: two 32-edge blocks a / b per wave, the six limb products of a K step issued for a and b alternately on the same A fragments (one fragment read serves
// both blocks), 54 MFMAs per tile, accumulators ping-pong between tiles so that the previous tile's 2 x 16 accumulator registers go to LDS in the shadows of the
// next tile's first MFMAs.  Waves 4-7 are CONSUMERS: ring traffic (the whole record: four 16-B chunks per lane), and per tile the epilogue of both blocks of their
// SIMD's producer read back from LDS (bias, 2 x 16 fmac + 2 x 16 tensor-product FMAs, every eighth tile ~60 more instructions as a flush).  One s_barrier per tile,
// which the producers reach in the middle of their burst.  Question: cycles per 256-edge tile against the floor 54 x 32 = 1728 (k_conv_x.hip: 2146, probe 15: 1940).
// Not a kernel: the LDS of the real thing would not fit (the F rows' 102 KB + ring 56 KB + 64 KB of accumulator buffers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_BYTES = 13968, LIMB = 4608, BIAS_OFF = 3 * LIMB;
constexpr int RING_BYTES = 4 * TILE_BYTES, DBUF = 4096, D_BYTES = 4 * 2 * 2 * DBUF, F_BYTES = 8192;

#define MF16Z(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(D) : "v"(a), "v"(b))
#define MF16(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "v"(b))
#define DSR128(v, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off))
#define DSW128(addr, v, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(off) : "memory")
#define BUFLD(v, voff, rsrc, soff) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define VMC(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FMAC(d, x, y) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define SUB4(D, i) __builtin_shufflevector(D, D, 4 * (i), 4 * (i) + 1, 4 * (i) + 2, 4 * (i) + 3)

struct HB { f16x8 hh[5], hm[5], hl[5]; };   // B operands of one block: four K steps + the packed tail as a fifth "step" of three products

// one K step of both blocks: six limb products each, a / b alternately on the same A fragments (ah, am, al); E0..E5: what rides in the shadows
#define STEP(FIRST, s, ah, am, al, E0, E1, E2, E3, E4, E5)                                                                 \
  if (FIRST) { MF16Z(Da, ah, A.hl[s]); } else { MF16(Da, ah, A.hl[s]); }                                                   \
  if (FIRST) { MF16Z(Db, ah, B.hl[s]); } else { MF16(Db, ah, B.hl[s]); }  E0                                               \
  MF16(Da, al, A.hh[s]); MF16(Db, al, B.hh[s]); E1                                                                         \
  MF16(Da, am, A.hm[s]); MF16(Db, am, B.hm[s]); E2                                                                         \
  MF16(Da, ah, A.hm[s]); MF16(Db, ah, B.hm[s]); E3                                                                         \
  MF16(Da, am, A.hh[s]); MF16(Db, am, B.hh[s]); E4                                                                         \
  MF16(Da, ah, A.hh[s]); MF16(Db, ah, B.hh[s]); E5

// a producer's tile: Da / Db = this tile's accumulators, Pa / Pb = the previous tile's (written to dprev while the first MFMAs run)
#define PRODUCER_TILE(XA_, XB_, Pa, Pb, dprev)                                                                               \
  {                                                                                                                        \
    f16x8 b_h, b_m, b_l;                                                                                                   \
    STEP(true, 0, a0h, a0m, a0l, DSR128(b_h, fa, 1024);, DSR128(b_m, fa, LIMB + 1024);, DSR128(b_l, fa, 2 * LIMB + 1024);, \
         DSW128(dprev, SUB4(Pa, 0), 0); DSW128(dprev, SUB4(Pa, 1), 1024);, DSW128(dprev, SUB4(Pa, 2), 2048); DSW128(dprev, SUB4(Pa, 3), 3072);, \
         DSW128(dprev, SUB4(Pb, 0), DBUF); DSW128(dprev, SUB4(Pb, 1), DBUF + 1024);)                                       \
    LGKM(6);                                                                                                               \
    STEP(false, 1, b_h, b_m, b_l, DSW128(dprev, SUB4(Pb, 2), DBUF + 2048); DSW128(dprev, SUB4(Pb, 3), DBUF + 3072);, DSR128(a0h, fa, 2048);, \
         DSR128(a0m, fa, LIMB + 2048);, DSR128(a0l, fa, 2 * LIMB + 2048);, ;, ;)                                           \
    LGKM(0);               /* the previous tile's accumulators are in LDS: meet the consumers (they are waiting) */        \
    if (BARRIER) __builtin_amdgcn_s_barrier();                                                                             \
    STEP(false, 2, a0h, a0m, a0l, DSR128(b_h, fa, 3072);, DSR128(b_m, fa, LIMB + 3072);, DSR128(b_l, fa, 2 * LIMB + 3072);, ;, ;, ;) \
    LGKM(0);                                                                                                               \
    STEP(false, 3, b_h, b_m, b_l, DSR128(a0h, fa, 4096);, DSR128(a0m, fa, LIMB + 4096);, DSR128(a0l, fa, 2 * LIMB + 4096 - 512);, ;, ;, ;) \
    LGKM(0);                                                                                                               \
    /* packed tail: three products per block */                                                                            \
    MF16(Da, a0h, A.hl[4]); MF16(Db, a0h, B.hl[4]); DSR128(b_h, fan, 0);                                                   \
    MF16(Da, a0m, A.hm[4]); MF16(Db, a0m, B.hm[4]); DSR128(b_m, fan, LIMB);                                                \
    MF16(Da, a0l, A.hh[4]); MF16(Db, a0l, B.hh[4]); DSR128(b_l, fan, 2 * LIMB);                                            \
    LGKM(0);                                                                                                               \
    a0h = b_h; a0m = b_m; a0l = b_l;                                                                                       \
  }

template <bool BARRIER>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const float* in, const char* w, float* out, unsigned* cyc, int tiles, int n_rec) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  for (int i = tid; i < (RING_BYTES + D_BYTES + F_BYTES) / 4; i += 512) reinterpret_cast<float*>(lds)[i] = in[i & 1023] * 0.01f;
  __syncthreads();
  const unsigned ring0 = 0u, d0 = (unsigned)RING_BYTES, f0a = (unsigned)(RING_BYTES + D_BYTES);
  float sink = 0.f;
  unsigned t0 = 0, t1 = 0;
  if (wave < 4) {
    // ---------------- producer ----------------
    HB A, B;
    for (int s = 0; s < 5; ++s)
      for (int i = 0; i < 8; ++i) {
        A.hh[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023]; A.hm[s][i] = (_Float16)(in[(lane * 3 + s * 8 + i + 5) & 1023] * 0.001f); A.hl[s][i] = (_Float16)(in[(lane * 5 + s * 8 + i + 9) & 1023] * 1e-6f);
        B.hh[s][i] = (_Float16)in[(lane * 11 + s * 8 + i) & 1023]; B.hm[s][i] = (_Float16)(in[(lane * 13 + s * 8 + i + 5) & 1023] * 0.001f); B.hl[s][i] = (_Float16)(in[(lane * 9 + s * 8 + i + 9) & 1023] * 1e-6f);
      }
    f32x16 D0a, D0b, D1a, D1b;
    for (int r = 0; r < 16; ++r) { D0a[r] = 0.f; D0b[r] = 0.f; D1a[r] = 0.f; D1b[r] = 0.f; }
    const unsigned fa_l = ring0 + lane * 16;
    const unsigned dw = d0 + (unsigned)wave * (2 * 2 * DBUF) + lane * 16;     // [buffer][block][4 x 1 KB]
    f16x8 a0h, a0m, a0l;
    DSR128(a0h, fa_l, 0); DSR128(a0m, fa_l, LIMB); DSR128(a0l, fa_l, 2 * LIMB);
    LGKM(0);
    t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; t += 2) {
      {
        const unsigned fa = fa_l + (unsigned)((t & 3) * TILE_BYTES), fan = fa_l + (unsigned)(((t + 1) & 3) * TILE_BYTES), dprev = dw + 2 * DBUF;
#define Da D0a
#define Db D0b
        PRODUCER_TILE(D0a, D0b, D1a, D1b, dprev)
#undef Da
#undef Db
      }
      {
        const unsigned fa = fa_l + (unsigned)(((t + 1) & 3) * TILE_BYTES), fan = fa_l + (unsigned)(((t + 2) & 3) * TILE_BYTES), dprev = dw;
#define Da D1a
#define Db D1b
        PRODUCER_TILE(D1a, D1b, D0a, D0b, dprev)
#undef Da
#undef Db
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    t1 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < 16; ++r) sink += D0a[r] + D0b[r] + D1a[r] + D1b[r];
  } else {
    // ---------------- consumer ----------------
    const int c = wave - 4, ct = tid - 256;      // pairs with producer c (same SIMD)
    const unsigned dr = d0 + (unsigned)c * (2 * 2 * DBUF) + lane * 16, ea_l = ring0 + hh * 64, fy = f0a + (unsigned)((lane & 31) * 64);
    const unsigned ck0 = 16u * ct, ck1 = 16u * (ct + 256), ck2 = 16u * (ct + 512), ck3 = 16u * min(ct + 768, 872);
    i32x4 rsrc;
    { const unsigned long long p = (unsigned long long)w; rsrc[0] = (int)(unsigned)p; rsrc[1] = (int)(unsigned)(p >> 32); rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000; }
    i32x4 c0, c1, c2, c3;
    for (int i = 0; i < 4; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    float accA[4], accB[4], fl[8];
    for (int k = 0; k < 4; ++k) { accA[k] = 0.f; accB[k] = 0.f; }
    for (int k = 0; k < 8; ++k) fl[k] = 0.f;
    const float bscA = in[lane & 255], bscB = in[(lane + 7) & 255];
    int soff = 0;
    for (int t = 0; t < tiles; ++t) {
      const unsigned dbuf = dr + (unsigned)(((t + 1) & 1) * 2 * DBUF);      // the accumulators of tile t-1
      const unsigned sts = ring0 + (unsigned)(((t + 2) & 3) * TILE_BYTES), ea = ea_l + (unsigned)(((t + 3) & 3) * TILE_BYTES);
      if (BARRIER) __builtin_amdgcn_s_barrier();
      f32x4 da[4], db[4], ba[4], bb[4], f0, f1;
      BUFLD(c0, ck0, rsrc, soff); BUFLD(c1, ck1, rsrc, soff); BUFLD(c2, ck2, rsrc, soff); BUFLD(c3, ck3, rsrc, soff);
      DSR128(da[0], dbuf, 0); DSR128(da[1], dbuf, 1024); DSR128(da[2], dbuf, 2048); DSR128(da[3], dbuf, 3072);
      DSR128(db[0], dbuf, DBUF); DSR128(db[1], dbuf, DBUF + 1024); DSR128(db[2], dbuf, DBUF + 2048); DSR128(db[3], dbuf, DBUF + 3072);
      DSR128(ba[0], ea, BIAS_OFF); DSR128(ba[1], ea, BIAS_OFF + 16); DSR128(ba[2], ea, BIAS_OFF + 32); DSR128(ba[3], ea, BIAS_OFF + 48);
      DSR128(f0, fy, 0); DSR128(f1, fy, 16);
      LGKM(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bb[q] = ba[q];
        FMAC(da[q].x, ba[q].x, bscA); FMAC(da[q].y, ba[q].y, bscA); FMAC(da[q].z, ba[q].z, bscA); FMAC(da[q].w, ba[q].w, bscA);
        FMAC(db[q].x, bb[q].x, bscB); FMAC(db[q].y, bb[q].y, bscB); FMAC(db[q].z, bb[q].z, bscB); FMAC(db[q].w, bb[q].w, bscB);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        FMAC(accA[q], f0.x, da[q].x); FMAC(accA[q], f0.y, da[q].y); FMAC(accA[q], f0.z, da[q].z); FMAC(accA[q], f0.w, da[q].w);
        FMAC(accB[q], f1.x, db[q].x); FMAC(accB[q], f1.y, db[q].y); FMAC(accB[q], f1.z, db[q].z); FMAC(accB[q], f1.w, db[q].w);
      }
      if ((t & 7) == 7) {      // a flush's worth of extra work: 60 dependent-ish VALU instructions
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) FMAC(fl[k], accA[k & 3], accB[(k + r) & 3]);
      }
      VMC(0);
      DSW128(sts + ck0, c0, 0); DSW128(sts + ck1, c1, 0); DSW128(sts + ck2, c2, 0); DSW128(sts + ck3, c3, 0);
      LGKM(0);
      soff += TILE_BYTES;
      if (soff >= n_rec * TILE_BYTES) soff = 0;
    }
    for (int k = 0; k < 4; ++k) sink += accA[k] + accB[k];
    for (int k = 0; k < 8; ++k) sink += fl[k];
    sink += (float)(c0[0] + c1[0] + c2[0] + c3[0]);
  }
  out[blockIdx.x * 512 + tid] = sink;
  if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool BARRIER>
void run(const char* name, float* in, char* w, float* out, unsigned* cyc, int n_rec) {
  const int tiles = 1180, grid = 256;
  const int ldsb = RING_BYTES + D_BYTES + F_BYTES;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((probe<BARRIER>), dim3(grid), dim3(512), ldsb, 0, in, w, out, cyc, 40, n_rec);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<BARRIER>), dim3(grid), dim3(512), ldsb, 0, in, w, out, cyc, tiles, n_rec);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0, mx = 0;
  for (int i = 0; i < 1024; ++i) { m += h[i]; if (h[i] > mx) mx = h[i]; }
  m /= 1024;
  printf("%-64s mean %7.1f  max %7.1f cycles per tile of 256 edges (54 MFMAs per SIMD = 1728)%s\n", name, m / tiles, mx / tiles, e == hipSuccess ? "" : " ** ERROR **");
}

int main() {
  float *in, *out; unsigned* cyc; char* w;
  const int n_rec = 4 * 59 + 3;
  (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 1024 * 4); (void)hipMalloc(&w, (size_t)n_rec * TILE_BYTES + 65536);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMemset(w, 0x11, (size_t)n_rec * TILE_BYTES + 65536);
  run<true>("specialised waves: producers + consumers, barrier per tile", in, w, out, cyc, n_rec);
  run<false>("... without the barrier (LDS races: timing only)", in, w, out, cyc, n_rec);
  return 0;
}
