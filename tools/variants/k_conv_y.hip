// Fused tensor-product convolution, round 5: the software-pipelined form of k_conv_x.hip's exact three-limb f16 kernel for the score model's conv
// layers (models/tensor_layers.py:147-159 with the FasterTensorProduct of :65-116; gather path with the GEMM1 node-term split, fp32 atomics).
//
// Same algorithm, tile tables, W2 / W1 records, F row, range scaling and limb split as k_conv_x.hip (see its header); what changes is WHO runs WHEN.
// k_conv_x.hip puts two waves on every SIMD and alternates them (one bursts 28 MFMAs while its partner runs the VALU epilogue): the tile period is the
// wave-serial chain burst + epilogue + barrier = 2580 cycles against a matrix-pipe floor of 1792.  Two probes of this round changed the premise:
//   tools/probes/mfma_probe14.hip: for ONE wave per SIMD, up to FIVE single-issue instructions (v_fmac / v_fma / v_add / ds_read_b128) hand-placed behind
//     every v_mfma_f32_32x32x16_f16 cost nothing (32.3 cycles per MFMA with five, 32.05 with none) - also on ONE accumulator chain; round 3's probe 9
//     had measured +12 cycles per VALU, but the compiler had packed, copied and clustered those VALUs instead of interleaving them;
//   tools/probes/mfma_probe15.hip: the skeleton of the loop below - fragments from the LDS ring, ring refill from L2, bias + tensor-product FMAs of the
//     previous tile in the shadows, one s_barrier per tile - runs 1940 cycles per tile.
// So: 256-thread workgroups, ONE wave per SIMD with the whole register file (B operands in AGPRs), every wave owns TWO 32-edge column blocks a / b of
// the 256-edge unit; a tile is two half-bursts HB(a), HB(b) of 28 MFMAs on ONE accumulator chain each (the six limb products of a K step in k_conv_x.hip's
// ONE_ACC order; the 2^-22 terms first; measured as accurate as the two-accumulator form: test_three_limb_product_at_least_as_accurate_as_fp32_chain),
// and the epilogue of block Y rides, instruction by instruction, in the MFMA shadows of block X's half-burst.  The placement is generated
// (tools/gen_conv_y.py -> k_conv_y_gen.inc) as `asm volatile` statements: their order is the issue order; the compiler only allocates registers.
// A column's flush (segmented scan + atomics) is deferred to the first segment of the block's NEXT epilogue slot.
//
// Ring protocol (4 stages, tile i of a unit in stage i & 3, ONE barrier per tile): during tile i every thread stores its four 16-B chunks of record i+2
// (requested during tile i-1) into the stage tile i-2 has left - nobody reads it between barriers i-1 and i+1 - and requests record i+3.
#include <stdlib.h>

#include "k_conv_common.h"

namespace ddk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16((a), (b), (c), 0, 0, 0)

// ---- hand-placed instructions outside the generated segment statements (operands are C++ variables: the compiler allocates, the order is ours) ----
#define MF16Z(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(D) : "v"(a), "a"(b))
#define MF16(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "a"(b))
#define MF8(D, a, b) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "a"(b))
#define DSR128(v, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off))
#define DSR64(v, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off))
#define DSR2ST64(v, addr, o0, o1) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "i"(o0), "i"(o1))
#define DSW128(addr, v, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(off) : "memory")
#define BUFLDA(v, voff, rsrc, soff) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(v) : "v"(voff), "s"(rsrc), "s"(soff))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define VMC(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PINA(x) asm volatile("" : "+a"(x))

#include "k_conv_y_gen.inc"      // (tools/variants/, next to this file)

// Exact power-of-two range scale (k_conv_x.hip)
__device__ __forceinline__ float y_range_scale(float m, float& inv) {
  const uint32_t eb = max((__float_as_uint(m) >> 23) & 0xffu, 87u);
  inv = __uint_as_float((eb - 14u) << 23);
  return __uint_as_float((268u - eb) << 23);
}
struct YLimb3 { _Float16 h, m, l; };
__device__ __forceinline__ YLimb3 y_split3(float v) {
  YLimb3 q;
  q.h = (_Float16)v;
  const float r1 = v - (float)q.h;
  q.m = (_Float16)r1;
  q.l = (_Float16)(r1 - (float)q.m);
  return q;
}
// B operands of one column block: three limbs of the 36 hidden values a lane half holds (4 K steps of 8 + the 4-value tail, packed as in k_conv_x.hip)
struct YBops { f16x8 hh[4], hm[4], hl[4], tmh, thl; f16x4 thi, tmid; };

__device__ __forceinline__ void y_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#define Y3_STEP(MF, ah, am, al, bh, bm, bl)   \
  D1 = MF(ah, bl, D1);                        \
  D0 = MF(ah, bm, D0);                        \
  D1 = MF(al, bh, D1);                        \
  D0 = MF(ah, bh, D0);                        \
  D1 = MF(am, bm, D1);                        \
  D0 = MF(am, bh, D0);

struct YSeg { float m1, m2, m4, m8, m16; };

// The prologue of ONE column block (32 edges, lane = (edge el, K half hh)): indices, gathers, GEMM1 with the node-term split, limbs of h, F row.
// (k_conv_x.hip's SPLIT prologue, restated per block.)
struct YBlock {
  int sn;
  float osc, bsc2, s0, vx, vy, vz;
  YSeg seg;
  bool tail, valid;
};
__device__ __forceinline__ void y_block_prologue(const ConvKArgs& A, int e0, int gend, int gw, int lane, float* Fr, YBops& H, YBlock& B) {
  constexpr int FS = FX_STRIDE;
  const int el = lane & 31, hh = lane >> 5;
  const int nvalid = min(32, gend - e0);
  const bool valid = el < nvalid;
  const int e = nvalid > 0 ? e0 + min(el, nvalid - 1) : gend - 1;
  const int sn = A.src[e], dn = A.dst[e];
  B.sn = sn;
  B.valid = valid;
  {
    const SegCtl c = make_segctl(sn, el, nvalid, valid);
    B.seg.m1 = c.m1 ? 1.0f : 0.0f; B.seg.m2 = c.m2 ? 1.0f : 0.0f; B.seg.m4 = c.m4 ? 1.0f : 0.0f; B.seg.m8 = c.m8 ? 1.0f : 0.0f; B.seg.m16 = c.m16 ? 1.0f : 0.0f;
    B.tail = c.tail;
  }
  const float4 shv = ld4(A.sh + (size_t)e * 4);
  float4 mainv[NS / 4];
  float2 pv2[3 * NV / 2];
  {
    const float* xr = A.x + (size_t)dn * XW;
#pragma unroll
    for (int j = 0; j < NS / 4; ++j) mainv[j] = ld4(xr + (hh ? OFF_C : 0) + 4 * j);
#pragma unroll
    for (int j = 0; j < 3 * NV / 2; ++j) pv2[j] = ld2(xr + (hh ? OFF_Q : OFF_P) + 2 * j);
  }
  {
    float h[36];
    const char* w1 = reinterpret_cast<const char*>(A.w1x) + (size_t)gw * 3 * W1X_TILE_BYTES;
    float4 psv[9], pdv[9];
    {
      const float* ps = A.pre + ((size_t)sn * 4 + (gw & 1)) * NE + 36 * hh;
      const float* pd = A.pre + ((size_t)dn * 4 + 2 + (gw >> 1)) * NE + 36 * hh;
#pragma unroll
      for (int j = 0; j < 9; ++j) { psv[j] = ld4(ps + 4 * j); pdv[j] = ld4(pd + 4 * j); }
    }
    float bin[12];
    {
      const float* pe = A.edge_attr + (size_t)e * NS + 12 * hh;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 a = ld4(pe + 4 * j);
        bin[4 * j + 0] = a.x; bin[4 * j + 1] = a.y; bin[4 * j + 2] = a.z; bin[4 * j + 3] = a.w;
      }
    }
    float m1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 12; ++j) m1 = fmaxf(m1, fabsf(bin[j]));
    m1 = fmaxf(m1, __shfl_xor(m1, 32));
    float inv1;
    const float s1 = y_range_scale(m1, inv1);
    f16x8 b0h, b0m, b0l;
    f16x4 b1h, b1m, b1l;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const YLimb3 q = y_split3(bin[i] * s1); b0h[i] = q.h; b0m[i] = q.m; b0l[i] = q.l; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const YLimb3 q = y_split3(bin[8 + i] * s1); b1h[i] = q.h; b1m[i] = q.m; b1l[i] = q.l; }
    const float bsc = s1 * A.w1s[gw], usc = inv1 * A.w1u[gw];
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      f32x16 D0, D1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (T < 2 || j == 0) {
          const float4 u = psv[4 * T + j], w = pdv[4 * T + j];
          D0[4 * j + 0] = (u.x + w.x) * bsc; D0[4 * j + 1] = (u.y + w.y) * bsc; D0[4 * j + 2] = (u.z + w.z) * bsc; D0[4 * j + 3] = (u.w + w.w) * bsc;
        } else {
          D0[4 * j + 0] = 0.0f; D0[4 * j + 1] = 0.0f; D0[4 * j + 2] = 0.0f; D0[4 * j + 3] = 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) D1[r] = 0.0f;
      const char* wt = w1 + (size_t)T * W1X_TILE_BYTES;
      {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(wt + lane * 16);
        const f16x8 am = *reinterpret_cast<const f16x8*>(wt + W2X_LIMB_BYTES + lane * 16);
        const f16x8 al = *reinterpret_cast<const f16x8*>(wt + 2 * W2X_LIMB_BYTES + lane * 16);
        Y3_STEP(MFMA16, ah, am, al, b0h, b0m, b0l)
      }
      {
        const f16x4 ah = *reinterpret_cast<const f16x4*>(wt + 1024 + lane * 16);
        const f16x4 am = *reinterpret_cast<const f16x4*>(wt + W2X_LIMB_BYTES + 1024 + lane * 16);
        const f16x4 al = *reinterpret_cast<const f16x4*>(wt + 2 * W2X_LIMB_BYTES + 1024 + lane * 16);
        Y3_STEP(MFMA8, ah, am, al, b1h, b1m, b1l)
      }
      const int nr = T < 2 ? 16 : 4;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r < nr) h[16 * T + r] = fmaxf(D0[r] + D1[r], 0.0f) * usc;
    }
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 36; ++j) m2 = fmaxf(m2, h[j]);
    m2 = fmaxf(m2, __shfl_xor(m2, 32));
    float inv2;
    const float s2 = y_range_scale(m2, inv2);
    f16x4 thi, tmid, tlo;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const YLimb3 q = y_split3(h[8 * s + i] * s2); H.hh[s][i] = q.h; H.hm[s][i] = q.m; H.hl[s][i] = q.l; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const YLimb3 q = y_split3(h[32 + i] * s2); thi[i] = q.h; tmid[i] = q.m; tlo[i] = q.l; }
    H.thi = thi; H.tmid = tmid;
    H.tmh = __builtin_shufflevector(tmid, thi, 0, 1, 2, 3, 4, 5, 6, 7);      // {h_mid, h_hi}: x {W_hi, W_mid}
    H.thl = __builtin_shufflevector(thi, tlo, 0, 1, 2, 3, 4, 5, 6, 7);       // {h_hi, h_lo}: x {W_lo, W_hi}
    B.bsc2 = s2 * A.w2s[gw];
    B.osc = inv2 * A.w2u[gw];
  }
  B.s0 = shv.x; B.vx = shv.y; B.vy = shv.z; B.vz = shv.w;
  {
    const float inv_s3 = 0.57735026918962576451f;
    const int o_main_dst = hh ? FX_C : FX_A, r0 = hh ? NV : 0;
#pragma unroll
    for (int j = 0; j < NS / 4; ++j) *reinterpret_cast<float4*>(Fr + o_main_dst + 4 * j) = mainv[j];
    float pv[3 * NV];
#pragma unroll
    for (int j = 0; j < 3 * NV / 2; ++j) { pv[2 * j] = pv2[j].x; pv[2 * j + 1] = pv2[j].y; }
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const float px = pv[3 * m], py = pv[3 * m + 1], pz = pv[3 * m + 2];
      Fr[FX_PQ + (m < 4 ? 4 * hh + m : 8 + 2 * hh + (m - 4))] = (px * B.vx + py * B.vy + pz * B.vz) * inv_s3;
      const int r = r0 + m;
      float* Pr = Fr + FX_R + 12 * (r >> 2) + (r & 3);
      Pr[0] = px;
      Pr[4] = py;
      Pr[8] = pz;
    }
  }
  (void)FS;
}

#ifndef Y_PROLOGUE_TILES
#define Y_PROLOGUE_TILES 12.0f
#endif
constexpr int Y_WAVES = 4;
constexpr int K_NONE = 0, K_RA = 1, K_RT = 2, K_TV0 = 3, K_RTS = 7;
__device__ __forceinline__ int y_kind_code(int w0) {
  const int kind = w0 & 3;
  return kind == T_RA ? K_RA : (kind == T_RT ? K_RT : (kind == T_TV ? K_TV0 + ((w0 >> 14) & 3) : K_RTS));
}

// TRACE: workgroup 0 writes one s_memtime record per UNIT (tools/conv_trace.py --coarse): slot 4 unit start, 5 / 6 block a / b prologue done, 7 before the
// barrier in front of the tile loop, 0 tile loop done, 1 tiles, 3 drain done, 2 unit handed over
template <bool TRACE>
__global__ __launch_bounds__(64 * Y_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_y_kernel(ConvXArgs AX) {
  const ConvKArgs& A = AX.k;
  int trace_n = 0;
  auto stamp = [&](int slot, int value) {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && trace_n < CONV_TRACE_TILES)
        AX.trace[((threadIdx.x >> 6) * CONV_TRACE_TILES + trace_n) * 8 + slot] = slot == 1 ? (uint32_t)value : (uint32_t)__builtin_amdgcn_s_memtime();
      if (slot == 2) ++trace_n;
    }
  };
  constexpr int FS = FX_STRIDE, BLOCK_EDGES = 64 * Y_WAVES, TB = W2X_TILE_BYTES;
  static_assert(BLOCK_EDGES == CONV_BLOCK_EDGES, "the unit (256 edges) is k_conv_x.hip's");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int el = lane & 31, hh = lane >> 5;
  float* Fa = lds + (wave * 64 + el) * FS;             // F rows of this lane's two edges (block a: edges 64 w .. + 31, block b: + 32 ..)
  float* Fb = Fa + 32 * FS;
  char* ring = reinterpret_cast<char*>(lds + BLOCK_EDGES * FS);
  int* blk_slot = reinterpret_cast<int*>(ring + W2X_STAGES * TB);
  const bool g2_shared = A.sum_g2 != nullptr;
  int gb_v = 0, ge_v = 0;
  if (lane < A.n_active) {
    gb_v = A.gbeg[lane];
    ge_v = A.gend[lane];
  }
  const int nb_v = (ge_v - gb_v + BLOCK_EDGES - 1) / BLOCK_EDGES;
  int pend_v = nb_v;
#pragma unroll
  for (int d = 1; d < 16; d *= 2) {
    const int t = __shfl_up(pend_v, d, 64);
    if (lane >= d) pend_v += t;
  }
  const int pbeg_v = pend_v - nb_v;
  const int bs4 = __builtin_amdgcn_readlane(pend_v, 15);
  const int n_tiles = A.n_tiles;
  constexpr int REC16 = TB / 16;                               // 873 x 16 B per tile record
  // ring chunks of this thread: 16-B chunks tid, tid + 256, tid + 512, min(tid + 768, 872)
  const unsigned ck0 = 16u * tid, ck1 = 16u * (tid + 256), ck2 = 16u * (tid + 512), ck3 = 16u * min(tid + 768, REC16 - 1);
  const unsigned ring0 = (unsigned)(size_t)(reinterpret_cast<char*>(ring) - reinterpret_cast<char*>(lds));
  const unsigned ringl = ring0 + lane * 16, ringt = ring0 + lane * 8, ringb = ring0 + hh * 64;
  const unsigned fra = (unsigned)((wave * 64 + el) * FS * 4), frb = fra + 32 * FS * 4;
  const unsigned hh4 = 4u * hh, hh12 = 12u * hh;

  const int nwg = gridDim.x;
  const int full = bs4 >= nwg ? (bs4 / nwg) * nwg : 0;
  const int rest = bs4 - full;
  int split = 1;
  if (rest > 0) {
    float best = 1e30f;
    for (int sp = 1; sp <= A.n_cols; ++sp) {
      const float cost = (float)((rest * sp + nwg - 1) / nwg) * ((float)n_tiles / (float)sp + Y_PROLOGUE_TILES);
      if (cost < best - 1e-6f) { best = cost; split = sp; }
    }
  }
  const int n_units = full + rest * split;

  int unit = blockIdx.x;
  for (;;) {
    if (unit >= n_units) break;
    stamp(4, 0);
    int unit_next = 0;
    if (tid == 0) unit_next = nwg + atomicAdd(A.counter, 1);
    int blk = unit, t_begin = 0, t_end = n_tiles;
    if (unit >= full) {
      const int r = unit - full, c = r % split;
      blk = full + r / split;
      t_begin = A.col_start[(c * A.n_cols) / split];
      t_end = A.col_start[((c + 1) * A.n_cols) / split];
    }
    const int g = __popcll(__ballot(lane < 16 && blk >= pend_v));
    const int gbeg = __builtin_amdgcn_readlane(gb_v, g), gend = __builtin_amdgcn_readlane(ge_v, g);
    const int bstart = __builtin_amdgcn_readlane(pbeg_v, g);
    const int e0a = gbeg + BLOCK_EDGES * (blk - bstart) + 64 * wave, e0b = e0a + 32;
    const int gw = (int)((A.wmap >> (4 * g)) & 15);
    const char* wrec = reinterpret_cast<const char*>(A.w2x) + (size_t)gw * n_tiles * TB;
    i32x4 rsrc;
    {
      const unsigned long long p = (unsigned long long)wrec;
      rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
      rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
      rsrc[2] = 0x7fffffff;
      rsrc[3] = 0x00020000;
    }
    // ---- stage records 0 and 1 of this unit, request record 2 (the ring is idle: the previous unit ended with a barrier) ----
    i32x4 c00, c10, c20, c30;      // ring chunks in flight (AGPRs): record i+2 during tile i
    {
      const char* wr0 = wrec + (size_t)t_begin * TB;
      const char* wr1 = wrec + (size_t)min(t_begin + 1, t_end - 1) * TB;
      const float4 r0 = *reinterpret_cast<const float4*>(wr0 + ck0), r1 = *reinterpret_cast<const float4*>(wr0 + ck1);
      const float4 r2 = *reinterpret_cast<const float4*>(wr0 + ck2), r3 = *reinterpret_cast<const float4*>(wr0 + ck3);
      const float4 r4 = *reinterpret_cast<const float4*>(wr1 + ck0), r5 = *reinterpret_cast<const float4*>(wr1 + ck1);
      const float4 r6 = *reinterpret_cast<const float4*>(wr1 + ck2), r7 = *reinterpret_cast<const float4*>(wr1 + ck3);
      *reinterpret_cast<float4*>(ring + ck0) = r0; *reinterpret_cast<float4*>(ring + ck1) = r1;
      *reinterpret_cast<float4*>(ring + ck2) = r2; *reinterpret_cast<float4*>(ring + ck3) = r3;
      *reinterpret_cast<float4*>(ring + TB + ck0) = r4; *reinterpret_cast<float4*>(ring + TB + ck1) = r5;
      *reinterpret_cast<float4*>(ring + TB + ck2) = r6; *reinterpret_cast<float4*>(ring + TB + ck3) = r7;
    }
    // ---- the two column blocks: indices, gathers, GEMM1, limbs, F rows; the B operands end up in AGPRs ----
    YBops Ha, Hb;
    YBlock Ba, Bb;
    y_block_prologue(A, e0a, gend, gw, lane, Fa, Ha, Ba);
#pragma unroll
    for (int s = 0; s < 4; ++s) { PINA(Ha.hh[s]); PINA(Ha.hm[s]); PINA(Ha.hl[s]); }
    PINA(Ha.tmh); PINA(Ha.thl); PINA(Ha.thi); PINA(Ha.tmid);
    stamp(5, 0);
    y_block_prologue(A, e0b, gend, gw, lane, Fb, Hb, Bb);
#pragma unroll
    for (int s = 0; s < 4; ++s) { PINA(Hb.hh[s]); PINA(Hb.hm[s]); PINA(Hb.hl[s]); }
    PINA(Hb.tmh); PINA(Hb.thl); PINA(Hb.thi); PINA(Hb.tmid);
    stamp(6, 0);

    // per-block epilogue constants
    const float inv_s2 = 0.70710678118654752440f;
    float bsc2a = Ba.bsc2, bsc2b = Bb.bsc2;
    float oscva = Ba.valid ? Ba.osc : 0.0f, oscvb = Bb.valid ? Bb.osc : 0.0f;      // lanes past the group's end contribute zeros
    float s0a = Ba.s0, vxa = Ba.vx, vya = Ba.vy, vza = Ba.vz, s0b = Bb.s0, vxb = Bb.vx, vyb = Bb.vy, vzb = Bb.vz;
    float wxa = vxa * inv_s2, wya = vya * inv_s2, wza = vza * inv_s2, wxb = vxb * inv_s2, wyb = vyb * inv_s2, wzb = vzb * inv_s2;
    float sm1a = Ba.seg.m1, sm2a = Ba.seg.m2, sm4a = Ba.seg.m4, sm8a = Ba.seg.m8, sm16a = Ba.seg.m16;
    float sm1b = Bb.seg.m1, sm2b = Bb.seg.m2, sm4b = Bb.seg.m4, sm8b = Bb.seg.m8, sm16b = Bb.seg.m16;
    const unsigned long long taila = __ballot(Ba.tail), tailb = __ballot(Bb.tail);
    // the node row of this lane's edge as a byte offset from the accumulator array (global_atomic with an SGPR base + 32-bit VGPR offset)
    const bool shared2 = g2_shared && g == 2;
    const float* sumbase = shared2 ? A.sum_g2 : A.sum;
    unsigned vrowa, vrowb;
    if (shared2) {
      vrowa = (unsigned)(Ba.sn - A.g2_node_off) * (unsigned)(XW * 4);
      vrowb = (unsigned)(Bb.sn - A.g2_node_off) * (unsigned)(XW * 4);
    } else {
      const unsigned slot = (A.slots >> (2 * g)) & 3;
      vrowa = ((unsigned)Ba.sn * (unsigned)A.n_slots + slot) * (unsigned)(XW * 4);
      vrowb = ((unsigned)Bb.sn * (unsigned)A.n_slots + slot) * (unsigned)(XW * 4);
    }
    float accAa[4], accVa[4][3], accXa[4][3], Ra[4], accAb[4], accVb[4][3], accXb[4][3], Rb[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      accAa[rq] = 0.0f; accAb[rq] = 0.0f; Ra[rq] = 0.0f; Rb[rq] = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { accVa[rq][c] = 0.0f; accXa[rq][c] = 0.0f; accVb[rq][c] = 0.0f; accXb[rq][c] = 0.0f; }
    }
    // record 2 of the unit (stored during tile 0); everything the prologue requested has landed behind the wait below, so the loop's vmcnt counts are
    // its own.  Requests past the unit's last tile re-read it (soffmax)
    const int soffmax = (t_end - 1) * TB;
    int soff = min((t_begin + 2) * TB, soffmax);
    BUFLDA(c00, ck0, rsrc, soff); BUFLDA(c10, ck1, rsrc, soff); BUFLDA(c20, ck2, rsrc, soff); BUFLDA(c30, ck3, rsrc, soff);
    soff = (t_begin + 3) * TB;
    stamp(7, 0);
    y_lds_barrier();      // ring stages 0 / 1 and the F rows are visible
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f16x8 a0h, a0m, a0l, a1h, a1m, a1l;
    f32x16 Da, Db;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Da[r] = 0.0f; Db[r] = 0.0f; }
    i32x2 dq;
    unsigned long long sv_;
    DSR128(a0h, ringl, 0); DSR128(a0m, ringl, W2X_LIMB_BYTES); DSR128(a0l, ringl, 2 * W2X_LIMB_BYTES);
    {
      const unsigned d0 = ring0 + W2X_DESC_OFF;
      DSR64(dq, d0, 0);
    }
    LGKM(0);
    int wC = __builtin_amdgcn_readfirstlane(dq[0]), chC = __builtin_amdgcn_readfirstlane(dq[1]);     // descriptor of the current tile
    int wP = 0, chP = 0;                                                                              // ... of the previous tile
    int penda = 0, pendb = 0, pchana = 0, pchanb = 0;      // pending flush of the column a block's previous tile closed: 0 none, FL_S, FL_V; chan0 * 4
    const int n_t = t_end - t_begin;
    // state the half-burst statements carry from one to the next (k_conv_y_gen.inc): fragment / tail-fragment address of the current tile, bias address of
    // the tile whose epilogue comes next; scratch scalars of the statements
    unsigned vfa = ringl, vft = ringt, vea = ringb;
    const unsigned ringw0 = ring0 + ck0, ringw1 = ring0 + ck1, ringw2 = ring0 + ck2, ringw3 = ring0 + ck3;
    const int ring0d = (int)(ring0 + W2X_DESC_OFF);
    int t0_, t1_, t2_, t3_, sel2_, pk0_, pk1_, pk2_;
    // half-burst a: the MFMAs of block a, tile i; epilogue slot of block b, tile i-1 (descriptor wP).  half-burst b: block b, tile i; epilogue slot of
    // block a, tile i (descriptor wC); it ends the tile: next descriptor, barrier.
#ifdef Y_EXP_NO_FLUSH         // timing experiment (wrong results): no column is ever flushed inside the tile loop
#define Y_SEL1_A (i == 0 ? 3 : 0)
#define Y_SEL1_B 0
#else
#define Y_SEL1_A (i == 0 ? 3 : pendb)
#define Y_SEL1_B penda
#endif
#ifdef Y_EXP_SAME_RECORD      // timing experiment (wrong results): every ring request asks for the same record (L2 hits only)
#define Y_EXP_SOFF soff = (t_begin + 3) * TB;
#else
#define Y_EXP_SOFF
#endif
    for (int i = 0;;) {
      { const int sel1_ = Y_SEL1_A; Y_HB_a0(); }
      { const int sel1_ = Y_SEL1_B; Y_HB_b0(); }
      Y_EXP_SOFF
      if (++i >= n_t) break;
    }
#undef Y_SEL1_A
#undef Y_SEL1_B
#undef Y_EXP_SOFF
    (void)t0_; (void)t1_; (void)t2_; (void)t3_; (void)sel2_; (void)pk0_; (void)pk1_; (void)pk2_;
    stamp(0, 0);
    stamp(1, n_t);
    // ================= drain: block a's last flush; block b's last tile (the flush of the tile before it first) and its flush =================
    {
      const unsigned stl = (unsigned)(((n_t - 1) & 3) * TB);
      // the last tile's ring requests (records past the unit's end) are dropped: wait for them HERE, before the drain's atomics go out, so that the unit
      // can end without waiting for those (the chunk registers are reused by the next unit's prologue)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int sel_ = penda;
      if (sel_ != 0) Y_DRAIN_FLUSH_a();
      sel_ = pendb;
      if (sel_ != 0) Y_DRAIN_FLUSH_b();
      const unsigned ea = ringb + stl, fy = frb + 4u * (unsigned)((wP >> 16) & 0xff), gy = frb + 4u * (unsigned)((wP >> 8) & 0x3c);
      const int xp = (wP & 0x80) ? ((wP >> 8) & 3) : 3;
      const float pk0b = xp == 0 ? 1.0f : 0.0f, pk1b = xp == 1 ? 1.0f : 0.0f, pk2b = xp == 2 ? 1.0f : 0.0f;
      sel_ = y_kind_code(wP);
      Y_DRAIN_MAIN_b();
      pendb = (wP >> 2) & 3; pchanb = chP * 4;
      sel_ = pendb;
      if (sel_ != 0) Y_DRAIN_FLUSH_b();
    }
    stamp(3, 0);
    // hand the next unit to the workgroup; this barrier also retires the ring and the F rows before the next unit's staging writes
    if (tid == 0) *blk_slot = unit_next;
    y_lds_barrier();
    unit = __builtin_amdgcn_readfirstlane(*blk_slot);
    stamp(2, 0);
  }
}

hipError_t conv_prepare_device_y() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_y_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_X_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_y_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_X_LDS_BYTES);
  return e;
}

// The layer's tile table qualifies when every flush closes a full scalar column (4 quads) or a 6-channel vector column (3 quads): the score model's
// conv layers (ns = 24, nv = 6).  Checked once per layer at finalize time (ConvLayerDev::epi_ok).
hipError_t launch_conv_y(const ConvXArgs& X, int n_cu, hipStream_t s) {
  if (X.trace != nullptr) hipLaunchKernelGGL(conv_y_kernel<true>, dim3(n_cu), dim3(64 * Y_WAVES), CONV_X_LDS_BYTES, s, X);
  else hipLaunchKernelGGL(conv_y_kernel<false>, dim3(n_cu), dim3(64 * Y_WAVES), CONV_X_LDS_BYTES, s, X);
  return hipGetLastError();
}

}  // namespace ddk
