// Fused tensor-product convolution on the f16 matrix pipe with EXACT fp32 operands (the default conv kernel of the score model).
//
// Same algorithm, tile tables, epilogue and scatter as k_conv.hip (the fp32-MFMA kernel, kept as ddk_config.conv_kernel = 1 and for
// the confidence model); only the two radial-MLP GEMMs of models/tensor_layers.py:140-143,154-155 change pipe.  Every fp32 operand x (after an
// exact power-of-two range scaling, below) is split into three fp16 limbs that carry their own weight
//        x = hi + mid + lo,     hi = fp16(x),  mid = fp16(x - hi),  lo = fp16(x - hi - mid)
// A 24-bit significand minus its 11-bit rounding leaves <= 13 bits, of which mid takes 11 and lo the rest: the split is EXACT as long as the
// last bit of x is a multiple of the fp16 subnormal step 2^-24 (v_mfma_f32_32x32x16_f16 and v_cvt_f16_f32 honour fp16 subnormals:
// tools/probes/mfma_probe10.hip), i.e. for every |x| >= 0.5 after scaling; smaller values are off by <= 2^-25 absolute.  The product of two
// such sums is formed with v_mfma_f32_32x32x16_f16 (products of fp16 numbers are exact in fp32) keeping the six terms
//        hi.hi + hi.mid + mid.hi   (accumulator D0)        hi.lo + lo.hi + mid.mid   (accumulator D1, everything of relative order 2^-22)
// and dropping mid.lo + lo.mid + lo.lo <= 3 * 2^-33 relative - 500 times below the fp32 rounding of the accumulation itself.  D0 + D1 is
// formed once per tile; D0 rounds like an fp32 FMA chain (one rounding per MFMA, round to nearest even), D1 keeps the small terms among
// themselves.  6 MFMAs of 8 passes replace 8 x 16 passes of v_mfma_f32_32x32x2_f32 per 16 K values: 2.7x fewer matrix-pipe cycles, and the
// f16 pipe does not share issue with the VALU the way the fp32 MFMA does.
//
// Range: fp16 spans 2^-24 .. 65504, a checkpoint does not.  Each split operand is first multiplied by an exact power of two that brings the
// maximum of its group into [2^14, 2^15): W1 and W2 per (layer, edge group) at pack time, the GEMM1 inputs and the hidden vector per
// EDGE in the kernel (exponent arithmetic only); the results are multiplied back by the inverse powers.  Every value within 2^-15 of its
// group's maximum is represented exactly; smaller ones are off by <= 2^-39 of the maximum.
//
// K = 72 = 4 steps of 16 + one of 8 (v_mfma_f32_32x32x8_f16): register 8s+i (< 36) of a lane half is element i of step s.
// Tile record in the LDS ring (13,968 B): three limbs x [4 x 1 KB fragments | 512 B tail fragment] | bias [2][16] f32 (added, scaled like the
// products, when D0 + D1 is formed); the tile descriptors ride in the kernel arguments (scalar loads).
// The fragments are streamed from the ring INSIDE the burst (no register double buffer: h's three limbs need the registers).
//
// Two waves share a SIMD and with it ONE matrix pipe; a wave's tile is a burst of 27 MFMAs (864 pipe cycles; the K=8 tail packs two products per
// 32x32x16) followed by a VALU / LDS epilogue.  The workgroup runs as two half-groups in ALTERNATION with ONE barrier per tile: between two
// barriers waves 0-3 (group A, one per SIMD) run [burst t, epilogue t] and waves 4-7 (group B, their SIMD partners) [epilogue t-1, burst t], so
// that a burst always sits beside the partner's epilogue and the pipe sees one burst after the other; where an epilogue is shorter than the
// partner's burst the next burst simply starts early and shares the pipe for a while (round 3 separated the two halves of a tile by a second
// barrier: every half phase then cost max(burst, partner's epilogue) + a barrier, 29 % of a wave's cycles were waits).
// Ring protocol (4 stages, tile t in stage t & 3, both groups read tile t between the same two barriers): every thread requests its two 16-B
// chunks of tile t+3 from L2 during its burst of tile t and stores them in its epilogue of tile t into the stage tile t-1 has left - group A
// between barriers t and t+1, group B one interval later - so tile t+3 is complete at barrier t+2 and each wave fetches the first two K steps of
// its NEXT tile at the start of its epilogue.
// Experiment switches.  X3_ABL_* are TIMING-ONLY ablations (results invalid), ONE_ACC / X3_PF2 / X3_CXX_EPI / X_PROLOGUE_TILES are measured-and-dropped
// alternatives kept for same-box A/Bs (tools/build_variant.sh builds them into ab_libs/, never into libddk.so).  A stray -D cannot reach the product library:
#if (defined(X3_ABL_NOBARRIER) || defined(X3_ABL_BAR2) || defined(X3_ABL_PRIO_B2) || defined(X3_ABL_NOW1) || defined(X3_ABL_NOGATHER)) && !defined(DDK_TIMING_ONLY_BUILD)
#error "X3_ABL_* switches give WRONG RESULTS (timing-only ablations): they need -DDDK_TIMING_ONLY_BUILD as well (tools/build_variant.sh adds it)"
#endif
#if (defined(ONE_ACC) || defined(X3_PF2) || defined(X3_CXX_EPI) || defined(X_PROLOGUE_TILES) || defined(X3_KEEP_MIDMID)) && !defined(DDK_VARIANT_BUILD) && !defined(DDK_TIMING_ONLY_BUILD)
#error "experiment switch of k_conv_x.hip outside a variant build: add -DDDK_VARIANT_BUILD (tools/build_variant.sh adds it)"
#endif
#include <stdlib.h>

#include "k_conv_common.h"
#ifdef X3_TWO_LIMBS
// k_conv_x2.hip compiles this file a second time as the TWO-LIMB form (ddk_config.conv_kernel = 0, include/ddk.h): the same kernel under its own names
#define conv_x3_kernel conv_x2_kernel
#define launch_conv_fused_x launch_conv_fused_x2
#define conv_prepare_device_x conv_prepare_device_x2
#include "k_conv_x_epi2_gen.inc"     // ... generated with GEN_ONE_ACC=1: one accumulator, nothing to fold
#else
#include "k_conv_x_epi_gen.inc"      // the tile epilogue as one asm statement per ring stage (tools/gen_conv_x_epi.py; round 5)
#endif

namespace ddk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((vector_size(16)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16((a), (b), (c), 0, 0, 0)

// Exact power-of-two range scale: 2^(14 - floor(log2 max(m, 2^-40))) and its inverse (max|x| lands in [2^14, 2^15))
__device__ __forceinline__ float range_scale(float m, float& inv) {
  const uint32_t eb = max((__float_as_uint(m) >> 23) & 0xffu, 87u);   // biased exponent; 0 and subnormals map to the 2^-40 floor
  inv = __uint_as_float((eb - 14u) << 23);
  return __uint_as_float((268u - eb) << 23);
}

// the three limbs of one (range-scaled) value, each carrying its own weight: the subtractions are exact, the conversions of the remainders
// round only below the fp16 subnormal step
struct Limb3 { _Float16 h, m, l; };
__device__ __forceinline__ Limb3 split3(float v) {
  Limb3 q;
  q.h = (_Float16)v;
  const float r1 = v - (float)q.h;
  q.m = (_Float16)r1;
  q.l = (_Float16)(r1 - (float)q.m);
  return q;
}

// B-operand limbs of the 36 values a lane half holds: steps 0..3 (8 values each) and the 4-value tail
struct Limbs {
  f16x8 hi[4], mid[4], lo[4];
  f16x4 thi, tmid, tlo;
};
// The K = 8 tail step costs the matrix pipe as much as a K = 16 step, and four of its six limb products come in pairs on the same accumulator:
// hi.mid + mid.hi and hi.lo + lo.hi each fit ONE 32x32x16 MFMA on concatenated operands ({W_hi, W_mid} x {h_mid, h_hi}, {W_lo, W_hi} x {h_hi, h_lo}),
// so do hi.hi + mid.mid once mid.mid (relative order 2^-22, the one small term of the tail that then rounds with D0's sum instead of among the small
// terms: the MFMA adds the products of its K range exactly and rounds once, so nothing is rounded that was not before) joins D0:
// {W_hi, W_mid} x {h_hi, h_mid}.  The tail is 3 MFMAs instead of 6 (27 per tile; round 4: 28).  B-side operands, built once per unit:
struct TailB { f16x8 mh, hl, hm; };      // {h_mid, h_hi}, {h_hi, h_lo}, {h_hi, h_mid} of the 4-value tail
__device__ __forceinline__ TailB make_tailb(const Limbs& L) {
  TailB t;
  t.mh = __builtin_shufflevector(L.tmid, L.thi, 0, 1, 2, 3, 4, 5, 6, 7);
  t.hl = __builtin_shufflevector(L.thi, L.tlo, 0, 1, 2, 3, 4, 5, 6, 7);
  t.hm = __builtin_shufflevector(L.thi, L.tmid, 0, 1, 2, 3, 4, 5, 6, 7);
  return t;
}

__device__ __forceinline__ void make_limbs(Limbs& L, const float (&v)[36], float scale) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const Limb3 q = split3(v[8 * s + i] * scale); L.hi[s][i] = q.h; L.mid[s][i] = q.m; L.lo[s][i] = q.l; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { const Limb3 q = split3(v[32 + i] * scale); L.thi[i] = q.h; L.tmid[i] = q.m; L.tlo[i] = q.l; }
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. the global atomics of a flush and
// the tile record requests that are meant to stay in flight across the barrier
#if defined(X3_ABL_NOBARRIER)      // (timing-only ablation: no barrier inside the tile loop - ring races, results invalid)
#define X3_LOOP_BARRIER(ST) ((void)0)
#elif defined(X3_ABL_BAR2)         // (timing-only: a barrier every second tile)
#define X3_LOOP_BARRIER(ST) do { if (((ST) & 1) == 0) lds_barrier(); } while (0)
#else
#define X3_LOOP_BARRIER(ST) lds_barrier()
#endif
#ifdef X3_ABL_PRIO_B2              // (timing-only: the later-dispatched half bursts at priority 2 - the arbiter prefers the older wave at equal priority)
#define X3_BURST_PRIO() do { if (grp_s) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); } while (0)
#else
#define X3_BURST_PRIO() __builtin_amdgcn_s_setprio(1)
#endif
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // s_waitcnt lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct Frag16 { f16x8 h, m, l; };     // the three limbs of one K step's A fragment
__device__ __forceinline__ Frag16 lds_frag16(const char* stage, int s, int lane) {
  Frag16 f;
  f.h = *reinterpret_cast<const f16x8*>(stage + s * 1024 + lane * 16);
  f.m = *reinterpret_cast<const f16x8*>(stage + W2X_LIMB_BYTES + s * 1024 + lane * 16);
  f.l = *reinterpret_cast<const f16x8*>(stage + 2 * W2X_LIMB_BYTES + s * 1024 + lane * 16);
  return f;
}

// Segmented inclusive scan over runs of equal edge_src with the step masks as 0 / 1 FLOATS: every Hillis-Steele step is one
// v_fmac_f32 whose first source is the DPP-shifted value (x += shr(x) * m), a third of the mov / select / add form of k_conv_common.h;
// the masks cost five VGPRs, which this kernel has and the fp32 kernel has not.
struct SegF { float m1, m2, m4, m8, m16; bool tail, valid; };
__device__ __forceinline__ SegF make_segf(const SegCtl& c) {
  SegF f;
  f.m1 = c.m1 ? 1.0f : 0.0f; f.m2 = c.m2 ? 1.0f : 0.0f; f.m4 = c.m4 ? 1.0f : 0.0f; f.m8 = c.m8 ? 1.0f : 0.0f; f.m16 = c.m16 ? 1.0f : 0.0f;
  f.tail = c.tail; f.valid = c.valid;
  return f;
}
// one scan step of N interleaved channels: x += dpp(x) * m as ONE v_fmac_f32 with the DPP modifier on its first source (the compiler keeps a
// v_mov_b32_dpp in front of an fma).  A VALU result needs two wait states before a DPP read of it: the other channels' instructions provide
// them for N >= 3, a lone channel gets an s_nop.
#define SEGF_STEP(CTRL, m)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < N; ++i) {                                                                           \
    if (N >= 3) asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:1" : "+v"(xv[i]) : "v"(m));                     \
    else asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:1" : "+v"(xv[i]) : "v"(m));                 \
  }
template <bool DET, int N>
__device__ __forceinline__ void segf_add_n(float* dst, int stride, float (&xv)[N], const SegF& c) {
#pragma unroll
  for (int i = 0; i < N; ++i) xv[i] = c.valid ? xv[i] : 0.0f;
  // the inline-asm DPP reads below are invisible to the compiler's hazard recogniser: two wait states behind whatever VALU instruction wrote xv
  asm volatile("s_nop 1" ::: "memory");
  SEGF_STEP("row_shr:1 row_mask:0xf bank_mask:0xf", c.m1)
  SEGF_STEP("row_shr:2 row_mask:0xf bank_mask:0xf", c.m2)
  SEGF_STEP("row_shr:4 row_mask:0xf bank_mask:0xf", c.m4)
  SEGF_STEP("row_shr:8 row_mask:0xf bank_mask:0xf", c.m8)
  SEGF_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", c.m16)      // lane 15 of rows 0 / 2 into rows 1 / 3
  if (c.tail) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (DET) dst[i * stride] = xv[i];
      else unsafeAtomicAdd(dst + i * stride, xv[i]);
    }
  }
}
#undef SEGF_STEP

// six-term product of one K step into the two accumulators, alternating (D0: hi.hi and the 2^-11 terms, D1: the 2^-22 terms)
#if defined(X3_TWO_LIMBS)
// TWO limbs per operand (hi = fp16(x), mid = fp16(x - hi): |x - hi - mid| <= 2^-22 |x|) and the three products hi.hi + hi.mid + mid.hi in ONE accumulator: 14 MFMAs per tile.
// The fourth product mid.mid is <= 2^-22 relative - the size of the operands' own truncation - and is dropped (it rides for free in the packed K = 8 tail, where
// {W_hi, W_mid} x {h_hi, h_mid} is one MFMA either way); a product is off by <= 3 * 2^-22, a K = 72 dot product by less than the classical fp32 bound 72 * 2^-24
#define X3_SUM(r) D0[r]                      /* one accumulator: nothing to fold */
#define X3_TAIL3(a_lh, a_hm) (void)(a_lh); D0 = MFMA16(a_hm, HT.mh, D0); D0 = MFMA16(a_hm, HT.hm, D0);
#ifdef X3_KEEP_MIDMID      /* (variant builds, tools/build_variant2.sh: the fourth product mid.mid as well - 18 MFMAs per tile, +10 % conv time, no measurable accuracy) */
#define X3_MM(MF, am, bm) D0 = MF(am, bm, D0);
#else
#define X3_MM(MF, am, bm)
#endif
#define X3_STEP(MF, ah, am, al, bh, bm, bl)   \
  (void)(al);                                 \
  X3_MM(MF, am, bm)                           \
  D0 = MF(ah, bm, D0);                        \
  D0 = MF(am, bh, D0);                        \
  D0 = MF(ah, bh, D0);
#elif defined(ONE_ACC)
#define X3_SUM(r) (D0[r] + D1[r])
#define X3_TAIL3(a_lh, a_hm) D0 = MFMA16(a_lh, HT.hl, D0); D0 = MFMA16(a_hm, HT.mh, D0); D0 = MFMA16(a_hm, HT.hm, D0);
#define X3_STEP(MF, ah, am, al, bh, bm, bl)   \
  D0 = MF(ah, bl, D0);                        \
  D0 = MF(al, bh, D0);                        \
  D0 = MF(am, bm, D0);                        \
  D0 = MF(ah, bm, D0);                        \
  D0 = MF(am, bh, D0);                        \
  D0 = MF(ah, bh, D0);
#else
#define X3_SUM(r) (D0[r] + D1[r])
#define X3_TAIL3(a_lh, a_hm) D0 = MFMA16(a_hm, HT.mh, D0); D1 = MFMA16(a_lh, HT.hl, D1); D0 = MFMA16(a_hm, HT.hm, D0);
#define X3_STEP(MF, ah, am, al, bh, bm, bl)   \
  D1 = MF(ah, bl, D1);                        \
  D0 = MF(ah, bm, D0);                        \
  D1 = MF(al, bh, D1);                        \
  D0 = MF(ah, bh, D0);                        \
  D1 = MF(am, bm, D1);                        \
  D0 = MF(am, bh, D0);
#endif

// scalar-accumulator tensor-product epilogue of one W2 tile (wave-uniform branch on the tile kind; the f16 pipe does not compete with the VALU
// for issue: fewer registers beat fewer instructions here).  T_RTS: only rows j = 0,1 belong to the column that is about to be flushed.
constexpr int TVQ = 3;      // a vector column has nv = 6 channels = three accumulator quads (quad 3: nothing, or the packed extra unit)
// the tensor product of a vector tile: rows 0,1 / rows 2,3 into the "times s0" set accV or the "cross v" set accX (X01 / X23)
template <bool X01, bool X23>
__device__ __forceinline__ void tv_rows(const f32x16& D, f32x4 f0, f32x4 f1, f32x4 f2, float (&accV)[4][3], float (&accX)[4][3]) {
#pragma unroll
  for (int rq = 0; rq < TVQ; ++rq) {
    const float d0 = D[4 * rq], d1 = D[4 * rq + 1], d2 = D[4 * rq + 2], d3 = D[4 * rq + 3];
    float (&a01)[3] = X01 ? accX[rq] : accV[rq];
    a01[0] = fmaf(f0.x, d0, fmaf(f0.y, d1, a01[0]));
    a01[1] = fmaf(f1.x, d0, fmaf(f1.y, d1, a01[1]));
    a01[2] = fmaf(f2.x, d0, fmaf(f2.y, d1, a01[2]));
    float (&a23)[3] = X23 ? accX[rq] : accV[rq];
    a23[0] = fmaf(f0.z, d2, fmaf(f0.w, d3, a23[0]));
    a23[1] = fmaf(f1.z, d2, fmaf(f1.w, d3, a23[1]));
    a23[2] = fmaf(f2.z, d2, fmaf(f2.w, d3, a23[2]));
  }
}
// ... all four rows into the l = 2 set accY (confidence model: the six p or six q rows contracted with v^ v^T - |v^|^2 I/3 when the column is flushed)
__device__ __forceinline__ void tv_rows_l2(const f32x16& D, f32x4 f0, f32x4 f1, f32x4 f2, float (&accY)[3][3]) {
#pragma unroll
  for (int rq = 0; rq < TVQ; ++rq) {
    const float d0 = D[4 * rq], d1 = D[4 * rq + 1], d2 = D[4 * rq + 2], d3 = D[4 * rq + 3];
    accY[rq][0] = fmaf(f0.x, d0, fmaf(f0.y, d1, fmaf(f0.z, d2, fmaf(f0.w, d3, accY[rq][0]))));
    accY[rq][1] = fmaf(f1.x, d0, fmaf(f1.y, d1, fmaf(f1.z, d2, fmaf(f1.w, d3, accY[rq][1]))));
    accY[rq][2] = fmaf(f2.x, d0, fmaf(f2.y, d1, fmaf(f2.z, d2, fmaf(f2.w, d3, accY[rq][2]))));
  }
}
template <int MODE>
__device__ __forceinline__ void tile_epilogue_s(int w0, const f32x16& D, const float* Fp, f32x4 f0, float (&accA)[4], float (&accV)[4][3],
                                                float (&accX)[4][3], float (&accY)[3][3]) {
  const int kind = w0 & 3;
  if (MODE == 1 && kind == T_TV && (w0 & X_TILE_L2)) {
    const f32x4 f1 = ldv4(Fp + 4), f2 = ldv4(Fp + 8);
    tv_rows_l2(D, f0, f1, f2, accY);
  } else if (kind == T_TV) {
    // raw p / q rows: rows whose product is "times s0" accumulate into accV, rows that are crossed with v into accX (bits 14 / 15 of the tile
    // word: rows j = 0,1 / j = 2,3 are cross rows); both factors are applied when the column is flushed
    const f32x4 f1 = ldv4(Fp + 4), f2 = ldv4(Fp + 8);      // y / z components of the 4 feature rows
    // one specialised body per (rows 0,1 cross?, rows 2,3 cross?) combination: with a run-time choice of the accumulator set inside ONE body the
    // compiler computes into temporaries and picks the set with 48 v_cndmask per tile
    switch ((w0 >> 14) & 3) {
      case 0: tv_rows<false, false>(D, f0, f1, f2, accV, accX); break;
      case 1: tv_rows<true, false>(D, f0, f1, f2, accV, accX); break;
      case 2: tv_rows<false, true>(D, f0, f1, f2, accV, accX); break;
      default: tv_rows<true, true>(D, f0, f1, f2, accV, accX); break;
    }
  } else if (kind == T_RA) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      accA[rq] = fmaf(f0.x, D[4 * rq], fmaf(f0.y, D[4 * rq + 1], fmaf(f0.z, D[4 * rq + 2], fmaf(f0.w, D[4 * rq + 3], accA[rq]))));
  } else if (kind == T_RT) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      accV[rq][0] = fmaf(f0.x, D[4 * rq], fmaf(f0.y, D[4 * rq + 1], fmaf(f0.z, D[4 * rq + 2], fmaf(f0.w, D[4 * rq + 3], accV[rq][0]))));
  } else {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) accV[rq][0] = fmaf(f0.x, D[4 * rq], fmaf(f0.y, D[4 * rq + 1], accV[rq][0]));
  }
}

// TRACE: workgroup 0 stamps s_memtime at the four edges of every tile's two half phases (slots 0-3) and, in the first tile's record of a unit, at
// four points of the unit's prologue (slots 4-7: unit start, indices + ring staging done, GEMM1 done, limbs + F rows done); ddk_debug_conv_trace,
// tools/conv_trace.py.  Every stamp costs the wave ~100 cycles (s_memtime round trip): read the spans as upper bounds.
// MODE 1: the confidence model's l <= 2 tensor product (a third accumulator set for the 1o(x)2e / 1e(x)2e row groups, nine edge groups, three accumulator slots)
#ifndef X_PROLOGUE_TILES
#define X_PROLOGUE_TILES 6.5f
#endif
template <bool GATHER, bool SPLIT, bool DET, bool TRACE = false, int MODE = 0>
__global__ __launch_bounds__(64 * CONV_WAVES) void conv_x3_kernel(ConvXArgs AX) {
  static_assert(!SPLIT || GATHER, "the GEMM1 split exists for the gather path");
  static_assert(MODE == 0 || (!SPLIT && !DET && !TRACE), "the confidence model runs the plain paths");
  int trace_n = 0;
  auto stamp = [&](int phase) {
    if constexpr (TRACE) {
      if (AX.trace_coarse == 1 && phase < 4) return;
      if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && trace_n < CONV_TRACE_TILES)
        AX.trace[((threadIdx.x >> 6) * CONV_TRACE_TILES + trace_n) * 8 + phase] = (uint32_t)__builtin_amdgcn_s_memtime();
      if (phase == 3) ++trace_n;
    }
  };
  bool first_tile = true;
  auto stamp_epi = [&](int slot) {     // trace mode 2: four stamps INSIDE the epilogue (slots 4-7 of every tile but a unit's first, whose slots hold the prologue)
    if constexpr (TRACE) {
      if (AX.trace_coarse != 2 || first_tile) return;
      if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && trace_n < CONV_TRACE_TILES)
        AX.trace[((threadIdx.x >> 6) * CONV_TRACE_TILES + trace_n) * 8 + slot] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
  };
  auto stamp_unit = [&](int slot, int value) {      // coarse trace: slot 0 / 2 take the time, slot 1 the given value; slot 2 closes the record
    if constexpr (TRACE) {
      if (AX.trace_coarse != 1) return;
      if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && trace_n < CONV_TRACE_TILES)
        AX.trace[((threadIdx.x >> 6) * CONV_TRACE_TILES + trace_n) * 8 + slot] = slot == 1 ? (uint32_t)value : (uint32_t)__builtin_amdgcn_s_memtime();
      if (slot == 2) ++trace_n;
    }
  };
  const ConvKArgs& A = AX.k;
  constexpr int WAVES = CONV_WAVES, FS = FX_STRIDE, BLOCK_EDGES = 32 * WAVES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* F = lds + wave * (32 * FS);                                   // this wave's 32 F rows
  char* ring = reinterpret_cast<char*>(lds + WAVES * (32 * FS));       // [W2X_STAGES][W2X_TILE_BYTES]
  int* blk_slot = reinterpret_cast<int*>(ring + W2X_STAGES * W2X_TILE_BYTES);
  const int el = lane & 31;
  const int hh = lane >> 5;
  const bool g2_shared = A.sum_g2 != nullptr;                 // group 2 is the shared rec-rec copy (layer-0 de-duplication): its own accumulator
  // work unit: a block of BLOCK_EDGES consecutive edges of ONE edge group (its radial-MLP weights are shared by the workgroup); lane g < n_active
  // keeps group g's edge range and its block range of the work queue; a block index is mapped to its group with one ballot
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
  int bs4 = __builtin_amdgcn_readlane(pend_v, 15);
  if constexpr (DET) {
    if (A.det_nr > 0) bs4 = A.det_rng[A.det_nr];      // sample-aligned units: the work queue walks the ranges' blocks (k_conv_common.h)
  }
  float* Fr = F + el * FS;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  const int n_tiles = A.n_tiles;
  constexpr int REC16 = W2X_TILE_BYTES / 16;                   // 873 x 16 B per tile record
  const int grp = wave >> 2;                                    // half-group: 0 = waves 0-3 (A), 1 = waves 4-7 (B, their SIMD partners)
  // every thread moves two 16-B chunks of a tile record: chunk tid and chunk tid + 512 (threads past the record's end move its last chunk
  // again: same bytes to the same place, no branch in the burst)
  const uint32_t fo0 = 16u * tid, fo1 = 16u * min(tid + 64 * WAVES, REC16 - 1);
  // the hot instantiation (score-model conv layers: gather + node terms, atomics) runs its tile epilogue as ONE asm statement (k_conv_x_epi_gen.inc): the
  // accumulators are in/out operands of every tile kind's code alike, so the compiler has nothing to copy or select at a merge (VERDICT r04 #1a)
#ifdef X3_CXX_EPI          // (A/B switch: the compiler-scheduled epilogue everywhere)
  constexpr bool ASM_EPI = false;
#else
  constexpr bool ASM_EPI = GATHER && SPLIT && !DET && MODE == 0;     // (the trace build too: its stamps 4-6 inside the epilogue exist only under X3_CXX_EPI)
#endif
  const unsigned ring0_u = (unsigned)(WAVES * 32 * FS * 4), ringb_u = ring0_u + 64u * hh, ringw0_u = ring0_u + fo0, ringw1_u = ring0_u + fo1;
  const unsigned fra_u = (unsigned)((wave * 32 + el) * FS * 4), hh4_u = 4u * hh, hh12_u = 12u * hh;
  int t0_, t1_, t2_, sel_, pk0_, pk1_, pk2_;
  unsigned long long sv_;
  (void)ringb_u; (void)ringw0_u; (void)ringw1_u; (void)fra_u; (void)hh4_u; (void)hh12_u;

  // work units: whole blocks, except that the last (bs4 mod #workgroups) blocks are split into column chunks so that the final round of the
  // persistent workgroups is a fraction of a block long (same rule as k_conv.hip)
  const int nwg = gridDim.x;
  const int full = bs4 >= nwg ? (bs4 / nwg) * nwg : 0;
  const int rest = bs4 - full;
  int split = 1;
  if (rest > 0) {
    float best = 1e30f;
    // rounds of the last blocks' chunks x the length of a chunk in tile periods: its share of the tiles + the unit prologue every chunk repeats
    // (gathers, GEMM1, limbs: ~17 k cycles = X_PROLOGUE_TILES tile periods; round 3's rule charged 5 % of a block per extra chunk - a third of the
    // real cost at 59 tiles - and cut the small launches of the late reverse steps into too many chunks)
    for (int sp = 1; sp <= A.n_cols; ++sp) {
      const float cost = (float)((rest * sp + nwg - 1) / nwg) * ((float)n_tiles / (float)sp + X_PROLOGUE_TILES);
      if (cost < best - 1e-6f) { best = cost; split = sp; }
    }
  }
  const int n_units = full + rest * split;

  int unit = blockIdx.x;
  for (;;) {
    if (unit >= n_units) break;
    stamp(4);
    first_tile = true;
    int unit_next = 0;
    if (tid == 0) unit_next = nwg + atomicAdd(A.counter, 1);      // fetched at the START of this unit: the round trip hides under the tile loop
    int blk = unit, t_begin = 0, t_end = n_tiles;
    if (unit >= full) {
      const int r = unit - full, c = r % split;
      blk = full + r / split;
      t_begin = A.col_start[(c * A.n_cols) / split];
      t_end = A.col_start[((c + 1) * A.n_cols) / split];
    }
    int g = __popcll(__ballot(lane < 16 && blk >= pend_v));
    int gbeg = __builtin_amdgcn_readlane(gb_v, g), gend = __builtin_amdgcn_readlane(ge_v, g);
    int bstart = __builtin_amdgcn_readlane(pbeg_v, g);
    if constexpr (DET) {
      if (A.det_nr > 0) {
        const DetRange R_ = det_find(A.det_rng, A.det_nr, blk);
        g = R_.g; gbeg = R_.beg; gend = R_.end; bstart = R_.bstart;
      }
    }
    const int e0 = gbeg + BLOCK_EDGES * (blk - bstart) + 32 * wave;
    const int nvalid = min(32, gend - e0);                    // <= 0: this wave's slice lies past the end of the group
    const bool valid = el < nvalid;
    const int e = nvalid > 0 ? e0 + min(el, nvalid - 1) : gend - 1;
    const int sn = A.src[e], dn = A.dst[e];

    // ---- stage the first three W2 tiles of this unit (the ring is idle: the previous unit ended with a barrier) ----
    const int gw = (int)((A.wmap >> (4 * g)) & 15);                 // weight set / node-term roles of this group
    const char* wrec = reinterpret_cast<const char*>(A.w2x) + (size_t)gw * n_tiles * W2X_TILE_BYTES;
    {
      const char* wr0 = wrec + (size_t)t_begin * W2X_TILE_BYTES;
      const char* wr1 = wrec + (size_t)min(t_begin + 1, t_end - 1) * W2X_TILE_BYTES;
      const char* wr2 = wrec + (size_t)min(t_begin + 2, t_end - 1) * W2X_TILE_BYTES;
      const float4 r0 = *reinterpret_cast<const float4*>(wr0 + fo0), r1 = *reinterpret_cast<const float4*>(wr0 + fo1);
      const float4 r2 = *reinterpret_cast<const float4*>(wr1 + fo0), r3 = *reinterpret_cast<const float4*>(wr1 + fo1);
      const float4 r4 = *reinterpret_cast<const float4*>(wr2 + fo0), r5 = *reinterpret_cast<const float4*>(wr2 + fo1);
      *reinterpret_cast<float4*>(ring + fo0) = r0;
      *reinterpret_cast<float4*>(ring + fo1) = r1;
      *reinterpret_cast<float4*>(ring + W2X_TILE_BYTES + fo0) = r2;
      *reinterpret_cast<float4*>(ring + W2X_TILE_BYTES + fo1) = r3;
      *reinterpret_cast<float4*>(ring + 2 * W2X_TILE_BYTES + fo0) = r4;
      *reinterpret_cast<float4*>(ring + 2 * W2X_TILE_BYTES + fo1) = r5;
    }

    // ---- segmented-scan control words (identical for every output channel of this wave's 32 edges) ----
    const SegF seg = make_segf(make_segctl(sn, el, nvalid, valid));
    stamp(5);

    // ---- the F row's inputs (x[dst] row, sh), requested here so that their latency runs under GEMM1 ----
    const float4 shv = ld4(A.sh + (size_t)e * 4);
    float4 mainv[NS / 4];
    float2 pv2[3 * NV / 2];
    {
#ifdef X3_ABL_NOGATHER      // (timing-only: every lane reads the rows of the wave's FIRST edge - one cache line per request instead of 32: what would hiding the gathers be worth?)
      const float* xr = A.x + (size_t)__shfl(dn, 0, 32) * XW;
#else
      const float* xr = A.x + (size_t)dn * XW;
#endif
#pragma unroll
      for (int j = 0; j < NS / 4; ++j) mainv[j] = ld4(xr + (hh ? OFF_C : 0) + 4 * j);
#pragma unroll
      for (int j = 0; j < 3 * NV / 2; ++j) pv2[j] = ld2(xr + (hh ? OFF_Q : OFF_P) + 2 * j);
    }

    // ---- GEMM1: h = relu(W1 [edge_emb | x_src[:ns] | x_dst[:ns]] + b1), K order kappa(s,hh) = 24*(s/12)+12*hh+s%12, three-limb product ----
    Limbs H;
    TailB HT;
    float osc, bsc2;      // GEMM2 accumulators hold (s2 h) x (w2s W2): bias goes in times bsc2, flushed sums come out times osc
    {
      float h[36];
      const char* w1 = reinterpret_cast<const char*>(A.w1x) + (size_t)gw * 3 * W1X_TILE_BYTES;
      if constexpr (SPLIT) {
        // W1a edge_emb on top of the per-node terms (W1b x[src][:ns] + b1) + W1c x[dst][:ns] (node_finalize_pre_kernel), which arrive in the
        // accumulator's own register order: K = 24 = one step of 16 + one of 8.  The node terms sit at random nodes of a 15 MB array (past the
        // L2): all 18 requests of a lane go out together, right behind the indices - one exposed memory latency, not one per row tile
        float4 psv[9], pdv[9];
        {
#ifdef X3_ABL_NOGATHER
          const float* ps = A.pre + ((size_t)__shfl(sn, 0, 32) * 4 + (gw & 1)) * NE + 36 * hh;
          const float* pd = A.pre + ((size_t)__shfl(dn, 0, 32) * 4 + 2 + (gw >> 1)) * NE + 36 * hh;
#else
          const float* ps = A.pre + ((size_t)sn * 4 + (gw & 1)) * NE + 36 * hh;
          const float* pd = A.pre + ((size_t)dn * 4 + 2 + (gw >> 1)) * NE + 36 * hh;
#endif
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
        m1 = fmaxf(m1, __shfl_xor(m1, 32));          // both lane halves hold K slices of the same edge
        float inv1;
        const float s1 = range_scale(m1, inv1);
        f16x8 b0h, b0m, b0l;
        f16x4 b1h, b1m, b1l;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const Limb3 q = split3(bin[i] * s1); b0h[i] = q.h; b0m[i] = q.m; b0l[i] = q.l; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const Limb3 q = split3(bin[8 + i] * s1); b1h[i] = q.h; b1m[i] = q.m; b1l[i] = q.l; }
        const f16x8 b1_mh = __builtin_shufflevector(b1m, b1h, 0, 1, 2, 3, 4, 5, 6, 7), b1_hl = __builtin_shufflevector(b1h, b1l, 0, 1, 2, 3, 4, 5, 6, 7),
                    b1_hm = __builtin_shufflevector(b1h, b1m, 0, 1, 2, 3, 4, 5, 6, 7);
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
#ifdef X3_ABL_NOW1      // (timing-only ablation: no W1 fragment loads)
            f16x8 ah, am, al;
            for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(float)(lane + T); am[i] = (_Float16)(float)(lane + 2 * T); al[i] = (_Float16)(float)(lane + 3 * T); }
#else
            const f16x8 ah = *reinterpret_cast<const f16x8*>(wt + lane * 16);
            const f16x8 am = *reinterpret_cast<const f16x8*>(wt + W2X_LIMB_BYTES + lane * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(wt + 2 * W2X_LIMB_BYTES + lane * 16);
#endif
            X3_STEP(MFMA16, ah, am, al, b0h, b0m, b0l)
          }
          {     // registers 8..11: the first half of step 1's fragment
#ifdef X3_ABL_NOW1
            f16x4 ah, am, al;
            for (int i = 0; i < 4; ++i) { ah[i] = (_Float16)(float)(lane + T + 1); am[i] = (_Float16)(float)(lane + 2 * T + 1); al[i] = (_Float16)(float)(lane + 3 * T + 1); }
#else
            const f16x4 ah = *reinterpret_cast<const f16x4*>(wt + 1024 + lane * 16);
            const f16x4 am = *reinterpret_cast<const f16x4*>(wt + W2X_LIMB_BYTES + 1024 + lane * 16);
            const f16x4 al = *reinterpret_cast<const f16x4*>(wt + 2 * W2X_LIMB_BYTES + 1024 + lane * 16);
#endif
            // (the K = 8 half step packed like the tile tail: hi.mid + mid.hi, lo.hi + hi.lo, hi.hi + mid.mid as one K = 16 MFMA each: 9 instead of 12 per row tile)
            const f16x8 a_hm = __builtin_shufflevector(ah, am, 0, 1, 2, 3, 4, 5, 6, 7), a_lh = __builtin_shufflevector(al, ah, 0, 1, 2, 3, 4, 5, 6, 7);
            D0 = MFMA16(a_hm, b1_mh, D0);
#ifndef X3_TWO_LIMBS
            D1 = MFMA16(a_lh, b1_hl, D1);
#else
            (void)a_lh; (void)b1_hl;
#endif
            D0 = MFMA16(a_hm, b1_hm, D0);
          }
          const int nr = T < 2 ? 16 : 4;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r < nr) h[16 * T + r] = fmaxf(X3_SUM(r), 0.0f) * usc;
        }
      } else {
        float bin[36];
        {
          const float *pe, *pxs, *pxd;
          if (GATHER) {
            pe = A.edge_attr + (size_t)e * NS + 12 * hh;
            pxs = A.x + (size_t)sn * XW + 12 * hh;
            pxd = A.x + (size_t)dn * XW + 12 * hh;
          } else {
            pe = A.edge_attr + (size_t)e * NE + 12 * hh;
            pxs = pe + NS;
            pxd = pe + 2 * NS;
          }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float4 a = ld4(pe + 4 * j), b = ld4(pxs + 4 * j), c = ld4(pxd + 4 * j);
            bin[4 * j + 0] = a.x; bin[4 * j + 1] = a.y; bin[4 * j + 2] = a.z; bin[4 * j + 3] = a.w;
            bin[12 + 4 * j + 0] = b.x; bin[12 + 4 * j + 1] = b.y; bin[12 + 4 * j + 2] = b.z; bin[12 + 4 * j + 3] = b.w;
            bin[24 + 4 * j + 0] = c.x; bin[24 + 4 * j + 1] = c.y; bin[24 + 4 * j + 2] = c.z; bin[24 + 4 * j + 3] = c.w;
          }
        }
        float m1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 36; ++j) m1 = fmaxf(m1, fabsf(bin[j]));
        m1 = fmaxf(m1, __shfl_xor(m1, 32));
        float inv1;
        const float s1 = range_scale(m1, inv1);
        Limbs Bn;
        make_limbs(Bn, bin, s1);
        const float bsc = s1 * A.w1s[gw], usc = inv1 * A.w1u[gw];
        const float* b1 = A.b1p + (size_t)gw * (3 * 2 * 16);
#pragma unroll
        for (int T = 0; T < 3; ++T) {
          f32x16 D0, D1;
          const float* bp = b1 + (T * 2 + hh) * 16;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 b = ld4(bp + 4 * j);
            D0[4 * j + 0] = b.x * bsc; D0[4 * j + 1] = b.y * bsc; D0[4 * j + 2] = b.z * bsc; D0[4 * j + 3] = b.w * bsc;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) D1[r] = 0.0f;
          const char* wt = w1 + (size_t)T * W1X_TILE_BYTES;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(wt + s * 1024 + lane * 16);
            const f16x8 am = *reinterpret_cast<const f16x8*>(wt + W2X_LIMB_BYTES + s * 1024 + lane * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(wt + 2 * W2X_LIMB_BYTES + s * 1024 + lane * 16);
            X3_STEP(MFMA16, ah, am, al, Bn.hi[s], Bn.mid[s], Bn.lo[s])
          }
          {
            const f16x4 ah = *reinterpret_cast<const f16x4*>(wt + 4096 + lane * 8);
            const f16x4 am = *reinterpret_cast<const f16x4*>(wt + W2X_LIMB_BYTES + 4096 + lane * 8);
            const f16x4 al = *reinterpret_cast<const f16x4*>(wt + 2 * W2X_LIMB_BYTES + 4096 + lane * 8);
            X3_STEP(MFMA8, ah, am, al, Bn.thi, Bn.tmid, Bn.tlo)
          }
          const int nr = T < 2 ? 16 : 4;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r < nr) h[16 * T + r] = fmaxf(X3_SUM(r), 0.0f) * usc;
        }
      }
      float m2 = 0.0f;
#pragma unroll
      for (int j = 0; j < 36; ++j) m2 = fmaxf(m2, h[j]);
      m2 = fmaxf(m2, __shfl_xor(m2, 32));
      float inv2;
      const float s2 = range_scale(m2, inv2);
      stamp(6);
      make_limbs(H, h, s2);
      HT = make_tailb(H);
      bsc2 = s2 * A.w2s[gw];
      osc = inv2 * A.w2u[gw];
    }

    // ---- F row of this edge: TP row operands derived from x[dst] and sh (written by both lane halves) ----
    const float s0 = shv.x, vx = shv.y, vy = shv.z, vz = shv.w;
    {
      // half 0: a -> FX_A, p: (p.v)/sqrt3 -> FX_PQ, raw p -> rows 0..nv-1 of FX_R;  half 1: c -> FX_C, q: (q.v)/sqrt3 -> FX_PQ, raw q -> rows nv..
      const int o_main_dst = hh ? FX_C : FX_A, r0 = hh ? NV : 0;
#pragma unroll
      for (int j = 0; j < NS / 4; ++j) *reinterpret_cast<float4*>(Fr + o_main_dst + 4 * j) = mainv[j];
      float pv[3 * NV];
#pragma unroll
      for (int j = 0; j < 3 * NV / 2; ++j) { pv[2 * j] = pv2[j].x; pv[2 * j + 1] = pv2[j].y; }
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const float px = pv[3 * m], py = pv[3 * m + 1], pz = pv[3 * m + 2];
        // FX_PQ = [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]
        Fr[FX_PQ + (m < 4 ? 4 * hh + m : 8 + 2 * hh + (m - 4))] = (px * vx + py * vy + pz * vz) * inv_s3;
        // row r of the 12 raw rows lives at 12*(r/4) + 4*c + r%4 (component-major inside a quad of rows)
        const int r = r0 + m;
        float* Pr = Fr + FX_R + 12 * (r >> 2) + (r & 3);
        Pr[0] = px;
        Pr[4] = py;
        Pr[8] = pz;
      }
    }
    stamp(7);
    lds_barrier();   // ring stages 0 / 1 and the F rows are visible

    // ---- GEMM2 over the W2 tiles + fused tensor-product epilogue ----
    float* node_row = (g2_shared && g == 2) ? A.sum_g2 + (size_t)(sn - A.g2_node_off) * XW
                                            : A.sum + ((size_t)sn * A.n_slots + ((A.slots >> (2 * g)) & 3)) * XW;
    if constexpr (DET) {
      // runs of equal edge_src are contiguous inside a group: only the tile's first / last run can continue in the neighbouring tile
      if (nvalid > 0) {
        const int sn0 = __shfl(sn, 0, 32), snl = __shfl(sn, nvalid - 1, 32);
        const bool first_cont = e0 > gbeg && A.src[e0 - 1] == sn0;
        const bool last_cont = e0 + nvalid < gend && A.src[e0 + nvalid] == snl;
        float* prow = A.part + ((size_t)(blk * WAVES + wave) * 2) * XW;    // this tile's two partial rows (tile id = global block index x 8 + wave)
        if (sn == sn0 && first_cont) node_row = prow;
        else if (sn == snl && last_cont) node_row = prow + XW;
      }
    }
    // (asm epilogue) the node row as a 32-bit byte offset from the accumulator array, run tails as an exec mask, lanes past the group's end scaled to zero
    const bool shared2_ = g2_shared && g == 2;
    const float* sumbase = shared2_ ? A.sum_g2 : A.sum;
    const unsigned vrow = shared2_ ? (unsigned)(sn - A.g2_node_off) * (unsigned)(XW * 4)
                                   : ((unsigned)sn * (unsigned)A.n_slots + ((A.slots >> (2 * g)) & 3u)) * (unsigned)(XW * 4);
    const unsigned long long tail_mask = __ballot(seg.tail);
    const float oscv = seg.valid ? osc : 0.0f;
    (void)sumbase; (void)vrow; (void)tail_mask; (void)oscv;
    float accA[4], accV[4][3], accX[4][3];      // accX: sums over the rows that are crossed with v (vector columns)
    float accY[3][3];                           // MODE 1: sums over the rows of the l = 2 groups
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f; accX[rq][0] = 0.0f; accX[rq][1] = 0.0f; accX[rq][2] = 0.0f;
      if (rq < 3) { accY[rq][0] = 0.0f; accY[rq][1] = 0.0f; accY[rq][2] = 0.0f; }
    }
    // ---- per-lane ring addresses: with the tile loop unrolled over the four ring stages every ring access is ONE base register + an immediate
    // (the ring spans 4 x 13,968 B < 2^16: the 16-bit ds offset reaches all of it), and the records of tile t+3 come through a buffer
    // descriptor (per-lane offset in a VGPR, the tile's offset in ONE SGPR that advances by a record per tile) ----
    const char* ringl = ring + lane * 16;       // 16-B fragment reads
    const char* ringt = ring + lane * 8;        // 8-B tail-fragment reads
    const char* ringb = ring + hh * 64;         // the lane half's 16 bias floats
    char* ringw0 = ring + fo0;                  // this thread's two chunks of a record
    char* ringw1 = ring + fo1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrec), 0, 0x7fffffff, 0x00020000);
    int rec_soff = (t_begin + 3) * W2X_TILE_BYTES;      // (records past a group's last tile exist: the next group's, or the three pad records)
    // the first K step of the first tile and its descriptor (the two words ride behind the bias in the ring record: an LDS read, not a scalar
    // load - a scalar load in flight turns every counted LDS wait into a full one); later tiles: fetched in the previous burst's tail
    Frag16 p0;
    p0.h = *reinterpret_cast<const f16x8*>(ringl); p0.m = *reinterpret_cast<const f16x8*>(ringl + W2X_LIMB_BYTES);
#ifndef X3_TWO_LIMBS
    p0.l = *reinterpret_cast<const f16x8*>(ringl + 2 * W2X_LIMB_BYTES);
#endif
#ifdef X3_PF2          // (experiment: fragments requested TWO K steps ahead: p0 / p1 of a tile in the previous burst's last two steps, +12 VGPRs across the epilogue)
    Frag16 p1;
    p1.h = *reinterpret_cast<const f16x8*>(ringl + 1024); p1.m = *reinterpret_cast<const f16x8*>(ringl + W2X_LIMB_BYTES + 1024); p1.l = *reinterpret_cast<const f16x8*>(ringl + 2 * W2X_LIMB_BYTES + 1024);
#endif
    int w0, chan0;
    {
      const int2 dq = *reinterpret_cast<const int2*>(ring + W2X_DESC_OFF);
      w0 = __builtin_amdgcn_readfirstlane(dq.x); chan0 = __builtin_amdgcn_readfirstlane(dq.y);
    }
    const bool grp_s = __builtin_amdgcn_readfirstlane(grp) != 0;      // provably wave-uniform: the barrier sits behind a scalar branch
    const int grp_i = __builtin_amdgcn_readfirstlane(grp); (void)grp_i;
    // (asm epilogue generated with GEN_DEFER=1 - an experiment, default off) group A's flush runs one tile late, in the interval in which group B flushes the same
    // column: the column's values, its kind (0 none / 1 scalar / 2 vector) and its channel offset wait here
    float P_[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int pend_ = 0, pchan_ = 0;
    (void)P_; (void)pend_; (void)pchan_;
    // the rare part of an epilogue: the packed extra quad of a 6-channel column and the flush of a column the tile closes
    auto finish_tile = [&](int w0x, int chan0x, const f32x16& D, f32x4 f0x) {
      if (w0x & 0x80) {   // 6-channel column: accumulator quad 3 holds another a / c row quad for channel pair xp
        const f32x4 g0 = ldv4(Fr + ((w0x >> 8) & 0x3c));
        const float xv = fmaf(g0.x, D[12], fmaf(g0.y, D[13], fmaf(g0.z, D[14], g0.w * D[15])));
        const int xp = (w0x >> 8) & 3;
        if (xp == 0) accA[0] += xv; else if (xp == 1) accA[1] += xv; else accA[2] += xv;
      }
      const int fl = (w0x >> 2) & 3;
      if (fl) {
        const int nrq = (w0x >> 4) & 7;
        if (fl == FL_S && nrq == 4) {   // a full scalar column: its four channels in one pass
          float m4[4];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) m4[rq] = osc * fmaf(accA[rq], s0, accV[rq][0]);
          segf_add_n<DET, 4>(node_row + chan0x + hh, 2, m4, seg);
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          if (rq < nrq && !(fl == FL_S && nrq == 4)) {
            if (fl == FL_S) {
              float m1v[1] = {osc * fmaf(accA[rq], s0, accV[rq][0])};
              segf_add_n<DET, 1>(node_row + chan0x + 2 * rq + hh, 1, m1v, seg);
            } else {
              float* d = node_row + chan0x + 3 * (2 * rq + hh);
              // a (x) v  +  s0 * (sum of the "times s0" rows)  +  (sum of the cross rows) x v / sqrt2
              const float sa = accA[rq], wx = vx * inv_s2, wy = vy * inv_s2, wz = vz * inv_s2;
              const float X0 = accX[rq][0], X1 = accX[rq][1], X2 = accX[rq][2];
              float m3[3] = {osc * fmaf(X1, wz, fmaf(-X2, wy, fmaf(sa, vx, s0 * accV[rq][0]))),
                             osc * fmaf(X2, wx, fmaf(-X0, wz, fmaf(sa, vy, s0 * accV[rq][1]))),
                             osc * fmaf(X0, wy, fmaf(-X1, wx, fmaf(sa, vz, s0 * accV[rq][2])))};
              if constexpr (MODE == 1) {
                // + (v^ v^T - |v^|^2 I/3) y with v = sqrt3 v^: (y.v / 3) v - (|v|^2 / 9) y  (|v^| is 1, or 0 for a zero-length edge); constants folded into the weights
                const float Y0 = accY[rq < 3 ? rq : 0][0], Y1 = accY[rq < 3 ? rq : 0][1], Y2 = accY[rq < 3 ? rq : 0][2];
                const float dv = (Y0 * vx + Y1 * vy + Y2 * vz) * (1.0f / 3.0f), n3 = (vx * vx + vy * vy + vz * vz) * (1.0f / 9.0f);
                m3[0] += osc * (dv * vx - Y0 * n3); m3[1] += osc * (dv * vy - Y1 * n3); m3[2] += osc * (dv * vz - Y2 * n3);
              }
              segf_add_n<DET, 3>(d, 1, m3, seg);
            }
          }
          accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f; accX[rq][0] = 0.0f; accX[rq][1] = 0.0f; accX[rq][2] = 0.0f;
          if (MODE == 1 && rq < 3) { accY[rq][0] = 0.0f; accY[rq][1] = 0.0f; accY[rq][2] = 0.0f; }
        }
        if ((w0x & 3) == T_RTS) {   // rows j = 2,3 of the shared tail open the next (0o) column
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) accV[rq][0] = fmaf(f0x.z, D[4 * rq + 2], f0x.w * D[4 * rq + 3]);
        }
      }
    };
    // one tile: burst + epilogue; ST = the tile's ring stage (compile time)
#define X3_PAIR(mask) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); }
#define X3_BARE { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#ifdef X3_TWO_LIMBS
#define X3_FRAG(f, off) f.h = *reinterpret_cast<const f16x8*>(ringl + (off)); f.m = *reinterpret_cast<const f16x8*>(ringl + W2X_LIMB_BYTES + (off));
#define X3_TRIO(mask) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#else
#define X3_FRAG(f, off) f.h = *reinterpret_cast<const f16x8*>(ringl + (off)); f.m = *reinterpret_cast<const f16x8*>(ringl + W2X_LIMB_BYTES + (off)); f.l = *reinterpret_cast<const f16x8*>(ringl + 2 * W2X_LIMB_BYTES + (off));
#endif
#if defined(X3_TWO_LIMBS)
/* three products per K step + the packed tail: 14 MFMAs; the same riders (two limbs of every fragment), three shadows per region */
#ifndef X3_KEEP_MIDMID
#define X3_L2_R0 X3_PAIR(0x100) X3_TRIO(0x020) X3_TRIO(0x020)
#define X3_L2_R1 X3_PAIR(0x100) X3_PAIR(0x100) X3_TRIO(0x100)
#define X3_L2_R2 X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE
#define X3_L2_R3 X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100)
#else
#define X3_L2_R0 X3_PAIR(0x100) X3_PAIR(0x020) X3_PAIR(0x100) X3_TRIO(0x020)
#define X3_L2_R1 X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100)
#define X3_L2_R2 X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE X3_BARE
#define X3_L2_R3 X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE
#endif
#define X3_BURST_BODY(ST)                                                                                                                    \
      Frag16 p1; X3_FRAG(p1, SO + 1024)                                                                                                      \
      const f32x4 f0 = ldv4(Fp);                                                                                                             \
      const u32x4 st0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo0, rec_soff, 0);                                                  \
      const u32x4 st1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo1, rec_soff, 0);                                                  \
      rec_soff += W2X_TILE_BYTES;                                                                                                            \
      X3_STEP(MFMA16, p0.h, p0.m, p0.l, H.hi[0], H.mid[0], H.lo[0])                                                                         \
      X3_L2_R0                                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      Frag16 q2; X3_FRAG(q2, SO + 2048)                                                                                                      \
      const f16x4 th = *reinterpret_cast<const f16x4*>(ringt + SO + 4096);                                                                   \
      const f16x4 tm = *reinterpret_cast<const f16x4*>(ringt + SO + W2X_LIMB_BYTES + 4096);                                                  \
      const f16x4 tl = th;                                                                                                                   \
      X3_STEP(MFMA16, p1.h, p1.m, p1.l, H.hi[1], H.mid[1], H.lo[1])                                                                         \
      X3_L2_R1                                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      Frag16 q3; X3_FRAG(q3, SO + 3072)                                                                                                      \
      X3_STEP(MFMA16, q2.h, q2.m, q2.l, H.hi[2], H.mid[2], H.lo[2])                                                                         \
      X3_L2_R2                                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      const int2 dq = *reinterpret_cast<const int2*>(ring + SN + W2X_DESC_OFF);                                                              \
      X3_FRAG(p0, SN)                                                                                                                        \
      X3_STEP(MFMA16, q3.h, q3.m, q3.l, H.hi[3], H.mid[3], H.lo[3])                                                                         \
      X3_L2_R3                                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);
#elif defined(X3_PF2)
#define X3_BURST_BODY(ST)                                                                                                                    \
      Frag16 q2; X3_FRAG(q2, SO + 2048)                                                                                                      \
      const f32x4 f0 = ldv4(Fp);                                                                                                             \
      const u32x4 st0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo0, rec_soff, 0);                                                  \
      const u32x4 st1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo1, rec_soff, 0);                                                  \
      rec_soff += W2X_TILE_BYTES;                                                                                                            \
      X3_STEP(MFMA16, p0.h, p0.m, p0.l, H.hi[0], H.mid[0], H.lo[0])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x020) X3_PAIR(0x100) X3_PAIR(0x020) X3_PAIR(0x100) X3_PAIR(0x100)                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      Frag16 q3; X3_FRAG(q3, SO + 3072)                                                                                                      \
      const f16x4 th = *reinterpret_cast<const f16x4*>(ringt + SO + 4096);                                                                   \
      const f16x4 tm = *reinterpret_cast<const f16x4*>(ringt + SO + W2X_LIMB_BYTES + 4096);                                                  \
      const f16x4 tl = *reinterpret_cast<const f16x4*>(ringt + SO + 2 * W2X_LIMB_BYTES + 4096);                                              \
      X3_STEP(MFMA16, p1.h, p1.m, p1.l, H.hi[1], H.mid[1], H.lo[1])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100)                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      const int2 dq = *reinterpret_cast<const int2*>(ring + SN + W2X_DESC_OFF);                                                              \
      X3_FRAG(p0, SN)                                                                                                                        \
      X3_STEP(MFMA16, q2.h, q2.m, q2.l, H.hi[2], H.mid[2], H.lo[2])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE X3_BARE                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      X3_FRAG(p1, SN + 1024)                                                                                                                 \
      X3_STEP(MFMA16, q3.h, q3.m, q3.l, H.hi[3], H.mid[3], H.lo[3])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE X3_BARE X3_BARE                                                                   \
      __builtin_amdgcn_sched_barrier(0);
#else
#define X3_BURST_BODY(ST)                                                                                                                    \
      Frag16 p1; X3_FRAG(p1, SO + 1024)                                                                                                      \
      const f32x4 f0 = ldv4(Fp);                                                                                                             \
      /* this thread's two chunks of record t+3, requested first: ~900 cycles until the epilogue stores them */                              \
      const u32x4 st0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo0, rec_soff, 0);                                                  \
      const u32x4 st1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)fo1, rec_soff, 0);                                                  \
      rec_soff += W2X_TILE_BYTES;                                                                                                            \
      X3_STEP(MFMA16, p0.h, p0.m, p0.l, H.hi[0], H.mid[0], H.lo[0])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x020) X3_PAIR(0x100) X3_PAIR(0x020) X3_PAIR(0x100) X3_PAIR(0x100)                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      Frag16 q2; X3_FRAG(q2, SO + 2048)                                                                                                      \
      const f16x4 th = *reinterpret_cast<const f16x4*>(ringt + SO + 4096);                                                                   \
      const f16x4 tm = *reinterpret_cast<const f16x4*>(ringt + SO + W2X_LIMB_BYTES + 4096);                                                  \
      const f16x4 tl = *reinterpret_cast<const f16x4*>(ringt + SO + 2 * W2X_LIMB_BYTES + 4096);                                              \
      X3_STEP(MFMA16, p1.h, p1.m, p1.l, H.hi[1], H.mid[1], H.lo[1])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100)                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      Frag16 q3; X3_FRAG(q3, SO + 3072)                                                                                                      \
      X3_STEP(MFMA16, q2.h, q2.m, q2.l, H.hi[2], H.mid[2], H.lo[2])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE X3_BARE X3_BARE                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      const int2 dq = *reinterpret_cast<const int2*>(ring + SN + W2X_DESC_OFF);                                                              \
      X3_FRAG(p0, SN)       /* (p0's last use was K step 0 of this burst: the next tile's first K step goes straight into its registers) */    \
      X3_STEP(MFMA16, q3.h, q3.m, q3.l, H.hi[3], H.mid[3], H.lo[3])                                                                         \
      X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_PAIR(0x100) X3_BARE X3_BARE                                                            \
      __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef X3_TWO_LIMBS
#define X3_TAIL_BARES X3_BARE X3_BARE
#else
#define X3_TAIL_BARES X3_BARE X3_BARE X3_BARE
#endif
#define X3_TILE(ST)                                                                                                                          \
    {                                                                                                                                        \
      constexpr int SO = (ST) * W2X_TILE_BYTES, SN = (((ST) + 1) & 3) * W2X_TILE_BYTES, SW = (((ST) + 3) & 3) * W2X_TILE_BYTES;              \
      /* the bursting wave wins issue arbitration against its SIMD partner's epilogue (round 3's static priority for the later-dispatched   \
         half costs 7 % once the half phases are not separated by a barrier; priority during the epilogue instead: +2.5 %) */                \
      X3_BURST_PRIO();                                                                                                                       \
      stamp(0);                                                                                                                              \
      /* ===== burst: 27 MFMAs; every other instruction rides in an MFMA shadow, pinned region by region (one K step each): the LDS reads    \
         of the fragments one step ahead, the feature rows, this thread's two chunks of record t+3 and - in the tail - the next tile's       \
         descriptor and first K step (complete in the ring since the last barrier) ===== */                                                  \
      const float* Fp = Fr + ((w0 >> 16) & 0xff);                                                                                            \
      f32x16 D0, D1;                                                                                                                         \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { D0[r] = 0.0f; D1[r] = 0.0f; }                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      X3_BURST_BODY(ST)                                                                                                                      \
      {     /* packed tail: D0 += hi.mid + mid.hi, D1 += lo.hi + hi.lo, D0 += hi.hi + mid.mid as one K = 16 MFMA each */                     \
        const f16x8 a_hm = __builtin_shufflevector(th, tm, 0, 1, 2, 3, 4, 5, 6, 7), a_lh = __builtin_shufflevector(tl, th, 0, 1, 2, 3, 4, 5, 6, 7); \
        X3_TAIL3(a_lh, a_hm)                                                                                                                 \
      }                                                                                                                                      \
      X3_TAIL_BARES                                                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
      __builtin_amdgcn_s_setprio(0);                                                                                                         \
      stamp(1);                                                                                                                              \
      if (grp_s) X3_LOOP_BARRIER(ST);          /* group B: [epilogue t-1, burst t] | barrier | [epilogue t, burst t+1] */                          \
      stamp(2);                                                                                                                              \
      /* ===== epilogue (beside the SIMD partner's burst): the tile's bias, then this thread's chunks of tile t+3 into the stage tile t-1     \
         has left (nobody reads it between barriers t and t+2) ===== */                                                                      \
      if constexpr (ASM_EPI) {                                                                                                               \
        const unsigned fpa_u = fra_u + 4u * (unsigned)((w0 >> 16) & 0xff);                                                                   \
        const int chan4_ = chan0 * 4;                                                                                                        \
        X3_EPI_##ST();                                                                                                                       \
      } else {                                                                                                                               \
      const float4 bs0 = ld4(reinterpret_cast<const float*>(ringb + SO + W2X_BIAS_OFF)), bs1 = ld4(reinterpret_cast<const float*>(ringb + SO + W2X_BIAS_OFF + 16)); \
      const float4 bs2 = ld4(reinterpret_cast<const float*>(ringb + SO + W2X_BIAS_OFF + 32)), bs3 = ld4(reinterpret_cast<const float*>(ringb + SO + W2X_BIAS_OFF + 48)); \
      *reinterpret_cast<u32x4*>(ringw0 + SW) = st0;                                                                                          \
      *reinterpret_cast<u32x4*>(ringw1 + SW) = st1;                                                                                          \
      stamp_epi(4);                                                                                                                          \
      /* the two accumulators and the bias into one */                                                                                       \
      D0[0] = fmaf(bs0.x, bsc2, X3_SUM(0)); D0[1] = fmaf(bs0.y, bsc2, X3_SUM(1)); D0[2] = fmaf(bs0.z, bsc2, X3_SUM(2)); D0[3] = fmaf(bs0.w, bsc2, X3_SUM(3)); \
      D0[4] = fmaf(bs1.x, bsc2, X3_SUM(4)); D0[5] = fmaf(bs1.y, bsc2, X3_SUM(5)); D0[6] = fmaf(bs1.z, bsc2, X3_SUM(6)); D0[7] = fmaf(bs1.w, bsc2, X3_SUM(7)); \
      D0[8] = fmaf(bs2.x, bsc2, X3_SUM(8)); D0[9] = fmaf(bs2.y, bsc2, X3_SUM(9)); D0[10] = fmaf(bs2.z, bsc2, X3_SUM(10)); D0[11] = fmaf(bs2.w, bsc2, X3_SUM(11)); \
      D0[12] = fmaf(bs3.x, bsc2, X3_SUM(12)); D0[13] = fmaf(bs3.y, bsc2, X3_SUM(13)); D0[14] = fmaf(bs3.z, bsc2, X3_SUM(14)); D0[15] = fmaf(bs3.w, bsc2, X3_SUM(15)); \
      stamp_epi(5);                                                                                                                          \
      if ((w0 & 0x8e) == 0) {        /* the common tile: scalar rows (T_RA / T_RT), no packed quad, no flush */                              \
        if (w0 & 1) {                                                                                                                        \
          _Pragma("unroll") for (int rq = 0; rq < 4; ++rq)                                                                                   \
            accV[rq][0] = fmaf(f0.x, D0[4 * rq], fmaf(f0.y, D0[4 * rq + 1], fmaf(f0.z, D0[4 * rq + 2], fmaf(f0.w, D0[4 * rq + 3], accV[rq][0])))); \
        } else {                                                                                                                             \
          _Pragma("unroll") for (int rq = 0; rq < 4; ++rq)                                                                                   \
            accA[rq] = fmaf(f0.x, D0[4 * rq], fmaf(f0.y, D0[4 * rq + 1], fmaf(f0.z, D0[4 * rq + 2], fmaf(f0.w, D0[4 * rq + 3], accA[rq])))); \
        }                                                                                                                                    \
        stamp_epi(6);                                                                                                                        \
      } else {                                                                                                                               \
        tile_epilogue_s<MODE>(w0, D0, Fp, f0, accA, accV, accX, accY);                                                                       \
        stamp_epi(6);                                                                                                                        \
        finish_tile(w0, chan0, D0, f0);                                                                                                      \
      }                                                                                                                                      \
      }                                                                                                                                      \
      w0 = __builtin_amdgcn_readfirstlane(dq.x); chan0 = __builtin_amdgcn_readfirstlane(dq.y);                                               \
      stamp_epi(7);                                                                                                                          \
      stamp(3);                                                                                                                              \
      first_tile = false;                                                                                                                    \
      if (!grp_s) X3_LOOP_BARRIER(ST);         /* group A: [burst t, epilogue t] | barrier */                                                      \
    }
    for (int t = t_begin;;) {       // unrolled over the ring's four stages: tile t_begin + i sits in stage i & 3
      X3_TILE(0)
      if (++t >= t_end) break;
      X3_TILE(1)
      if (++t >= t_end) break;
      X3_TILE(2)
      if (++t >= t_end) break;
      X3_TILE(3)
      if (++t >= t_end) break;
    }
    if constexpr (ASM_EPI) { X3_EPI_DRAIN(); }      // group A's last column (group B is still in its last epilogue: nobody waits for this)
#undef X3_PAIR
#undef X3_BARE
#undef X3_FRAG
#undef X3_TILE
#undef X3_BURST_BODY
    // hand the next unit to the workgroup.  Group A waits here through group B's last epilogue: this barrier also retires the ring
    // (nobody reads it any more) before the next unit's staging writes
    stamp_unit(0, 0);
    stamp_unit(1, t_end - t_begin);
    if (tid == 0) *blk_slot = unit_next;
    lds_barrier();
    unit = __builtin_amdgcn_readfirstlane(*blk_slot);
    stamp_unit(2, 0);
  }
}

#ifndef X3_TWO_LIMBS
// Test hook kernel: the in-kernel limb split of n fp32 values (one range scale per group of `group` consecutive values, like the kernel's
// per-edge scaling): limbs as fp32, and the scale
__global__ void split3_probe_kernel(const float* x, int64_t n, int group, float* hi, float* mid, float* lo, float* scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g0 = (i / group) * group;
  float m = 0.0f;
  for (int64_t k = g0; k < g0 + group && k < n; ++k) m = fmaxf(m, fabsf(x[k]));
  float inv;
  const float s = range_scale(m, inv);
  const Limb3 q = split3(x[i] * s);
  hi[i] = (float)q.h; mid[i] = (float)q.m; lo[i] = (float)q.l; scale[i] = s;
}

hipError_t launch_split3_probe(const float* x, int64_t n, int group, float* hi, float* mid, float* lo, float* scale, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(split3_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, group, hi, mid, lo, scale);
  return hipGetLastError();
}

#endif      // !X3_TWO_LIMBS

template <bool GATHER, bool SPLIT, bool DET>
static hipError_t launch_x_t(const ConvXArgs& k, int n_cu, hipStream_t s) {
  hipLaunchKernelGGL((conv_x3_kernel<GATHER, SPLIT, DET>), dim3(n_cu), dim3(64 * CONV_WAVES), CONV_X_LDS_BYTES, s, k);
  return hipGetLastError();
}

template <bool GATHER, bool SPLIT, bool DET>
static hipError_t attr_x_t() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<GATHER, SPLIT, DET>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)CONV_X_LDS_BYTES);
}

#ifndef X3_TWO_LIMBS
bool conv_epilogue_shapes_ok(const std::vector<TileDesc>& tiles) {
  if (tiles.empty()) return false;
  for (const TileDesc& t : tiles) {
    const int w0 = x_tile_word(t.w0);
    if (w0 < 0 || (w0 & X_TILE_L2)) return false;
    const int fl = (w0 >> 2) & 3, nrq = (w0 >> 4) & 7, kind = w0 & 3;
    if (fl == FL_S && nrq != 4) return false;
    if (fl == FL_V && nrq != 3) return false;
    if (kind == T_RTS && fl != FL_S) return false;
  }
  const int last = x_tile_word(tiles.back().w0);
  return ((last >> 2) & 3) != 0;      // (a unit ends with a flush)
}
#endif      // !X3_TWO_LIMBS

hipError_t conv_prepare_device_x() {
  hipError_t e = attr_x_t<true, true, false>();
  if (e == hipSuccess) e = attr_x_t<true, false, false>();
  if (e == hipSuccess) e = attr_x_t<false, false, false>();
  if (e == hipSuccess) e = attr_x_t<true, true, true>();
  if (e == hipSuccess) e = attr_x_t<false, false, true>();
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CONV_X_LDS_BYTES);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<true, false, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CONV_X_LDS_BYTES);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<false, false, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CONV_X_LDS_BYTES);
#if defined(DDK_VARIANT_CONV_Y) && !defined(X3_TWO_LIMBS)
  if (e == hipSuccess) e = conv_prepare_device_y();
#endif
  return e;
}

void conv_det_fix(const ConvKArgs& k, const ConvLaunch& a, int dout, hipStream_t s);   // k_conv.hip

hipError_t launch_conv_fused_x(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  if (a.mode != 0 && a.mode != 1) return hipErrorInvalidValue;
  ConvXArgs X;
  ConvKArgs& k = X.k;
  k.x = a.x; k.src = a.src; k.dst = a.dst; k.edge_attr = a.edge_attr; k.sh = a.sh; k.sum = a.sum;
  k.counter = a.counter;
  k.w1p = nullptr; k.b1p = L.b1p[0]; k.w2r = nullptr; k.w1x = L.w1x; k.w2x = L.w2x; k.n_tiles = L.n_tiles;
  for (int g = 0; g < CONV_MAX_GROUPS; ++g) { k.w1s[g] = L.w1s[g]; k.w1u[g] = 1.0f / L.w1s[g]; k.w2s[g] = L.w2s[g]; k.w2u[g] = 1.0f / L.w2s[g]; }
  k.n_cols = L.n_cols;
  for (int c = 0; c <= L.n_cols; ++c) k.col_start[c] = L.col_start[c];
  k.sum_g2 = a.sum_g2; k.g2_node_off = a.g2_node_off;
  k.n_groups = a.n_groups; k.n_active = a.n_active; k.n_slots = a.n_slots; k.slots = a.slots; k.wmap = a.wmap;
  if (a.gbeg) { k.gbeg = a.gbeg; k.gend = a.gend; }
  else { k.gbeg = a.tile_info + 5; k.gend = a.tile_info + 6; }   // 4 contiguous groups go[g] .. go[g+1] (explicit-boundary entry point)
  k.pre = a.pre; k.part = a.part; k.det_rng = a.det_rng; k.det_nr = a.det_nr;
  X.trace = nullptr; X.trace_coarse = a.trace_coarse;
  if (a.mode == 1) {      // the confidence model: plain gather path (no node-term split, atomics)
    if (a.pre != nullptr || a.part != nullptr || a.trace != nullptr) return hipErrorInvalidValue;
    if (a.gather) hipLaunchKernelGGL((conv_x3_kernel<true, false, false, false, 1>), dim3(n_cu), dim3(64 * CONV_WAVES), CONV_X_LDS_BYTES, s, X);
    else hipLaunchKernelGGL((conv_x3_kernel<false, false, false, false, 1>), dim3(n_cu), dim3(64 * CONV_WAVES), CONV_X_LDS_BYTES, s, X);
    return hipGetLastError();
  }
  if (a.part != nullptr) {       // deterministic scatter
    hipError_t e = (a.gather && a.pre != nullptr) ? launch_x_t<true, true, true>(X, n_cu, s)
                                                  : (!a.gather ? launch_x_t<false, false, true>(X, n_cu, s) : hipErrorInvalidValue);
    if (e != hipSuccess) return e;
    conv_det_fix(k, a, L.dout, s);
    return hipGetLastError();
  }
#if defined(DDK_VARIANT_CONV_Y) && !defined(X3_TWO_LIMBS)      // tools/variants/k_conv_y.hip (round 5's one-wave-per-SIMD form, +9 %): linked by tools/build_variant_y.sh only
  if (a.gather && a.pre != nullptr && (a.trace == nullptr || a.trace_coarse == 1) && L.epi_ok && a.use_y) {
    X.trace = a.trace;      // (its TRACE instantiation writes the per-unit records only)
    return launch_conv_y(X, n_cu, s);
  }
#endif
  // the generated asm epilogue of the hot instantiation hard-codes the column shapes of this model family (a scalar flush closes 4 row quads, a vector
  // flush 3, the shared tail sits behind a scalar flush, no l = 2 rows): a tile table outside them must not reach it
  if (a.gather && a.pre != nullptr && !L.epi_ok) return hipErrorInvalidValue;
  X.trace = a.trace;
  if (a.trace != nullptr) {
    if (!(a.gather && a.pre != nullptr)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_x3_kernel<true, true, false, true>), dim3(n_cu), dim3(64 * CONV_WAVES), CONV_X_LDS_BYTES, s, X);
    return hipGetLastError();
  }
  if (a.gather && a.pre != nullptr) return launch_x_t<true, true, false>(X, n_cu, s);
  return a.gather ? launch_x_t<true, false, false>(X, n_cu, s) : launch_x_t<false, false, false>(X, n_cu, s);
}

}  // namespace ddk
