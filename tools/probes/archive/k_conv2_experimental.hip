// Fused tensor-product convolution, version 2: same algorithm and packed-weight format as k_conv.hip (read its
// header first), restructured so that ONE wave keeps its SIMD's matrix pipe busy on its own:
//   * the 36 MFMAs of weight tile t+1 and the tensor-product epilogue of tile t live in the same branch-free
//     basic block (two accumulator sets), so the VALU / LDS work of the epilogue issues in the 64-cycle shadows
//     of the fp32 MFMAs instead of after them (v1 measured MfmaUtil = 58 %: the pipe idled during every epilogue);
//   * the epilogue is branch-free: unit kinds select per-lane multipliers instead of code paths; finished
//     (block, k-pair) values are written to a per-wave LDS message tile m[32 edges][84 channels];
//   * the scatter is done once per 32-edge tile: lanes = output channels walk the 32 edges, accumulate runs of
//     equal edge_src and flush each run with one coalesced fp32 atomic per channel (deterministic order inside
//     the tile, 64 consecutive addresses per atomic instruction).
// Occupancy target: 1 wave per SIMD (4-5 single-wave workgroups per CU, 29.5 KB LDS each).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "ddk_internal.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKArgs2 {
  const float* x;
  const int32_t* src;
  const int32_t* dst;
  const float* edge_attr;
  const float* sh;
  float* sum;
  const int32_t* tile_info;
  int32_t* counter;
  const float* w1p;
  const float* b1p;
  const float* w2p;
  const float* b2p;
  const Unit* units;
  int n_tiles;
  int split;          // WAVES == 2: first weight tile of wave 1 (a (block, k-pair) boundary)
  long long* tstamps; // VAR & 16: [wg < 64][wave][tile < 8][8] s_memtime stamps (phase timing experiment)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct UnitQuad2 { int w0[4], w1[4]; float scale[4]; };
typedef const int32_t __attribute__((address_space(4))) cint32;
__device__ __forceinline__ UnitQuad2 load_units(const Unit* p) {
  cint32* q = (cint32*)(uintptr_t)p;
  UnitQuad2 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.w0[i] = q[4 * i + 0];
    r.w1[i] = q[4 * i + 1];
    r.scale[i] = __int_as_float(q[4 * i + 2]);
  }
  return r;
}

// tensor-product epilogue of one weight tile: 4 units x 4 accumulator registers (see k_conv.hip)
__device__ __forceinline__ void tp_epilogue(const f32x16& D, const UnitQuad2& u, const float* Fr, float* Mr, int hh,
                                            float s0, float vx, float vy, float vz, float& acc0, float& acc1, float& acc2) {
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const int w0 = u.w0[rq], w1 = u.w1[rq];
    const int kind = w0 & 15;
    const bool last = (w0 >> 5) & 1;
    const int ncomp = (w0 >> 8) & 15;
    const float* Fp = Fr + (w0 >> 16);
    const float4 f0 = ld4(Fp), f1 = ld4(Fp + 4), f2 = ld4(Fp + 8);
    const float d0 = D[4 * rq + 0], d1 = D[4 * rq + 1], d2 = D[4 * rq + 2], d3 = D[4 * rq + 3];
    const float ps = f0.x * d0 + f0.y * d1 + f0.z * d2 + f0.w * d3;          // rows are scalars  F[r]
    const float pv0 = f0.x * d0 + f0.w * d1 + f1.z * d2 + f2.y * d3;         // rows are vectors  F[3r + c]
    const float pv1 = f0.y * d0 + f1.x * d1 + f1.w * d2 + f2.z * d3;
    const float pv2 = f0.z * d0 + f1.y * d1 + f2.x * d2 + f2.w * d3;
    const bool kS0 = kind == U_R1_S0, kV = kind == U_R1_V, kTS = kind == U_T_S, kTV = kind == U_T_V;
    const float ms0 = kS0 ? s0 : (kV ? vx : (kTS ? 1.0f : 0.0f));
    const float ms1 = kV ? vy : 0.0f, ms2 = kV ? vz : 0.0f;
    const float mv = kTV ? 1.0f : 0.0f;
    acc0 = fmaf(ms0, ps, fmaf(mv, pv0, acc0));
    acc1 = fmaf(ms1, ps, fmaf(mv, pv1, acc1));
    acc2 = fmaf(ms2, ps, fmaf(mv, pv2, acc2));
    const float sc = u.scale[rq];
    const int chan = (w1 & 0xffff) + hh * (w1 >> 16);
    const int cstep = ncomp == 3 ? 1 : 0;
    // partial values of an unfinished (block, k-pair) are overwritten by its later units; scalar outputs write the
    // same slot three times, component 0 last
    Mr[chan + 2 * cstep] = acc2 * sc;
    Mr[chan + cstep] = acc1 * sc;
    Mr[chan] = acc0 * sc;
    acc0 = last ? 0.0f : acc0;
    acc1 = last ? 0.0f : acc1;
    acc2 = last ? 0.0f : acc2;
  }
}

// WAVES = 1: one wave owns a 32-edge tile and all weight tiles.  WAVES = 2: a 128-thread workgroup shares the tile's
// F / message rows in LDS; both waves run GEMM1 (redundantly, 5 % of the MFMA work) and split the weight tiles at a
// (block, k-pair) boundary (A.split) - two waves per SIMD hide each other's epilogue and LDS latencies.
// VAR (experiment bits): 1 = sched_barrier between the MFMA burst and the epilogue (no interleave); 2 = two
// accumulator chains over K (no dependent back-to-back MFMAs); 4 = ablation: skip the epilogue; 8 = ablation: do not
// stream weight tiles (re-use tile 0's fragments).
template <bool GATHER, int WAVES, int VAR>
__global__ __launch_bounds__(64 * WAVES, WAVES) void conv_fused2_kernel(ConvKArgs2 A) {
  __shared__ __attribute__((aligned(16))) float F[32 * F_STRIDE + 16];
  __shared__ float Mv[32 * M_STRIDE];
  __shared__ int srcs[32];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int el = lane & 31;
  const int hh = lane >> 5;
  const int ts1 = A.tile_info[1], ts2 = A.tile_info[2], ts3 = A.tile_info[3], ts4 = A.tile_info[4];
  const int go0 = A.tile_info[5], go1 = A.tile_info[6], go2 = A.tile_info[7], go3 = A.tile_info[8], go4 = A.tile_info[9];
  float* Fr = F + el * F_STRIDE;
  float* Mr = Mv + el * M_STRIDE;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  if (threadIdx.x < 16) F[32 * F_STRIDE + threadIdx.x] = 0.0f;
  const int n_tiles = A.n_tiles;
  // weight tiles of this wave: [t_beg, t_end)
  const int t_beg = (WAVES == 2 && wave == 1) ? A.split : 0;
  const int t_end = (WAVES == 2 && wave == 0) ? A.split : n_tiles;

  // Static round-robin tile assignment (all tiles cost the same): the next tile of this workgroup is known in
  // advance, so its edge indices are loaded and the cache lines its gathers will need are touched one tile ahead
  // (the dependent index -> gather -> GEMM1 chain of HBM round trips was ~1/4 of a tile's time when exposed).
  auto tile_geom = [&](int tile, int& g, int& e, int& nvalid) {
    g = (tile >= ts1) + (tile >= ts2) + (tile >= ts3);
    const int tstart = g == 0 ? 0 : (g == 1 ? ts1 : (g == 2 ? ts2 : ts3));
    const int gbeg = g == 0 ? go0 : (g == 1 ? go1 : (g == 2 ? go2 : go3));
    const int gend = g == 0 ? go1 : (g == 1 ? go2 : (g == 2 ? go3 : go4));
    const int e0 = gbeg + 32 * (tile - tstart);
    nvalid = min(32, gend - e0);
    e = e0 + min(el, nvalid - 1);
  };
  int tile = blockIdx.x;
  if (tile >= ts4) return;
  int g, e, nvalid;
  tile_geom(tile, g, e, nvalid);
  int sn = A.src[e], dn = A.dst[e];
  for (;;) {
    const int ntile = tile + (int)gridDim.x;
    const bool has_next = ntile < ts4;
    int g2, e2, nvalid2;
    tile_geom(has_next ? ntile : tile, g2, e2, nvalid2);
    const int sn2 = A.src[e2], dn2 = A.dst[e2];      // consumed after GEMM2: a whole tile of slack
    const int tcount = (tile - (int)blockIdx.x) / (int)gridDim.x;
    const bool stamp = (VAR & 16) && blockIdx.x < 64 && tcount < 8 && lane == 0;
    long long* ts = A.tstamps + (((size_t)blockIdx.x * 2 + wave) * 8 + (tcount & 7)) * 8;
#define STAMP(i) do { if (stamp) ts[i] = (long long)__builtin_readcyclecounter(); } while (0)
    STAMP(0);
    if (hh == 0 && wave == 0) srcs[el] = sn;

    // ---- GEMM1 (identical to v1) ----
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
    STAMP(1);   // gathers issued+landed (bin consumed below)
    float h[36];
    {
      const float* w1 = A.w1p + (size_t)g * (3 * 9 * 64 * 4);
      const float* b1 = A.b1p + (size_t)g * (3 * 2 * 16);
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        f32x16 acc;
        const float* bp = b1 + (T * 2 + hh) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = ld4(bp + 4 * j);
          acc[4 * j + 0] = b.x; acc[4 * j + 1] = b.y; acc[4 * j + 2] = b.z; acc[4 * j + 3] = b.w;
        }
        const float* wp = w1 + ((size_t)T * 9 * 64 + lane) * 4;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          const float4 a = ld4(wp + s4 * 64 * 4);
          acc = MFMA(a.x, bin[4 * s4 + 0], acc);
          acc = MFMA(a.y, bin[4 * s4 + 1], acc);
          acc = MFMA(a.z, bin[4 * s4 + 2], acc);
          acc = MFMA(a.w, bin[4 * s4 + 3], acc);
        }
        if (T < 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) h[16 * T + r] = fmaxf(acc[r], 0.0f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[32 + r] = fmaxf(acc[r], 0.0f);
        }
      }
    }

    STAMP(2);   // GEMM1 done
    // ---- F row (identical to v1) ----
    const float4 shv = ld4(A.sh + (size_t)e * 4);
    const float s0 = shv.x, vx = shv.y, vy = shv.z, vz = shv.w;
    if (wave == 0) {
      const float* xr = A.x + (size_t)dn * XW;
      const int o_main_src = hh ? OFF_C : 0, o_main_dst = hh ? F_C : F_A;
      const int o_vec_src = hh ? OFF_Q : OFF_P, o_dot = hh ? F_QV : F_PV;
      const int o_s0 = hh ? (F_T1E + 3 * NV) : F_T1O, o_cross = hh ? (F_T1O + 3 * NV) : F_T1E;
#pragma unroll
      for (int j = 0; j < NS / 4; ++j) *reinterpret_cast<float4*>(Fr + o_main_dst + 4 * j) = ld4(xr + o_main_src + 4 * j);
      float pv[3 * NV];
#pragma unroll
      for (int j = 0; j < 3 * NV / 2; ++j) {
        const float2 t = ld2(xr + o_vec_src + 2 * j);
        pv[2 * j] = t.x; pv[2 * j + 1] = t.y;
      }
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const float px = pv[3 * m], py = pv[3 * m + 1], pz = pv[3 * m + 2];
        Fr[o_dot + m] = (px * vx + py * vy + pz * vz) * inv_s3;
        Fr[o_s0 + 3 * m + 0] = px * s0;
        Fr[o_s0 + 3 * m + 1] = py * s0;
        Fr[o_s0 + 3 * m + 2] = pz * s0;
        Fr[o_cross + 3 * m + 0] = (py * vz - pz * vy) * inv_s2;
        Fr[o_cross + 3 * m + 1] = (pz * vx - px * vz) * inv_s2;
        Fr[o_cross + 3 * m + 2] = (px * vy - py * vx) * inv_s2;
      }
      Fr[o_dot + NV] = 0.0f;
      Fr[o_dot + NV + 1] = 0.0f;
      if (hh) *reinterpret_cast<float4*>(Fr + F_SH) = shv;
    }
    __syncthreads();
    STAMP(3);   // F built + barrier
    // L2/L1 touch of the lines the next tile's gathers will read (one dword per 128-B line); the values are only
    // "consumed" at the end of the tile so that no wait is placed in front of GEMM2
    const float* pe2 = GATHER ? A.edge_attr + (size_t)e2 * NS : A.edge_attr + (size_t)e2 * NE;
    const float* px2 = A.x + (size_t)(hh ? dn2 : sn2) * XW;
    float pf0 = pe2[16 * hh], pf1, pf2, pf3 = A.sh[(size_t)e2 * 4];
    if (GATHER) { pf1 = px2[0]; pf2 = px2[32 + 32 * hh]; }
    else { pf1 = pe2[32 + 8 * hh]; pf2 = pe2[64 + 4 * hh]; }

    // ---- GEMM2, software pipelined: MFMAs of tile t+1 share a basic block with the epilogue of tile t ----
    const float* w2 = A.w2p + ((size_t)g * n_tiles + t_beg) * (9 * 64 * 4) + (size_t)lane * 4;
    const float* b2 = A.b2p + ((size_t)g * n_tiles + t_beg) * 32 + hh * 16;
    const Unit* units = A.units + 4 * t_beg;
    const int nt = t_end - t_beg;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    f32x16 Dc;
    UnitQuad2 uc = load_units(units);
    {
      float4 a0[9];
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) a0[s4] = ld4(w2 + s4 * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = ld4(b2 + 4 * j);
        Dc[4 * j + 0] = b.x; Dc[4 * j + 1] = b.y; Dc[4 * j + 2] = b.z; Dc[4 * j + 3] = b.w;
      }
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) {
        Dc = MFMA(a0[s4].x, h[4 * s4 + 0], Dc);
        Dc = MFMA(a0[s4].y, h[4 * s4 + 1], Dc);
        Dc = MFMA(a0[s4].z, h[4 * s4 + 2], Dc);
        Dc = MFMA(a0[s4].w, h[4 * s4 + 3], Dc);
      }
    }
    float4 an[9], bn[4];
    UnitQuad2 un;
    {
      const int t1 = min(1, nt - 1);
      const float* wn = w2 + (size_t)t1 * (9 * 64 * 4);
      const float* bp = b2 + (size_t)t1 * 32;
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) an[s4] = ld4(wn + s4 * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = ld4(bp + 4 * j);
      un = load_units(units + 4 * t1);
    }
    for (int t = 0; t < nt - 1; ++t) {
      float4 ac[9];
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) ac[s4] = an[s4];
      f32x16 Dn;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Dn[4 * j + 0] = bn[j].x; Dn[4 * j + 1] = bn[j].y; Dn[4 * j + 2] = bn[j].z; Dn[4 * j + 3] = bn[j].w;
      }
      const UnitQuad2 ucn = un;
      if (!(VAR & 8)) {   // prefetch tile t+2 (clamped: the final iterations re-read the last tile instead of branching)
        const int t2 = min(t + 2, nt - 1);
        const float* wn = w2 + (size_t)t2 * (9 * 64 * 4);
        const float* bp = b2 + (size_t)t2 * 32;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) an[s4] = ld4(wn + s4 * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = ld4(bp + 4 * j);
        un = load_units(units + 4 * t2);
      }
      if (VAR & 2) {
        f32x16 Db;
#pragma unroll
        for (int r = 0; r < 16; ++r) Db[r] = 0.0f;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          Dn = MFMA(ac[s4].x, h[4 * s4 + 0], Dn);
          Db = MFMA(ac[s4].y, h[4 * s4 + 1], Db);
          Dn = MFMA(ac[s4].z, h[4 * s4 + 2], Dn);
          Db = MFMA(ac[s4].w, h[4 * s4 + 3], Db);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Dn[r] += Db[r];
      } else {
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          Dn = MFMA(ac[s4].x, h[4 * s4 + 0], Dn);
          Dn = MFMA(ac[s4].y, h[4 * s4 + 1], Dn);
          Dn = MFMA(ac[s4].z, h[4 * s4 + 2], Dn);
          Dn = MFMA(ac[s4].w, h[4 * s4 + 3], Dn);
        }
      }
      if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
      if (VAR & 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(Dc[r]));
      } else {
        tp_epilogue(Dc, uc, Fr, Mr, hh, s0, vx, vy, vz, acc0, acc1, acc2);
      }
      Dc = Dn;
      uc = ucn;
    }
    tp_epilogue(Dc, uc, Fr, Mr, hh, s0, vx, vy, vz, acc0, acc1, acc2);
    STAMP(4);   // GEMM2 done
    __syncthreads();
    STAMP(5);

    // ---- scatter: lanes = channels, walk the tile's edges, flush runs of equal edge_src ----
#pragma unroll
    for (int pass = 0; pass < (WAVES == 1 ? 2 : 1); ++pass) {
      const int ch = (int)threadIdx.x + 64 * pass;
      if (ch < XW) {
        int cur = srcs[0];
        float s = 0.0f;
        for (int ee = 0; ee < nvalid; ++ee) {
          const int se = srcs[ee];
          if (se != cur) {
            unsafeAtomicAdd(A.sum + (size_t)cur * XW + ch, s);
            s = 0.0f;
            cur = se;
          }
          s += Mv[ee * M_STRIDE + ch];
        }
        unsafeAtomicAdd(A.sum + (size_t)cur * XW + ch, s);
      }
    }
    __syncthreads();
    STAMP(6);   // scatter + barrier done
    asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));
    if (!has_next) break;
    tile = ntile; g = g2; e = e2; nvalid = nvalid2; sn = sn2; dn = dn2;
  }
}

static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  int v = e ? atoi(e) : dflt;
  return (v < lo || v > hi) ? dflt : v;
}

template <int VAR>
static void launch_var(const ConvKArgs2& k, bool gather, int waves, int grid, hipStream_t s) {
  if (waves == 2) {
    if (gather) hipLaunchKernelGGL((conv_fused2_kernel<true, 2, VAR>), dim3(grid), dim3(128), 0, s, k);
    else hipLaunchKernelGGL((conv_fused2_kernel<false, 2, VAR>), dim3(grid), dim3(128), 0, s, k);
  } else {
    if (gather) hipLaunchKernelGGL((conv_fused2_kernel<true, 1, VAR>), dim3(grid), dim3(64), 0, s, k);
    else hipLaunchKernelGGL((conv_fused2_kernel<false, 1, VAR>), dim3(grid), dim3(64), 0, s, k);
  }
}

hipError_t launch_conv_fused_v2(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  static const int waves = env_int("DDK_CONV_WG_WAVES", 2, 1, 2);          // waves per workgroup
  static const int wg_per_cu = env_int("DDK_CONV_WG_PER_CU", 4, 1, 8);
  static const int var = env_int("DDK_CONV_VAR", 0, 0, 63);
  static long long* tstamps = nullptr;
  if (var & 16) {
    if (!tstamps) { hipMalloc((void**)&tstamps, 64 * 2 * 8 * 8 * 8); }
    hipMemsetAsync(tstamps, 0, 64 * 2 * 8 * 8 * 8, s);
  }
  ConvKArgs2 k;
  k.x = a.x; k.src = a.src; k.dst = a.dst; k.edge_attr = a.edge_attr; k.sh = a.sh; k.sum = a.sum;
  k.tile_info = a.tile_info; k.counter = a.counter;
  k.w1p = L.w1p[0]; k.b1p = L.b1p[0]; k.w2p = L.w2p[0]; k.b2p = L.b2p[0]; k.units = L.units; k.n_tiles = L.n_tiles;
  k.split = L.split;
  k.tstamps = tstamps;
  const int grid = n_cu * wg_per_cu;
  const int w = (waves == 2 && L.split > 0 && L.split < L.n_tiles) ? 2 : 1;
  switch (var) {
    case 1: launch_var<1>(k, a.gather, w, grid, s); break;
    case 2: launch_var<2>(k, a.gather, w, grid, s); break;
    case 3: launch_var<3>(k, a.gather, w, grid, s); break;
    case 4: launch_var<4>(k, a.gather, w, grid, s); break;
    case 8: launch_var<8>(k, a.gather, w, grid, s); break;
    case 12: launch_var<12>(k, a.gather, w, grid, s); break;
    case 16: launch_var<16>(k, a.gather, w, grid, s); break;
    case 32: launch_var<32>(k, a.gather, w, grid, s); break;
    default: launch_var<0>(k, a.gather, w, grid, s); break;
  }
  if (var & 16) {
    static int dumped = 0;
    if (dumped++ == 2) {
      std::vector<long long> h(64 * 2 * 8 * 8);
      hipStreamSynchronize(s);
      hipMemcpy(h.data(), tstamps, h.size() * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen("gpurun_out/conv_stamps.txt", "w");
      if (f) {
        for (int wg = 0; wg < 64; ++wg) for (int wv = 0; wv < 2; ++wv) for (int t = 0; t < 8; ++t) {
          fprintf(f, "%d %d %d", wg, wv, t);
          for (int i = 0; i < 8; ++i) fprintf(f, " %lld", h[((wg * 2 + wv) * 8 + t) * 8 + i]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
  }
  return hipGetLastError();
}

hipError_t launch_conv_fused(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  static int use_v1 = -1;
  if (use_v1 < 0) {
    const char* e = getenv("DDK_CONV_V1");
    use_v1 = (e && atoi(e)) ? 1 : 0;
  }
  return use_v1 ? launch_conv_fused_v1(L, a, n_cu, s) : launch_conv_fused_v2(L, a, n_cu, s);
}

}  // namespace ddk
