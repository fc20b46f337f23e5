// Device helpers shared by the fused conv kernels (k_conv.hip: fp32 MFMA; k_conv_x.hip: exact three-limb f16 MFMA).
#pragma once
#include "ddk_internal.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKArgs {
  const float* x;
  const int32_t* src;
  const int32_t* dst;
  const float* edge_attr;
  const float* sh;
  const float* pre;   // [N, PRE_W] per-node terms of GEMM1 (SPLIT kernels)
  float* sum;
  int32_t* counter;
  const float* w1p;   // [4][3][9][64][4]
  const float* b1p;   // [4][3][2][16]
  const float* w2r;   // [4][n_tiles][W2_TILE_FLOATS]
  const uint8_t* w1x;    // three-limb f16 kernel: [groups][3][W1X_TILE_BYTES] GEMM1 fragments
  const uint8_t* w2x;    // three-limb f16 kernel: [groups][n_tiles][W2X_TILE_BYTES] tile records
  float w1s[CONV_MAX_GROUPS], w1u[CONV_MAX_GROUPS], w2s[CONV_MAX_GROUPS], w2u[CONV_MAX_GROUPS];   // three-limb f16 kernel: weight range scales of GEMM1 / GEMM2 per weight set and their inverses
  int n_tiles;
  int n_cols;         // flush columns; col_start[c] .. col_start[c+1] = tiles of column c
  int col_start[17];
  float* part;        // deterministic mode: [tiles][2][XW] partial rows of the runs that straddle a 32-edge tile (see seg_add)
  float* sum_g2;      // != null: group 2 is the shared rec-rec copy; its messages go to sum_g2[(src - g2_node_off)] (see ConvLaunch)
  int g2_node_off;
  int n_groups, n_active, n_slots;   // edge groups [gbeg[g], gend[g]); the first n_active run; sum row = (node*n_slots + slot(g))
  uint32_t slots;
  uint64_t wmap;      // 4 bits per group: which radial MLP (weight set, node-term role) group g uses (identity unless a group is split in two)
  const int32_t* gbeg;
  const int32_t* gend;
  // deterministic mode, sample-aligned work units (round 6): det_nr > 0 -> the launch's edges as det_nr RANGES (one per edge group, level segment and
  // SAMPLE), det_rng = [block prefix pb[0 .. nr] | beg[nr] | end[nr] | group[nr]] (device, det_ranges_kernel).  A 256-edge block never crosses a range, so
  // where the 32-edge tiles cut a node's run of edges - and with it the association of its fp32 sum - depends on the sample's own edge list only, not on which
  // other samples share the batch (samples sharded over ranks see the same bits as the 40-sample batch).
  const int32_t* det_rng = nullptr;
  int det_nr = 0;
};
struct DetRange { int beg, end, g, bstart; };
// the range that holds work-queue block blk: the largest r with pb[r] <= blk (empty ranges share their successor's prefix and are skipped by that rule)
__device__ __forceinline__ DetRange det_find(const int32_t* rng, int nr, int blk) {
  int lo = 0, hi = nr;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rng[mid] <= blk) lo = mid; else hi = mid;
  }
  DetRange r;
  r.bstart = rng[lo]; r.beg = rng[nr + 1 + lo]; r.end = rng[2 * nr + 1 + lo]; r.g = rng[3 * nr + 1 + lo];
  return r;
}

struct ConvXArgs {      // kernel arguments of the three-limb f16 kernels (k_conv_x.hip, k_conv_y.hip)
  ConvKArgs k;
  uint32_t* trace;                  // TRACE instantiation: [8 waves][CONV_TRACE_TILES][8] s_memtime stamps of workgroup 0
  int trace_coarse;                 // 1: one record per UNIT (slots 4-7 prologue, 0 = tile loop done, 1 = tiles, 2 = unit handed over): no stamp inside the tile loop
};
// tools/variants/k_conv_y.hip (round 5's software-pipelined one-wave-per-SIMD form; measured +9 %, not part of libddk.so: tools/build_variant_y.sh)
hipError_t launch_conv_y(const ConvXArgs& X, int n_cu, hipStream_t s);
hipError_t conv_prepare_device_y();
// every flush of the tile table closes a 4-quad scalar or a 3-quad vector column, the shared tail sits behind a scalar flush, no l = 2 rows: the column
// shapes the generated asm epilogue of k_conv_x.hip (and k_conv_y.hip) hard-codes.  Checked by pack_x3 (ConvLayerDev::epi_ok)
bool conv_epilogue_shapes_ok(const std::vector<TileDesc>& tiles);

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct TileQ { int w0, chan0; };     // the two descriptor words of a W2 tile (they ride in the tile's record)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
#define D_LO(D, rq) (f32x2{(D)[4 * (rq)], (D)[4 * (rq) + 1]})
#define D_HI(D, rq) (f32x2{(D)[4 * (rq) + 2], (D)[4 * (rq) + 3]})
#define V_LO(f) __builtin_shufflevector(f, f, 0, 1)
#define V_HI(f) __builtin_shufflevector(f, f, 2, 3)

// kind-specialised tensor-product epilogue of one W2 tile (wave-uniform branch).  Every accumulator is a register PAIR
// {sum over even rows j, sum over odd rows j} so that the whole epilogue is v_pk_fma_f32 on adjacent registers
// (D[4rq+j], D[4rq+j+1]) x (f[j], f[j+1]) without any shuffling moves; the pair is added up when the column is flushed.
__device__ __forceinline__ void tile_epilogue(int kind, const f32x16& D, f32x4 f0, f32x4 f1, f32x4 f2, f32x2 (&accA)[4],
                                              f32x2 (&accV)[4][3]) {
  // three independent wave-uniform branches, each updating its own accumulators in place (a four-way if / else-if chain made the
  // compiler merge the accumulator sets with ~25 register copies per tile)
  f32x2 fhi = V_HI(f0);
  if (kind == T_RTS) fhi = 0.0f;     // shared tail: only rows j = 0,1 belong to the column that is about to be flushed (the caller adds j = 2,3 behind the flush)
  if (kind == T_RA) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) accA[rq] = __builtin_elementwise_fma(D_HI(D, rq), fhi, __builtin_elementwise_fma(D_LO(D, rq), V_LO(f0), accA[rq]));
  }
  if (kind != T_RA) {                // T_RT, T_RTS, and the x components of T_TV
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) accV[rq][0] = __builtin_elementwise_fma(D_HI(D, rq), fhi, __builtin_elementwise_fma(D_LO(D, rq), V_LO(f0), accV[rq][0]));
  }
  if (kind == T_TV) {                // f1 / f2 = y / z components of the 4 feature rows
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      accV[rq][1] = __builtin_elementwise_fma(D_HI(D, rq), V_HI(f1), __builtin_elementwise_fma(D_LO(D, rq), V_LO(f1), accV[rq][1]));
      accV[rq][2] = __builtin_elementwise_fma(D_HI(D, rq), V_HI(f2), __builtin_elementwise_fma(D_LO(D, rq), V_LO(f2), accV[rq][2]));
    }
  }
}

// Control words of the segmented inclusive scan over runs of equal edge_src (identical for every channel of an edge tile).  The scan runs
// on the VALU's DPP lane crossbar, not through the LDS: four Hillis-Steele steps inside each 16-lane row (row_shr:1,2,4,8) and one
// row_bcast:15 step that hands lane 15's sum of the half's first row to the lanes of the second row whose run started in the first
// (a 64-lane wave = two independent halves of 32 edges = rows {0,1} and {2,3}).
struct SegCtl { bool m1, m2, m4, m8, m16, tail, valid; };

// sn = edge_src of this lane's edge (lanes past nvalid are clamped duplicates of the last edge)
__device__ __forceinline__ SegCtl make_segctl(int sn, int el, int nvalid, bool valid) {
  SegCtl c;
  const int prev = __shfl_up(sn, 1, 32);
  const int next = __shfl_down(sn, 1, 32);
  const int r = (el == 0) || (prev != sn);          // a run starts at this lane
  c.tail = valid && ((el == nvalid - 1) || (next != sn));
  c.valid = valid;
  const int rl = el & 15;
  int f = r | (rl == 0);                            // in-row flag: "a start (or the row start) lies inside the window behind this lane"
  int g = r;                                        // real starts only: inclusive OR over the lane's row (for the cross-row step)
  int fu, gu;
  fu = __shfl_up(f, 1, 32); gu = __shfl_up(g, 1, 32); c.m1 = !f; if (rl >= 1) { f |= fu; g |= gu; }
  fu = __shfl_up(f, 2, 32); gu = __shfl_up(g, 2, 32); c.m2 = !f; if (rl >= 2) { f |= fu; g |= gu; }
  fu = __shfl_up(f, 4, 32); gu = __shfl_up(g, 4, 32); c.m4 = !f; if (rl >= 4) { f |= fu; g |= gu; }
  fu = __shfl_up(f, 8, 32); gu = __shfl_up(g, 8, 32); c.m8 = !f; if (rl >= 8) { f |= fu; g |= gu; }
  c.m16 = (el >= 16) && !g;                         // the lane's run began in the first row of this half
  return c;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float x) {    // lanes without a source (outside the row / row mask) read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, true));
}

// DET: the run tail STORES (every (node, accumulator slot, channel) has exactly one writer per launch: complete runs write the node row,
// runs that straddle a 32-edge tile write that tile's partial row and conv_det_fix_kernel folds the chain in tile order) instead of
// adding atomically - a fixed summation order per node (ddk_config.deterministic)
template <bool DET = false>
__device__ __forceinline__ void seg_add(float* dst, float xv, const SegCtl& c) {
  xv = c.valid ? xv : 0.0f;
  float up;
  up = dpp_f<0x111, 0xF>(xv); if (c.m1) xv += up;     // row_shr:1
  up = dpp_f<0x112, 0xF>(xv); if (c.m2) xv += up;     // row_shr:2
  up = dpp_f<0x114, 0xF>(xv); if (c.m4) xv += up;     // row_shr:4
  up = dpp_f<0x118, 0xF>(xv); if (c.m8) xv += up;     // row_shr:8
  up = dpp_f<0x142, 0xA>(xv); if (c.m16) xv += up;    // row_bcast:15 into rows 1 and 3
  if (c.tail) {
    if (DET) *dst = xv;
    else unsafeAtomicAdd(dst, xv);
  }
}

// N channels of the same edge tile at once (dst + i * stride): the N dependent DPP chains interleave, which fills the two wait states a
// DPP read needs behind the VALU write of its source, and the run tails issue their N atomics under one branch
template <bool DET, int N>
__device__ __forceinline__ void seg_add_n(float* dst, int stride, float (&xv)[N], const SegCtl& c) {
  float up[N];
#pragma unroll
  for (int i = 0; i < N; ++i) xv[i] = c.valid ? xv[i] : 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) up[i] = dpp_f<0x111, 0xF>(xv[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) if (c.m1) xv[i] += up[i];
#pragma unroll
  for (int i = 0; i < N; ++i) up[i] = dpp_f<0x112, 0xF>(xv[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) if (c.m2) xv[i] += up[i];
#pragma unroll
  for (int i = 0; i < N; ++i) up[i] = dpp_f<0x114, 0xF>(xv[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) if (c.m4) xv[i] += up[i];
#pragma unroll
  for (int i = 0; i < N; ++i) up[i] = dpp_f<0x118, 0xF>(xv[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) if (c.m8) xv[i] += up[i];
#pragma unroll
  for (int i = 0; i < N; ++i) up[i] = dpp_f<0x142, 0xA>(xv[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) if (c.m16) xv[i] += up[i];
  if (c.tail) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (DET) dst[i * stride] = xv[i];
      else unsafeAtomicAdd(dst + i * stride, xv[i]);
    }
  }
}

}  // namespace ddk
