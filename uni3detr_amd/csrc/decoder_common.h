// Building blocks of the fused decoder-layer kernels (decoder.hip, decoder_bwd.hip): a 32-row block of the layer state lives in LDS
// through a whole chain of GEMMs / LayerNorms; weights stream from L2 straight into MFMA fragments (each wave owns 64 output
// columns, so no two waves share a weight row and LDS staging of the weights would buy nothing).
//
// MFMA orientation: D^T = W . X^T, i.e. a-operand = weight rows (lane l: row n = l&15, 8 consecutive k at (l>>4)*8), b-operand =
// activation rows from LDS (lane l: row m = l&15, same 8 k) -> a lane ends up with 4 CONSECUTIVE output columns of one row:
// 8-byte bf16 / 16-byte f32 accesses in every epilogue.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define DC_BM 32              /* rows per workgroup */
#define DC_THREADS 256
#define DC_C 256              /* embed dim */
#define DC_FF 512             /* FFN hidden */
#define DC_TS 260             /* row stride (floats) of the f32 tiles: 8-lane store groups land on distinct banks */
#define DC_NHEAD 8
#define DC_HD 32

// LDS map (bytes) of the row-chain kernels
#define DC_OFF_A0 0                                  /* bf16 [32][512] */
#define DC_OFF_A1 (DC_OFF_A0 + DC_BM * 512 * 2)      /* bf16 [32][512] */
#define DC_OFF_A2 (DC_OFF_A1 + DC_BM * 512 * 2)      /* bf16 [32][256] */
#define DC_OFF_F (DC_OFF_A2 + DC_BM * 256 * 2)       /* f32 [32][260] */
#define DC_OFF_G (DC_OFF_F + DC_BM * DC_TS * 4)      /* f32 [32][260] */
#define DC_MISC_LD 48                                 /* floats per row of the misc tile: [0,16) scalars, [16,48) narrow gradients */
#define DC_OFF_MISC (DC_OFF_G + DC_BM * DC_TS * 4)   /* f32 [32][48] */
#define DC_LDS_BYTES (DC_OFF_MISC + DC_BM * DC_MISC_LD * 4)

__device__ __forceinline__ float dc_bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 dc_f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ float dc_round(float f) { return dc_bf2f(dc_f2bf(f)); }      // value after a bf16 store
__device__ __forceinline__ u16x4 dc_pack4(f32x4 v) { return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4)); }
__device__ __forceinline__ f32x4 dc_unpack4(u16x4 v) {
  f32x4 r = {dc_bf2f(v[0]), dc_bf2f(v[1]), dc_bf2f(v[2]), dc_bf2f(v[3])};
  return r;
}
__device__ __forceinline__ f32x4 dc_round4(f32x4 v) { return dc_unpack4(dc_pack4(v)); }
__device__ __forceinline__ float dc_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

// element offset of (row, col) in a bf16 activation tile whose rows hold `ldk` elements: the 16-byte chunk index is XORed with
// row & 15, so the ds_read_b128 of one MFMA operand (16 rows x one chunk column per 16-lane group) touches all 64 banks once
__device__ __forceinline__ int dc_aoff(int row, int col, int ldk) { return row * ldk + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7)); }

// ---- dropout: keep decision of element idx at (layer, site); identical in forward and backward ----------------------------
struct DcRng { unsigned lo, hi; };
__device__ __forceinline__ DcRng dc_rng_load(const unsigned long long* p) {
  const unsigned long long v = p ? *p : 0ull;
  DcRng r = {(unsigned)v, (unsigned)(v >> 32)};
  return r;
}
__device__ __forceinline__ unsigned dc_site_key(int layer, int site) { return (unsigned)(layer * 8 + site + 1) * 0x85EBCA77u; }
__device__ __forceinline__ bool dc_keep(DcRng g, unsigned site_key, unsigned idx, unsigned thresh) {
  unsigned h = idx * 0x9E3779B1u + (g.lo ^ site_key);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  h ^= g.hi; h *= 0x9E3779B1u; h ^= h >> 15;
  return (h >> 8) >= thresh;                       // thresh = p * 2^24
}
__host__ __device__ static inline unsigned dc_thresh(float p) { return p <= 0.f ? 0u : (unsigned)(p * 16777216.0f); }
__host__ __device__ static inline float dc_inv_keep(float p) { return p <= 0.f ? 1.f : 1.f / (1.f - p); }

struct DcDrop {
  DcRng rng; unsigned thresh; float inv_keep; int layer;
  __device__ __forceinline__ f32x4 apply(f32x4 v, int site, unsigned idx0) const {     // 4 consecutive elements idx0..idx0+3
    if (thresh == 0u) return v;
    const unsigned key = dc_site_key(layer, site);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = dc_keep(rng, key, idx0 + r, thresh) ? v[r] * inv_keep : 0.f;
    return v;
  }
};

// ---- the block GEMM: acc[mt][nt] (+)= X[32 rows] . W[NT*16 rows]^T over K ---------------------------------------------------
// A: LDS activation tile (ldk = K); W: global weight rows, already offset to the wave's first output column.
// The weight fragments of DC_PF k-steps are requested in ONE burst and consumed behind sched_barriers: left alone, hipcc's
// scheduler sinks each load to just before its MFMA (2-3 loads in flight per wave), and with one wave per SIMD every k-step then
// waits out a full L2 round trip (measured: 13 us per 256x256 stage, 4x the burst schedule).
#ifndef DC_PF
#define DC_PF 8
#endif
constexpr int dc_burst(int ks, int want) {          // largest divisor of ks that is <= want
  int b = 1;
  for (int d = 1; d <= want && d <= ks; ++d)
    if (ks % d == 0) b = d;
  return b;
}
template <int K, int NT>
__device__ __forceinline__ void dc_gemm(const u16* A, const u16* __restrict__ W, f32x4 (&acc)[2][NT], int lane) {
  constexpr int KS = K / 32;
  constexpr int PF = dc_burst(KS, DC_PF);
  const int r16 = lane & 15, kq = lane >> 4;
  const u16* wp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = W + (size_t)(nt * 16 + r16) * K + kq * 8;
  const u16* ap0 = A + r16 * K;
  const u16* ap1 = A + (16 + r16) * K;
#pragma unroll
  for (int kb = 0; kb < KS; kb += PF) {
    bf16x8 wf[PF][NT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[p][nt] = *(const bf16x8*)(wp[nt] + (kb + p) * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int ch = (((kb + p) * 4 + kq) ^ r16) << 3;
      const bf16x8 a0 = *(const bf16x8*)(ap0 + ch);
      const bf16x8 a1 = *(const bf16x8*)(ap1 + ch);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[p][nt], a0, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[p][nt], a1, acc[1][nt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// y[32 rows][ncol0 .. ncol0 + NT*16) = X . W^T; epi(row, col, v) gets 4 consecutive columns col..col+3 of row `row`
template <int K, int NT, typename Epi>
__device__ __forceinline__ void dc_linear(const u16* A, const u16* __restrict__ W, int ncol0, int lane, Epi epi) {
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  dc_gemm<K, NT>(A, W + (size_t)ncol0 * K, acc, lane);
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) epi(mt * 16 + r16, ncol0 + nt * 16 + kq * 4, acc[mt][nt]);
}

__device__ __forceinline__ f32x4 dc_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 dc_bias4(const float* b, int col) {
  return b ? *(const f32x4*)(b + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 dc_relu4(f32x4 v) {
  f32x4 r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  return r;
}

// ---- control flow discipline -------------------------------------------------------------------------------------------------
// Every loop below has a trip count the compiler can see is the same for all lanes (DC_FOR_TID), and no branch depends on the row
// index: row matrices owned by this library (save / gradient slots, outputs) hold whole 32-row blocks (u3d_decoder_layer_blocks(m)
// * 32 rows), caller-owned inputs of m rows are read through a clamped row index.  Reason: hipcc (ROCm 7.2) was seen to place
// register-allocator copies (v_accvgpr_write) in the exit block of an exec-masked `for (i = tid; i < N; i += 256)` loop AHEAD of the
// `s_or_b64 exec` that restores the lane mask - the copies ran with EXEC = 0 and the "saved" LDS addresses were garbage afterwards
// (tools/check_exec_restore.py scans the ISA for that pattern; tests/test_build_cpu.py runs it).
#define DC_FOR_TID(c, N) _Pragma("unroll") for (int it_ = 0, c = tid; it_ < (N) / DC_THREADS; ++it_, c += DC_THREADS)
__device__ __forceinline__ int dc_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- tile movers ------------------------------------------------------------------------------------------------------------
// global bf16 rows [.., ld] (columns [0, K)) -> activation tile (ldk = K).  CLAMP: the source has only M rows (rows >= M re-read row M-1)
template <int K, bool CLAMP>
__device__ __forceinline__ void dc_load_a(u16* A, const u16* __restrict__ src, int ld, int row0, int M, int tid) {
  constexpr int CPR = K / 8;
  static_assert((DC_BM * CPR) % DC_THREADS == 0, "tile chunks must split evenly over the workgroup");
  DC_FOR_TID(c, DC_BM * CPR) {
    const int row = c / CPR, ch = c % CPR;
    const int gr = CLAMP ? min(row0 + row, M - 1) : row0 + row;
    *(u16x8*)(A + dc_aoff(row, ch * 8, K)) = *(const u16x8*)(src + (size_t)gr * ld + ch * 8);
  }
}
// activation tile -> global bf16 rows (padded destination)
template <int K>
__device__ __forceinline__ void dc_store_a(const u16* A, u16* __restrict__ dst, int ld, int row0, int tid) {
  constexpr int CPR = K / 8;
  DC_FOR_TID(c, DC_BM * CPR) {
    const int row = c / CPR, ch = c % CPR;
    *(u16x8*)(dst + (size_t)(row0 + row) * ld + ch * 8) = *(const u16x8*)(A + dc_aoff(row, ch * 8, K));
  }
}
// f32 tile (stride DC_TS) <- global f32 rows [M, 256]: rows >= M are zero (ZERO_TAIL, gradients) or repeat row M-1 (inputs);
// src == nullptr (uniform) fills zeros
template <bool ZERO_TAIL>
__device__ __forceinline__ void dc_load_f(float* T, const float* __restrict__ src, int row0, int M, int tid) {
  if (src == nullptr) {
    DC_FOR_TID(c, DC_BM * 64) *(f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  DC_FOR_TID(c, DC_BM * 64) {
    const int row = c >> 6, q = c & 63;
    f32x4 v = *(const f32x4*)(src + (size_t)min(row0 + row, M - 1) * DC_C + q * 4);
    if (ZERO_TAIL) {
      const float keep = row0 + row < M ? 1.f : 0.f;
      v *= keep;
    }
    *(f32x4*)(T + row * DC_TS + q * 4) = v;
  }
}
// padded workspace rows -> tile / tile -> padded rows
__device__ __forceinline__ void dc_load_f_rows(float* T, const float* __restrict__ src, int row0, int tid) {
  DC_FOR_TID(c, DC_BM * 64) *(f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4) = *(const f32x4*)(src + (size_t)(row0 + (c >> 6)) * DC_C + (c & 63) * 4);
}
__device__ __forceinline__ void dc_store_f(const float* T, float* __restrict__ dst, int row0, int tid) {
  DC_FOR_TID(c, DC_BM * 64) *(f32x4*)(dst + (size_t)(row0 + (c >> 6)) * DC_C + (c & 63) * 4) = *(const f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4);
}

// ---- LayerNorm over the 256 columns of an f32 tile: wave w owns rows 8w..8w+7, a lane 4 consecutive columns ------------------
struct DcLnOut {
  float* tile;        // f32 tile to receive y (may alias the input tile), or null
  u16* a; int a_ldk;  // activation tile to receive bf16(y), or null
  float* g32;         // global f32 [M,256], or null
  u16* g16;           // global bf16 [M,256], or null
  float* mr;          // global f32 [M,16]: (mean, rstd) at columns 2*idx, 2*idx+1 (saved for the backward), or null
  int mr_idx;
  bool round_out;     // y rounded through bf16 before it is used as f32 (outputs that are bf16 tensors in the layer-by-layer formulation)
  float* gpre;        // global f32 [M,256] to receive the INPUT rows (saved for the backward), or null
};
__device__ __forceinline__ void dc_layernorm(const float* T, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                             bool relu, const DcLnOut& o, int row0, int wave, int lane) {
  const f32x4 ga = *(const f32x4*)(gamma + lane * 4), be = *(const f32x4*)(beta + lane * 4);
#pragma unroll 2
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
    const f32x4 v = *(const f32x4*)(T + row * DC_TS + lane * 4);
    const float mu = u3d_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / DC_C);
    const f32x4 d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
    const float rs = rsqrtf(u3d_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / DC_C) + eps);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = d[j] * rs * ga[j] + be[j];
      if (relu) y[j] = fmaxf(y[j], 0.f);
    }
    const u16x4 yb = dc_pack4(y);
    if (o.round_out) y = dc_unpack4(yb);
    const size_t grow = (size_t)(row0 + row);
    if (o.gpre) *(f32x4*)(o.gpre + grow * DC_C + lane * 4) = v;
    if (o.tile) *(f32x4*)(o.tile + row * DC_TS + lane * 4) = y;
    if (o.a) *(u16x4*)(o.a + dc_aoff(row, lane * 4, o.a_ldk)) = yb;
    if (o.g32) *(f32x4*)(o.g32 + grow * DC_C + lane * 4) = y;
    if (o.g16) *(u16x4*)(o.g16 + grow * DC_C + lane * 4) = yb;
    if (o.mr) {                       // every lane stores the same two values: no lane-dependent branch
      o.mr[grow * 16 + 2 * o.mr_idx] = mu;
      o.mr[grow * 16 + 2 * o.mr_idx + 1] = rs;
    }
  }
}

// ---- trilinear corner setup (F.grid_sample, align_corners=False, zeros padding; grid = (sigmoid(ref) - 0.5) * 2) --------------
struct DcCorners { int row[8]; float w[8], dwx[8], dwy[8], dwz[8]; };
// ref3: the three reference-point logits of ONE query, identical in all lanes -> row ids are made wave-uniform (scalar branches)
__device__ __forceinline__ void dc_corners(const float* ref3, int b, int D, int H, int W, DcCorners& tc) {
  const float gx = (dc_sigmoid(ref3[0]) - 0.5f) * 2.f, gy = (dc_sigmoid(ref3[1]) - 0.5f) * 2.f, gz = (dc_sigmoid(ref3[2]) - 0.5f) * 2.f;
  const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f, iz = ((gz + 1.f) * D - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float tx = ix - fx, ty = iy - fy, tz = iz - fz;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
    const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
    const float wx = dx ? tx : 1.f - tx, wy = dy ? ty : 1.f - ty, wz = dz ? tz : 1.f - tz;
    const bool ok = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H && (unsigned)z < (unsigned)D;
    tc.row[c] = __builtin_amdgcn_readfirstlane(ok ? ((b * D + z) * H + y) * W + x : -1);
    tc.w[c] = wx * wy * wz;
    tc.dwx[c] = (dx ? 1.f : -1.f) * wy * wz;
    tc.dwy[c] = (dy ? 1.f : -1.f) * wx * wz;
    tc.dwz[c] = (dz ? 1.f : -1.f) * wx * wy;
  }
}

// debugging aid (-DDC_POISON_LDS=1): fill the whole dynamic LDS block with NaN patterns before a row kernel starts, so that any
// read-before-write shows up as NaN instead of depending on what the previous workgroup left behind
#ifndef DC_POISON_LDS
#define DC_POISON_LDS 0
#endif
__device__ __forceinline__ void dc_poison_lds(unsigned char* lds, int tid) {
#if DC_POISON_LDS == 1
  for (int i = tid; i < DC_LDS_BYTES / 4; i += DC_THREADS) ((unsigned*)lds)[i] = 0xFFFFFFFFu;
  __syncthreads();
#elif DC_POISON_LDS == 2
  __syncthreads();
#elif DC_POISON_LDS == 3
  for (int i = tid; i < DC_LDS_BYTES / 4; i += DC_THREADS) ((unsigned*)lds)[i] = 0xFFFFFFFFu;
#endif
}
