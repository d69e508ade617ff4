// Building blocks of the fused decoder-layer kernels (decoder.hip, decoder_bwd.hip): a block of rows of the layer state lives in LDS
// through a whole chain of GEMMs / LayerNorms; weights stream from L2 straight into MFMA fragments (each wave owns 64 output
// columns, so no two waves share a weight row and LDS staging of the weights would buy nothing).
//
// Every kernel is a template over an ELEMENT TRAIT E - the storage type of activations / weights / slots and the MFMA that consumes
// them - and is instantiated twice:
//   EB  bf16 storage, v_mfma_f32_16x16x32_bf16, 32 rows per workgroup            (throughput mode, U3D_BF16)
//   EF  f32 storage,  v_mfma_f32_16x16x4_f32 (bitwise an fma chain), 16 rows     (parity mode, U3D_F32: the 1e-3 reference goldens)
// Same source, same launch structure, same slot layout (element size aside): what the parity tests exercise IS the benchmarked code.
//
// MFMA orientation: D^T = W . X^T, i.e. a-operand = weight rows (lane l: row n = l&15, one 16-byte chunk of k at chunk (l>>4)),
// b-operand = activation rows from LDS (lane l: row m = l&15, same chunk) -> a lane ends up with 4 CONSECUTIVE output columns of
// one row: 8-byte bf16 / 16-byte f32 accesses in every epilogue.  A 16-byte chunk is 8 bf16 (one 16x16x32 MFMA) or 4 f32 (four
// 16x16x4 MFMAs, MFMA j taking element j of every lane's chunk: the reduction index is permuted identically in both operands).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define DC_THREADS 256
#define DC_C 256              /* embed dim */
#define DC_FF 512             /* FFN hidden */
#define DC_TS 260             /* row stride (floats) of the f32 tiles: 8-lane store groups land on distinct banks */
#define DC_NHEAD 8
#define DC_HD 32
#define DC_MISC_LD 48         /* floats per row of the misc tile: [0,16) scalars, [16,48) narrow gradients */

__device__ __forceinline__ float dc_bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 dc_f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ float dc_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- wave reduction on the DPP network --------------------------------------------------------------------------------------------
// u3d_wave_sum (common.h) is six dependent __shfl_xor = ds_bpermute_b32 round trips through the LDS crossbar (~100+ clocks each): a
// LayerNorm row (two dependent sums) cost ~1.4 k clocks - the phase stamps of tools/dec_bench.py put the seven LayerNorms of
// k_dec_post at 23 % of the kernel.  The same sum on the VALU's data-parallel primitives: quad permutes, row mirrors and the two
// row broadcasts (6 dependent v_add_f32 with DPP modifiers), the total read from lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dc_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float dc_wave_sum(float v) {
  v += dc_dpp<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
  v += dc_dpp<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
  v += dc_dpp<0x141, 0xF>(v);       // row_half_mirror
  v += dc_dpp<0x140, 0xF>(v);       // row_mirror: every lane of a 16-lane row now holds the row's sum
  v += dc_dpp<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3
  v += dc_dpp<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- element traits -----------------------------------------------------------------------------------------------------------
struct EB {                    // bf16 storage
  typedef u16 T;
  typedef u16x4 V4;            // 4 consecutive elements (a lane's share of one MFMA output row)
  typedef u16x8 VC;            // one 16-byte chunk
  static constexpr int CH = 8;        // elements per 16-byte chunk
  static constexpr int BM = 32;       // rows per workgroup of the row-chain kernels
  static constexpr int MT = 2;        // 16-row MFMA tiles per workgroup
  static constexpr int KSTEP = 32;    // reduction elements per dc_gemm step (4 chunk columns, one per 16-lane group)
  static constexpr int DT = U3D_BF16;
  static constexpr int MHA_KC = 320;  // rows per LDS chunk of the attention kernels: a 300-query group is ONE chunk
  __device__ static __forceinline__ float to_f(T v) { return dc_bf2f(v); }
  __device__ static __forceinline__ T from_f(float f) { return dc_f2bf(f); }
  __device__ static __forceinline__ float round(float f) { return dc_bf2f(dc_f2bf(f)); }         // value after a store in T
  __device__ static __forceinline__ V4 pack4(f32x4 v) { return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4)); }
  __device__ static __forceinline__ f32x4 unpack4(V4 v) {
    f32x4 r = {dc_bf2f(v[0]), dc_bf2f(v[1]), dc_bf2f(v[2]), dc_bf2f(v[3])};
    return r;
  }
  __device__ static __forceinline__ f32x4 round4(f32x4 v) { return unpack4(pack4(v)); }
  // transcendentals: the hardware approximations (v_exp_f32 / v_rsq_f32, ~1 ulp) - far below bf16 storage rounding
  __device__ static __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
  __device__ static __forceinline__ float sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
  __device__ static __forceinline__ float rsqrt(float x) { return rsqrtf(x); }
  __device__ static __forceinline__ VC zero_chunk() { return (VC){0, 0, 0, 0, 0, 0, 0, 0}; }
  __device__ static __forceinline__ float chunk_elem(const VC& v, int e) { return dc_bf2f(v[e]); }
  // acc += sum over the chunk's reduction elements of w . a
  __device__ static __forceinline__ void mma(const VC& w, const VC& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
};
struct EF {                    // f32 storage, exact-f32 MFMA
  typedef float T;
  typedef f32x4 V4;
  typedef f32x4 VC;
  static constexpr int CH = 4;
  static constexpr int BM = 16;
  static constexpr int MT = 1;
  static constexpr int KSTEP = 16;
  static constexpr int DT = U3D_F32;
  static constexpr int MHA_KC = 64;
  __device__ static __forceinline__ float to_f(T v) { return v; }
  __device__ static __forceinline__ T from_f(float f) { return f; }
  __device__ static __forceinline__ float round(float f) { return f; }
  __device__ static __forceinline__ V4 pack4(f32x4 v) { return v; }
  __device__ static __forceinline__ f32x4 unpack4(V4 v) { return v; }
  __device__ static __forceinline__ f32x4 round4(f32x4 v) { return v; }
  // parity mode: correctly rounded library functions (the softmax backward multiplies p by a difference of near-equal terms, which
  // amplifies a 1-ulp error in p into 1e-4 of the gradient)
  __device__ static __forceinline__ float exp2(float x) { return exp2f(x); }
  __device__ static __forceinline__ float sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
  __device__ static __forceinline__ float rsqrt(float x) { return 1.f / sqrtf(x); }
  __device__ static __forceinline__ VC zero_chunk() { return (VC){0.f, 0.f, 0.f, 0.f}; }
  __device__ static __forceinline__ float chunk_elem(const VC& v, int e) { return v[e]; }
  __device__ static __forceinline__ void mma(const VC& w, const VC& a, f32x4& acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[j], acc, 0, 0, 0);
  }
};

// LDS map (bytes) of the row-chain kernels
template <typename E>
struct DcLds {
  static constexpr int ES = (int)sizeof(typename E::T);
  static constexpr int A0 = 0;                                /* T [BM][512] */
  static constexpr int A1 = A0 + E::BM * 512 * ES;            /* T [BM][512] */
  static constexpr int A2 = A1 + E::BM * 512 * ES;            /* T [BM][256] */
  static constexpr int F = A2 + E::BM * 256 * ES;             /* f32 [BM][260] */
  static constexpr int G = F + E::BM * DC_TS * 4;             /* f32 [BM][260] */
  static constexpr int MISC = G + E::BM * DC_TS * 4;          /* f32 [BM][48] */
  static constexpr int BYTES = MISC + E::BM * DC_MISC_LD * 4;
};
static_assert(DcLds<EB>::BYTES <= 160 * 1024 && DcLds<EF>::BYTES <= 160 * 1024, "row-chain tiles must fit the CU's LDS");

// element offset of (row, col) in an activation tile whose rows hold `ldk` elements: the 16-byte chunk index is XORed with
// row & 15, so the ds_read_b128 of one MFMA operand (16 rows x one chunk column per 16-lane group) spreads over the banks
template <typename E>
__device__ __forceinline__ int dc_aoff(int row, int col, int ldk) {
  return row * ldk + ((((col / E::CH) ^ (row & 15)) * E::CH) | (col % E::CH));
}

// ---- dropout: keep decision of element idx at (layer, site); identical in forward and backward ----------------------------
// One 32-bit hash word serves TWO neighbouring elements (idx >> 1 selects the word, idx & 1 its 16-bit half; keep <=> half >= p * 2^16,
// i.e. p is honoured to 2^-16).  The attention kernels hash every score element: with a word per element and four 32-bit multiplies
// per word the hashing cost three times the softmax itself (quarter-rate v_mul_lo_u32).  Attention rows are padded to an even length
// in index space (dc_att_idx) so that the 4 consecutive keys a lane owns always start on a pair boundary.
struct DcRng { unsigned lo, hi; };
__device__ __forceinline__ DcRng dc_rng_load(const unsigned long long* p) {
  const unsigned long long v = p ? *p : 0ull;
  DcRng r = {(unsigned)v, (unsigned)(v >> 32)};
  return r;
}
__device__ __forceinline__ unsigned dc_site_key(int layer, int site) { return (unsigned)(layer * 8 + site + 1) * 0x85EBCA77u; }
__device__ __forceinline__ unsigned dc_hash_pair(DcRng g, unsigned site_key, unsigned pair) {
  unsigned h = pair * 0x9E3779B1u + (g.lo ^ site_key);
  h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h ^= g.hi; h *= 0xC2B2AE3Du; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool dc_keep_half(unsigned word, unsigned odd, unsigned thresh) { return ((word >> (odd * 16u)) & 0xFFFFu) >= thresh; }
__device__ __forceinline__ bool dc_keep(DcRng g, unsigned site_key, unsigned idx, unsigned thresh) {
  return dc_keep_half(dc_hash_pair(g, site_key, idx >> 1), idx & 1u, thresh);
}
// keep decisions of 4 consecutive elements idx0 .. idx0+3, idx0 EVEN: two hash words
__device__ __forceinline__ void dc_keep4(DcRng g, unsigned site_key, unsigned idx0, unsigned thresh, bool (&k)[4]) {
  const unsigned w0 = dc_hash_pair(g, site_key, idx0 >> 1), w1 = dc_hash_pair(g, site_key, (idx0 >> 1) + 1u);
  k[0] = dc_keep_half(w0, 0u, thresh); k[1] = dc_keep_half(w0, 1u, thresh);
  k[2] = dc_keep_half(w1, 0u, thresh); k[3] = dc_keep_half(w1, 1u, thresh);
}
// index of attention weight (row = (group * 8 + head) * nq + query, key): rows of nq_pad = nq rounded up to even
__device__ __forceinline__ unsigned dc_att_idx(unsigned row, unsigned key, unsigned nq_pad) { return row * nq_pad + key; }
__host__ __device__ static inline unsigned dc_thresh(float p) { return p <= 0.f ? 0u : (unsigned)(p * 65536.0f); }
__host__ __device__ static inline float dc_inv_keep(float p) { return p <= 0.f ? 1.f : 1.f / (1.f - p); }

struct DcDrop {
  DcRng rng; unsigned thresh; float inv_keep; int layer;
  __device__ __forceinline__ f32x4 apply(f32x4 v, int site, unsigned idx0) const {     // 4 consecutive elements idx0..idx0+3, idx0 % 4 == 0
    if (thresh == 0u) return v;
    bool k[4];
    dc_keep4(rng, dc_site_key(layer, site), idx0, thresh, k);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = k[r] ? v[r] * inv_keep : 0.f;
    return v;
  }
};

// ---- the block GEMM: acc[mt][nt] (+)= X[BM rows] . W[NT*16 rows]^T over K ---------------------------------------------------
// A: LDS activation tile (ldk = K); W: global weights, fragment-packed, already offset to the wave's first 16-row tile.
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
template <typename E, int K, int NT>
__device__ __forceinline__ void dc_gemm(const typename E::T* A, const typename E::T* __restrict__ W, f32x4 (&acc)[E::MT][NT], int lane) {
  typedef typename E::T T;
  typedef typename E::VC VC;
  constexpr int KS = K / E::KSTEP;
  constexpr int PF = dc_burst(KS, DC_PF);
  static_assert((K / E::CH) % 16 == 0, "the chunk swizzle permutes aligned groups of 16 chunks");
#ifdef DC_ABL_NOGEMM
  return;
#endif
  const int r16 = lane & 15, kq = lane >> 4;
  // weights are FRAGMENT-PACKED (u3d_wpack): block (16-row tile, k-step) = 64 lanes x 16 bytes in lane order, so one wave load is
  // 1 KiB contiguous (8 cache lines).  Row-major weights made every 16-lane group of a dwordx4 load touch 16 different lines - the
  // vector L1 then serialised ~64 tag lookups per instruction and the row kernels ran at ~8 B/clk/CU of weight stream
  // (phase stamps: 16 k clocks per 256 x 256 linear against 1.1 k clocks of MFMAs).
  constexpr int WBLK = 64 * E::CH;                   // elements per (tile, k-step) block
  const T* wp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = W + (size_t)nt * KS * WBLK + lane * E::CH;
  const T* ap[E::MT];
#pragma unroll
  for (int mt = 0; mt < E::MT; ++mt) ap[mt] = A + (mt * 16 + r16) * K;
  // k-steps are walked from a per-workgroup starting point: every workgroup of a launch streams the SAME weight matrix at about the
  // same time, and in lock step they all queue on the one or two L2 channels that hold the current lines
#ifndef DC_NO_KROT
  const int rot = (int)(blockIdx.x % KS);
#else
  const int rot = 0;
#endif
#pragma unroll
  for (int kb = 0; kb < KS; kb += PF) {
    VC wf[PF][NT];
    int kk[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      kk[p] = kb + p + rot;
      kk[p] = kk[p] >= KS ? kk[p] - KS : kk[p];
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[p][nt] = *(const VC*)(wp[nt] + kk[p] * WBLK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int ch = ((kk[p] * 4 + kq) ^ r16) * E::CH;
      VC a[E::MT];
#pragma unroll
      for (int mt = 0; mt < E::MT; ++mt) a[mt] = *(const VC*)(ap[mt] + ch);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < E::MT; ++mt) E::mma(wf[p][nt], a[mt], acc[mt][nt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// y[BM rows][ncol0 .. ncol0 + NT*16) = X . W^T; epi(row, col, v) gets 4 consecutive columns col..col+3 of row `row`
template <typename E, int K, int NT, typename Epi>
__device__ __forceinline__ void dc_linear(const typename E::T* A, const typename E::T* __restrict__ W, int ncol0, int lane, Epi epi) {
  f32x4 acc[E::MT][NT];
#pragma unroll
  for (int mt = 0; mt < E::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  dc_gemm<E, K, NT>(A, W + (size_t)ncol0 * K, acc, lane);
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < E::MT; ++mt)
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
// index: row matrices owned by this library (save / gradient slots, outputs) hold whole BM-row blocks (u3d_decoder_layer_blocks_dt(m)
// * BM rows), caller-owned inputs of m rows are read through a clamped row index.  Reason: hipcc (ROCm 7.2) was seen to place
// register-allocator copies (v_accvgpr_write) in the exit block of an exec-masked `for (i = tid; i < N; i += 256)` loop AHEAD of the
// `s_or_b64 exec` that restores the lane mask - the copies ran with EXEC = 0 and the "saved" LDS addresses were garbage afterwards
// (tools/check_exec_restore.py scans the ISA for that pattern; tests/test_build_cpu.py runs it).
#define DC_FOR_TID(c, N) _Pragma("unroll") for (int it_ = 0, c = tid; it_ < (N) / DC_THREADS; ++it_, c += DC_THREADS)
__device__ __forceinline__ int dc_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- tile movers ------------------------------------------------------------------------------------------------------------
// global rows [.., ld] (columns [0, K)) -> activation tile (ldk = K).  CLAMP: the source has only M rows (rows >= M re-read row M-1)
template <typename E, int K, bool CLAMP>
__device__ __forceinline__ void dc_load_a(typename E::T* A, const typename E::T* __restrict__ src, int ld, int row0, int M, int tid) {
  typedef typename E::VC VC;
  constexpr int CPR = K / E::CH;
  static_assert((E::BM * CPR) % DC_THREADS == 0, "tile chunks must split evenly over the workgroup");
  DC_FOR_TID(c, E::BM * CPR) {
    const int row = c / CPR, ch = c % CPR;
    const int gr = CLAMP ? min(row0 + row, M - 1) : row0 + row;
    *(VC*)(A + dc_aoff<E>(row, ch * E::CH, K)) = *(const VC*)(src + (size_t)gr * ld + ch * E::CH);
  }
}
// activation tile -> global rows (padded destination)
template <typename E, int K>
__device__ __forceinline__ void dc_store_a(const typename E::T* A, typename E::T* __restrict__ dst, int ld, int row0, int tid) {
  typedef typename E::VC VC;
  constexpr int CPR = K / E::CH;
  DC_FOR_TID(c, E::BM * CPR) {
    const int row = c / CPR, ch = c % CPR;
    *(VC*)(dst + (size_t)(row0 + row) * ld + ch * E::CH) = *(const VC*)(A + dc_aoff<E>(row, ch * E::CH, K));
  }
}
// f32 tile (stride DC_TS) <- global f32 rows [M, 256]: rows >= M are zero (ZERO_TAIL, gradients) or repeat row M-1 (inputs);
// src == nullptr (uniform) fills zeros
template <typename E, bool ZERO_TAIL>
__device__ __forceinline__ void dc_load_f(float* T, const float* __restrict__ src, int row0, int M, int tid) {
  if (src == nullptr) {
    DC_FOR_TID(c, E::BM * 64) *(f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  DC_FOR_TID(c, E::BM * 64) {
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
template <typename E>
__device__ __forceinline__ void dc_load_f_rows(float* T, const float* __restrict__ src, int row0, int tid) {
  DC_FOR_TID(c, E::BM * 64) *(f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4) = *(const f32x4*)(src + (size_t)(row0 + (c >> 6)) * DC_C + (c & 63) * 4);
}
template <typename E>
__device__ __forceinline__ void dc_store_f(const float* T, float* __restrict__ dst, int row0, int tid) {
  DC_FOR_TID(c, E::BM * 64) *(f32x4*)(dst + (size_t)(row0 + (c >> 6)) * DC_C + (c & 63) * 4) = *(const f32x4*)(T + (c >> 6) * DC_TS + (c & 63) * 4);
}

// ---- LayerNorm over the 256 columns of an f32 tile: wave w owns rows RPW*w .. RPW*w + RPW-1, a lane 4 consecutive columns ------
template <typename E>
struct DcLnOut {
  float* tile;                 // f32 tile to receive y (may alias the input tile), or null
  typename E::T* a; int a_ldk; // activation tile to receive T(y), or null
  float* g32;                  // global f32 [M,256], or null
  typename E::T* g16;          // global T [M,256], or null
  float* mr;                   // global f32 [M,16]: (mean, rstd) at columns 2*idx, 2*idx+1 (saved for the backward), or null
  int mr_idx;
  bool round_out;              // y rounded through T before it is used as f32 (outputs that are T tensors in the layer-by-layer formulation)
  float* gpre;                 // global f32 [M,256] to receive the INPUT rows (saved for the backward), or null
  typename E::T* gpre16;       // global T [M,256] to receive the INPUT rows (values that are already T-representable), or null
};
template <typename E>
__device__ __forceinline__ void dc_layernorm(const float* T, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                             bool relu, const DcLnOut<E>& o, int row0, int wave, int lane) {
  typedef typename E::V4 V4;
  constexpr int RPW = E::BM / 4;
  const f32x4 ga = *(const f32x4*)(gamma + lane * 4), be = *(const f32x4*)(beta + lane * 4);
#pragma unroll 2
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr;
    const f32x4 v = *(const f32x4*)(T + row * DC_TS + lane * 4);
    const float mu = dc_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / DC_C);
    const f32x4 d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
    const float rs = E::rsqrt(dc_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / DC_C) + eps);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = d[j] * rs * ga[j] + be[j];
      if (relu) y[j] = fmaxf(y[j], 0.f);
    }
    const V4 yb = E::pack4(y);
    if (o.round_out) y = E::unpack4(yb);
    const size_t grow = (size_t)(row0 + row);
    if (o.gpre) *(f32x4*)(o.gpre + grow * DC_C + lane * 4) = v;
    if (o.gpre16) *(V4*)(o.gpre16 + grow * DC_C + lane * 4) = E::pack4(v);
    if (o.tile) *(f32x4*)(o.tile + row * DC_TS + lane * 4) = y;
    if (o.a) *(V4*)(o.a + dc_aoff<E>(row, lane * 4, o.a_ldk)) = yb;
    if (o.g32) *(f32x4*)(o.g32 + grow * DC_C + lane * 4) = y;
    if (o.g16) *(V4*)(o.g16 + grow * DC_C + lane * 4) = yb;
    if (o.mr) {                       // every lane stores the same two values: no lane-dependent branch
      o.mr[grow * 16 + 2 * o.mr_idx] = mu;
      o.mr[grow * 16 + 2 * o.mr_idx + 1] = rs;
    }
  }
}

// ---- trilinear corner setup (F.grid_sample, align_corners=False, zeros padding; grid = (sigmoid(ref) - 0.5) * 2) --------------
struct DcCorners { int row[8]; float w[8], dwx[8], dwy[8], dwz[8]; };
// ref3: the three reference-point logits of ONE query, identical in all lanes -> row ids are made wave-uniform (scalar branches)
template <typename E>
__device__ __forceinline__ void dc_corners(const float* ref3, int b, int D, int H, int W, DcCorners& tc) {
  const float gx = (E::sigmoid(ref3[0]) - 0.5f) * 2.f, gy = (E::sigmoid(ref3[1]) - 0.5f) * 2.f, gz = (E::sigmoid(ref3[2]) - 0.5f) * 2.f;
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
template <typename E>
__device__ __forceinline__ void dc_poison_lds(unsigned char* lds, int tid) {
#if DC_POISON_LDS == 1
  for (int i = tid; i < DcLds<E>::BYTES / 4; i += DC_THREADS) ((unsigned*)lds)[i] = 0xFFFFFFFFu;
  __syncthreads();
#elif DC_POISON_LDS == 2
  __syncthreads();
#elif DC_POISON_LDS == 3
  for (int i = tid; i < DcLds<E>::BYTES / 4; i += DC_THREADS) ((unsigned*)lds)[i] = 0xFFFFFFFFu;
#endif
}

// ablation hook (tools/dec_bench.py): -DDC_ABL_NOFRAGSTORE drops the MFMA-fragment-layout global stores (timing experiments only)
#ifdef DC_ABL_NOFRAGSTORE
#define DC_FRAG_STORE(stmt)
#else
#define DC_FRAG_STORE(stmt) stmt
#endif

// ---- slot geometry shared by the forward and the backward translation units ---------------------------------------------------------
__host__ static inline int dc_esize(int dtype) { return dtype == U3D_BF16 ? 2 : 4; }
__host__ static inline int dc_bm(int dtype) { return dtype == U3D_BF16 ? EB::BM : EF::BM; }

// ---- attention building blocks (k_mha_fwd in decoder.hip, k_mha_bwd_* in decoder_bwd.hip) -----------------------------------------
// A head slice of a row is 32 elements = NP 16-byte parts.  A chunk of KC rows is staged in LDS row-major.
//   EB: rows of RS = 48 elements (96 bytes): the 16 rows of a ds_read_b128 fragment and the 4 x 4 blocks of a ds_read_b64_tr_b16
//       transpose read both land on distinct banks, so ONE image serves as row operand (scores) and as transposed operand (the
//       products that reduce over the rows) - no transposed copy, no 2-byte LDS stores.  KC = 320: the 300 keys of a SUN RGB-D /
//       ScanNet / KITTI group are one chunk (one staging round, no running-max rescale).
//   EF: f32 has no transpose read: swizzled 32-element rows plus an explicit transposed copy [32][KC + 8]; KC = 64.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <typename E>
struct Mha {
  typedef typename E::T T;
  typedef typename E::VC VC;
  static constexpr bool TR = E::CH == 8;         // hardware transpose reads available (16-bit elements)
  static constexpr int KC = E::MHA_KC;
  static constexpr int NP = DC_HD / E::CH;       // parts per row: 4 (bf16) / 8 (f32)
  static constexpr int RS = TR ? 48 : 32;        // row stride (elements) of a row-major image
  static constexpr int TLD = KC + 8;             // row stride of a transposed copy [32][KC] (EF only)
  static constexpr int RM_ELEMS = KC * RS;       // elements of a row-major image
  static constexpr int TP_ELEMS = TR ? 8 : 32 * TLD;     // elements of a transposed copy (EB: a stub, never touched)
  __device__ static __forceinline__ int roff(int row, int part) {
    if constexpr (TR) return row * RS + part * 8;
    else return row * 32 + (((part ^ row) & 7) << 2);
  }
  // stage KC rows (head slice) of a row matrix into LDS: the row-major image and (EF) the transposed copy
  __device__ static __forceinline__ void stage(const T* __restrict__ src, int ld, long long base_row, int first, int nvalid, T* rowmajor,
                                               T* transposed, int tid) {
    DC_FOR_TID(c, KC * NP) {
      const int key = c / NP, part = c % NP;
      VC v = E::zero_chunk();
      if (first + key < nvalid) v = *(const VC*)(src + (base_row + first + key) * ld + part * E::CH);
      if (rowmajor) *(VC*)(rowmajor + roff(key, part)) = v;
      if constexpr (!TR) {
        if (transposed) {
#pragma unroll
          for (int e = 0; e < E::CH; ++e) transposed[(part * E::CH + e) * TLD + key] = v[e];
        }
      }
    }
  }
  // a lane's share of a 32-element row as the b-operand of the score product (lane: row l&15, 16-lane group kq)
  struct RowFrag { VC c[NP / 4]; };
  __device__ static __forceinline__ RowFrag zero_frag() {
    RowFrag f;
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) f.c[j] = E::zero_chunk();
    return f;
  }
  __device__ static __forceinline__ RowFrag load_frag(const T* __restrict__ row32, int kq) {       // from global: row32 -> the head slice
    RowFrag f;
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) f.c[j] = *(const VC*)(row32 + (j * 4 + kq) * E::CH);
    return f;
  }
  __device__ static __forceinline__ float dot(const RowFrag& a, const RowFrag& b) {               // partial dot over this lane's elements
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < NP / 4; ++j)
#pragma unroll
      for (int e = 0; e < E::CH; ++e) d += E::chunk_elem(a.c[j], e) * E::chunk_elem(b.c[j], e);
    return d;
  }
  // S^T tile [16 rows of the image starting at row0][16 columns of f] = R . frag^T
  __device__ static __forceinline__ f32x4 scores(const T* rowmajor, int row0, int r16, int kq, const RowFrag& f) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) E::mma(*(const VC*)(rowmajor + roff(row0 + r16, j * 4 + kq)), f.c[j], acc);
    return acc;
  }
  // acc[dt] += X^T[dt*16 .. +16)[rows of the tile pair tp] . P^T: p0 / p1 = the lane's 4 values (rows kq*4 + r) of tiles 2tp, 2tp+1.
  // EB: X = the row-major image (transpose reads); EF: X = the transposed copy.
  __device__ static __forceinline__ void pv(const T* rowmajor, const T* transposed, int tp, int r16, int kq, f32x4 p0, f32x4 p1,
                                            f32x4 (&acc)[2]) {
    if constexpr (TR) {
      (void)transposed;
      const u16x4 a4 = EB::pack4(p0), b4 = EB::pack4(p1);
      const u16x8 pb = {a4[0], a4[1], a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};
      // 8 reduction values per lane: rows tp*32 + 4*kq + {0..3} and + 16 of column dt*16 + r16 (the order p0 | p1 has)
      const int j = r16 >> 2, q = r16 & 3;
      const u16* p0a = (const u16*)rowmajor + (tp * 32 + 4 * kq + j) * RS + 4 * q;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0a + dt * 16));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0a + dt * 16 + 16 * RS));
        const s16x8 vb = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vb), __builtin_bit_cast(bf16x8, pb), acc[dt], 0, 0, 0);
      }
    } else {
      (void)rowmajor;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 p = u ? p1 : p0;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x4 v = *(const f32x4*)((const float*)transposed + (dt * 16 + r16) * TLD + (tp * 2 + u) * 16 + kq * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], p[j], acc[dt], 0, 0, 0);
        }
      }
    }
  }
};

// query / key tiles (64 rows) one workgroup walks: up to 3 when a group fits one staged chunk (5 tiles of a 300-query group -> two
// workgroups of 3 + 2 tiles, every workgroup resident in one round), else one (chunks are re-staged per tile)
template <typename E>
static inline int mha_tiles_per_wg(int nq) { return nq <= Mha<E>::KC ? 3 : 1; }
// bf16 layers whose attention groups fit one LDS chunk run in-projection + attention as ONE launch (k_mha_proj_fwd, decoder.hip), which
// also publishes the attention-dropout keep bits (U3D_DS_AMASK) that the layer's backward then reads instead of hashing again
#ifndef MHA_FUSED_INPROJ
#define MHA_FUSED_INPROJ 1
#endif
static inline bool mha_fused_inproj(int dtype, int nq) { return MHA_FUSED_INPROJ && dtype == U3D_BF16 && nq <= Mha<EB>::KC; }

