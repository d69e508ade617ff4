// bf16 implicit-GEMM convolution kernels, second generation (large tiles, double-buffered LDS, register-staged
// prefetch, LDS transpose reads).  They serve BOTH the sparse levels and the dense SECOND3D/FPN lattice: the only
// difference is where the neighbour table comes from.
//
//   forward / dgrad :  out[m, n] = sum_kappa sum_k  in[nbr[kappa][m], k] * W[kappa](k, n)
//   weight gradient :  dW[kappa](ci, co) = sum_m in[nbr[kappa][m], ci] * dout[m, co]
//
// MFMA: v_mfma_f32_16x16x32_bf16, f32 accumulation.  Tiles up to 256x256 per workgroup (8 waves) so that the
// L2->CU traffic per MFMA cycle stays below ~36 B/clk (a 128x128 tile would need ~64 B/clk: DESIGN.md §kernels).
// Operands whose reduction index is the LDS row (W[k][n] in forward, both operands in wgrad) are fetched with
// ds_read_b64_tr_b16; rows are padded by 16 elements so that those reads are bank-conflict free.
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// BatchNorm-BACKWARD statistics in the epilogue of an input-gradient launch (glds_epilogue.inc; == u3d_bn_epi of the C ABI): the
// tensor this launch writes is dy of the BatchNorm that produced the conv's input, so sum(g) and sum(g * xhat) (g = dy under the
// ReLU mask) can leave per row tile with the store - the separate pass over dy and x (k_col_stats_vec) disappears.
struct BnEpi {
  const u16* x = nullptr;        // the BatchNorm's input (bf16 [n][C]); nullptr: off
  const u16* y = nullptr;        // its output, for the ReLU mask of layers with a residual; nullptr: mask recomputed from x
  const float *mean = nullptr, *invstd = nullptr, *gamma = nullptr, *beta = nullptr;
  int relu = 0, pad = 0;
};
// BatchNorm-backward sums of one BM x BN output tile whose rounded dy the epilogue left in LDS (16 B chunks, XOR-swizzled rows):
// the per-tile form of k_col_stats_vec<., 1> (rowops.hip) - a thread owns 8 columns and every R-th row, x (and y for the mask of
// layers with a residual) arrive by coalesced 16 B loads, the R row lanes are added in a fixed order, one f64 partial per
// (tile, column) leaves.  It does not touch the accumulators (dead by then): a pass over the MFMA fragments cost the kernels their
// register allocation (128 x 128 tiles: 180 -> 288 VGPRs; so did calling this out of line), and fragment-shaped 8 B global loads of
// x made every (row block, column block) step wait for its own round trip (+35 us per launch).
template <int BM, int BN, int NT>
__device__ __forceinline__ void bn_bwd_tile_sums(const BnEpi bn, u16* smem, int m0, int col0, int n_out, int cout, int tile,
                                              double* __restrict__ stats, int tid) {
  constexpr int CPR = BN / 8, SWZ = (CPR - 1) & 15, R = NT / CPR, ROWS = BM / R, BATCH = ROWS < 8 ? ROWS : (ROWS % 8 == 0 ? 8 : (ROWS % 6 == 0 ? 6 : 4));
  static_assert(NT % CPR == 0 && BM % R == 0 && ROWS % BATCH == 0, "tile / thread mismatch");
  const int cc = tid % CPR, r0 = tid / CPR;
  const int col = col0 + cc * 8;
  const bool from_y = bn.relu && bn.y, remask = bn.relu && !bn.y;
  float s0[8], s1[8], mu[8], is[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; ga[e] = 0.f; be[e] = 0.f; }
  __syncthreads();                                      // every wave's dy fragments are in LDS
  if (col < cout) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = bn.mean[col + e]; is[e] = bn.invstd[col + e]; }
    if (remask) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ga[e] = bn.gamma[col + e]; be[e] = bn.beta[col + e]; }
    }
    for (int i0 = 0; i0 < ROWS; i0 += BATCH) {
      bf16x8 xv[BATCH], yv[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        const int r = r0 + (i0 + i) * R;
        const long long o = (long long)(m0 + r) * cout + col;
        if (m0 + r < n_out) {
          xv[i] = *(const bf16x8*)(bn.x + o);
          if (from_y) yv[i] = *(const bf16x8*)(bn.y + o);
        }
      }
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        const int r = r0 + (i0 + i) * R;
        if (m0 + r >= n_out) continue;
        const bf16x8 dv = *(const bf16x8*)(smem + (r * CPR + (cc ^ (r & SWZ))) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = ((float)xv[i][e] - mu[e]) * is[e];
          float gm = (float)dv[e];
          const float yy = from_y ? (float)yv[i][e] : xh * ga[e] + be[e];
          if (bn.relu && !(yy > 0.f)) gm = 0.f;
          s0[e] += gm;
          s1[e] += gm * xh;
        }
      }
    }
  }
  __syncthreads();                                      // the dy tile is dead: the buffers take the row lanes' sums
  float* red = (float*)smem;                            // [2][R][BN]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(0 * R + r0) * BN + cc * 8 + e] = s0[e];
    red[(1 * R + r0) * BN + cc * 8 + e] = s1[e];
  }
  __syncthreads();
  for (int t = tid; t < 2 * BN; t += NT) {
    const int which = t / BN, cl = t % BN;
    if (col0 + cl < cout) {
      double a = 0.0;
#pragma unroll 4
      for (int k = 0; k < R; ++k) a += (double)red[(which * R + k) * BN + cl];
      stats[((long long)tile * 2 + which) * cout + col0 + cl] = a;
    }
  }
}

#ifndef IGEMM_SMALL_C
#define IGEMM_SMALL_C 1
#endif
#ifndef IGEMM_DIRECT
#define IGEMM_DIRECT 1     /* narrow sparse levels on the direct-operand kernel (igemm_direct.hip); 0: the tiled BK = 32 kernels below */
#endif
int u3d_launch_igemm_direct(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev, int n_out_cap,
                            int cin, int cout, int kvol, int transpose_w, hipStream_t s, const void* addend = nullptr,
                            double* stats = nullptr, int* stats_blocks = nullptr, int f32acc = 0);
#ifndef IGEMM_SMALL_PF
#define IGEMM_SMALL_PF 1   /* stages of operand loads in flight in the 16/32-channel kernels (1: the wide layers' one-stage pipeline) */
#endif
#ifndef IGEMM_SMALL_WM
#define IGEMM_SMALL_WM 4   /* 16-row blocks per wave: 4 waves x 4 = 256-row tiles */
#endif
#ifndef GLDS256_CFG
#define GLDS256_CFG 2, 4, 8, 4   /* waves (rows x cols) and 16x16 blocks per wave (rows x cols) of the 256x256-tile LDS-DMA kernel */
#endif
#ifndef WGRAD_DMA_SPREAD
#define WGRAD_DMA_SPREAD 1   /* the same for the LDS-DMA weight-gradient kernel */
#endif
#ifndef GLDS_DMA_SPREAD
#define GLDS_DMA_SPREAD 2   /* 1: one LDS-DMA instruction behind each MFMA group of the stage, 2: all of them within the first k-step */
#endif
#ifndef GLDS_EXP
#define GLDS_EXP 0          /* timing experiments only (wrong results): 1 no LDS-DMA after the first stage, 2 no fragment reads in the loop, 4 activation rows from a 1024-row (cache-resident) window */
#endif
#ifndef GLDS_FRAG_B128
#define GLDS_FRAG_B128 1   /* fragments as one ds_read_b128 per lane (0: two ds_read_b64 in the instruction's nominal k-order) */
#endif
#ifndef GLDS_PIPE
#define GLDS_PIPE 0     /* 1: unit-level fragment pipeline in the LDS-DMA forward/dgrad kernels (igemm_glds_body); measured time-neutral */
#endif
#ifndef IGEMM_STORE_KS
#define IGEMM_STORE_KS 0   /* k-step after which the next stage's registers are written to LDS (0: mid-stage, 1: end of stage) */
#endif

#define LDS_PTR(p) ((s16x4 __attribute__((address_space(3)))*)(p))

__device__ __forceinline__ u16 f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// 8 reduction-index values for one MFMA operand, reduction index = LDS row.  tile: row-major, `stride` elements per
// row; returns values (rows k0 + r(g,e), column c0 + (lane&15)), r(g,e) = e<4 ? 4g+e : 16+4g+(e-4), g = lane>>4.
__device__ __forceinline__ bf16x8 tr_frag(const u16* tile, int stride, int k0, int c0, int lane) {
  const int g = lane >> 4, L = lane & 15, j = L >> 2, q = L & 3;
  const u16* p0 = tile + (k0 + 4 * g + j) * stride + c0 + 4 * q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0 + 16 * stride));
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// 8 reduction-index values, reduction index contiguous in the LDS row: row r0 + (lane&15), elements k0 + kmap(g,e) with the
// SAME r(g,e) permutation as tr_frag (so a tr operand and a direct operand can be paired in one MFMA).
__device__ __forceinline__ bf16x8 direct_frag(const u16* tile, int stride, int r0, int k0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const u16* p = tile + (r0 + i) * stride + k0 + 4 * g;
  // two SEPARATE ds_read_b64 (volatile keeps hipcc from fusing them into ds_read2_b64, whose 16-lane groups / mod-32 banking
  // put rows r and r+8 of the 36-dword-stride tile on the same banks: PMC showed SQ_LDS_BANK_CONFLICT = 36 % of LDS cycles)
  typedef const volatile s16x4 __attribute__((address_space(3))) * lds_vptr;
  s16x4 a = *(lds_vptr)(p);
  s16x4 b = *(lds_vptr)(p + 16);
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// =============================================================================================
// forward / dgrad
//   tile BM x BN, BM = WAVES_M*WM*16, BN = WAVES_N*WN*16, BK = 64 reduction elements per stage.
//   W_KMAJOR = true : global W[kappa][k][n]  (forward; staged row-major [k][n], fetched with transpose reads)
//   W_KMAJOR = false: global W[kappa][n][k]  (dgrad: the forward weight read transposed; staged [n][k], direct reads)
// =============================================================================================
template <int WAVES_M, int WAVES_N, int WM, int WN, bool W_KMAJOR, int BK = 64, int PF = 1>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void k_igemm_fwd(const u16* __restrict__ in, const u16* __restrict__ w,
                                                                      const int* __restrict__ nbr, int ld, u16* __restrict__ out,
                                                                      const int* __restrict__ n_out_dev, int n_out_cap, int cin,
                                                                      int cout, int kvol, const float* __restrict__ bias, int relu) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;   // BK = 64, or 32 for the 16/32-channel sparse levels (one MFMA k-step)
  constexpr int LDA = BK + 8;                           // A tile [BM][BK] (k contiguous): 36- / 20-dword stride -> conflict-free b64 reads
  constexpr int LDW = W_KMAJOR ? BN + 16 : BK + 8;      // W tile [BK][BN] (transpose reads: +16) or [BN][BK] (direct: +8)
  constexpr int A_ELEMS = BM * LDA;
  constexpr int W_ELEMS = W_KMAJOR ? BK * LDW : BN * LDW;
  constexpr int A_SEGS = BM * (BK / 8) / NT;            // 16-byte segments per thread per stage
  constexpr int W_TOTAL = BK * BN / 8;                  // 16-byte segments of the W tile; small tiles leave some threads without one
  constexpr int W_SEGS = (W_TOTAL + NT - 1) / NT;
  static_assert(BM * (BK / 8) % NT == 0 && (BK == 64 || BK == 32), "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  constexpr int STAGE_ELEMS = A_ELEMS + W_ELEMS;        // buffer b: A at smem + b*STAGE_ELEMS, W right behind it

  const int n_out = min(*n_out_dev, n_out_cap);
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only).  Every XCD gets one
  // CONTIGUOUS slab of row tiles, so the halo rows a tile re-gathers for its 27 offsets are shared through that XCD's L2
  // instead of being pulled into all eight L2s.  (Measured: time-neutral on the 3x3x3 layers — they are MFMA-pipeline bound.)
  const int tile = u3d_xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);      // XCD-contiguous ranges of the LIVE tiles (capacity-sized grids)
  if (tile < 0) return;
  const int m0 = tile * BM;
  const int col0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int kchunks = (cin + BK - 1) / BK;              // cin % 16 == 0 (dispatch); a short last chunk is zero-filled
  const int nstage = kvol * kchunks;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Operand fetch = raw buffer loads (32-bit byte offsets, scalar stage offset, hardware zero fill for out-of-range offsets):
  // a missing neighbour is the offset 0xFFFFFFFF, so the whole stage body is ONE branch-free basic block and the scheduler can
  // interleave the next stage's address arithmetic / loads with this stage's MFMAs (with per-load `if (idx >= 0)` branches the
  // ~230 VALU/SALU instructions of the load section ran with the matrix pipe idle: ISA + SQ counters in DESIGN.md §kernels).
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 2u;
  const unsigned a_part16 = (unsigned)(tid % (BK / 8)) * 16u;          // NT % (BK/8) == 0: the same 16-byte part for every segment
  const int a_row0 = tid / (BK / 8);                                   // segment u covers row a_row0 + u * (NT / (BK/8))
  constexpr int A_ROW_STEP = NT / (BK / 8);
  unsigned w_voff[W_SEGS];
  int w_k[W_SEGS];                                      // reduction index (within the chunk) of the segment's first element
#pragma unroll
  for (int u = 0; u < W_SEGS; ++u) {
    int sgi = tid + u * NT;
    const bool live = sgi < W_TOTAL;
    if (W_KMAJOR) {
      int k = sgi / (BN / 8), part = sgi % (BN / 8);
      int n = col0 + part * 8;
      w_k[u] = k;
      w_voff[u] = (live && n < cout) ? (unsigned)(k * cout + n) * 2u : 0xFFFFFFFFu;
    } else {
      int n = sgi / (BK / 8), part = sgi % (BK / 8);
      w_k[u] = part * 8;
      w_voff[u] = (live && col0 + n < cout) ? (unsigned)((col0 + n) * cin + part * 8) * 2u : 0xFFFFFFFFu;
    }
  }

  u32x4 ra[A_SEGS], rw[W_SEGS];
  int idx_cur[A_SEGS], idx_nxt[A_SEGS];

  // Stage order: channel chunk OUTER, kernel offset INNER (stage st = chunk st / kvol, offset st % kvol).  Within one chunk
  // round a tile re-reads the same 128-byte row pieces for all kvol offsets (its 3x3x3 halo window, ~130 KB), so the window of
  // the ~32 tiles an XCD runs at a time (~1.5 MB) stays in that XCD's 4 MB L2; with the offset outer, the re-reads of a row
  // for the next dy / dz offsets came 12 / 36 stages (12 / 37 MB of other traffic per XCD) later and went back to MALL/HBM.
  // The neighbour indices of stage st+2 are fetched while stage st computes: the idx -> row gather chain (two dependent
  // L2 round trips) must never sit in front of a stage.
  auto load_idx_next = [&](int stage) {
    const int kap = stage % kvol;
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) {
      int m = m0 + a_row0 + u * A_ROW_STEP;
      int mc = m < n_out ? m : n_out - 1;                              // clamped, branch-free (n_out >= 1 here)
      // the loaded value is NOT touched here: any use (even a select) makes the compiler wait for it on the spot, and
      // vmcnt is in-order - that wait would also cover the stage loads issued just before it (measured: the whole L2
      // latency exposed once per stage).  Rows past n_out are masked when the index is consumed (advance_idx).
      idx_nxt[u] = nbr ? nbr[(long long)kap * ld + mc] : mc;
    }
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) idx_cur[u] = (m0 + a_row0 + u * A_ROW_STEP < n_out) ? idx_nxt[u] : -1;
  };
  auto issue_loads = [&](int st) {
    const int kap = st % kvol, c0 = (st / kvol) * BK;
    const unsigned a_soff = (unsigned)c0 * 2u;
    const bool a_in = BK == 64 || (c0 + (int)(a_part16 >> 1) < cin);   // BK = 32 with cin = 16: the upper half of the chunk is zero
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) {
      unsigned voff = (idx_cur[u] >= 0 && a_in) ? (unsigned)idx_cur[u] * row_bytes + a_part16 : 0xFFFFFFFFu;
#ifdef IGEMM_EXP_SKIP_A
      if (st > 0) continue;
#endif
      ra[u] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff, a_soff, 0);
    }
    const unsigned w_soff = W_KMAJOR ? (unsigned)((kap * cin + c0) * cout) * 2u : (unsigned)(kap * cin * cout + c0) * 2u;
#ifdef IGEMM_EXP_SKIP_W
    if (st > 0) return;
#endif
#pragma unroll
    for (int u = 0; u < W_SEGS; ++u) {
      const unsigned vo = (BK == 64 || c0 + w_k[u] < cin) ? w_voff[u] : 0xFFFFFFFFu;
      rw[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, vo, w_soff, 0);
    }
  };
  auto store_lds = [&](int buf) {
    u16* Ab = smem + buf * STAGE_ELEMS;
    u16* Wb = Ab + A_ELEMS;
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) *(u32x4*)(Ab + (a_row0 + u * A_ROW_STEP) * LDA + (tid % (BK / 8)) * 8) = ra[u];
#pragma unroll
    for (int u = 0; u < W_SEGS; ++u) {
      int sgi = tid + u * NT;
      if (sgi >= W_TOTAL) continue;
      if (W_KMAJOR) { int k = sgi / (BN / 8), part = sgi % (BN / 8); *(u32x4*)(Wb + k * LDW + part * 8) = rw[u]; }
      else { int n = sgi / (BK / 8), part = sgi % (BK / 8); *(u32x4*)(Wb + n * LDW + part * 8) = rw[u]; }
    }
  };

  if constexpr (PF > 1) {
    // Deep-prefetch form for the narrow sparse levels (BK = 32: one MFMA k-step = ~130 clk of matrix work per stage).  In the loop
    // below a stage's gather is requested one stage before it is stored to LDS, so every stage waits out a full L2/HBM round trip
    // (measured: 1.5 us per stage with three workgroups per CU interleaved - the kernel moved ~1.1 TB/s).  Two things keep a
    // deeper pipeline from forming there: the register set, and the gather INDICES - they come from a vector load too, vmcnt
    // retires in order, so consuming an index waits for every row load issued before it.  Here the tile's whole neighbour table
    // (kvol x BM ints, coalesced: the table is offset-major) is staged in LDS once (LDS reads count on lgkmcnt, not vmcnt), and
    // the operand registers exist PF times: the loads of stage st + PF are issued when stage st + 1 has been written to LDS.
    int* sidx = (int*)(smem + 2 * STAGE_ELEMS);
    {
      constexpr int IB = 9;                               // table loads in flight per thread (a plain loop waits out one round trip per entry)
      const int total = kvol * BM;
      for (int i0 = 0; i0 < total; i0 += IB * NT) {
        int v[IB];
#pragma unroll
        for (int j = 0; j < IB; ++j) {
          const int i = min(i0 + j * NT + tid, total - 1);
          const int kap = i / BM, m = m0 + i % BM;
          const int mc = m < n_out ? m : n_out - 1;
          v[j] = nbr ? nbr[(long long)kap * ld + mc] : mc;
        }
#pragma unroll
        for (int j = 0; j < IB; ++j) {
          const int i = min(i0 + j * NT + tid, total - 1);
          sidx[i] = (m0 + i % BM < n_out) ? v[j] : -1;
        }
      }
    }
    __syncthreads();
    u32x4 pa[PF][A_SEGS], pw[PF][W_SEGS];
    auto issue_p = [&](int st, u32x4* qa, u32x4* qw) {
      const int kap = st % kvol, c0 = (st / kvol) * BK;
      const unsigned a_soff = (unsigned)c0 * 2u;
      const bool a_in = BK == 64 || (c0 + (int)(a_part16 >> 1) < cin);
#pragma unroll
      for (int u = 0; u < A_SEGS; ++u) {
        const int idx = sidx[kap * BM + a_row0 + u * A_ROW_STEP];
        unsigned voff = (idx >= 0 && a_in) ? (unsigned)idx * row_bytes + a_part16 : 0xFFFFFFFFu;
        qa[u] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff, a_soff, 0);
      }
      const unsigned w_soff = W_KMAJOR ? (unsigned)((kap * cin + c0) * cout) * 2u : (unsigned)(kap * cin * cout + c0) * 2u;
#pragma unroll
      for (int u = 0; u < W_SEGS; ++u) {
        const unsigned vo = (BK == 64 || c0 + w_k[u] < cin) ? w_voff[u] : 0xFFFFFFFFu;
        qw[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, vo, w_soff, 0);
      }
    };
    // LDS position of each weight segment; threads without one (small tiles) write their zeros to a scratch slot behind the
    // neighbour table: no branch in the stage body (a divergent branch there makes hipcc fall back to `vmcnt(0)`)
    int w_lds[W_SEGS];
#pragma unroll
    for (int u = 0; u < W_SEGS; ++u) {
      const int sgi = tid + u * NT;
      if (W_KMAJOR) w_lds[u] = A_ELEMS + (sgi / (BN / 8)) * LDW + (sgi % (BN / 8)) * 8;
      else w_lds[u] = A_ELEMS + (sgi / (BK / 8)) * LDW + (sgi % (BK / 8)) * 8;
      if (sgi >= W_TOTAL) w_lds[u] = -1;
    }
    u16* const scratch = (u16*)(sidx + kvol * BM) + tid * 8;
    auto store_p = [&](int buf, const u32x4* qa, const u32x4* qw) {
      u16* Ab = smem + buf * STAGE_ELEMS;
#pragma unroll
      for (int u = 0; u < A_SEGS; ++u) *(u32x4*)(Ab + (a_row0 + u * A_ROW_STEP) * LDA + (tid % (BK / 8)) * 8) = qa[u];
#pragma unroll
      for (int u = 0; u < W_SEGS; ++u) *(u32x4*)(w_lds[u] >= 0 ? Ab + w_lds[u] : scratch) = qw[u];
    };
    const int last = nstage - 1;
#pragma unroll
    for (int p = 0; p < PF; ++p) issue_p(p < last ? p : last, pa[p], pw[p]);
    store_p(0, pa[0], pw[0]);
    issue_p(PF < last ? PF : last, pa[0], pw[0]);
    __syncthreads();
    for (int st0 = 0; st0 < nstage; st0 += PF) {          // nstage % PF == 0 (dispatch)
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int st = st0 + p;
        {
          const int buf = st & 1;
          const u16* A = smem + buf * STAGE_ELEMS;
          const u16* W = A + A_ELEMS;
#pragma unroll
          for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[WM];
#pragma unroll
            for (int a = 0; a < WM; ++a) af[a] = direct_frag(A, LDA, (wm * WM + a) * 16, ks * 32, lane);
#pragma unroll
            for (int b = 0; b < WN; ++b) {
              bf16x8 bfr = W_KMAJOR ? tr_frag(W, LDW, ks * 32, (wn * WN + b) * 16, lane)
                                    : direct_frag(W, LDW, (wn * WN + b) * 16, ks * 32, lane);
#pragma unroll
              for (int a = 0; a < WM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr, acc[a][b], 0, 0, 0);
            }
          }
          const int q = (p + 1) % PF;                     // register set of stage st + 1 (and, once stored, of stage st + 1 + PF)
          store_p(buf ^ 1, pa[q], pw[q]);
          const int nx = st + 1 + PF;
          issue_p(nx < last ? nx : last, pa[q], pw[q]);
          __syncthreads();
        }
      }
    }
  } else {
  load_idx_next(0);
  advance_idx();
  load_idx_next(1);
  issue_loads(0);
  store_lds(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    // the last stage re-fetches itself into the idle buffer (never read): no "is there a next stage" branch in the body
    const int nx = st + 1 < nstage ? st + 1 : st;
    advance_idx();                                      // indices of stage st+1 (fetched a whole stage ago)
    load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
    issue_loads(nx);                                    // global loads in flight while this stage computes
    const u16* A = smem + buf * STAGE_ELEMS;
    const u16* W = A + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) af[a] = direct_frag(A, LDA, (wm * WM + a) * 16, ks * 32, lane);
#pragma unroll
      for (int b = 0; b < WN; ++b) {
        bf16x8 bfr = W_KMAJOR ? tr_frag(W, LDW, ks * 32, (wn * WN + b) * 16, lane)
                              : direct_frag(W, LDW, (wn * WN + b) * 16, ks * 32, lane);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr, acc[a][b], 0, 0, 0);
      }
      // the next stage's tile goes to the OTHER buffer (free since the previous barrier)
#ifndef IGEMM_EXP_SKIP_STORE
      if (ks == IGEMM_STORE_KS) store_lds(buf ^ 1);
#endif
    }
    __syncthreads();
  }
  }
  // epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + r
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int m = m0 + (wm * WM + a) * 16 + g * 4 + r;
      if (m >= n_out) continue;
#pragma unroll
      for (int b = 0; b < WN; ++b) {
        int col = col0 + (wn * WN + b) * 16 + li;
        if (col < cout) {
          float v = acc[a][b][r];
          if (bias) v += bias[col];
          if (relu) v = fmaxf(v, 0.f);
          out[(long long)m * cout + col] = f2bf(v);
        }
      }
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool WK, int BK = 64, int PF = 1>
static int launch_igemm_fwd(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev,
                            int n_out_cap, int cin, int cout, int kvol, hipStream_t s, const float* bias = nullptr, int relu = 0) {
  constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
  constexpr int LDA = BK + 8, LDW = WK ? BN + 16 : BK + 8;
  if (PF > 1 && (kvol * u3d_cdiv(cin, BK)) % PF != 0)      // the deep-prefetch loop is unrolled PF times without a tail
    return launch_igemm_fwd<WAVES_M, WAVES_N, WM, WN, WK, BK, 1>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, bias, relu);
  const size_t lds = 2 * (size_t)(BM * LDA + (WK ? BK * LDW : BN * LDW)) * 2 +
                     (PF > 1 ? (size_t)kvol * BM * 4 + (size_t)WAVES_M * WAVES_N * 64 * 16 : 0);   // + the tile's neighbour table + scratch
  auto kern = k_igemm_fwd<WAVES_M, WAVES_N, WM, WN, WK, BK, PF>;
  if (lds > 64 * 1024) U3D_ALLOW_LDS(kern, lds);      // one call site per template instantiation: per-kernel, per-device
  dim3 grid(u3d_cdiv(n_out_cap, BM), u3d_cdiv(cout, BN));
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, (const u16*)in, (const u16*)w, nbr, ld, (u16*)out, n_out_dev,
                     n_out_cap, cin, cout, kvol, bias, relu);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

// =============================================================================================
// forward / dgrad with LDS-DMA staging (256 x 256 tile, n-major weights w[kappa][n][k]).
//
// Ablation of k_igemm_fwd (tools/conv_bench.py): without the register->LDS stores of the next stage it runs at 1.2-1.4 PFLOP/s
// instead of 0.9 - the ds_write_b128 stream (13 clk each on the shared VGPR->LDS path, behind a vmcnt wait) is its largest
// overhead; and the k-major weight tile (transpose reads) costs another ~15 % against the n-major one.  Here both operand tiles go
// global -> LDS with `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write, zero fill for missing neighbours by an
// out-of-range offset), issued at the top of a stage into the idle buffer and in flight during the whole stage.  LDS-DMA writes
// lane-linearly (1 KiB per wave-instruction = 8 rows x 128 B), so the tiles are unpadded [256][64] and bank conflicts are avoided
// by an XOR swizzle applied on the SOURCE side: 16-byte part p of row r is stored in slot p ^ ((r >> 1) & 7); a fragment read
// (16 rows x 8 B per k-group) then covers all 64 banks exactly once.
// =============================================================================================
#ifndef IGEMM_GLDS
#define IGEMM_GLDS 1
#endif
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int WAVES_M, int WAVES_N, int WM, int WN, bool F32OUT = false>
__device__ __forceinline__ void igemm_glds_body(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr,
                                                int ld, u16* __restrict__ out, const int* __restrict__ n_out_dev, int n_out_cap,
                                                int cin, int cout, int kvol, const float* __restrict__ bias, int relu,
                                                double* __restrict__ stats, const BnEpi bn) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16, BK = 64;
  constexpr int A_ELEMS = BM * BK, W_ELEMS = BN * BK;   // unpadded tiles, 128 B per row
  constexpr int STAGE_ELEMS = A_ELEMS + W_ELEMS;
  constexpr int SEGS_A = BM / 8 / NW, SEGS_W = BN / 8 / NW;   // wave-instructions (8 rows x 128 B = 1 KiB) per wave per tile
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];

  const int n_out = min(*n_out_dev, n_out_cap);
  const int tile = u3d_xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);      // XCD-contiguous ranges of the LIVE tiles (capacity-sized grids)
  if (tile < 0) return;
  const int m0 = tile * BM;
  const int col0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int kchunks = cin / BK;
  const int nstage = kvol * kchunks;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 2u;
  // loader role: wave-instruction u of this wave fills rows (wv*4+u)*8 .. +7 of a tile; lane = (row in group, 16-byte slot)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_part16[SEGS_A], w_voff[SEGS_W];
  int arow[SEGS_A];
#pragma unroll
  for (int u = 0; u < SEGS_A; ++u) {
    const int r = (wv * SEGS_A + u) * 8 + lrow;
    arow[u] = r;
    a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
  }
#pragma unroll
  for (int u = 0; u < SEGS_W; ++u) {
    const int r = (wv * SEGS_W + u) * 8 + lrow;
    const int part = lslot ^ ((r >> 1) & 7);
    w_voff[u] = (col0 + r < cout) ? (unsigned)((col0 + r) * cin + part * 8) * 2u : 0xFFFFFFFFu;
  }
  int idx_cur[SEGS_A], idx_nxt[SEGS_A];
  auto load_idx_next = [&](int stage) {
    const int kap = stage % kvol;
#pragma unroll
    for (int u = 0; u < SEGS_A; ++u) {
      int m = m0 + arow[u];
      int mc = m < n_out ? m : n_out - 1;
      idx_nxt[u] = nbr ? nbr[(long long)kap * ld + mc] : mc;           // raw: masked when consumed (see k_igemm_fwd)
    }
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int u = 0; u < SEGS_A; ++u) idx_cur[u] = (m0 + arow[u] < n_out) ? idx_nxt[u] : -1;
  };
  auto issue = [&](int st, int buf) {
    const int kap = st % kvol, c0 = (st / kvol) * BK;
    u16* Ab = smem + buf * STAGE_ELEMS + wv * (SEGS_A * 512);
    u16* Wb = smem + buf * STAGE_ELEMS + A_ELEMS + wv * (SEGS_W * 512);
    const unsigned a_soff = (unsigned)c0 * 2u;
    const unsigned w_soff = (unsigned)(kap * cin * cout + c0) * 2u;
#pragma unroll
    for (int u = 0; u < SEGS_A; ++u) {
#if GLDS_EXP & 4
      unsigned voff = idx_cur[u] >= 0 ? (unsigned)(idx_cur[u] & 1023) * row_bytes + a_part16[u] : 0xFFFFFFFFu;      // timing experiment: 1024-row working set (cache hits)
#else
      unsigned voff = idx_cur[u] >= 0 ? (unsigned)idx_cur[u] * row_bytes + a_part16[u] : 0xFFFFFFFFu;
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(Ab + u * 512), 16, voff, a_soff, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SEGS_W; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)(Wb + u * 512), 16, w_voff[u], w_soff, 0, 0);
  };
#if GLDS_FRAG_B128
  // one LDS-DMA instruction of stage `st` (q < SEGS_A: activation rows, else weight rows): lets the stage loop feed the address unit
  // between MFMA groups instead of queueing all of a stage's loads in front of them
  auto issue_one = [&](int st, int buf, int q) {
    const int kap = st % kvol, c0 = (st / kvol) * BK;
    if (q < SEGS_A) {
      u16* Ab = smem + buf * STAGE_ELEMS + wv * (SEGS_A * 512);
      unsigned voff = idx_cur[q] >= 0 ? (unsigned)idx_cur[q] * row_bytes + a_part16[q] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(Ab + q * 512), 16, voff, (unsigned)c0 * 2u, 0, 0);
    } else {
      const int u = q - SEGS_A;
      u16* Wb = smem + buf * STAGE_ELEMS + A_ELEMS + wv * (SEGS_W * 512);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)(Wb + u * 512), 16, w_voff[u], (unsigned)(kap * cin * cout + c0) * 2u, 0, 0);
    }
  };
  // fragment = ONE 16-byte LDS read: lane (g, row li) takes the 8 consecutive reduction elements of part ks*4+g of its row.  The MFMA
  // only needs A and B to agree on which reduction element sits in which (lane group, position) - both go through this function -
  // so the nominal k-order of the instruction does not matter.  Conflict-free under the source-side swizzle: in a 16-lane group
  // (one g, rows 0..15) the slots (ks*4+g) ^ ((row>>1)&7) take all 8 values per row parity = every bank once.
  const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = ((ks * 4 + g) ^ fsw) << 3;
  typedef const volatile s16x8 __attribute__((address_space(3))) * lds_vptr;
  auto frag = [&](const u16* rowp, int ks) {
    s16x8 v = *(lds_vptr)(rowp + foff[ks]);
    return __builtin_bit_cast(bf16x8, v);
  };
#else
  // fragment offsets inside a 64-element row: slot of part (ks*4 + g/2 [+2]) under this lane's row swizzle, plus the 8-byte half
  const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) foff[ks][j] = (((ks * 4 + (g >> 1) + 2 * j) ^ fsw) << 3) + (g & 1) * 4;
  typedef const volatile s16x4 __attribute__((address_space(3))) * lds_vptr;
  auto frag = [&](const u16* rowp, int ks) {
    s16x4 x = *(lds_vptr)(rowp + foff[ks][0]);
    s16x4 y = *(lds_vptr)(rowp + foff[ks][1]);
    s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

#endif
#if GLDS_PIPE
  // Fragment pipeline.  A stage (64 reduction elements) is four UNITS: (k-step 0|1) x (lower|upper half of this wave's row blocks);
  // a unit = WM/2 x WN MFMAs on fragments already in registers, issued right after the LDS reads of the NEXT unit's fragments, so
  // every fragment read has a whole unit of MFMAs (>= 256 clk) to land.  Registers: two half-sets of A fragments and two sets of B
  // fragments - as many as the unpipelined loop held (all A fragments of a k-step + one B fragment).  The stage barrier sits
  // before the LAST unit: by then every fragment of the stage is in registers, so the last unit's MFMAs cover the first reads of
  // the next stage (other buffer) and the LDS-DMA of stage st+2 starts into the buffer just released.
  static_assert(WM % 2 == 0, "row blocks are processed in two halves");
  constexpr int HM = WM / 2;
  bf16x8 af[2][HM], bw[2][WN];
  // (macros, not lambdas: the slot / half indices must be literal for the accumulators to stay in place in registers)
#define GLDS_LOAD_A(SLOT, BUF, KS, HALF)                                                                             \
  {                                                                                                                  \
    const u16* A_ = smem + (BUF) * STAGE_ELEMS + ((wm * WM + (HALF) * HM) * 16 + li) * BK;                           \
    _Pragma("unroll") for (int a = 0; a < HM; ++a) af[SLOT][a] = frag(A_ + a * 16 * BK, KS);                         \
  }
#define GLDS_LOAD_B(SLOT, BUF, KS)                                                                                   \
  {                                                                                                                  \
    const u16* W_ = smem + (BUF) * STAGE_ELEMS + A_ELEMS + (wn * WN * 16 + li) * BK;                                 \
    _Pragma("unroll") for (int b = 0; b < WN; ++b) bw[SLOT][b] = frag(W_ + b * 16 * BK, KS);                         \
  }
#define GLDS_MMA(AS, BS, HALF)                                                                                       \
  {                                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < WN; ++b) {                                                                 \
      _Pragma("unroll") for (int a = 0; a < HM; ++a)                                                                 \
        acc[(HALF) * HM + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[BS][b], af[AS][a], acc[(HALF) * HM + a][b], 0, 0, 0); \
    }                                                                                                                \
  }
  load_idx_next(0);
  advance_idx();
  load_idx_next(1 < nstage ? 1 : 0);
  issue(0, 0);
  advance_idx();
  load_idx_next(2 < nstage ? 2 : nstage - 1);
  issue(1 < nstage ? 1 : 0, 1);
  __syncthreads();
  GLDS_LOAD_B(0, 0, 0)
  GLDS_LOAD_A(0, 0, 0, 0)
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    GLDS_LOAD_A(1, buf, 0, 1)                           // unit (k0, lower): fetch (k0, upper) and k-step 1's B fragments
    GLDS_LOAD_B(1, buf, 1)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_MMA(0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_LOAD_A(0, buf, 1, 0)                           // unit (k0, upper): fetch (k1, lower)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_MMA(1, 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_LOAD_A(1, buf, 1, 1)                           // unit (k1, lower): fetch (k1, upper)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_MMA(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                    // every wave holds all of stage st; stage st+1 has landed in the other buffer
    advance_idx();
    load_idx_next(st + 3 < nstage ? st + 3 : nstage - 1);
    issue(st + 2 < nstage ? st + 2 : nstage - 1, buf);  // beyond the end: a harmless re-fetch into a buffer nobody reads again
    GLDS_LOAD_B(0, buf ^ 1, 0)                          // unit (k1, upper): fetch the next stage's first unit
    GLDS_LOAD_A(0, buf ^ 1, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    GLDS_MMA(1, 1, 1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef GLDS_LOAD_A
#undef GLDS_LOAD_B
#undef GLDS_MMA
#else
  load_idx_next(0);
  advance_idx();
  load_idx_next(1 < nstage ? 1 : 0);
  issue(0, 0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    const int nx = st + 1 < nstage ? st + 1 : st;       // the last stage re-fetches itself into the idle buffer: branch-free body
    advance_idx();
    load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
#if !(GLDS_EXP & 1) && !GLDS_DMA_SPREAD
    issue(nx, buf ^ 1);                                 // in flight during the whole stage; buffer free since the last barrier
#endif
#if GLDS_EXP & 2
    const u16* A = smem + (wm * WM * 16 + li) * BK;     // timing experiment: fragments of stage 0 re-read by every stage -> loop-invariant, hoisted
    const u16* W = smem + A_ELEMS + (wn * WN * 16 + li) * BK;
    typedef const s16x8 __attribute__((address_space(3))) * lds_cptr;
#define GLDS_FRAG(P, KS) __builtin_bit_cast(bf16x8, *(lds_cptr)((P) + foff[KS]))
#else
    const u16* A = smem + buf * STAGE_ELEMS + (wm * WM * 16 + li) * BK;
    const u16* W = smem + buf * STAGE_ELEMS + A_ELEMS + (wn * WN * 16 + li) * BK;
#define GLDS_FRAG(P, KS) frag(P, KS)
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) af[a] = GLDS_FRAG(A + a * 16 * BK, ks);
#pragma unroll
      for (int b = 0; b < WN; ++b) {
        bf16x8 bfr = GLDS_FRAG(W + b * 16 * BK, ks);
#pragma unroll
        // operands swapped: the MFMA produces the TRANSPOSED 16x16 block, i.e. this lane ends up with 4 consecutive output
        // COLUMNS (4g..4g+3) of row li - one 8-byte store per block in the epilogue instead of four 2-byte ones
        for (int a = 0; a < WM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, af[a], acc[a][b], 0, 0, 0);
#if GLDS_DMA_SPREAD && !(GLDS_EXP & 1)
        {   // the next stage's LDS-DMA instructions, dealt out behind the MFMA groups of the first part of this stage
          constexpr int NQ = SEGS_A + SEGS_W, NG = (GLDS_DMA_SPREAD == 1) ? 2 * WN : (GLDS_DMA_SPREAD == 3 ? (WN > 1 ? WN / 2 : 1) : (GLDS_DMA_SPREAD == 4 ? WN + WN / 2 : WN));      // groups that carry loads
          constexpr int PER = (NQ + NG - 1) / NG;
          const int grp = ks * WN + b;
          if (grp < NG) {
#pragma unroll
            for (int j = 0; j < PER; ++j)
              if (grp * PER + j < NQ) issue_one(nx, buf ^ 1, grp * PER + j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
    }
    __syncthreads();                                    // also drains this wave's LDS-DMA (vmcnt) before anyone reads the next buffer
  }
#endif
#define GLDS_EPI_ADDEND (BM != 256 || BN != 256)
#include "glds_epilogue.inc"
#undef GLDS_EPI_ADDEND
}

// =============================================================================================
// 256 x 256 tile, EIGHT-PHASE schedule (two k-tiles of 64 = one trip through both LDS buffers; 4 phases per k-tile).
//
// igemm_glds_body walks all eight waves through a k-tile together: 24 fragment reads, 8 LDS-DMA instructions and 64 MFMAs per wave
// between two __syncthreads() that drain vmcnt - the two waves of a SIMD are always in the same part of the stage, and the LDS-DMA
// instructions (the 35 % its ablation prices) sit among the MFMAs of the wave that issues them.  Here (the schedule the CDNA4
// guide's 256^2 template describes) a k-tile is four PHASES of 16 MFMAs (one 64 x 32 quadrant of the wave's 128 x 64 block, both
// k-steps), each phase = a LOAD segment (the phase's fragment reads + 2 LDS-DMA instructions of the next k-tile + a COUNTED vmcnt
// wait) | s_barrier | an MFMA segment (16 MFMAs at raised priority) | s_barrier, and the two wave rows (wm = 0 / 1: one wave of
// each per SIMD) run ONE barrier apart: while one wave of a SIMD is in its MFMA segment its partner is in its load segment.
//
// LDS image of a k-tile = four 16 KiB pieces, each exactly what one load segment reads: A0 / A1 = the first / second 64 rows of
// BOTH wave rows, B0 / B1 = the first / second 32 columns of all four wave columns.  Phase p reads: 1: A0 + B0, 2: B1, 3: A1, 4: -
// (B0 stays in registers for the last quadrant) and requests for the NEXT k-tile: 1: A0, 2: B0, 3: B1, 4: A1 (+ the gather
// indices two k-tiles ahead, requested first thing in phase 1 and consumed at the end of phase 4).  vmcnt is in order, so
// "everything up to piece X has landed" is one count: at the end of the load segments 1 and 2 exactly 8 younger requests are
// allowed in flight (2 pieces x 2 + 4 index loads), at the end of segment 4 four (B1, A1) - which retires B1 / A1 /
// A0 + B0 one barrier before their first reader - two barriers for the wave row that runs behind.  Nothing waits for vmcnt(0)
// inside the loop.  Rows / swizzle / fragment layout / epilogue as igemm_glds_body.
// =============================================================================================
#ifndef IGEMM_GLDS8
#define IGEMM_GLDS8 1
#endif
#ifndef GLDS8_PRIO
#define GLDS8_PRIO 1
#endif
#ifndef GLDS8_BALANCE
#define GLDS8_BALANCE 1     /* the next k-tile's first A half is read in phase 4 (second register set): 4 / 4 / 8 / 8 fragment reads per segment */
#endif
#ifndef GLDS8_STAGGER
#define GLDS8_STAGGER 1     /* 0 (experiment): both wave rows in the same segment at the same time */
#endif
// RB = 16-row blocks per wave and row half: 4 = the 256-row tile; 3 = a 192-row tile in the SAME LDS image and schedule (the last
// 16 rows of every 64-row piece quarter are dead: never requested - their LDS-DMA lanes carry the "no row" offset -, never read,
// never multiplied).  For row counts where 256-row tiles leave a quarter of the CUs without a workgroup (48 000 rows x 256 columns:
// 188 tiles for 256 CUs; 192-row tiles: 250) - see igemm_rows192().
template <bool F32OUT = false, int RB = 4>
__device__ __forceinline__ void igemm_glds8_body(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr,
                                                 int ld, u16* __restrict__ out, const int* __restrict__ n_out_dev, int n_out_cap,
                                                 int cin, int cout, int kvol, const float* __restrict__ bias, int relu,
                                                 double* __restrict__ stats, const BnEpi bn) {
  constexpr int WAVES_M = 2, WAVES_N = 4, WM = 2 * RB, WN = 4, NW = 8;
  constexpr int BM = 64 * RB, BN = 256, BK = 64;
  constexpr int PIECE = 128 * BK;                         // elements of one piece (16 KiB)
  constexpr int STAGE_ELEMS = 4 * PIECE;                  // A0 | A1 | B0 | B1
  extern __shared__ __attribute__((aligned(16))) u16 smem[];

  const int n_out = min(*n_out_dev, n_out_cap);
  const int tile = u3d_xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);      // XCD-contiguous ranges of the LIVE tiles (capacity-sized grids)
  if (tile < 0) return;
  const int m0 = tile * BM;
  const int col0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int nstage = kvol * (cin / BK);

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 2u;
  // loader role: LDS-DMA instruction u (0 / 1) of this wave fills piece rows (wv*2+u)*8 .. +7; lane = (row in group, 16-byte slot)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_part16[2], w_voff[2][2];                     // [u], [piece t][u]
  int arow[2][2];                                         // tile row of (piece s, u); -1: a dead row of the 192-row tile
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (wv * 2 + u) * 8 + lrow;                // piece row 0..127
    a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      arow[sp][u] = ((r & 63) < 16 * RB) ? (r >> 6) * (32 * RB) + sp * (16 * RB) + (r & 63) : -1;
      const int tc = (r >> 5) * 64 + sp * 32 + (r & 31);  // piece B_sp row r = tile column tc
      w_voff[sp][u] = (col0 + tc < cout) ? (unsigned)((col0 + tc) * cin + (lslot ^ ((r >> 1) & 7)) * 8) * 2u : 0xFFFFFFFFu;
    }
  }
  int idx_cur[2][2], idx_nxt[2][2];
  // (needs a neighbour table: the plain GEMM callers, nbr == nullptr, stay on igemm_glds_body)
  int mcl[2][2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + (arow[sp][u] < 0 ? 0 : arow[sp][u]);
      mcl[sp][u] = m < n_out ? m : n_out - 1;
      idx_nxt[sp][u] = mcl[sp][u];
    }
  // the loads must land in the loop-carried registers themselves: a copy at the loop's back edge needs the value, i.e. a
  // vmcnt(0) per k-tile (what hipcc emitted with a select or a branch "nbr ? load : m" between the load and the variable)
  // Requested at the top of phase 1 and consumed at the end of phase 4 of the SAME loop trip: a load pending across the back
  // edge made hipcc copy the register there, i.e. wait vmcnt(0) once per k-tile.  Between the request and advance_idx() lie
  // exactly the 8 LDS-DMA requests of the trip: the wait hipcc inserts for the indices is the vmcnt(8) the schedule wants.
  auto load_idx_next = [&](int stage) {
    const int* row = nbr + (long long)(stage % kvol) * ld;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_nxt[sp][u] = row[mcl[sp][u]];
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_cur[sp][u] = (arow[sp][u] >= 0 && m0 + arow[sp][u] < n_out) ? idx_nxt[sp][u] : -1;
  };
  auto issue_a = [&](int st, int buf, int sp) {           // piece A_sp of k-tile st
    const unsigned soff = (unsigned)((st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + sp * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned voff = idx_cur[sp][u] >= 0 ? (unsigned)idx_cur[sp][u] * row_bytes + a_part16[u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(dst + u * 512), 16, voff, soff, 0, 0);
    }
  };
  auto issue_b = [&](int st, int buf, int sp) {           // piece B_sp of k-tile st
    const unsigned soff = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)(dst + u * 512), 16, w_voff[sp][u], soff, 0, 0);
  };
  const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = ((ks * 4 + g) ^ fsw) << 3;
  typedef const volatile s16x8 __attribute__((address_space(3))) * lds_vptr;
  auto frag = [&](const u16* rowp, int ks) {
    s16x8 v = *(lds_vptr)(rowp + foff[ks]);
    return __builtin_bit_cast(bf16x8, v);
  };
#if GLDS8_BALANCE
  bf16x8 af2[2][RB][2], bf[2][2][2];                      // A: [half][row block][k-step]; B: [half][col block][k-step]
#define G8_AF(SP) af2[SP]
#else
  bf16x8 af1[RB][2], bf[2][2][2];                         // A: [row block][k-step] of the current half; B: [half][col block][k-step]
#define G8_AF(SP) af1
#endif
#define G8_READ_A(BUF, SP)                                                                                   \
  {                                                                                                          \
    const u16* A_ = smem + (BUF) * STAGE_ELEMS + (SP) * PIECE + (wm * 64 + li) * BK;                         \
    _Pragma("unroll") for (int a = 0; a < RB; ++a) {                                                         \
      G8_AF(SP)[a][0] = frag(A_ + a * 16 * BK, 0);                                                           \
      G8_AF(SP)[a][1] = frag(A_ + a * 16 * BK, 1);                                                           \
    }                                                                                                        \
  }
#define G8_READ_B(BUF, SP)                                                                                   \
  {                                                                                                          \
    const u16* B_ = smem + (BUF) * STAGE_ELEMS + (2 + (SP)) * PIECE + (wn * 32 + li) * BK;                   \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                          \
      bf[SP][b][0] = frag(B_ + b * 16 * BK, 0);                                                              \
      bf[SP][b][1] = frag(B_ + b * 16 * BK, 1);                                                              \
    }                                                                                                        \
  }
#define G8_MMA(SA, SB)                                                                                       \
  {                                                                                                          \
    __builtin_amdgcn_s_setprio(GLDS8_PRIO);                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                          \
        _Pragma("unroll") for (int a = 0; a < RB; ++a)                                                       \
          acc[(SA) * RB + a][(SB) * 2 + b] =                                                                 \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[SB][b][ks], G8_AF(SA)[a][ks], acc[(SA) * RB + a][(SB) * 2 + b], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  }
#define G8_BAR()                                  \
  {                                               \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_barrier();                 \
    __builtin_amdgcn_sched_barrier(0);            \
  }
#define G8_VMCNT8() __builtin_amdgcn_s_waitcnt(0x0F78)    /* vmcnt(8): expcnt / lgkmcnt untouched */

  // prologue: indices of k-tiles 0, 1, 2; all four pieces of k-tile 0; everything landed before the first read
  load_idx_next(0);
  advance_idx();
  issue_a(0, 0, 0); issue_b(0, 0, 0); issue_b(0, 0, 1); issue_a(0, 0, 1);
  load_idx_next(1 < nstage ? 1 : 0);
  advance_idx();
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
  G8_BAR();
#if GLDS8_BALANCE
  G8_READ_A(0, 0)                                         // phase 4 reads the NEXT k-tile's first row half: the first one here
#endif
  if (GLDS8_STAGGER && wm == 1) G8_BAR();                 // the second wave row runs one barrier behind the first
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    const int nx = st + 1 < nstage ? st + 1 : st;         // past the end: a harmless re-fetch into the buffer nobody reads again
    // ---- phase 1: quadrant (rows 0-63, cols 0-31)
    load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
    __builtin_amdgcn_sched_barrier(0);
    G8_READ_B(buf, 0)
#if !GLDS8_BALANCE
    __builtin_amdgcn_sched_barrier(0);
    G8_READ_A(buf, 0)
#endif
    issue_a(nx, buf ^ 1, 0);
    G8_VMCNT8();                                          // B1 of this k-tile has landed (every wave's share after the barrier)
    G8_BAR();
    G8_MMA(0, 0)
    G8_BAR();
    // ---- phase 2: (rows 0-63, cols 32-63)
    G8_READ_B(buf, 1)
    issue_b(nx, buf ^ 1, 0);
    G8_VMCNT8();                                          // A1 of this k-tile
    G8_BAR();
    G8_MMA(0, 1)
    G8_BAR();
    // ---- phase 3: (rows 64-127, cols 32-63)
    G8_READ_A(buf, 1)
    issue_b(nx, buf ^ 1, 1);
#if GLDS8_BALANCE
    __builtin_amdgcn_s_waitcnt(0x0F74);                   // vmcnt(4): A0 of the next k-tile (read in phase 4)
#endif
    G8_BAR();
    G8_MMA(1, 1)
    G8_BAR();
    // ---- phase 4: (rows 64-127, cols 0-31): B0 is still in registers
#if GLDS8_BALANCE
    G8_READ_A(buf ^ 1, 0)                                 // 12 / 4 / 8 / 0 reads per segment -> 4 / 4 / 8 / 8
#endif
    issue_a(nx, buf ^ 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    advance_idx();                                        // (hipcc waits vmcnt(8) here: the indices requested in phase 1)
    __builtin_amdgcn_s_waitcnt(0x0F74);                   // vmcnt(4): A0 and B0 of the next k-tile
    G8_BAR();
    G8_MMA(1, 0)
    G8_BAR();
  }
  if (GLDS8_STAGGER && wm == 0) G8_BAR();                 // same number of barriers for both wave rows
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // the tail's re-fetch requests: landed before the epilogue reuses the buffers
  __syncthreads();
#undef G8_AF
#undef G8_READ_A
#undef G8_READ_B
#undef G8_MMA
#undef G8_BAR
#undef G8_VMCNT8
#define GLDS_EPI_ADDEND 1
#include "glds_epilogue.inc"
#undef GLDS_EPI_ADDEND
}
__global__ __launch_bounds__(512) void k_igemm_glds8_256x256(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                             const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                             const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8_body(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_256x256_f32o(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                                  const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                                  const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8_body<true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_192x256(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                             const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                             const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8_body<false, 3>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_192x256_f32o(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                                  const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                                  const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8_body<true, 3>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
// 192-row tiles instead of 256-row ones when they finish sooner on 256 CUs with one workgroup each (time ~ rounds x tile rows): the
// mid-size layers of the dense stack (48 000 rows x 256 columns: 188 -> 250 workgroups, 12 000 rows x 512 columns in 128-column
// tiles: 188 -> 252).  Convolutions with a neighbour table and more than one offset only (the plain-GEMM callers stay on 256 rows).
#ifndef IGEMM_ROWS192
#define IGEMM_ROWS192 1
#endif
static inline bool igemm_rows192(const int32_t* nbr, int n_out_cap, int col_blocks, int kvol) {
  if (!IGEMM_ROWS192 || !nbr || kvol <= 1) return false;
  const long long w256 = (long long)u3d_cdiv(n_out_cap, 256) * col_blocks, w192 = (long long)u3d_cdiv(n_out_cap, 192) * col_blocks;
  return (double)u3d_cdiv(w192, 256) * 192.0 * 1.05 < (double)u3d_cdiv(w256, 256) * 256.0;
}

// =============================================================================================
// 256 x 128 tile on the same idea, for LONG reductions with 128-column output tiles (the 12 000-row 512-channel layers: 72 k-tiles per
// tile; they ran on 128 x 128 tiles, two independent workgroups per CU whose relative phase nobody controls).  Eight waves = two
// GROUPS of four (waves w and w + 4 share a SIMD): group g owns rows g*128 .. +127 as 2 x 2 waves of 64 x 64, both groups share the
// weight tile (1.5 x the LDS-DMA bytes per MFMA of the 256 x 256 tile instead of 2 x) and run one barrier apart.  A k-tile is only
// TWO phases of 16 MFMAs here (column halves of the wave's block), so two stage buffers would leave a request one phase to land:
// THREE buffers (3 x 48 KiB), k-tile st + 2 requested during k-tile st.  Pieces: A0 / A1 = the rows of group 0 / 1 (16 KiB),
// B0 / B1 = the first / second 32 columns of both wave columns (8 KiB).  Per trip: segment 1 reads B0 + B1 and requests A(st+2)
// (+ the gather indices of st + 3), segment 2 reads the NEXT k-tile's A (second register set) and requests B(st+2); one counted
// wait per segment, vmcnt(10) both times (what is younger than the piece that must have landed: 2 + 4 + 4 and 4 + 4 + 2 requests).
// Not for short reductions: with 18 k-tiles per tile a quarter of a workgroup's life is prologue + epilogue, which a second
// resident workgroup hides and this 144 KiB one cannot (measured on par there).
// =============================================================================================
#ifndef IGEMM_GLDS8N
#define IGEMM_GLDS8N 1
#endif
template <bool F32OUT = false, int RB = 4>      // RB = 3: 192-row tiles (48 live rows per wave), see igemm_glds8_body
__device__ __forceinline__ void igemm_glds8n_body(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr,
                                                  int ld, u16* __restrict__ out, const int* __restrict__ n_out_dev, int n_out_cap,
                                                  int cin, int cout, int kvol, const float* __restrict__ bias, int relu,
                                                  double* __restrict__ stats, const BnEpi bn) {
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = RB, WN = 4, NW = 8;
  constexpr int BM = 64 * RB, BN = 128, BK = 64;
  constexpr int APIECE = 128 * BK, BPIECE = 64 * BK;
  constexpr int STAGE_ELEMS = 2 * APIECE + 2 * BPIECE;    // A0 | A1 | B0 | B1 = 48 KiB
  extern __shared__ __attribute__((aligned(16))) u16 smem[];

  const int n_out = min(*n_out_dev, n_out_cap);
  const int tile = u3d_xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);      // XCD-contiguous ranges of the LIVE tiles (capacity-sized grids)
  if (tile < 0) return;
  const int m0 = tile * BM;
  const int col0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wv >> 2, wm = wv >> 1, wn = wv & 1;     // wm = grp*2 + (row half inside the group): the epilogue's row-block index
  const int nstage = kvol * (cin / BK);

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 2u;
  const int lrow = lane >> 3, lslot = lane & 7;
  // A piece g: 16 instructions, this wave issues u = 0 / 1: piece rows (wv*2+u)*8 + lrow = tile rows g*128 + ...
  // B piece t:  8 instructions, this wave issues one: piece row wv*8 + lrow = tile column (r>>5)*64 + t*32 + (r&31)
  unsigned a_part16[2], w_voff[2];
  int arow[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (wv * 2 + u) * 8 + lrow;
    a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) arow[gp][u] = ((r & 63) < 16 * RB) ? gp * (32 * RB) + (r >> 6) * (16 * RB) + (r & 63) : -1;
  }
  {
    const int r = wv * 8 + lrow;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tc = (r >> 5) * 64 + t * 32 + (r & 31);
      w_voff[t] = (col0 + tc < cout) ? (unsigned)((col0 + tc) * cin + (lslot ^ ((r >> 1) & 7)) * 8) * 2u : 0xFFFFFFFFu;
    }
  }
  int idx_cur[2][2], idx_nxt[2][2], mcl[2][2];
#pragma unroll
  for (int gp = 0; gp < 2; ++gp)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + (arow[gp][u] < 0 ? 0 : arow[gp][u]);
      mcl[gp][u] = m < n_out ? m : n_out - 1;
      idx_nxt[gp][u] = mcl[gp][u];
    }
  auto load_idx_next = [&](int stage) {                   // requested and consumed inside one loop trip (see igemm_glds8_body)
    const int* row = nbr + (long long)(stage % kvol) * ld;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_nxt[gp][u] = row[mcl[gp][u]];
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_cur[gp][u] = (arow[gp][u] >= 0 && m0 + arow[gp][u] < n_out) ? idx_nxt[gp][u] : -1;
  };
  auto issue_a = [&](int st, int sl) {                    // both A pieces of k-tile st into stage slot sl (rows idx_cur describes)
    const unsigned soff = (unsigned)((st / kvol) * BK) * 2u;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      u16* dst = smem + sl * STAGE_ELEMS + gp * APIECE + wv * 1024;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned voff = idx_cur[gp][u] >= 0 ? (unsigned)idx_cur[gp][u] * row_bytes + a_part16[u] : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(dst + u * 512), 16, voff, soff, 0, 0);
      }
    }
  };
  auto issue_b = [&](int st, int sl) {                    // both B pieces of k-tile st
    const unsigned soff = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u16* dst = smem + sl * STAGE_ELEMS + 2 * APIECE + t * BPIECE + wv * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)dst, 16, w_voff[t], soff, 0, 0);
    }
  };
  const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = ((ks * 4 + g) ^ fsw) << 3;
  typedef const volatile s16x8 __attribute__((address_space(3))) * lds_vptr;
  auto frag = [&](const u16* rowp, int ks) {
    s16x8 v = *(lds_vptr)(rowp + foff[ks]);
    return __builtin_bit_cast(bf16x8, v);
  };
  bf16x8 af[2][RB][2], bf[2][2][2];                       // A: [register set][row block][k-step]; B: [column half][col block][k-step]
#define N8_READ_A(SL, SET)                                                                                   \
  {                                                                                                          \
    const u16* A_ = smem + (SL) * STAGE_ELEMS + grp * APIECE + (((wv >> 1) & 1) * 64 + li) * BK;             \
    _Pragma("unroll") for (int a = 0; a < RB; ++a) {                                                         \
      af[SET][a][0] = frag(A_ + a * 16 * BK, 0);                                                             \
      af[SET][a][1] = frag(A_ + a * 16 * BK, 1);                                                             \
    }                                                                                                        \
  }
#define N8_READ_B(SL)                                                                                        \
  {                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                          \
      const u16* B_ = smem + (SL) * STAGE_ELEMS + 2 * APIECE + t * BPIECE + (wn * 32 + li) * BK;             \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
        bf[t][b][0] = frag(B_ + b * 16 * BK, 0);                                                             \
        bf[t][b][1] = frag(B_ + b * 16 * BK, 1);                                                             \
      }                                                                                                      \
    }                                                                                                        \
  }
#define N8_MMA(SET, T)                                                                                       \
  {                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                          \
        _Pragma("unroll") for (int a = 0; a < RB; ++a)                                                       \
          acc[a][(T) * 2 + b] =                                                                              \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[T][b][ks], af[SET][a][ks], acc[a][(T) * 2 + b], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  }
#define N8_BAR()                                  \
  {                                               \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_barrier();                 \
    __builtin_amdgcn_sched_barrier(0);            \
  }
  // one k-tile: SET = the A register set holding k-tile st (the other one receives k-tile st + 1)
#define N8_TRIP(SET)                                                                                         \
  {                                                                                                          \
    const int s2 = st + 2 < nstage ? st + 2 : nstage - 1;                                                    \
    load_idx_next(st + 3 < nstage ? st + 3 : nstage - 1);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    N8_READ_B(sl0)                                                                                           \
    issue_a(s2, sl2);                                                                                        \
    __builtin_amdgcn_s_waitcnt(0x0F7A);                   /* vmcnt(10): A of k-tile st + 1 */                \
    N8_BAR();                                                                                                \
    N8_MMA(SET, 0)                                                                                           \
    N8_BAR();                                                                                                \
    N8_READ_A(sl1, (SET) ^ 1)                                                                                \
    issue_b(s2, sl2);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    advance_idx();                                                                                           \
    __builtin_amdgcn_s_waitcnt(0x0F7A);                   /* vmcnt(10): B of k-tile st + 1 */                \
    N8_BAR();                                                                                                \
    N8_MMA(SET, 1)                                                                                           \
    N8_BAR();                                                                                                \
    { const int t_ = sl0; sl0 = sl1; sl1 = sl2; sl2 = t_; }                                                  \
    ++st;                                                                                                    \
  }
  // prologue: k-tiles 0 and 1 requested (slots 0, 1), indices of k-tile 2 current, A of k-tile 0 in register set 0
  load_idx_next(0);
  advance_idx();
  issue_a(0, 0); issue_b(0, 0);
  load_idx_next(1 < nstage ? 1 : 0);
  advance_idx();
  issue_a(1 < nstage ? 1 : 0, 1); issue_b(1 < nstage ? 1 : 0, 1);
  load_idx_next(2 < nstage ? 2 : nstage - 1);
  advance_idx();
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
  N8_BAR();
  N8_READ_A(0, 0)
  if (grp == 1) N8_BAR();                                 // the second group runs one barrier behind the first
  int st = 0, sl0 = 0, sl1 = 1, sl2 = 2;
  while (st + 1 < nstage) {
    N8_TRIP(0)
    N8_TRIP(1)
  }
  if (st < nstage) N8_TRIP(0)
  if (grp == 0) N8_BAR();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
#undef N8_READ_A
#undef N8_READ_B
#undef N8_MMA
#undef N8_BAR
#undef N8_TRIP
#define GLDS_EPI_ADDEND 1
#include "glds_epilogue.inc"
#undef GLDS_EPI_ADDEND
}
__global__ __launch_bounds__(512) void k_igemm_glds8_256x128(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                             const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                             const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8n_body(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_256x128_f32o(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                                  const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                                  const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8n_body<true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_192x128(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                             const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                             const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8n_body<false, 3>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
__global__ __launch_bounds__(512) void k_igemm_glds8_192x128_f32o(const u16* in, const u16* w, const int* nbr, int ld, u16* out,
                                                                  const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                                  const float* bias, int relu, double* stats, BnEpi bn) {
  igemm_glds8n_body<true, 3>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);
}
// the 256 x 128 eight-phase kernel for the shapes it is dispatched on: long reductions (>= GLDS8N_MIN_KTILES k-tiles), enough
// workgroups to keep most CUs busy with ONE per CU
#ifndef GLDS8N_MIN_KTILES
#define GLDS8N_MIN_KTILES 48
#endif
static bool igemm_glds8n_shape(const int32_t* nbr, int n_out_cap, int cin, int cout, int kvol) {
  return IGEMM_GLDS8N && nbr && cin % 64 == 0 && cout % 128 == 0 && kvol * (cin / 64) >= GLDS8N_MIN_KTILES
         && (long long)u3d_cdiv(n_out_cap, 256) * (cout / 128) >= 160;
}
static int launch_igemm_glds8n(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev,
                               int n_out_cap, int cin, int cout, int kvol, hipStream_t s, const float* bias = nullptr, int relu = 0,
                               double* stats = nullptr, const BnEpi bn = BnEpi{}, bool f32o = false) {
  constexpr size_t lds = 3 * (size_t)(256 + 128) * 64 * 2;      // 144 KiB
  U3D_ALLOW_LDS(k_igemm_glds8_256x128, lds);
  U3D_ALLOW_LDS(k_igemm_glds8_256x128_f32o, lds);
  U3D_ALLOW_LDS(k_igemm_glds8_192x128, lds);
  U3D_ALLOW_LDS(k_igemm_glds8_192x128_f32o, lds);
  const bool r192 = igemm_rows192(nbr, n_out_cap, cout / 128, kvol);
  dim3 grid(u3d_cdiv(n_out_cap, r192 ? 192 : 256), cout / 128);
  hipLaunchKernelGGL(r192 ? (f32o ? k_igemm_glds8_192x128_f32o : k_igemm_glds8_192x128) : (f32o ? k_igemm_glds8_256x128_f32o : k_igemm_glds8_256x128),
                     grid, dim3(512), lds, s, (const u16*)in, (const u16*)w, nbr, ld, (u16*)out, n_out_dev, n_out_cap,
                     cin, cout, kvol, bias, relu, stats, bn);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

// concrete kernels (a __global__ TEMPLATE with this body lost its host stub under hipcc 7.2: undefined symbol at load time)
#define U3D_GLDS_KERNEL(NAME, A, B, C, D)                                                                                        \
  __global__ __launch_bounds__(A* B * 64) void NAME(const u16* in, const u16* w, const int* nbr, int ld, u16* out,               \
                                                    const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,            \
                                                    const float* bias, int relu, double* stats, BnEpi bn) {                      \
    igemm_glds_body<A, B, C, D>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);               \
  }
#define U3D_GLDS_KERNEL_X(NAME, ...) U3D_GLDS_KERNEL(NAME, __VA_ARGS__)
U3D_GLDS_KERNEL_X(k_igemm_glds_256x256, GLDS256_CFG)
// (a 4-wave variant with 128 x 128 per wave - 1.5x fewer LDS fragment bytes per MFMA - compiled to 512 VGPRs + spills and ran at
//  877 vs 1076 TFLOP/s: it needs a hand-scheduled fragment pipeline, not another template instance)
U3D_GLDS_KERNEL(k_igemm_glds_256x128, 4, 2, 4, 4)
U3D_GLDS_KERNEL(k_igemm_glds_128x64, 4, 1, 2, 4)
U3D_GLDS_KERNEL(k_igemm_glds_128x128, 2, 2, 4, 4)      /* 4 waves, 64 KiB LDS: two workgroups per CU run out of phase */
#undef U3D_GLDS_KERNEL
// f32-output instantiations of the two small tiles (split-bf16 products, u3d_igemm_fwd_split_bf16)
#define U3D_GLDS_KERNEL_F32O(NAME, A, B, C, D)                                                                                   \
  __global__ __launch_bounds__(A* B * 64) void NAME(const u16* in, const u16* w, const int* nbr, int ld, u16* out,               \
                                                    const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,            \
                                                    const float* bias, int relu, double* stats, BnEpi bn) {                      \
    igemm_glds_body<A, B, C, D, true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, bias, relu, stats, bn);         \
  }
U3D_GLDS_KERNEL_F32O(k_igemm_glds_128x64_f32o, 4, 1, 2, 4)
U3D_GLDS_KERNEL_F32O(k_igemm_glds_128x128_f32o, 2, 2, 4, 4)
#undef U3D_GLDS_KERNEL_F32O
typedef void (*glds_kernel_t)(const u16*, const u16*, const int*, int, u16*, const int*, int, int, int, int, const float*, int, double*, BnEpi);

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int launch_igemm_glds(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev,
                             int n_out_cap, int cin, int cout, int kvol, hipStream_t s, const float* bias = nullptr, int relu = 0,
                             double* stats = nullptr, const BnEpi bn = BnEpi{}, bool f32o = false) {
  constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 64 * 2;        // 256 x 256: 128 KiB
  glds_kernel_t kern = (BM == 256 && BN == 256) ? ((IGEMM_GLDS8 && nbr) ? k_igemm_glds8_256x256 : k_igemm_glds_256x256)
                       : (BM == 256 ? k_igemm_glds_256x128 : (BN == 128 ? k_igemm_glds_128x128 : k_igemm_glds_128x64));
  if (f32o) {      // f32-output instantiations exist for the shapes u3d_igemm_fwd_split_bf16 dispatches: 256 x 256 (eight-phase), 128 x 128, 128 x 64
    if (BM == 256 && BN == 256 && IGEMM_GLDS8 && nbr) kern = k_igemm_glds8_256x256_f32o;
    else if (BM == 128 && BN == 128) kern = k_igemm_glds_128x128_f32o;
    else if (BM == 128 && BN == 64) kern = k_igemm_glds_128x64_f32o;
    else return U3D_ERR_UNSUPPORTED;
  }
  int rows = BM;
  if (BM == 256 && BN == 256 && IGEMM_GLDS8 && nbr && igemm_rows192(nbr, n_out_cap, u3d_cdiv(cout, BN), kvol)) {
    kern = f32o ? k_igemm_glds8_192x256_f32o : k_igemm_glds8_192x256;      // same LDS image, 192 live rows (igemm_glds8_body<., 3>)
    rows = 192;
  }
  if (lds > 64 * 1024) U3D_ALLOW_LDS(kern, lds);      // one call site per template instantiation: per-kernel, per-device
  dim3 grid(u3d_cdiv(n_out_cap, rows), u3d_cdiv(cout, BN));
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, (const u16*)in, (const u16*)w, nbr, ld, (u16*)out, n_out_dev, n_out_cap, cin,
                     cout, kvol, bias, relu, stats, bn);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

// (The register-staged "ping-pong" kernel of round 1 - two wave groups in opposite phases, measured on par with k_igemm_fwd -
//  was removed when igemm_glds8_body took that idea to the LDS-DMA kernels; DESIGN.md 3.1 keeps its numbers.)

// Dense layer on rows: out[M,N] = act(x[M,K] @ W[N,K]^T + bias) — nn.Linear layout, bf16 in/out, f32 accumulate/bias.
// Small M (decoder: B*900 rows): 128x64 tiles so that a few hundred workgroups exist.
extern "C" int32_t u3d_linear_bf16(const void* x, const void* w, const float* bias, int32_t relu, void* out,
                                   const int32_t* m_dev, int32_t m_cap, int32_t k, int32_t n, u3d_stream s) {
  U3D_REQUIRE(x && w && out && m_dev, U3D_ERR_ARG);
  if (k % 64 != 0 || n % 64 != 0) return U3D_ERR_UNSUPPORTED;
  if (m_cap <= 0) return U3D_OK;
  // nn.Linear's [N, K] weight IS the n-major layout of the LDS-DMA kernels
  const long long wg256 = (long long)u3d_cdiv(m_cap, 256) * (n / 256);
  if (n % 256 == 0 && wg256 >= 128) return launch_igemm_glds<GLDS256_CFG>(x, w, nullptr, 0, out, m_dev, m_cap, k, n, 1, s, bias, relu);
  if (n % 128 == 0 && (long long)u3d_cdiv(m_cap, 256) * (n / 128) >= 128)
    return launch_igemm_glds<4, 2, 4, 4>(x, w, nullptr, 0, out, m_dev, m_cap, k, n, 1, s, bias, relu);
  return launch_igemm_glds<4, 1, 2, 4>(x, w, nullptr, 0, out, m_dev, m_cap, k, n, 1, s, bias, relu ? 1 : 0);
}

// row-tile height of the n-major (LDS-DMA) kernel u3d_igemm_fwd_stats_bf16 would launch for this shape; 0 = not served by it
extern "C" int32_t u3d_igemm_fwd_stats_tile_rows(int32_t n_out_cap, int32_t cin, int32_t cout) {
  if (!IGEMM_GLDS || cin % 64 != 0 || cout % 64 != 0 || n_out_cap <= 0) return 0;
  const long long wg256 = (long long)u3d_cdiv(n_out_cap, 256) * (cout / 256 > 0 ? cout / 256 : 1);
  if (cout % 256 == 0 && wg256 >= 128) return 256;
  return 128;
}
// forward with n-major weights w[kappa][cout][cin] that also leaves the per-row-tile BatchNorm statistics of the (bf16-rounded)
// output: stats f64 [ceil(n_out_cap / tile_rows)][2][cout] = (sum, sum of squares) per tile and column
// number of statistics partials u3d_igemm_fwd_stats_bf16 writes for this shape (0: not served).  LDS-DMA kernels: one per row tile
// (ceil(n_out_cap / u3d_igemm_fwd_stats_tile_rows)); direct-operand kernels of the narrow 27-offset levels: one per WAVE of their
// persistent grid - u3d_igemm_fwd_stats_tile_rows is 0 there and u3d_bn_finalize_partials takes rows_per_block = 0 ("all of them")
// rows per statistics partial of u3d_igemm_fwd_stats_bf16 for a conv WITH a neighbour table and `kvol` offsets (what
// u3d_bn_finalize_partials takes as rows_per_block): the LDS-DMA kernels' row-tile height - which depends on kvol where the
// 256 x 128 eight-phase kernel serves long reductions -, 0 for the per-wave partials of the direct-operand kernels / unserved shapes
extern "C" int32_t u3d_igemm_fwd_stats_rows(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol) {
  // the dispatch order of u3d_igemm_fwd_stats_bf16 / _dgrad_bnstats: 256-column tiles first, then the 256 x 128 eight-phase kernel
  const int32_t* some = (const int32_t*)16;      // "there is a neighbour table"
  const int tr = u3d_igemm_fwd_stats_tile_rows(n_out_cap, cin, cout);
  if (tr == 256) return (IGEMM_GLDS8 && igemm_rows192(some, n_out_cap, cout / 256, kvol)) ? 192 : 256;      // (launch_igemm_glds's choice)
  if (igemm_glds8n_shape(some, n_out_cap, cin, cout, kvol)) return igemm_rows192(some, n_out_cap, cout / 128, kvol) ? 192 : 256;
  return tr;
}
#ifndef DIRECT_STATS
#define DIRECT_STATS 1
#endif
extern "C" int32_t u3d_igemm_fwd_stats_blocks(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol) {
#if IGEMM_DIRECT && DIRECT_STATS
  if (n_out_cap > 0) {
    int nb = 0;
    if (u3d_launch_igemm_direct(nullptr, nullptr, (const int32_t*)16, 1, nullptr, nullptr, n_out_cap, cin, cout, kvol, 1, nullptr, nullptr, nullptr, &nb) == U3D_OK)
      return nb;
  }
#endif
  const int tr = u3d_igemm_fwd_stats_rows(n_out_cap, cin, cout, kvol);
  return tr ? u3d_cdiv(n_out_cap, tr) : 0;
}
extern "C" int32_t u3d_igemm_fwd_stats_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                                            const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                            double* stats, u3d_stream s) {
  U3D_REQUIRE(in && w && out && n_out_dev && stats && (nbr || kvol == 1), U3D_ERR_ARG);
#if IGEMM_DIRECT && DIRECT_STATS
  {
    const int rc = u3d_launch_igemm_direct(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, 1, s, nullptr, stats);
    if (rc != U3D_ERR_UNSUPPORTED) return rc;
  }
#endif
  const int tr = u3d_igemm_fwd_stats_tile_rows(n_out_cap, cin, cout);
  if (tr == 0) return U3D_ERR_UNSUPPORTED;
  if (tr == 256) return launch_igemm_glds<GLDS256_CFG>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, nullptr, 0, stats);
  if (igemm_glds8n_shape(nbr, n_out_cap, cin, cout, kvol))          // (256-row tiles too: u3d_igemm_fwd_stats_rows)
    return launch_igemm_glds8n(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, nullptr, 0, stats);
  if (cout % 128 == 0) return launch_igemm_glds<2, 2, 4, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, nullptr, 0, stats);
  return launch_igemm_glds<4, 1, 2, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, nullptr, 0, stats);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 convolution: f32-grade products on the bf16 matrix pipe (the reference keeps SparseEncoderHD and SECOND3D in fp32 - ref:
// sparse_encoder_hd.py:62-64, uni3detr.py:150-151 - and the exact f32 MFMA runs at 1/16 of the bf16 rate).  An f32 tensor x is held
// as two bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits together), and x.w ~ hi.wh + hi.wl + lo.wh with f32
// accumulation (the dropped lo.wl term is 2^-16 of a 2^-8 term).  The kernels are the LDS-DMA implicit-GEMM kernels above, UNCHANGED:
// the three products are three sets of "offsets" - the caller stacks the planes as rows [hi ; lo] of one matrix, triples the
// neighbour table (nbr, nbr, nbr + plane stride) and the weights (wh, wl, wh) - and this entry only picks the f32-output instantiations.
//   in   bf16 [2 * n_in_cap][cin] (u3d_split_rows_f32), w bf16 [kvol3][cout][cin] (n-major, kvol3 = 3 * offsets), nbr int32 [kvol3][ld],
//   out  f32 [n_out_cap][cout]; stats (optional) f64 [ceil(n_out_cap / u3d_igemm_fwd_stats_rows(.., kvol3))][2][cout] of the f32 output
extern "C" int32_t u3d_igemm_fwd_split_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, float* out,
                                            const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol3,
                                            double* stats, const float* addend, u3d_stream s) {
  U3D_REQUIRE(in && w && out && n_out_dev && nbr && kvol3 > 0, U3D_ERR_ARG);
  if (!IGEMM_GLDS || cin % 64 != 0 || cout % 64 != 0) return U3D_ERR_UNSUPPORTED;
  if (n_out_cap <= 0) return U3D_OK;
  const int tr = u3d_igemm_fwd_stats_tile_rows(n_out_cap, cin, cout);
  if (tr == 256 && addend && !(IGEMM_GLDS8 && nbr)) return U3D_ERR_UNSUPPORTED;      // (the two-phase 256 x 256 kernel has no addend epilogue)
  if (tr == 256) return launch_igemm_glds<GLDS256_CFG>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol3, s, addend, addend ? 2 : 0, stats, BnEpi{}, true);
  if (igemm_glds8n_shape(nbr, n_out_cap, cin, cout, kvol3))
    return launch_igemm_glds8n(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol3, s, addend, addend ? 2 : 0, stats, BnEpi{}, true);
  if (cout % 128 == 0) return launch_igemm_glds<2, 2, 4, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol3, s, addend, addend ? 2 : 0, stats, BnEpi{}, true);
  return launch_igemm_glds<4, 1, 2, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol3, s, addend, addend ? 2 : 0, stats, BnEpi{}, true);
}

// hi / lo bf16 planes of an f32 row matrix: dst[r] = bf16(x[r]), dst[n_cap + r] = bf16(x[r] - dst[r]) (round to nearest even both)
__global__ __launch_bounds__(256) void k_split_rows_f32(const float* __restrict__ x, const int* __restrict__ n_dev, int n_cap, int c,
                                                        u16* __restrict__ dst) {
  // rows past the device-side count (capacity padding of a captured step) become ZERO rows of both planes: a table entry can then
  // never pick up a stale NaN pattern, whatever it names
  const long long n = (long long)min(*n_dev, n_cap) * c / 4, plane = (long long)n_cap * c, ncap = plane / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ncap; i += (long long)gridDim.x * 256) {
    const f32x4 v = i < n ? *(const f32x4*)(x + i * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const bf16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
    *(bf16x4*)(dst + i * 4) = h;
    *(bf16x4*)(dst + plane + i * 4) = l;
  }
}
extern "C" int32_t u3d_split_rows_f32(const float* x, const int32_t* n_dev, int32_t n_cap, int32_t c, void* dst, u3d_stream s) {
  U3D_REQUIRE(x && n_dev && dst && c > 0 && c % 4 == 0, U3D_ERR_ARG);
  if (n_cap <= 0) return U3D_OK;
  const long long n4 = (long long)n_cap * c / 4;
  const int blocks = (int)(n4 / 256 + 1 < 4096 ? n4 / 256 + 1 : 4096);
  hipLaunchKernelGGL(k_split_rows_f32, dim3(blocks), dim3(256), 0, s, x, n_dev, n_cap, c, (u16*)dst);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// the weight side of a split-bf16 product: dst bf16 [3][K][A][B] = (hi, lo, hi) of src[k * sk + a * sa + b * sb] (f32, any layout:
// the checkpoint layouts [kD,kH,kW,Cin,Cout] / [Cout,Cin,kD,kH,kW] are read in place, no re-laid-out f32 copy in between)
__global__ __launch_bounds__(256) void k_split3_weights(const float* __restrict__ src, long long sk, long long sa, long long sb, int K, int A,
                                                        int B, u16* __restrict__ dst) {
  const long long n = (long long)K * A * B;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i % B), a = (int)((i / B) % A), k = (int)(i / ((long long)A * B));
    const float v = src[k * sk + a * sa + b * sb];
    const u16 h = f2bf(v);
    const u16 l = f2bf(v - __uint_as_float((unsigned)h << 16));
    dst[i] = h; dst[n + i] = l; dst[2 * n + i] = h;
  }
}
extern "C" int32_t u3d_split3_weights(const float* src, int64_t sk, int64_t sa, int64_t sb, int32_t k, int32_t a, int32_t b, void* dst,
                                      u3d_stream s) {
  U3D_REQUIRE(src && dst && k > 0 && a > 0 && b > 0, U3D_ERR_ARG);
  const long long n = (long long)k * a * b;
  const int blocks = (int)(n / 256 + 1 < 2048 ? n / 256 + 1 : 2048);
  hipLaunchKernelGGL(k_split3_weights, dim3(blocks), dim3(256), 0, s, src, (long long)sk, (long long)sa, (long long)sb, k, a, b, (u16*)dst);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// All weight splits of a step in ONE launch (a `parity` step made 88 launches of k_split3_weights, ~6.6 us each whatever the size):
// jobs[j] describes one (parameter, layout) pair, blocks [first_block[j], first_block[j + 1]) of the grid work on it.
struct U3dSplit3Job { const float* src; u16* dst; long long sk, sa, sb; int K, A, B, first_block; };
#define SPLIT3_EPB 2048          /* elements per block */
__global__ __launch_bounds__(256) void k_split3_weights_batch(const U3dSplit3Job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs;                               // the job this block belongs to: last j with first_block[j] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const U3dSplit3Job jb = jobs[lo];
  const long long n = (long long)jb.K * jb.A * jb.B;
  const long long i0 = (long long)((int)blockIdx.x - jb.first_block) * SPLIT3_EPB;
  for (long long i = i0 + threadIdx.x; i < i0 + SPLIT3_EPB && i < n; i += 256) {
    const int b = (int)(i % jb.B), a = (int)((i / jb.B) % jb.A), k = (int)(i / ((long long)jb.A * jb.B));
    const float v = jb.src[k * jb.sk + a * jb.sa + b * jb.sb];
    const u16 h = f2bf(v);
    const u16 l = f2bf(v - __uint_as_float((unsigned)h << 16));
    jb.dst[i] = h; jb.dst[n + i] = l; jb.dst[2 * n + i] = h;
  }
}
// hi / lo planes of up to 32 (possibly strided) f32 row matrices in ONE launch (the decoder's parameter-gradient operands in `parity`
// mode: ~30 tensors of 7 200 rows per layer, each of which was a copy + a u3d_split_rows_f32 launch of ~6 us).  The job list travels
// BY VALUE in the kernel arguments: the sources are slots of per-step workspaces, so nothing about it can be uploaded ahead of time.
// (U3dSplitRowsJobs: include/u3d_hip.h - row r of job j starts at src[j] + r * ld[j]; dst[j] bf16 [2 * rows[j]][cols[j]], hi plane then lo
//  plane; blocks [first_block[j], first_block[j + 1]) work on job j, 1024 elements per block)
__global__ __launch_bounds__(256) void k_split_rows_batch(const U3dSplitRowsJobs jb) {
  int j = 0;
  while (j + 1 < jb.njobs && jb.first_block[j + 1] <= (int)blockIdx.x) ++j;
  const int cols = jb.cols[j], c4 = cols >> 2;
  const long long n4 = (long long)jb.rows[j] * c4, plane = (long long)jb.rows[j] * cols;
  const long long i = (long long)((int)blockIdx.x - jb.first_block[j]) * 256 + threadIdx.x;
  if (i >= n4) return;
  const int r = (int)(i / c4), c = (int)(i % c4) * 4;
  const f32x4 v = *(const f32x4*)(jb.src[j] + (long long)r * jb.ld[j] + c);
  const bf16x4 h = __builtin_convertvector(v, bf16x4);
  const bf16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
  u16* d = (u16*)jb.dst[j] + (long long)r * cols + c;
  *(bf16x4*)d = h;
  *(bf16x4*)(d + plane) = l;
}
// jobs->first_block is filled here (cols % 4 == 0, 16-byte aligned rows); njobs <= 32
extern "C" int32_t u3d_split_rows_batch(const U3dSplitRowsJobs* jobs, u3d_stream s) {
  U3D_REQUIRE(jobs && jobs->njobs > 0 && jobs->njobs <= 32, U3D_ERR_ARG);
  U3dSplitRowsJobs jb = *jobs;
  int fb = 0;
  for (int j = 0; j < jb.njobs; ++j) {
    U3D_REQUIRE(jb.src[j] && jb.dst[j] && jb.cols[j] > 0 && jb.cols[j] % 4 == 0 && jb.ld[j] % 4 == 0 && jb.rows[j] >= 0, U3D_ERR_ARG);
    jb.first_block[j] = fb;
    fb += (int)(((long long)jb.rows[j] * (jb.cols[j] / 4) + 255) / 256);
  }
  jb.first_block[jb.njobs] = fb;
  if (fb == 0) return U3D_OK;
  hipLaunchKernelGGL(k_split_rows_batch, dim3(fb), dim3(256), 0, s, jb);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
// dW of a split-bf16 product = the sum of its three bf16 products' f32 weight gradients (x^T dy ~ xh^T dyh + xl^T dyh + xh^T dyl):
// one pass instead of two element-wise additions
__global__ __launch_bounds__(256) void k_sum3_f32(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                  float* __restrict__ out, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    *(f32x4*)(out + i * 4) = (*(const f32x4*)(a + i * 4) + *(const f32x4*)(b + i * 4)) + *(const f32x4*)(c + i * 4);
}
extern "C" int32_t u3d_sum3_f32(const float* a, const float* b, const float* c, float* out, int64_t n, u3d_stream s) {
  U3D_REQUIRE(a && b && c && out && n >= 0 && n % 4 == 0, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(k_sum3_f32, dim3((int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048)), dim3(256), 0, s, a, b, c, out, n4);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int64_t u3d_split3_job_bytes(void) { return (int64_t)sizeof(U3dSplit3Job); }
extern "C" int32_t u3d_split3_job_blocks(int32_t k, int32_t a, int32_t b) {
  const long long n = (long long)k * a * b;
  return (int32_t)((n + SPLIT3_EPB - 1) / SPLIT3_EPB);
}
// jobs: DEVICE array of njobs records {src, dst, sk, sa, sb (element strides, int64), K, A, B, first_block (int32)} of u3d_split3_job_bytes()
// bytes each (natural C layout), first_block ascending from 0; total_blocks = sum of u3d_split3_job_blocks over the jobs
extern "C" int32_t u3d_split3_weights_batch(const void* jobs, int32_t njobs, int32_t total_blocks, u3d_stream s) {
  U3D_REQUIRE(jobs && njobs > 0 && total_blocks > 0, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_split3_weights_batch, dim3(total_blocks), dim3(256), 0, s, (const U3dSplit3Job*)jobs, njobs);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// returns U3D_ERR_UNSUPPORTED when the shape is better served by the first-generation kernel

// =============================================================================================
// The encoder's input convolution: 4 point features (padded to 8) -> 16 channels, 27 offsets, ~128 k rows
// (ref: sparse_encoder_hd.py:80-88).  0.9 GFLOP: far too small for a tiled MFMA kernel (the first-generation kernel spent 166 us on
// it, its weight gradient 126 us) - plain VALU, one thread per output row, weights (27 x 8 x 16 f32 = 13.5 KB) in LDS read by
// broadcast; rows without a neighbour at an offset skip it (6 % of the (offset,row) pairs exist at this level).
// =============================================================================================
#define CONVIN_CIN 8
#define CONVIN_COUT 16
#define CONVIN_MAXK 27
#define CONVIN_WG_ROWS 128          /* rows per workgroup of the weight-gradient kernel */

__device__ __forceinline__ float convin_bf(u16 v) { return __uint_as_float((unsigned)v << 16); }

__global__ __launch_bounds__(256) void k_conv_in_fwd(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr,
                                                     int ld, u16* __restrict__ out, const int* __restrict__ n_out_dev, int n_out_cap,
                                                     int kvol) {
  __shared__ __attribute__((aligned(16))) float ws[CONVIN_MAXK * CONVIN_CIN * CONVIN_COUT];
  for (int i = threadIdx.x; i < kvol * CONVIN_CIN * CONVIN_COUT; i += 256) ws[i] = convin_bf(w[i]);
  __syncthreads();
  const int n = min(*n_out_dev, n_out_cap);
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  float acc[CONVIN_COUT];
#pragma unroll
  for (int co = 0; co < CONVIN_COUT; ++co) acc[co] = 0.f;
  int idxs[CONVIN_MAXK];                       // all offsets' indices first: 27 independent loads in flight, not 27 round trips
#pragma unroll
  for (int k = 0; k < CONVIN_MAXK; ++k) idxs[k] = k < kvol ? (nbr ? nbr[(long long)k * ld + m] : m) : -1;
#pragma unroll
  for (int k = 0; k < CONVIN_MAXK; ++k) {
    const int idx = idxs[k];
    if (idx < 0) continue;
    const uint4 xv = *(const uint4*)(in + (long long)idx * CONVIN_CIN);
    const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
    const float* wk = ws + k * CONVIN_CIN * CONVIN_COUT;
#pragma unroll
    for (int ci = 0; ci < CONVIN_CIN; ++ci) {
      const float x = (ci & 1) ? __uint_as_float(xw[ci >> 1] & 0xffff0000u) : __uint_as_float(xw[ci >> 1] << 16);
#pragma unroll
      for (int q = 0; q < CONVIN_COUT / 4; ++q) {
        const float4 wv = *(const float4*)(wk + ci * CONVIN_COUT + q * 4);
        acc[q * 4 + 0] += x * wv.x; acc[q * 4 + 1] += x * wv.y; acc[q * 4 + 2] += x * wv.z; acc[q * 4 + 3] += x * wv.w;
      }
    }
  }
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x8_t f = {acc[h * 8], acc[h * 8 + 1], acc[h * 8 + 2], acc[h * 8 + 3], acc[h * 8 + 4], acc[h * 8 + 5], acc[h * 8 + 6], acc[h * 8 + 7]};
    *(bf16x8_t*)(out + (long long)m * CONVIN_COUT + h * 8) = __builtin_convertvector(f, bf16x8_t);
  }
}

// weight gradient: dW[k][ci][:] = sum_rows in[nbr[k][row]][ci] * dout[row][:].  A workgroup owns CONVIN_WG_ROWS output rows (their
// dout staged in LDS), thread (k, ci) walks them branch-free (absent neighbour -> factor 0) with the index and input loads of
// several rows in flight; per-workgroup partials [blocks][K*8*16] are summed in order by k_igemm_wgrad_reduce.
__global__ __launch_bounds__(256) void k_conv_in_wgrad(const u16* __restrict__ in, const u16* __restrict__ dout, const int* __restrict__ nbr,
                                                       int ld, float* __restrict__ partial, const int* __restrict__ n_out_dev,
                                                       int n_out_cap, int kvol) {
  __shared__ __attribute__((aligned(16))) u16 dys[CONVIN_WG_ROWS * CONVIN_COUT];
  const int n = min(*n_out_dev, n_out_cap);
  const int r0 = blockIdx.x * CONVIN_WG_ROWS;
  const int r1 = min(n, r0 + CONVIN_WG_ROWS);
  for (int i = threadIdx.x; i < CONVIN_WG_ROWS * 2; i += 256) {
    const int row = r0 + (i >> 1);
    uint4 v = {0u, 0u, 0u, 0u};
    if (row < r1) v = *(const uint4*)(dout + (long long)row * CONVIN_COUT + (i & 1) * 8);
    *(uint4*)(dys + (i >> 1) * CONVIN_COUT + (i & 1) * 8) = v;
  }
  __syncthreads();
  const int k = threadIdx.x >> 3, ci = threadIdx.x & 7;
  if (k >= kvol) return;
  float acc[CONVIN_COUT];
#pragma unroll
  for (int co = 0; co < CONVIN_COUT; ++co) acc[co] = 0.f;
  const int* nk = nbr ? nbr + (long long)k * ld : nullptr;
  for (int rb = r0; rb < r1; rb += 8) {
    int idx8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) idx8[j] = (rb + j < r1) ? (nk ? nk[rb + j] : rb + j) : -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
    const int r = rb + j, idx = idx8[j];
    if (__builtin_amdgcn_ballot_w64(idx >= 0) == 0) continue;       // none of this wave's 8 offsets has a neighbour for row r (most rows)
    const float x = idx >= 0 ? convin_bf(in[(long long)idx * CONVIN_CIN + ci]) : 0.f;
    const uint4 a = *(const uint4*)(dys + (r - r0) * CONVIN_COUT), b = *(const uint4*)(dys + (r - r0) * CONVIN_COUT + 8);
    const unsigned dw_[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[2 * q] += x * __uint_as_float(dw_[q] << 16);
      acc[2 * q + 1] += x * __uint_as_float(dw_[q] & 0xffff0000u);
    }
    }
  }
  float* p = partial + (long long)blockIdx.x * (kvol * CONVIN_CIN * CONVIN_COUT) + (k * CONVIN_CIN + ci) * CONVIN_COUT;
#pragma unroll
  for (int q = 0; q < 4; ++q) *(float4*)(p + q * 4) = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
}
// few outputs (K*8*16 = 3456), many partials (one per 128 rows): 64 columns x 16 partial-lanes per workgroup, fixed order
__global__ __launch_bounds__(1024) void k_conv_in_reduce(const float* __restrict__ partial, float* __restrict__ dw, int n, int nsplit) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, lane = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
  float a = 0.f, b = 0.f, cc = 0.f, d = 0.f;
  if (col < n) {
    const float* p = partial + col;
    int k = lane;
    for (; k + 48 < nsplit; k += 64) {
      a += p[(long long)k * n]; b += p[(long long)(k + 16) * n]; cc += p[(long long)(k + 32) * n]; d += p[(long long)(k + 48) * n];
    }
    for (; k < nsplit; k += 16) a += p[(long long)k * n];
  }
  red[lane][c] = (a + b) + (cc + d);
  __syncthreads();
  if (threadIdx.x < 64 && col < n) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][threadIdx.x];
    dw[col] = s;
  }
}
static inline bool convin_shape(int cin, int cout, int kvol) { return cin == CONVIN_CIN && cout == CONVIN_COUT && kvol >= 1 && kvol <= CONVIN_MAXK; }

// out = conv(in) + addend (bf16, out's shape) in one pass - the input gradient of a residual block's first conv with the residual
// branch's gradient summed in by the epilogue.  Shapes: the direct-operand kernels (16/32/64 channels, 27 offsets) and the n-major
// LDS-DMA kernels (transpose_w != 0, cin % 64 == 0; 256 x 256 tiles: the eight-phase kernel only); anything else: U3D_ERR_UNSUPPORTED (the caller adds).
extern "C" int32_t u3d_igemm_fwd_add_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, const void* addend, void* out,
                                          const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                          int32_t transpose_w, u3d_stream s) {
  U3D_REQUIRE(in && w && out && addend && n_out_dev && nbr, U3D_ERR_ARG);
  if (n_out_cap <= 0) return U3D_OK;
#if IGEMM_DIRECT
  {
    const int rc = u3d_launch_igemm_direct(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, transpose_w, s, addend);
    if (rc != U3D_ERR_UNSUPPORTED) return rc;
  }
#endif
#if IGEMM_GLDS
  if (transpose_w && cin % 64 == 0 && cout % 64 == 0) {
    const long long wg256 = (long long)u3d_cdiv(n_out_cap, 256) * (cout / 256 > 0 ? cout / 256 : 1);
    if (cout % 256 == 0 && wg256 >= 128) {                                            // 256 x 256: the eight-phase kernel's epilogue takes the addend
      if (!(IGEMM_GLDS8 && nbr)) return U3D_ERR_UNSUPPORTED;
      return launch_igemm_glds<GLDS256_CFG>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, (const float*)addend, 2);
    }
    if (igemm_glds8n_shape(nbr, n_out_cap, cin, cout, kvol))
      return launch_igemm_glds8n(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, (const float*)addend, 2);
    if (cout % 128 == 0) return launch_igemm_glds<2, 2, 4, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, (const float*)addend, 2);
    return launch_igemm_glds<4, 1, 2, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, (const float*)addend, 2);
  }
#endif
  return U3D_ERR_UNSUPPORTED;
}

// Input gradient (n-major weights [K][Cin][Cout] as the dgrad passes them) + optional addend + the BatchNorm-BACKWARD statistics of the
// layer that produced the conv's input, per row tile: stats f64 [ceil(n_out_cap / T)][2][cout] = (sum g, sum g * xhat), T =
// u3d_igemm_fwd_stats_rows(n_out_cap, cin, cout, kvol) (the dispatch below is u3d_igemm_fwd_stats_bf16's).  LDS-DMA kernels with the
// addend epilogue only: U3D_ERR_UNSUPPORTED for the direct-operand shapes and the two-phase 256 x 256 kernel (the caller then runs
// u3d_igemm_fwd_add_bf16 / u3d_igemm_fwd_bf16 and u3d_bn_bwd_stats).
extern "C" int32_t u3d_igemm_dgrad_bnstats_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, const void* addend, void* out,
                                                const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                                const u3d_bn_epi* bn, double* stats, u3d_stream s) {
  U3D_REQUIRE(in && w && out && n_out_dev && nbr && bn && bn->x && bn->mean && bn->invstd && stats, U3D_ERR_ARG);
  U3D_REQUIRE(!bn->relu || bn->y || (bn->gamma && bn->beta), U3D_ERR_ARG);
#if IGEMM_GLDS
  if (n_out_cap <= 0 || cin % 64 != 0 || cout % 64 != 0) return U3D_ERR_UNSUPPORTED;
#if IGEMM_DIRECT
  if (u3d_launch_igemm_direct(nullptr, nullptr, (const int32_t*)16, 1, nullptr, nullptr, n_out_cap, cin, cout, kvol, 1, nullptr, nullptr, nullptr, nullptr)
      != U3D_ERR_UNSUPPORTED) return U3D_ERR_UNSUPPORTED;                 // a direct-operand shape: its partial layout is per wave
#endif
  BnEpi e;
  e.x = (const u16*)bn->x; e.y = (const u16*)bn->y; e.mean = bn->mean; e.invstd = bn->invstd; e.gamma = bn->gamma; e.beta = bn->beta;
  e.relu = bn->relu;
  const float* add = (const float*)addend;
  const int fl = addend ? 2 : 0;
  const int tr = u3d_igemm_fwd_stats_tile_rows(n_out_cap, cin, cout);
  if (tr == 0) return U3D_ERR_UNSUPPORTED;
  if (tr == 256) {
    if (!(IGEMM_GLDS8 && nbr)) return U3D_ERR_UNSUPPORTED;
    return launch_igemm_glds<GLDS256_CFG>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, add, fl, stats, e);
  }
  if (igemm_glds8n_shape(nbr, n_out_cap, cin, cout, kvol))
    return launch_igemm_glds8n(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, add, fl, stats, e);
  if (cout % 128 == 0) return launch_igemm_glds<2, 2, 4, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, add, fl, stats, e);
  return launch_igemm_glds<4, 1, 2, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s, add, fl, stats, e);
#else
  return U3D_ERR_UNSUPPORTED;
#endif
}

extern "C" int32_t u3d_igemm_fwd_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                                      const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                      int32_t transpose_w, u3d_stream s) {
  U3D_REQUIRE(in && w && out && n_out_dev && (nbr || kvol == 1), U3D_ERR_ARG);
  if (!transpose_w && convin_shape(cin, cout, kvol)) {            // the encoder's input convolution (w = [K][8][16])
    if (n_out_cap <= 0) return U3D_OK;
    hipLaunchKernelGGL(k_conv_in_fwd, dim3(u3d_cdiv(n_out_cap, 256)), dim3(256), 0, s, (const u16*)in, (const u16*)w, nbr, ld, (u16*)out, n_out_dev,
                       n_out_cap, kvol);
    return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
  }
#if IGEMM_DIRECT
  // 16/32/64-channel sparse levels with 27 offsets (not 64 -> 64): activations gathered straight into the MFMA operand registers,
  // all weights LDS-resident, persistent barrier-free waves (igemm_direct.hip)
  {
    const int rc = u3d_launch_igemm_direct(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, transpose_w, s);
    if (rc != U3D_ERR_UNSUPPORTED) return rc;
  }
#endif
#if IGEMM_SMALL_C
  // 16/32-channel sparse levels (and the 32<->64 transitions): 256-row tiles, one MFMA k-step per stage (BK = 32), the same
  // register-staged, software-pipelined loop as the wide layers - the first-generation kernel it replaces does
  // "indices -> barrier -> gather -> barrier -> MFMA -> barrier" per offset with nothing in flight across the barriers
  if (n_out_cap > 0 && cin % 16 == 0 && cout % 16 == 0 && cin <= 64 && cout <= 64 && (cin < 64 || cout < 64)) {
#define IG_SMALL(WNV, BKV)                                                                                                                          \
    return transpose_w ? launch_igemm_fwd<4, 1, IGEMM_SMALL_WM, WNV, false, BKV, IGEMM_SMALL_PF>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s)   \
                       : launch_igemm_fwd<4, 1, IGEMM_SMALL_WM, WNV, true, BKV, IGEMM_SMALL_PF>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
    // (cin = 64 -> cout 32/16, the dgrad of the 32->64 transition, measured slower here than on the first-generation kernel: 212 vs 170 us)
    if (cin == 16 || cin == 32) {
      if (cout == 16) { IG_SMALL(1, 32) }
      if (cout == 32) { IG_SMALL(2, 32) }
      if (cout == 64) { IG_SMALL(4, 32) }
    }
#undef IG_SMALL
  }
#endif
  if (cin % 64 != 0 || cout % 8 != 0 || cout < 64) return U3D_ERR_UNSUPPORTED;
  if (n_out_cap <= 0) return U3D_OK;
#define IG_CASE(A, B, C, D)                                                                                                  \
  return transpose_w ? launch_igemm_fwd<A, B, C, D, false>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s)   \
                     : launch_igemm_fwd<A, B, C, D, true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
  // (a 128x128-tile variant for layers with few rows was measured SLOWER: 285 vs 512 TF/s at N=48000 — the L2->CU traffic of
  //  the smaller tile outweighs the better CU fill; not dispatched)
  // few row tiles (the stride-4 branch of SECOND3D: 12000 rows): 256 x 256 tiles leave most CUs idle -> narrower tiles
  const long long wg256 = (long long)u3d_cdiv(n_out_cap, 256) * (cout / 256 > 0 ? cout / 256 : 1);
#if IGEMM_GLDS
  if (transpose_w) {                                                // n-major weights: LDS-DMA staged kernels
    if (cout % 256 == 0 && wg256 >= 128) return launch_igemm_glds<GLDS256_CFG>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
    // 128 x 128 (4 waves, 64 KiB LDS, two workgroups per CU out of phase): +3...6 % over 256 x 128 on the 128- and 512-channel layers
    if (igemm_glds8n_shape(nbr, n_out_cap, cin, cout, kvol)) return launch_igemm_glds8n(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
    if (cout % 128 == 0) return launch_igemm_glds<2, 2, 4, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
    if (cout % 64 == 0) return launch_igemm_glds<4, 1, 2, 4>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, s);
  }
#endif
  if (cout % 256 == 0 && wg256 < 128) { IG_CASE(4, 2, 4, 4) }       // measured: 94 workgroups (N=12000, 512 ch) 0.136 -> 0.100 ms;
                                                                    // 188 workgroups (N=48000, 256 ch) stay faster on 256 x 256
  if (cout >= 256 && cout % 256 == 0) { IG_CASE(2, 4, 8, 4) }       // 256 x 256
  if (cout >= 128 && cout % 128 == 0) { IG_CASE(4, 2, 4, 4) }       // 256 x 128
  if (cout % 64 == 0) { IG_CASE(4, 1, 2, 4) }                        // 128 x 64: 57 KB LDS -> 2 workgroups per CU hide the gather latency
#undef IG_CASE
  return U3D_ERR_UNSUPPORTED;
}

// =============================================================================================
// weight gradient: workgroup (split, kappa, block) accumulates dW[kappa][ci0:+TM][co0:+TN] over its slice of output rows.
// stage = 64 output rows: A tile [64][TM] (gathered input rows), D tile [64][TN] (dout rows), both row-major with the
// reduction index as the LDS row -> both MFMA operands come from transpose reads.
// =============================================================================================
template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void k_igemm_wgrad(const u16* __restrict__ in, const u16* __restrict__ dout,
                                                                        const int* __restrict__ nbr, int ld, float* __restrict__ partial,
                                                                        const int* __restrict__ n_out_dev, int n_out_cap, int cin,
                                                                        int cout, int kvol, int co_blocks) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TM = WAVES_M * WM * 16, TN = WAVES_N * WN * 16, RK = 64;
  constexpr int LDA = TM + 16, LDD = TN + 16;
  constexpr int A_ELEMS = RK * LDA, D_ELEMS = RK * LDD;
  constexpr int A_SEGS = RK * TM / 8 / NT, D_SEGS = RK * TN / 8 / NT;
  static_assert(RK * TM / 8 % NT == 0 && RK * TN / 8 % NT == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  constexpr int STAGE_ELEMS = A_ELEMS + D_ELEMS;

  const int n_out = min(*n_out_dev, n_out_cap);
  const int nsplit = gridDim.x, split = blockIdx.x, kap = blockIdx.y;
  const int ci0 = (blockIdx.z / co_blocks) * TM, co0 = (blockIdx.z % co_blocks) * TN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (n_out + RK - 1) / RK;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

  // Two operand-fetch variants (measured, tools/conv_bench.py): raw buffer loads + branch-free stage body win for the 128 / 64 /
  // 32 / 16 tiles (+35...65 %), the 256 x 256 tile schedules better with the plain predicated loads (879 vs 707 TFLOP/s).
  if constexpr (TM < 256) {
    // operand fetch as in k_igemm_fwd: raw buffer loads, missing rows = out-of-range offset (hardware zero fill), one branch-free
    // stage body, gather indices consumed one stage after they were requested
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, -1, 0x00020000);
    const unsigned in_row_bytes = (unsigned)cin * 2u, d_row_bytes = (unsigned)cout * 2u;
    int a_row[A_SEGS], d_row[D_SEGS];
    unsigned a_col[A_SEGS], d_col[D_SEGS];
  #pragma unroll
    for (int u = 0; u < A_SEGS; ++u) {
      int sgi = tid + u * NT;
      a_row[u] = sgi / (TM / 8);
      int c = ci0 + (sgi % (TM / 8)) * 8;
      a_col[u] = c < cin ? (unsigned)c * 2u : 0xFFFFFFFFu;
    }
  #pragma unroll
    for (int u = 0; u < D_SEGS; ++u) {
      int sgi = tid + u * NT;
      d_row[u] = sgi / (TN / 8);
      int c = co0 + (sgi % (TN / 8)) * 8;
      d_col[u] = c < cout ? (unsigned)c * 2u : 0xFFFFFFFFu;
    }
    u32x4 ra[A_SEGS], rd[D_SEGS];
    int src_nxt[A_SEGS];
    auto load_src_next = [&](int t) {                     // gather indices of stage t, fetched one stage early (raw: masked at use)
      const int r0 = t * RK;
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) {
        int m = r0 + a_row[u];
        int mc = m < n_out ? m : n_out - 1;
        src_nxt[u] = nbr ? nbr[(long long)kap * ld + mc] : mc;
      }
    };
    auto issue_loads = [&](int t) {
      const int r0 = t * RK;
      const bool live = t < t_end;
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) {
        const bool ok = live && (r0 + a_row[u] < n_out) && src_nxt[u] >= 0 && a_col[u] != 0xFFFFFFFFu;
        unsigned voff = ok ? (unsigned)src_nxt[u] * in_row_bytes + a_col[u] : 0xFFFFFFFFu;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff, 0, 0);
      }
  #pragma unroll
      for (int u = 0; u < D_SEGS; ++u) {
        const int m = r0 + d_row[u];
        const bool ok = live && m < n_out && d_col[u] != 0xFFFFFFFFu;
        unsigned voff = ok ? (unsigned)m * d_row_bytes + d_col[u] : 0xFFFFFFFFu;
        rd[u] = __builtin_amdgcn_raw_buffer_load_b128(d_rs, voff, 0, 0);
      }
      load_src_next(t + 1);
    };
    auto store_lds = [&](int buf) {
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) { int sgi = tid + u * NT; *(u32x4*)(smem + buf * STAGE_ELEMS + a_row[u] * LDA + (sgi % (TM / 8)) * 8) = ra[u]; }
  #pragma unroll
      for (int u = 0; u < D_SEGS; ++u) { int sgi = tid + u * NT; *(u32x4*)(smem + buf * STAGE_ELEMS + A_ELEMS + d_row[u] * LDD + (sgi % (TN / 8)) * 8) = rd[u]; }
    };

    if (t_begin < t_end) {
      load_src_next(t_begin);
      issue_loads(t_begin);
      store_lds(0);
      __syncthreads();
      for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        issue_loads(t + 1);                               // past the last stage: all offsets out of range -> zeros into the idle buffer
        const u16* A = smem + buf * STAGE_ELEMS;
        const u16* D = A + A_ELEMS;
  #pragma unroll
        for (int ks = 0; ks < RK / 32; ++ks) {
          bf16x8 bfr[WN];
  #pragma unroll
          for (int b = 0; b < WN; ++b) bfr[b] = tr_frag(D, LDD, ks * 32, (wn * WN + b) * 16, lane);
  #pragma unroll
          for (int a = 0; a < WM; ++a) {
            bf16x8 af = tr_frag(A, LDA, ks * 32, (wm * WM + a) * 16, lane);
  #pragma unroll
            for (int b = 0; b < WN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[b], acc[a][b], 0, 0, 0);
          }
          if (ks == 0) store_lds(buf ^ 1);                // under the second k-step's MFMAs (see k_igemm_fwd)
        }
        __syncthreads();
      }
    }
  } else {
    uint4 ra[A_SEGS], rd[D_SEGS];
    int src_cur[A_SEGS], src_nxt[A_SEGS];
    auto load_src_next = [&](int t) {                     // gather indices of stage t, fetched one stage early
      const int r0 = t * RK;
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) {
        int m = r0 + (tid + u * NT) / (TM / 8);
        src_nxt[u] = (t < t_end && m < n_out) ? (nbr ? nbr[(long long)kap * ld + m] : m) : -1;
      }
    };
    auto issue_loads = [&](int t) {
      const int r0 = t * RK;
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) src_cur[u] = src_nxt[u];
      load_src_next(t + 1);
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) {
        int sgi = tid + u * NT;
        int part = sgi % (TM / 8);
        int src = src_cur[u];
        int c = ci0 + part * 8;
        ra[u] = (src >= 0 && c < cin) ? *(const uint4*)(in + (long long)src * cin + c) : make_uint4(0, 0, 0, 0);
      }
  #pragma unroll
      for (int u = 0; u < D_SEGS; ++u) {
        int sgi = tid + u * NT;
        int row = sgi / (TN / 8), part = sgi % (TN / 8);
        int m = r0 + row;
        int c = co0 + part * 8;
        rd[u] = (m < n_out && c < cout) ? *(const uint4*)(dout + (long long)m * cout + c) : make_uint4(0, 0, 0, 0);
      }
    };
    auto store_lds = [&](int buf) {
  #pragma unroll
      for (int u = 0; u < A_SEGS; ++u) { int sgi = tid + u * NT; *(uint4*)(smem + buf * STAGE_ELEMS + (sgi / (TM / 8)) * LDA + (sgi % (TM / 8)) * 8) = ra[u]; }
  #pragma unroll
      for (int u = 0; u < D_SEGS; ++u) { int sgi = tid + u * NT; *(uint4*)(smem + buf * STAGE_ELEMS + A_ELEMS + (sgi / (TN / 8)) * LDD + (sgi % (TN / 8)) * 8) = rd[u]; }
    };

    if (t_begin < t_end) {
      load_src_next(t_begin);
      issue_loads(t_begin);
      store_lds(0);
      __syncthreads();
      for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) issue_loads(t + 1);
        const u16* A = smem + buf * STAGE_ELEMS;
        const u16* D = A + A_ELEMS;
  #pragma unroll
        for (int ks = 0; ks < RK / 32; ++ks) {
          bf16x8 bfr[WN];
  #pragma unroll
          for (int b = 0; b < WN; ++b) bfr[b] = tr_frag(D, LDD, ks * 32, (wn * WN + b) * 16, lane);
  #pragma unroll
          for (int a = 0; a < WM; ++a) {
            bf16x8 af = tr_frag(A, LDA, ks * 32, (wm * WM + a) * 16, lane);
  #pragma unroll
            for (int b = 0; b < WN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[b], acc[a][b], 0, 0, 0);
          }
          if (ks == 0 && t + 1 < t_end) store_lds(buf ^ 1);     // under the second k-step's MFMAs (see k_igemm_fwd)
        }
        __syncthreads();
      }
    }
  }
  float* p = partial + ((long long)split * kvol + kap) * cin * cout;
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = ci0 + (wm * WM + a) * 16 + g * 4 + r;
        int co = co0 + (wn * WN + b) * 16 + li;
        if (ci < cin && co < cout) p[(long long)ci * cout + co] = acc[a][b][r];
      }
}

// Workgroup (blockIdx.x, blockIdx.y) -> the (row split, offset) it works on.  The `kvol` workgroups of one row split read the
// same `dout` rows and (offset-shifted) the same input rows, at about the same time: on ONE XCD they are fetched into that L2 once
// and hit by the other offsets; dealt round-robin over the XCDs (hardware order: linear id % 8) every L2 streams the whole of both
// tensors.  XCD x takes the splits x, x + 8, ... of the first 8 * floor(nsplit / 8); the workgroups of the remaining splits are
// dealt round-robin (they keep every CU busy: 27 offsets x 9 splits = 243 workgroups, 8 of the 9 splits L2-local).
#ifndef WGRAD_XCD_SPLITS
#define WGRAD_XCD_SPLITS 1
#endif
__device__ __forceinline__ void wgrad_xcd_remap(int& split, int& kap, int nsplit, int kvol) {
#if WGRAD_XCD_SPLITS
  if (nsplit >= 8 && gridDim.z == 1) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
    const int q8 = nsplit >> 3, aligned = q8 * kvol;      // slots of the XCD-local part
    if (slot < aligned) {
      split = xcd + 8 * (slot / kvol);
      kap = slot % kvol;
    } else {
      const int rem = (slot - aligned) * 8 + xcd;
      split = 8 * q8 + rem / kvol;
      kap = rem % kvol;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// weight gradient with LDS-DMA staging: both stage tiles ([64 rows][TM] gathered input rows, [64 rows][TN] dout rows) go global ->
// LDS with `buffer_load_dwordx4 ... lds`, lane-linear, unpadded.  Transpose reads of an unpadded tile would put the 8 rows of a
// 32-lane group on the same banks (row stride = multiple of 256 B), so 16-column tile T of row r is stored at tile position
// T ^ (r & 7) (source-side swizzle; r & 7 is a per-lane constant of the reading lane: its rows are k0 + 4g + j (+16)).
// Requires cin % TM == 0 and cout % TN == 0 (dispatch), TM, TN in {64, 128, 256}.
// ---------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int WM, int WN>
__device__ __forceinline__ void igemm_wgrad_glds_body(const u16* __restrict__ in, const u16* __restrict__ dout,
                                                      const int* __restrict__ nbr, int ld, float* __restrict__ partial,
                                                      const int* __restrict__ n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                      int co_blocks, int kap_override = -1) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int TM = WAVES_M * WM * 16, TN = WAVES_N * WN * 16, RK = 64;
  constexpr int A_ELEMS = RK * TM, D_ELEMS = RK * TN, STAGE_ELEMS = A_ELEMS + D_ELEMS;
  constexpr int A_LPR = TM / 8, D_LPR = TN / 8;                  // lanes (16-byte slots) per row
  constexpr int A_RPI = 64 / A_LPR, D_RPI = 64 / D_LPR;          // rows per wave-instruction (1 KiB)
  constexpr int A_SEGS = RK / A_RPI / NW, D_SEGS = RK / D_RPI / NW;
  constexpr int A_YMASK = (TM / 16 >= 8) ? 7 : (TM / 16 - 1), D_YMASK = (TN / 16 >= 8) ? 7 : (TN / 16 - 1);
  static_assert(RK % (A_RPI * NW) == 0 && RK % (D_RPI * NW) == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];

  const int n_out = min(*n_out_dev, n_out_cap);
  const int nsplit = gridDim.x;
  int split = blockIdx.x, kap = kap_override >= 0 ? kap_override : (int)blockIdx.y;
  if (kap_override < 0) wgrad_xcd_remap(split, kap, nsplit, kvol);
  const int ci0 = (blockIdx.z / co_blocks) * TM, co0 = (blockIdx.z % co_blocks) * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (n_out + RK - 1) / RK;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, -1, 0x00020000);
  const unsigned in_row_bytes = (unsigned)cin * 2u, d_row_bytes = (unsigned)cout * 2u;
  // loader role
  int a_row[A_SEGS], d_row[D_SEGS];
  unsigned a_col[A_SEGS], d_col[D_SEGS];
#pragma unroll
  for (int u = 0; u < A_SEGS; ++u) {
    const int r = (wv * A_SEGS + u) * A_RPI + lane / A_LPR, slot = lane % A_LPR;
    const int chunk = slot ^ (((r & 7) & A_YMASK) << 1);         // 16-column tile T = chunk >> 1 is XORed with r & 7
    a_row[u] = r;
    a_col[u] = (unsigned)(ci0 + chunk * 8) * 2u;
  }
#pragma unroll
  for (int u = 0; u < D_SEGS; ++u) {
    const int r = (wv * D_SEGS + u) * D_RPI + lane / D_LPR, slot = lane % D_LPR;
    const int chunk = slot ^ (((r & 7) & D_YMASK) << 1);
    d_row[u] = r;
    d_col[u] = (unsigned)(co0 + chunk * 8) * 2u;
  }
  int src_nxt[A_SEGS];
  auto load_src_next = [&](int t) {
    const int r0 = t * RK;
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) {
      int m = r0 + a_row[u];
      int mc = m < n_out ? m : n_out - 1;
      src_nxt[u] = nbr ? nbr[(long long)kap * ld + mc] : mc;
    }
  };
  auto issue = [&](int t, int buf) {
    const int r0 = t * RK;
    const bool live = t < t_end;
    u16* Ab = smem + buf * STAGE_ELEMS + wv * (A_SEGS * 512);
    u16* Db = smem + buf * STAGE_ELEMS + A_ELEMS + wv * (D_SEGS * 512);
#pragma unroll
    for (int u = 0; u < A_SEGS; ++u) {
      const bool ok = live && (r0 + a_row[u] < n_out) && src_nxt[u] >= 0;
      unsigned voff = ok ? (unsigned)src_nxt[u] * in_row_bytes + a_col[u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(Ab + u * 512), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < D_SEGS; ++u) {
      const int m = r0 + d_row[u];
      unsigned voff = (live && m < n_out) ? (unsigned)m * d_row_bytes + d_col[u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (lds_void_ptr)(Db + u * 512), 16, voff, 0, 0, 0);
    }
    load_src_next(t + 1);
  };
  // one LDS-DMA instruction of stage t (q < A_SEGS: gathered input rows, else gradient rows), dealt out behind MFMA groups (see
  // GLDS_DMA_SPREAD in the forward kernel: a stage's loads queued in front of its MFMAs hold the waves at the address unit)
  auto issue_one = [&](int t, int buf, int q) {
    const int r0 = t * RK;
    const bool live = t < t_end;
    if (q < A_SEGS) {
      u16* Ab = smem + buf * STAGE_ELEMS + wv * (A_SEGS * 512);
      const bool ok = live && (r0 + a_row[q] < n_out) && src_nxt[q] >= 0;
      unsigned voff = ok ? (unsigned)src_nxt[q] * in_row_bytes + a_col[q] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(Ab + q * 512), 16, voff, 0, 0, 0);
    } else {
      const int u = q - A_SEGS;
      u16* Db = smem + buf * STAGE_ELEMS + A_ELEMS + wv * (D_SEGS * 512);
      const int m = r0 + d_row[u];
      unsigned voff = (live && m < n_out) ? (unsigned)m * d_row_bytes + d_col[u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (lds_void_ptr)(Db + u * 512), 16, voff, 0, 0, 0);
    }
  };
  // reader role: transpose-read fragments; this lane's rows are k0 + 4g + j (+16): y = (4g + j) & 7
  const int g = lane >> 4, L = lane & 15, j = L >> 2, q = L & 3;
  const int ya = ((4 * g + j) & 7) & A_YMASK, yd = ((4 * g + j) & 7) & D_YMASK;
  auto trf = [&](const u16* tile, int stride, int k0, int T, int y) {
    const u16* p0 = tile + (k0 + 4 * g + j) * stride + ((T ^ y) << 4) + 4 * q;
    s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0));
    s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0 + 16 * stride));
    s16x8 v = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  if (t_begin < t_end) {
    load_src_next(t_begin);
    issue(t_begin, 0);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
      const int buf = (t - t_begin) & 1;
      // spreading pays on the 256 x 256 tile only (measured per tile size: +4.5 % there, a loss at step level when applied to all)
      constexpr bool SPREAD = WGRAD_DMA_SPREAD && TM >= 256 && TN >= 256;
      int src_nn[A_SEGS];                               // gather indices of stage t+2: requested now, moved into src_nxt at the end of the stage
      if constexpr (!SPREAD) {
        issue(t + 1, buf ^ 1);                          // past the last stage: all offsets out of range -> zeros into the idle buffer
      } else {
        const int r0n = (t + 2) * RK;
#pragma unroll
        for (int u = 0; u < A_SEGS; ++u) {
          int m = r0n + a_row[u];
          int mc = m < n_out ? m : n_out - 1;
          src_nn[u] = nbr ? nbr[(long long)kap * ld + mc] : mc;
        }
      }
      const u16* A = smem + buf * STAGE_ELEMS;
      const u16* D = A + A_ELEMS;
#pragma unroll
      for (int ks = 0; ks < RK / 32; ++ks) {
        bf16x8 bfr[WN];
#pragma unroll
        for (int b = 0; b < WN; ++b) bfr[b] = trf(D, TN, ks * 32, wn * WN + b, yd);
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          bf16x8 af = trf(A, TM, ks * 32, wm * WM + a, ya);
#pragma unroll
          for (int b = 0; b < WN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af, acc[a][b], 0, 0, 0);   // transposed block
          if constexpr (SPREAD) if (ks == 0) {            // the next stage's loads behind the MFMA groups of the first k-step
            constexpr int NQ = A_SEGS + D_SEGS, PER = (NQ + WM - 1) / WM;
#pragma unroll
            for (int jq = 0; jq < PER; ++jq)
              if (a * PER + jq < NQ) issue_one(t + 1, buf ^ 1, a * PER + jq);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (SPREAD) {
#pragma unroll
        for (int u = 0; u < A_SEGS; ++u) src_nxt[u] = src_nn[u];
      }
      __syncthreads();
    }
  }
  // acc[a][b][r] = dW[ci (wm*WM+a)*16 + li][co (wn*WN+b)*16 + 4g + r]: one 16-byte store per block
  float* p = partial + ((long long)split * kvol + kap) * cin * cout;
  const int li = lane & 15;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int ci = ci0 + (wm * WM + a) * 16 + li;
      const int co = co0 + (wn * WN + b) * 16 + 4 * g;
      *(f32x4*)(p + (long long)ci * cout + co) = acc[a][b];
    }
}
// ---------------------------------------------------------------------------------------------
// 256 x 256 weight-gradient tile on the EIGHT-PHASE schedule of igemm_glds8_body (same segments, counts and barriers; read that
// comment first).  What differs: the reduction index is the output ROW (64 per k-tile), both operands are [64 rows][256 channels]
// tiles read with transpose reads, and a k-tile's four 16 KiB pieces are COLUMN ranges: A0 / A1 = the first / second 64 input
// channels of both wave rows (128 columns, 256 B per row), D0 / D1 = the first / second 32 output channels of all four wave
// columns.  One LDS-DMA instruction = 4 rows x 256 B; 16-column tile T of piece row r sits at position T ^ (r & 7).  The two A
// pieces gather the SAME 64 rows: two index registers per lane.  Needs a neighbour table (the batched linear-layer form stays on
// igemm_wgrad_glds_body).
// ---------------------------------------------------------------------------------------------
#ifndef IGEMM_WGRAD_GLDS8
#define IGEMM_WGRAD_GLDS8 1
#endif
__device__ __forceinline__ void igemm_wgrad_glds8_body(const u16* __restrict__ in, const u16* __restrict__ dout,
                                                       const int* __restrict__ nbr, int ld, float* __restrict__ partial,
                                                       const int* __restrict__ n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                       int co_blocks) {
  constexpr int WAVES_N = 4, WM = 8, WN = 4, RK = 64;
  constexpr int PC = 128;                                  // columns of a piece
  constexpr int PIECE = RK * PC, STAGE_ELEMS = 4 * PIECE;  // A0 | A1 | D0 | D1
  extern __shared__ __attribute__((aligned(16))) u16 smem[];

  const int n_out = min(*n_out_dev, n_out_cap);
  const int nsplit = gridDim.x;
  int split = blockIdx.x, kap = (int)blockIdx.y;
  wgrad_xcd_remap(split, kap, nsplit, kvol);
  const int ci0 = (blockIdx.z / co_blocks) * 256, co0 = (blockIdx.z % co_blocks) * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (n_out + RK - 1) / RK;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t_begin = split * per, t_end = min(ntiles, t_begin + per);
  const int nstage = t_end - t_begin;

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, -1, 0x00020000);
  const unsigned in_row_bytes = (unsigned)cin * 2u, d_row_bytes = (unsigned)cout * 2u;
  // loader role: instruction u (0 / 1) of this wave fills piece rows (wv*2+u)*4 .. +3; lane = (row in group, 16-byte slot of 16)
  const int lrow = lane >> 4, lslot = lane & 15;
  int prow[2];
  unsigned a_col[2][2], d_col[2][2];                       // [piece][u]: byte offset of this lane's 16 bytes inside a source row
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (wv * 2 + u) * 4 + lrow;
    prow[u] = r;
    const int chunk = lslot ^ ((r & 7) << 1);              // 8-column chunk of the piece this slot receives (tile T = chunk >> 1 swizzled)
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const int ca = (chunk < 8) ? sp * 64 + chunk * 8 : 128 + sp * 64 + (chunk - 8) * 8;          // piece column -> input channel
      a_col[sp][u] = (unsigned)(ci0 + ca) * 2u;
      const int pc = chunk * 8;                                                                     // piece column 0..127
      const int cd = (pc >> 5) * 64 + sp * 32 + (pc & 31);                                          // -> output channel
      d_col[sp][u] = (unsigned)(co0 + cd) * 2u;
    }
  }
  const int* nrow = nbr + (long long)kap * ld;
  int idx_cur[2], idx_nxt[2];
  auto load_idx_next = [&](int t) {                        // t: row tile (clamped by the caller)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = t * RK + prow[u];
      idx_nxt[u] = nrow[m < n_out ? m : n_out - 1];
    }
  };
  auto advance_idx = [&](int t) {                          // indices of row tile t; -1 = zero row (past the end / missing neighbour)
#pragma unroll
    for (int u = 0; u < 2; ++u) idx_cur[u] = (t < t_end && t * RK + prow[u] < n_out) ? idx_nxt[u] : -1;
  };
#ifndef W8_EXP
#define W8_EXP 0   /* timing experiments only (wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop, 4 no wave-row offset */
#endif
  bool in_loop = false;
  auto issue_a = [&](int buf, int sp) {                    // piece A_sp of the row tile idx_cur describes
    if ((W8_EXP & 1) && in_loop) return;
    u16* dst = smem + buf * STAGE_ELEMS + sp * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned voff = idx_cur[u] >= 0 ? (unsigned)idx_cur[u] * in_row_bytes + a_col[sp][u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(dst + u * 512), 16, voff, 0, 0, 0);
    }
  };
  auto issue_d = [&](int t, int buf, int sp) {
    if ((W8_EXP & 1) && in_loop) return;
    u16* dst = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = t * RK + prow[u];
      const unsigned voff = (t < t_end && m < n_out) ? (unsigned)m * d_row_bytes + d_col[sp][u] : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (lds_void_ptr)(dst + u * 512), 16, voff, 0, 0, 0);
    }
  };
  // reader role: transpose-read fragments; this lane's rows are k0 + 4g + j (+16): y = (4g + j) & 7
  const int g = lane >> 4, L = lane & 15, j = L >> 2, q = L & 3;
  const int y = (4 * g + j) & 7;
  auto trf = [&](const u16* piece, int k0, int T) {
    const u16* p0 = piece + (k0 + 4 * g + j) * PC + ((T ^ y) << 4) + 4 * q;
    s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0));
    s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(p0 + 16 * PC));
    s16x8 v = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  bf16x8 af[2][4][2], df[2][2][2];                         // A: [half][ci block][k-step]; D: [half][co block][k-step]
#define W8_READ_A(BUF, SP)                                                                       \
  if (!((W8_EXP & 2) && in_loop)) {                                                              \
    const u16* A_ = smem + (BUF) * STAGE_ELEMS + (SP) * PIECE;                                   \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                              \
      af[SP][a][0] = trf(A_, 0, wm * 4 + a);                                                     \
      af[SP][a][1] = trf(A_, 32, wm * 4 + a);                                                    \
    }                                                                                            \
  }
#define W8_READ_D(BUF, SP)                                                                       \
  if (!((W8_EXP & 2) && in_loop)) {                                                              \
    const u16* D_ = smem + (BUF) * STAGE_ELEMS + (2 + (SP)) * PIECE;                             \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                              \
      df[SP][b][0] = trf(D_, 0, wn * 2 + b);                                                     \
      df[SP][b][1] = trf(D_, 32, wn * 2 + b);                                                    \
    }                                                                                            \
  }
#define W8_MMA(SA, SB)                                                                           \
  {                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                            \
          acc[(SA) * 4 + a][(SB) * 2 + b] =                                                      \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[SB][b][ks], af[SA][a][ks], acc[(SA) * 4 + a][(SB) * 2 + b], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                               \
  }
#define W8_BAR()                                  \
  {                                               \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_barrier();                 \
    __builtin_amdgcn_sched_barrier(0);            \
  }
  if (nstage > 0) {
    // prologue: row tile t_begin complete in buffer 0, its first A half in registers, indices of t_begin + 1 current
    load_idx_next(t_begin);
    advance_idx(t_begin);
    issue_a(0, 0); issue_d(t_begin, 0, 0); issue_d(t_begin, 0, 1); issue_a(0, 1);
    load_idx_next(t_begin + 1);
    advance_idx(t_begin + 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    W8_BAR();
    W8_READ_A(0, 0)
    if (!(W8_EXP & 4) && wm == 1) W8_BAR();                // the second wave row runs one barrier behind the first
#if W8_EXP & 2
    W8_READ_A(0, 1) W8_READ_D(0, 0) W8_READ_D(0, 1)
#endif
    in_loop = true;
    for (int st = 0; st < nstage; ++st) {
      const int buf = st & 1, t = t_begin + st;
      // ---- phase 1: (ci 0-63, co 0-31); requests: indices of t + 2, piece A0 of t + 1
      load_idx_next(t + 2);
      __builtin_amdgcn_sched_barrier(0);
      W8_READ_D(buf, 0)
      issue_a(buf ^ 1, 0);
      __builtin_amdgcn_s_waitcnt(0x0F76);                  // vmcnt(6) = A1 + 2 indices + A0': D1 of this row tile has landed
      W8_BAR();
      W8_MMA(0, 0)
      W8_BAR();
      // ---- phase 2: (ci 0-63, co 32-63)
      W8_READ_D(buf, 1)
      issue_d(t + 1, buf ^ 1, 0);
      __builtin_amdgcn_s_waitcnt(0x0F76);                  // vmcnt(6) = 2 indices + A0' + D0': A1 of this row tile
      W8_BAR();
      W8_MMA(0, 1)
      W8_BAR();
      // ---- phase 3: (ci 64-127, co 32-63)
      W8_READ_A(buf, 1)
      issue_d(t + 1, buf ^ 1, 1);
      __builtin_amdgcn_s_waitcnt(0x0F74);                  // vmcnt(4): the indices and A0 of the next row tile
      W8_BAR();
      W8_MMA(1, 1)
      W8_BAR();
      // ---- phase 4: (ci 64-127, co 0-31): D0 is still in registers; the next row tile's first A half is read here
      W8_READ_A(buf ^ 1, 0)
      issue_a(buf ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      advance_idx(t + 2);
      __builtin_amdgcn_s_waitcnt(0x0F74);                  // vmcnt(4): D0 of the next row tile
      W8_BAR();
      W8_MMA(1, 0)
      W8_BAR();
    }
    if (!(W8_EXP & 4) && wm == 0) W8_BAR();
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // the tail's zero-fill requests
  }
#undef W8_READ_A
#undef W8_READ_D
#undef W8_MMA
#undef W8_BAR
  float* p = partial + ((long long)split * kvol + kap) * cin * cout;
  const int li = lane & 15;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int ci = ci0 + (wm * WM + a) * 16 + li;
      const int co = co0 + (wn * WN + b) * 16 + 4 * g;
      *(f32x4*)(p + (long long)ci * cout + co) = acc[a][b];
    }
}
__global__ __launch_bounds__(512) void k_igemm_wgrad_glds8_256(const u16* in, const u16* dout, const int* nbr, int ld, float* partial,
                                                               const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol,
                                                               int co_blocks) {
  igemm_wgrad_glds8_body(in, dout, nbr, ld, partial, n_out_dev, n_out_cap, cin, cout, kvol, co_blocks);
}

#define U3D_WGRAD_GLDS_KERNEL(NAME, A, B, C, D)                                                                                   \
  __global__ __launch_bounds__(A* B * 64) void NAME(const u16* in, const u16* dout, const int* nbr, int ld, float* partial,        \
                                                    const int* n_out_dev, int n_out_cap, int cin, int cout, int kvol, int co_blocks) { \
    igemm_wgrad_glds_body<A, B, C, D>(in, dout, nbr, ld, partial, n_out_dev, n_out_cap, cin, cout, kvol, co_blocks);               \
  }
U3D_WGRAD_GLDS_KERNEL(k_igemm_wgrad_glds_256, 2, 4, 8, 4)
U3D_WGRAD_GLDS_KERNEL(k_igemm_wgrad_glds_128, 2, 2, 4, 4)
U3D_WGRAD_GLDS_KERNEL(k_igemm_wgrad_glds_64, 2, 2, 2, 2)
// (32- and 16-channel tiles were tried on this kernel too: correct, but no faster than k_igemm_wgrad - those layers are bound by
//  the L2 gather, not by staging - so they stay on the buffer-load kernel)
#undef U3D_WGRAD_GLDS_KERNEL
// ---- batched form for the decoder / head linears: `count` independent products dW_b = in_b^T @ dout_b of ONE shape in one launch
//      (grid.y = batch index; pointers arrive by value in the kernel arguments - no device-side table, capturable as is)
#define U3D_WGRAD_BATCH_MAX 48
struct WgradBatch {
  const u16* in[U3D_WGRAD_BATCH_MAX];
  const u16* dout[U3D_WGRAD_BATCH_MAX];
  float* dw[U3D_WGRAD_BATCH_MAX];
};
#define U3D_WGRAD_BATCH_KERNEL(NAME, A, B, C, D)                                                                                    \
  __global__ __launch_bounds__(A* B * 64) void NAME(WgradBatch bt, float* partial, const int* n_dev, int n_cap, int cin, int cout,   \
                                                    int co_blocks, long long partial_stride) {                                      \
    const int b = blockIdx.y;                                                                                                        \
    igemm_wgrad_glds_body<A, B, C, D>(bt.in[b], bt.dout[b], nullptr, 0, partial + (long long)b * partial_stride, n_dev, n_cap, cin,   \
                                      cout, 1, co_blocks, 0);                                                                         \
  }
U3D_WGRAD_BATCH_KERNEL(k_wgrad_batch_256, 2, 4, 8, 4)
U3D_WGRAD_BATCH_KERNEL(k_wgrad_batch_128, 2, 2, 4, 4)
U3D_WGRAD_BATCH_KERNEL(k_wgrad_batch_64, 2, 2, 2, 2)
#undef U3D_WGRAD_BATCH_KERNEL
// sum of the `nsplit` partials of one f32x4 in a FIXED order, as four independent chains: eight loads in flight per pass instead of
// one (the reductions were latency chains: 20 us for 16 MB of partials)
__device__ __forceinline__ f32x4 wgrad_sum_splits(const float* __restrict__ p, long long stride, int nsplit) {
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a, c = a, d = a;
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {
    const f32x4 v0 = *(const f32x4*)(p + (long long)k * stride), v1 = *(const f32x4*)(p + (long long)(k + 1) * stride);
    const f32x4 v2 = *(const f32x4*)(p + (long long)(k + 2) * stride), v3 = *(const f32x4*)(p + (long long)(k + 3) * stride);
    const f32x4 v4 = *(const f32x4*)(p + (long long)(k + 4) * stride), v5 = *(const f32x4*)(p + (long long)(k + 5) * stride);
    const f32x4 v6 = *(const f32x4*)(p + (long long)(k + 6) * stride), v7 = *(const f32x4*)(p + (long long)(k + 7) * stride);
    a += v0; b += v1; c += v2; d += v3; a += v4; b += v5; c += v6; d += v7;
  }
  for (; k < nsplit; ++k) a += *(const f32x4*)(p + (long long)k * stride);
  return (a + b) + (c + d);
}

// Products that name the SAME output in consecutive batch slots (a weight shared by several decoder layers: dW = sum over its uses)
// are summed here: the group's first slot reduces the nsplit partials of all its members (contiguous in the workspace), the others
// have no output (mult 0).  One fixed order, no separate accumulate launches.
struct WgradGroups { unsigned char mult[U3D_WGRAD_BATCH_MAX]; };
__global__ void k_wgrad_batch_reduce(WgradBatch bt, WgradGroups gr, const float* __restrict__ partial, long long n, int nsplit,
                                     long long partial_stride) {
  const int b = blockIdx.y;
  const int g = gr.mult[b];
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n || g == 0) return;
  *(f32x4*)(bt.dw[b] + i) = wgrad_sum_splits(partial + (long long)b * partial_stride + i, n, nsplit * g);
}

typedef void (*wgrad_glds_kernel_t)(const u16*, const u16*, const int*, int, float*, const int*, int, int, int, int, int);

__global__ void k_igemm_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, long long n, int nsplit) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  *(f32x4*)(dw + i) = wgrad_sum_splits(partial + i, n, nsplit);
}

#ifndef IGEMM_WGRAD_MIN_STAGES_K1
#define IGEMM_WGRAD_MIN_STAGES_K1 8   /* 4 and 2 measured slower end-to-end (33.4 / 33.9 vs 33.3 ms per step) */
#endif
// same reduction, result written as [Cout][Cin][K] (nn.Conv3d's checkpoint layout): the gradient lands in the parameter's own
// layout and autograd keeps it as is (a permuted view would be cloned into a contiguous tensor by AccumulateGrad: one more launch)
__global__ __launch_bounds__(256) void k_igemm_wgrad_reduce_oik(const float* __restrict__ partial, float* __restrict__ dw, long long n,
                                                                int nsplit, int kvol, int cin, int cout) {
  // workgroup = (ci, 64 output channels, a third of the offsets): reads run along co (coalesced; splits summed in a fixed order), the
  // [co][k] tile is turned in LDS, writes run along k (the innermost dimension of [Cout][Cin][K])
  __shared__ float tile[64][10];
  const int ci = blockIdx.x, co0 = blockIdx.y * 64;
  const int kper = (kvol + gridDim.z - 1) / gridDim.z, k0 = blockIdx.z * kper, k1 = min(kvol, k0 + kper);
  const int t = threadIdx.x, col = t & 63;
  for (int k = k0 + (t >> 6); k < k1; k += 4) {
    float s = 0.f;
    if (co0 + col < cout) {
      const float* p = partial + ((long long)k * cin + ci) * cout + co0 + col;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      int sp = 0;
      for (; sp + 8 <= nsplit; sp += 8) {            // eight independent loads in flight, fixed summation order
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p[(long long)(sp + j) * n];
      }
      for (; sp < nsplit; ++sp) a[0] += p[(long long)sp * n];
      s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    tile[col][k - k0] = s;
  }
  __syncthreads();
  const int nk = k1 - k0;
  for (int idx = t; idx < 64 * nk; idx += 256) {
    const int cl = idx / nk, k = idx % nk;
    if (co0 + cl < cout) dw[((long long)(co0 + cl) * cin + ci) * kvol + k0 + k] = tile[cl][k];
  }
}

struct WgPlan { int tile; int ci_blocks, co_blocks, nsplit; };
static WgPlan wgrad_plan_tile(int tile, int n_out_cap, int cin, int cout, int kvol) {
  WgPlan p;
  p.tile = tile;
  p.ci_blocks = u3d_cdiv(cin, p.tile);
  p.co_blocks = u3d_cdiv(cout, p.tile);
  int ntiles = u3d_cdiv(n_out_cap > 0 ? n_out_cap : 1, 64);
  int wgs_per_split = kvol * p.ci_blocks * p.co_blocks;
  int target = (p.tile == 256 ? 256 : (p.tile >= 64 ? 512 : 2048)) / wgs_per_split;
  if (target < 1) target = 1;
  int ns = ntiles < target ? ntiles : target;
  // keep at least 8 stages per split so the prologue is amortised (4 for the single-offset products of the decoder / head linears:
  // ~113 stages in all, latency-bound - more, shorter workgroups finish sooner)
  const int min_stages = (kvol == 1 && ntiles <= 256) ? IGEMM_WGRAD_MIN_STAGES_K1 : 8;
  while (ns > 1 && ntiles / ns < min_stages) --ns;
  p.nsplit = ns < 1 ? 1 : ns;
  return p;
}
static WgPlan wgrad_plan(int n_out_cap, int cin, int cout, int kvol) {
  int mn = cin < cout ? cin : cout;
  int tile;
  if (mn >= 256 && cin % 256 == 0 && cout % 256 == 0) tile = 256;
  else if (mn >= 128 && cin % 128 == 0 && cout % 128 == 0) tile = 128;
  else if (cin % 64 == 0 && cout % 64 == 0) tile = 64;
  else if (cin % 32 == 0 && cout % 32 == 0) tile = 32;       // sparse levels with 32 channels: load/latency bound
  else tile = 16;
  WgPlan p = wgrad_plan_tile(tile, n_out_cap, cin, cout, kvol);
  // few rows (the decoder / head linears: kvol 1, <= 10^4 rows): the row split cannot fill 256 CUs with big tiles -> smaller tiles
  while (p.tile > 64 && (long long)p.nsplit * kvol * p.ci_blocks * p.co_blocks < 192) p = wgrad_plan_tile(p.tile / 2, n_out_cap, cin, cout, kvol);
  return p;
}

#ifndef IGEMM_WGRAD_NARROW
#define IGEMM_WGRAD_NARROW 1   /* 16/32-channel 27-offset weight gradients on wgrad_narrow.hip (0: the tiled kernels below) */
#endif
bool u3d_wgrad_narrow_shape(int cin, int cout, int kvol);
int64_t u3d_wgrad_narrow_workspace(int n_out_cap, int cin, int cout);
int u3d_launch_wgrad_narrow(const void* in, const void* dout, const int32_t* nbr, int ld, float* dw, const int32_t* n_out_dev, int n_out_cap,
                            int cin, int cout, int kvol, int out_oik, void* workspace, int64_t workspace_bytes, hipStream_t s);

extern "C" int64_t u3d_igemm_wgrad_bf16_workspace(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol) {
#if IGEMM_WGRAD_NARROW
  if (u3d_wgrad_narrow_shape(cin, cout, kvol)) return u3d_wgrad_narrow_workspace(n_out_cap, cin, cout);
#endif
  if (convin_shape(cin, cout, kvol)) return (int64_t)u3d_cdiv(n_out_cap > 0 ? n_out_cap : 1, CONVIN_WG_ROWS) * kvol * cin * cout * 4;
  WgPlan p = wgrad_plan(n_out_cap, cin, cout, kvol);
  return (int64_t)p.nsplit * kvol * cin * cout * 4;
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int launch_igemm_wgrad(const void* in, const void* dout, const int32_t* nbr, int ld, float* partial, const int32_t* n_out_dev,
                              int n_out_cap, int cin, int cout, int kvol, const WgPlan& p, hipStream_t s) {
  constexpr int TM = WAVES_M * WM * 16, TN = WAVES_N * WN * 16, RK = 64;
  constexpr size_t lds = 2 * (size_t)(RK * (TM + 16) + RK * (TN + 16)) * 2;
  auto kern = k_igemm_wgrad<WAVES_M, WAVES_N, WM, WN>;
  if (lds > 64 * 1024) U3D_ALLOW_LDS(kern, lds);      // one call site per template instantiation: per-kernel, per-device
  dim3 grid(p.nsplit, kvol, p.ci_blocks * p.co_blocks);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, (const u16*)in, (const u16*)dout, nbr, ld, partial, n_out_dev,
                     n_out_cap, cin, cout, kvol, p.co_blocks);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

#ifndef IGEMM_WGRAD_GLDS
#define IGEMM_WGRAD_GLDS 1
#endif
#ifndef IGEMM_WGRAD_GLDS_MIN_TILE
#define IGEMM_WGRAD_GLDS_MIN_TILE 64
#endif
static int launch_igemm_wgrad_glds(int tile, const void* in, const void* dout, const int32_t* nbr, int ld, float* partial,
                                   const int32_t* n_out_dev, int n_out_cap, int cin, int cout, int kvol, const WgPlan& p, hipStream_t s) {
  wgrad_glds_kernel_t kern = tile == 256 ? ((IGEMM_WGRAD_GLDS8 && nbr) ? k_igemm_wgrad_glds8_256 : k_igemm_wgrad_glds_256)
                                         : (tile == 128 ? k_igemm_wgrad_glds_128 : k_igemm_wgrad_glds_64);
  const int nthreads = tile == 256 ? 512 : 256;
  const size_t lds = 2 * (size_t)(64 * tile + 64 * tile) * 2;
  if (lds > 64 * 1024) {                       // one per-device mask per kernel
    if (tile == 256 && IGEMM_WGRAD_GLDS8 && nbr) U3D_ALLOW_LDS(k_igemm_wgrad_glds8_256, lds);
    else if (tile == 256) U3D_ALLOW_LDS(k_igemm_wgrad_glds_256, lds);
    else if (tile == 128) U3D_ALLOW_LDS(k_igemm_wgrad_glds_128, lds);
    else U3D_ALLOW_LDS(k_igemm_wgrad_glds_64, lds);
  }
  dim3 grid(p.nsplit, kvol, p.ci_blocks * p.co_blocks);
  hipLaunchKernelGGL(kern, grid, dim3(nthreads), lds, s, (const u16*)in, (const u16*)dout, nbr, ld, partial, n_out_dev, n_out_cap, cin, cout,
                     kvol, p.co_blocks);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

extern "C" int32_t u3d_igemm_wgrad_bf16(const void* in, const void* dout, const int32_t* nbr, int32_t ld, float* dw,
                                        const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                        int32_t out_layout, void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(in && dout && dw && n_out_dev && workspace && (nbr || kvol == 1), U3D_ERR_ARG);
  if (convin_shape(cin, cout, kvol) && out_layout == 0) {         // the encoder's input convolution
    const int nb = u3d_cdiv(n_out_cap > 0 ? n_out_cap : 1, CONVIN_WG_ROWS);
    const long long nw = (long long)kvol * cin * cout;
    U3D_REQUIRE(workspace_bytes >= (int64_t)nb * nw * 4, U3D_ERR_WORKSPACE);
    hipLaunchKernelGGL(k_conv_in_wgrad, dim3(nb), dim3(256), 0, s, (const u16*)in, (const u16*)dout, nbr, ld, (float*)workspace, n_out_dev,
                       n_out_cap, kvol);
    hipLaunchKernelGGL(k_conv_in_reduce, dim3(u3d_cdiv((int)nw, 64)), dim3(1024), 0, s, (const float*)workspace, dw, (int)nw, nb);
    U3D_CHECK_LAUNCH();
    return U3D_OK;
  }
  if (cin % 16 != 0 || cout % 16 != 0) return U3D_ERR_UNSUPPORTED;
#if IGEMM_WGRAD_NARROW
  if (nbr && u3d_wgrad_narrow_shape(cin, cout, kvol)) {
    if (n_out_cap <= 0) { hipMemsetAsync(dw, 0, sizeof(float) * kvol * cin * cout, s); return U3D_OK; }
    return u3d_launch_wgrad_narrow(in, dout, nbr, ld, dw, n_out_dev, n_out_cap, cin, cout, kvol, out_layout, workspace, workspace_bytes, s);
  }
#endif
  WgPlan p = wgrad_plan(n_out_cap, cin, cout, kvol);
  long long n = (long long)kvol * cin * cout;
  U3D_REQUIRE(workspace_bytes >= (int64_t)p.nsplit * n * 4, U3D_ERR_WORKSPACE);
  int rc;
#if IGEMM_WGRAD_GLDS
  if (p.tile >= IGEMM_WGRAD_GLDS_MIN_TILE && cin % p.tile == 0 && cout % p.tile == 0)
    rc = launch_igemm_wgrad_glds(p.tile, in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  else
#endif
  if (p.tile == 256) rc = launch_igemm_wgrad<2, 4, 8, 4>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  else if (p.tile == 128) rc = launch_igemm_wgrad<2, 2, 4, 4>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  else if (p.tile == 64) rc = launch_igemm_wgrad<2, 2, 2, 2>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  else if (p.tile == 32) rc = launch_igemm_wgrad<2, 2, 1, 1>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  else rc = launch_igemm_wgrad<1, 1, 1, 1>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, p, s);
  if (rc != U3D_OK) return rc;
  if (out_layout == 1 && kvol > 27) return U3D_ERR_UNSUPPORTED;
  if (out_layout == 1)
    hipLaunchKernelGGL(k_igemm_wgrad_reduce_oik, dim3(cin, u3d_cdiv(cout, 64), kvol > 9 ? 3 : 1), dim3(256), 0, s, (const float*)workspace, dw, n, p.nsplit, kvol, cin, cout);
  else
    hipLaunchKernelGGL(k_igemm_wgrad_reduce, dim3(u3d_cdiv(n / 4, 256)), dim3(256), 0, s, (const float*)workspace, dw, n, p.nsplit);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// dW_b = in_b^T @ dout_b for b < count, all [n_rows, cin] x [n_rows, cout] bf16 -> f32 [cin, cout] (deterministic row split + ordered
// reduce, as u3d_igemm_wgrad_bf16 with kvol = 1).  Two launches for the whole batch.
// plan for `count` same-shape products in one launch: the batch itself fills the chip, so prefer big tiles and few, long row splits
static WgPlan wgrad_plan_batched(int count, int n_rows, int cin, int cout) {
  int mn = cin < cout ? cin : cout;
  int tile = (mn >= 256 && cin % 256 == 0 && cout % 256 == 0) ? 256 : ((mn >= 128 && cin % 128 == 0 && cout % 128 == 0) ? 128 : 64);
  const int ntiles = u3d_cdiv(n_rows > 0 ? n_rows : 1, 64);
  WgPlan p;
  for (;;) {
    p.tile = tile;
    p.ci_blocks = u3d_cdiv(cin, tile);
    p.co_blocks = u3d_cdiv(cout, tile);
    const int per = (count > 0 ? count : 1) * p.ci_blocks * p.co_blocks;
    int ns = u3d_cdiv(768, per);
    int max_ns = ntiles / 8 > 0 ? ntiles / 8 : 1;
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    p.nsplit = ns;
    if (tile == 64 || (long long)per * ns >= 192) break;
    tile /= 2;
  }
  return p;
}
extern "C" int64_t u3d_wgrad_batched_workspace(int32_t count, int32_t n_rows, int32_t cin, int32_t cout) {
  WgPlan p = wgrad_plan_batched(count, n_rows, cin, cout);
  return (int64_t)count * p.nsplit * cin * cout * 4;
}
extern "C" int32_t u3d_wgrad_batched_bf16(const void* const* in, const void* const* dout, float* const* dw, int32_t count,
                                          const int32_t* n_dev, int32_t n_rows, int32_t cin, int32_t cout, void* workspace,
                                          int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(in && dout && dw && n_dev && workspace && count >= 0 && count <= U3D_WGRAD_BATCH_MAX, U3D_ERR_ARG);
  if (count == 0) return U3D_OK;
  if (cin % 64 != 0 || cout % 64 != 0) return U3D_ERR_UNSUPPORTED;
  WgPlan p = wgrad_plan_batched(count, n_rows, cin, cout);
  U3D_REQUIRE(workspace_bytes >= u3d_wgrad_batched_workspace(count, n_rows, cin, cout), U3D_ERR_WORKSPACE);
  WgradBatch bt;
  for (int i = 0; i < count; ++i) { bt.in[i] = (const u16*)in[i]; bt.dout[i] = (const u16*)dout[i]; bt.dw[i] = dw[i]; }
  for (int i = count; i < U3D_WGRAD_BATCH_MAX; ++i) { bt.in[i] = nullptr; bt.dout[i] = nullptr; bt.dw[i] = nullptr; }
  const long long n = (long long)cin * cout, stride = (long long)p.nsplit * n;
  const size_t lds = 2 * (size_t)(64 * p.tile + 64 * p.tile) * 2;
  dim3 grid(p.nsplit, count, p.ci_blocks * p.co_blocks);
  if (p.tile == 256) {
    U3D_ALLOW_LDS(k_wgrad_batch_256, lds);
    hipLaunchKernelGGL(k_wgrad_batch_256, grid, dim3(512), lds, s, bt, (float*)workspace, n_dev, n_rows, cin, cout, p.co_blocks, stride);
  } else if (p.tile == 128) {
    if (lds > 64 * 1024) U3D_ALLOW_LDS(k_wgrad_batch_128, lds);
    hipLaunchKernelGGL(k_wgrad_batch_128, grid, dim3(256), lds, s, bt, (float*)workspace, n_dev, n_rows, cin, cout, p.co_blocks, stride);
  } else {
    hipLaunchKernelGGL(k_wgrad_batch_64, grid, dim3(256), lds, s, bt, (float*)workspace, n_dev, n_rows, cin, cout, p.co_blocks, stride);
  }
  WgradGroups gr;
  for (int i = 0; i < U3D_WGRAD_BATCH_MAX; ++i) gr.mult[i] = 0;
  for (int i = 0, lead = 0; i < count; ++i) {
    if (i > 0 && dw[i] == dw[i - 1]) { gr.mult[lead]++; } else { lead = i; gr.mult[i] = 1; }
  }
  hipLaunchKernelGGL(k_wgrad_batch_reduce, dim3(u3d_cdiv(n / 4, 256), count), dim3(256), 0, s, bt, gr, (const float*)workspace, n, p.nsplit, stride);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
