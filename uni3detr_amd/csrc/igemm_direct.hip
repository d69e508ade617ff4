// bf16 implicit-GEMM forward / dgrad for the NARROW sparse levels (16 / 32 / 64 channels, 27 offsets; ref: the SubMConv3d /
// SparseConv3d layers of models/pts_encoder/sparse_encoder_hd.py:80-138 on the 16-, 32- and 64-channel levels).
//
// Measured on the tiled kernel these levels ran on (k_igemm_fwd<4,1,4,*,*,32>, tools/sparse_bench.py, 338 k rows, 32 -> 32,
// 9.4 of 27 neighbours present): 70 us; without the register -> LDS stores of the activation tile 34 us; without the gather
// loads 64 us; with three stages of loads in flight 69-76 us.  It is neither memory nor latency: the activation tile's trip
// through LDS (ds_write_b128 + barrier + fragment reads, per offset, for ~130 clk of MFMAs) is the cost.
//
// With <= 64 input channels the whole reduction of one offset is one or two MFMA k-steps, and the MFMA operand layout of
// v_mfma_f32_16x16x32_bf16 (lane l: row l & 15, eight consecutive k at (l >> 4) * 8) IS a 16-byte piece of a gathered row.  So:
//   * activations never touch LDS: every lane gathers its operand piece straight into the MFMA register
//     (buffer_load_dwordx4, missing neighbour = out-of-range offset = hardware zero fill), three offsets ahead;
//   * the weights of ALL 27 offsets sit in LDS for the lifetime of the workgroup ([27][cout][cin_pad + 8] bf16, n-major, 80- /
//     144-byte rows: conflict-free ds_read_b128 fragments), loaded once; workgroups are persistent (each wave walks a strided
//     list of 64-row tiles inside its XCD's contiguous slab), so there is NO barrier after the prologue;
//   * neighbour indices: one coalesced dword load per offset and wave (lane r = row r of the tile), one tile ahead, handed to
//     the lane that needs them with ds_bpermute (LDS crossbar, no LDS memory); the slot of an index is reloaded for the next tile
//     right after its last use, so the gather chain index -> row never waits on vmcnt's in-order retirement;
//   * operands swapped (D^T = W X^T): a lane ends up with four consecutive output channels of one row -> 8-byte stores.
#include "common.h"

typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define DIR_K 27          /* offsets (3 x 3 x 3); the offset loop is fully unrolled */
#ifndef DIR_P1
#define DIR_P1 3          /* offsets of operand loads in flight, <= 32 input channels (3 or 9: divides DIR_K) */
#endif
#ifndef DIR_P2
#define DIR_P2 3          /* the same for 64 input channels (two loads per block and offset) */
#endif
#ifndef DIR_NW_A
#define DIR_NW_A 4        /* waves per workgroup, shapes whose weights take <= 80 KB of LDS */
#endif
#ifndef DIR_NW_B
#define DIR_NW_B 8        /* waves per workgroup, shapes whose weights take more (one workgroup per CU) */
#endif
#ifndef DIR_SB
#define DIR_SB 14         /* weight-staging loads in flight per thread (covers every shape in one batch) */
#endif
#ifndef DIR_WAVES_PER_SIMD
#define DIR_WAVES_PER_SIMD 2
#endif
#ifndef DIR_SPLIT_64
#define DIR_SPLIT_64 0    /* 1: 64 -> 64 layers as two 64 -> 32 launches.  Measured SLOWER on the 207 k-row level (17.5 of 27 neighbours
                             exist there: 2 x 78 us against 73 us on the LDS-DMA tiled kernel - each half gathers all 64 input channels
                             again and the level is bound by real L2 -> L1 bytes, not by empty slots) */
#endif
#ifndef DIR_SCHED_BARRIER
#define DIR_SCHED_BARRIER 1 /* pin the step's three phases (LDS requests | MFMAs | gathers): hipcc otherwise sinks the requests to their uses */
#endif
#define DIR_WM 4          /* 16-row blocks per wave: 64-row tiles */

// KS: MFMA k-steps per offset (cin_pad = 32 * KS; KS = 2 means cin = 64); WN: 16-column blocks of the output (cout = 16 * WN);
// NW: waves per workgroup; P: offsets of operand loads in flight (divides 27: the register sets are static).
//
// The loop body is written for INSTRUCTION COUNT: a wave issues one instruction per ~4 clk, two waves share a SIMD, and the first
// version of this body (~105 instructions per offset: per-block validity compares and selects, 64-bit address arithmetic for the
// index loads, ...) ran 30 of its 40 us with the gathers AND the MFMAs removed.  Now, per offset: 8 MFMAs, 4 * KS gathers, 1 index
// load, WN * KS + 4 LDS-pipe requests, and five VALU instructions:
//   * a missing neighbour needs no select: the index -1 shifted into a byte offset lies beyond the buffer's 2 GB bound -> zero fill;
//     rows past n_out (and tiles past this wave's slab) get -1 by OR-ing a per-tile lane mask into the index BEFORE it is permuted;
//   * index loads are buffer loads: per-tile row offset in a VGPR, the offset's table row as the scalar offset;
//   * weight fragments and permuted indices are requested one offset ahead (LDS round trips are not hidden by two waves per SIMD).
// F32ACC (split-bf16 products of the fp32 modules' narrow levels, u3d_igemm_direct_split_bf16): `out` is an F32 matrix and `addend`
// is only a FLAG - non-null: the tile is added to what `out` holds (the second and third of the three products hi.wh + hi.wl + lo.wh).
template <int KS, int WN, bool W_KMAJOR, int NW, int P, bool STATS = false, bool F32ACC = false>
__device__ __forceinline__ void igemm_direct_body(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr, int ld,
                                                  u16* __restrict__ out, const int* __restrict__ n_out_dev, int n_out_cap, int cin, int cout,
                                                  int co0, const u16* __restrict__ addend, double* __restrict__ stats) {
  // stats (STATS instantiations: the n-major kernels with <= 32 output columns - 32 more accumulator registers spill the 64-column
  // ones): BatchNorm statistics of the rounded output, one partial per WAVE: f64 [gridDim.x * NW][2][cout] = column sums
  // and sums of squares over the rows this wave wrote (f32 per lane over its ~20 rows, 16 lanes by shuffles at the end, f64 from
  // there on) - u3d_bn_finalize_partials with rows_per_block = 0 adds them up; waves without tiles write zeros.
  // cout = channels of a row of `out` (and of the weight tensor); this launch computes the COUT = 16 * WN columns from co0 on
  // (a 64 -> 64 layer is two launches of the 64 -> 32 kernel: the weights of 27 x 64 x 64 do not fit LDS, those of one half do)
  static_assert(DIR_K % P == 0, "operand register sets are indexed statically");
  constexpr int NT = NW * 64;
  constexpr int KP = KS * 32;                 // padded reduction length
  constexpr int LDW = KP + 8;                 // LDS row of one output channel (elements)
  constexpr int COUT = WN * 16;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int n_out = min(*n_out_dev, n_out_cap);

  // ---- this wave's tiles: XCD x (= blockIdx & 7, the observed dispatch rule; speed only) owns the contiguous slab x of 64-row
  //      tiles, its waves interleave inside the slab: the rows gathered at any time are close in memory and share one L2
  const int ntiles = (n_out + 63) >> 6;
  const int slab = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
  const int slab_end = min(ntiles, (xcd + 1) * slab);
  const int stride = wg_per_xcd * NW;
  int tile = xcd * slab + slot * NW + wv;

  // ld < 0: the table is read in REVERSED offset order (row 26 - q for offset q; `nbr` then points at table row 26) - the transposed
  // table of a submanifold convolution is its forward table with the offsets reversed, so the input gradient needs no table of its own
  const bool rev = ld < 0;
  const __amdgpu_buffer_rsrc_t nbr_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rev ? nbr + (long long)(DIR_K - 1) * ld : nbr), 0, 0x7FFFFFFC, 0x00020000);
  const unsigned ld4 = (unsigned)(rev ? -ld : ld) * 4u;
#define DIR_ROW(q) ((unsigned)(rev ? DIR_K - 1 - (q) : (q)) * ld4)
  auto row_off = [&](int t) -> unsigned {                       // byte offset of this lane's row of tile t in one table row (clamped)
    return (unsigned)max(0, min(t * 64 + lane, n_out - 1)) * 4u;
  };
  auto row_mask = [&](int t) -> int {                           // 0 for a live row of one of this wave's tiles, -1 otherwise
    return (t < slab_end && t * 64 + lane < n_out) ? 0 : -1;
  };
  int X[DIR_K];                                                 // X[q]: neighbour row of (tile row `lane`, offset q), tile = the one that uses it next
  // indices of the first tile: in flight underneath the weight staging
  {
    const unsigned r0 = row_off(tile);
#pragma unroll
    for (int q = 0; q < DIR_K; ++q) X[q] = __builtin_amdgcn_raw_buffer_load_b32(nbr_rs, r0, DIR_ROW(q), 0);
  }

  // 2 GB bound: every real row offset is below it, (-1 << shift) | part is above it (hardware zero fill)
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x80000000u, 0x00020000);
  const int row_shift = cin == 16 ? 5 : (cin == 32 ? 6 : 7);   // log2(bytes per input row)
  // this lane's 16-byte piece of a row (k-step 1 of a 64-channel row: + 64 bytes, as the instruction's immediate offset);
  // pieces beyond a 16-channel row are out of range by construction
  const unsigned part = (g * 8 < cin) ? (unsigned)g * 16u : 0x80000000u;
  int perm[DIR_WM];
#pragma unroll
  for (int a = 0; a < DIR_WM; ++a) perm[a] = (a * 16 + li) * 4;

  f32x4 acc[DIR_WM][WN];
  f32x4 cs[WN], cq[WN];
#pragma unroll
  for (int b = 0; b < WN; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int a = 0; a < DIR_WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 A[P][DIR_WM][KS];

  auto permute = [&](int q, int mask, int* dst) {               // rows a*16 + li of (X[q] | mask) -> lane (li, g)
    const int xm = X[q] | mask;
#pragma unroll
    for (int a = 0; a < DIR_WM; ++a) dst[a] = __builtin_amdgcn_ds_bpermute(perm[a], xm);
  };
  auto issue = [&](const int* idx, u32x4 (*dst)[KS]) {
#pragma unroll
    for (int a = 0; a < DIR_WM; ++a) {
      const unsigned voff = ((unsigned)idx[a] << row_shift) | part;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#if defined(DIR_EXP) && (DIR_EXP & 1)                           /* timing experiment (wrong results): no gather loads */
        dst[a][ks] = (u32x4){voff, voff, voff, voff};
#else
        dst[a][ks] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff + (unsigned)ks * 64u, 0, 0);
#endif
      }
    }
  };
  // weight fragments of offset s: three base registers (nine offsets each: immediate offsets stay below 64 KB), re-made opaque
  // once per tile - otherwise hipcc hoists all 27 x WN x KS loop-invariant fragments out of the tile loop and spills
  int wb[3];
  auto read_w = [&](int s, bf16x8 (*wf)[KS]) {
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wf[b][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(smem + wb[s / 9] + ((s % 9) * COUT + b * 16) * LDW + ks * 32));
  };
  auto make_wb = [&]() {
    int base = li * LDW + g * 8;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int i = 0; i < 3; ++i) wb[i] = base + i * 9 * COUT * LDW;
  };

  // ---- prologue, three round trips deep instead of five: the weight loads of the whole workgroup go out right behind the index
  //      loads (all SB of a thread in flight); while they travel, the first P offsets' operands are requested (they only need the
  //      indices, which retire first) and their index slots move on to the next tile; then the weights are written to LDS
  //      (n-major, zero-padded to KP), one barrier, and the loop starts with its first operands already on their way
  constexpr int SB = DIR_SB;
  constexpr int TOTAL = W_KMAJOR ? DIR_K * KP * (COUT / 8) : DIR_K * COUT * (KP / 8);
  static_assert(TOTAL <= SB * NT, "one staging batch per thread");
  u32x4 wv_[SB];
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int i = j * NT + tid;
    wv_[j] = (u32x4){0u, 0u, 0u, 0u};
    if constexpr (W_KMAJOR) {                 // global [27][cin][cout]: 16-byte pieces along n
      const int kap = i / (KP * (COUT / 8)), r = i % (KP * (COUT / 8)), k = r / (COUT / 8), n0 = (r % (COUT / 8)) * 8;
      if (i < TOTAL && k < cin) wv_[j] = *(const u32x4*)(w + ((long long)kap * cin + k) * cout + co0 + n0);
    } else {                                  // global [27][cout][cin]: rows as they are
      const int row = i / (KP / 8), k0 = (i % (KP / 8)) * 8;
      const long long grow = (long long)(row / COUT) * cout + co0 + row % COUT;
      if (i < TOTAL && k0 < cin) wv_[j] = *(const u32x4*)(w + grow * cin + k0);
    }
  }
  int mcur = row_mask(tile), mnxt = row_mask(tile + stride);
  unsigned roff1 = row_off(tile + stride), roff2 = row_off(tile + 2 * stride);
  int pi[2][DIR_WM];
  bf16x8 wf[2][WN][KS];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    permute(p, mcur, pi[0]);
    issue(pi[0], A[p]);
    X[p] = __builtin_amdgcn_raw_buffer_load_b32(nbr_rs, roff1, DIR_ROW(p), 0);
  }
  permute(P % DIR_K, mcur, pi[0]);
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const int i = j * NT + tid;
    if (i >= TOTAL) continue;
    if constexpr (W_KMAJOR) {                 // scattered as 2-byte LDS stores (transposition; callers that care pass n-major)
      const int kap = i / (KP * (COUT / 8)), r = i % (KP * (COUT / 8)), k = r / (COUT / 8), n0 = (r % (COUT / 8)) * 8;
      const u16* e = (const u16*)&wv_[j];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) smem[(kap * COUT + n0 + jj) * LDW + k] = e[jj];
    } else {
      const int row = i / (KP / 8), k0 = (i % (KP / 8)) * 8;
      *(u32x4*)(smem + row * LDW + k0) = wv_[j];
    }
  }
  __syncthreads();
  if (tile >= slab_end) {                                       // no tile for this wave (its statistics partial must still read as zero)
    if constexpr (STATS) {
      double* p = stats + ((long long)blockIdx.x * NW + wv) * 2 * cout + co0;
      if (lane < 2 * COUT) p[(lane / COUT) * cout + lane % COUT] = 0.0;
      if (2 * COUT > 64 && lane + 64 < 2 * COUT) p[((lane + 64) / COUT) * cout + (lane + 64) % COUT] = 0.0;
    }
    return;
  }
  make_wb();
  read_w(0, wf[0]);

  for (; tile < slab_end; tile += stride) {
#pragma unroll
    for (int s = 0; s < DIR_K; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;                     // DIR_K is odd: the roles swap across the loop edge, fixed below
      // next step's LDS traffic first
      read_w((s + 1) % DIR_K, wf[nxt]);
      permute((s + P + 1) % DIR_K, (s + P + 1 >= DIR_K) ? mnxt : mcur, pi[nxt]);
#if DIR_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);                        // keep the requests in front of the MFMAs that hide them
#endif
      // MFMAs of (tile, s): operand set s % P (requested P offsets ago)
#pragma unroll
      for (int a = 0; a < DIR_WM; ++a)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 xf = __builtin_bit_cast(bf16x8, A[s % P][a][ks]);
#pragma unroll
          for (int b = 0; b < WN; ++b) {
#if defined(DIR_EXP) && (DIR_EXP & 2)                           /* timing experiment (wrong results): no MFMAs */
            acc[a][b][0] += (float)xf[0] * (float)wf[cur][b][ks][0];
#else
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][b][ks], xf, acc[a][b], 0, 0, 0);
#endif
          }
        }
      if (s == DIR_K - 1) {                                     // tile finished: rows out (lane: row a*16 + li, channels b*16 + 4g .. +3)
#pragma unroll
        for (int a = 0; a < DIR_WM; ++a) {
          const int m = tile * 64 + a * 16 + li;
#pragma unroll
          for (int b = 0; b < WN; ++b) {
#if !(defined(DIR_EXP) && (DIR_EXP & 8))                        /* timing experiment (wrong results): nothing stored except by the last tile */
            if constexpr (F32ACC) {
              if (m < n_out) {
                float* of = (float*)out + (long long)m * cout + co0 + b * 16 + g * 4;
                f32x4 v = acc[a][b];
                if (addend) v += *(const f32x4*)of;
                *(f32x4*)of = v;
              }
            } else
            if (m < n_out) {
              f32x4 v = acc[a][b];
              // addend (nullable): a bf16 tensor of out's shape summed in before the rounding (the residual branch's gradient)
              if (addend) v += __builtin_convertvector(*(const bf16x4*)(addend + (long long)m * cout + co0 + b * 16 + g * 4), f32x4);
              const bf16x4 o = __builtin_convertvector(v, bf16x4);
              *(bf16x4*)(out + (long long)m * cout + co0 + b * 16 + g * 4) = o;
              if constexpr (STATS) {                            // of the ROUNDED values: what the BatchNorm behind this conv reads
                const f32x4 vr = __builtin_convertvector(o, f32x4);
                cs[b] += vr;
                cq[b] += vr * vr;
              }
            }
#else
            if (m < n_out && tile + stride >= slab_end) *(bf16x4*)(out + (long long)m * cout + co0 + b * 16 + g * 4) = __builtin_convertvector(acc[a][b], bf16x4);
#endif
            acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
#if DIR_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
      // request offset q = s + P (wrapping into this wave's next tile) into the set just consumed; then the index slot q moves on
      // to the tile after that one
      const int q = (s + P) % DIR_K;
      issue(pi[cur], A[s % P]);
#if !(defined(DIR_EXP) && (DIR_EXP & 4))                        /* timing experiment (wrong results): indices never reloaded */
      X[q] = __builtin_amdgcn_raw_buffer_load_b32(nbr_rs, (s + P >= DIR_K) ? roff2 : roff1, DIR_ROW(q), 0);
#endif
    }
    // 27 steps: what the last step left in slot 1 is what step 0 of the next tile reads from slot 0
#pragma unroll
    for (int a = 0; a < DIR_WM; ++a) pi[0][a] = pi[1][a];
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[0][b][ks] = wf[1][b][ks];
    mcur = mnxt;
    mnxt = row_mask(tile + 2 * stride);
    roff1 = roff2;
    roff2 = row_off(tile + 3 * stride);
    make_wb();
  }
  if constexpr (STATS) {
    double* p = stats + ((long long)blockIdx.x * NW + wv) * 2 * cout + co0;
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = cs[b][r], s2 = cq[b][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (li == 0) {
          p[b * 16 + g * 4 + r] = (double)s1;
          p[cout + b * 16 + g * 4 + r] = (double)s2;
        }
      }
  }
}

#define U3D_DIRECT_KERNEL(NAME, KS, WN, KM, NW, P)                                                                                    \
  __global__ __launch_bounds__(NW * 64, DIR_WAVES_PER_SIMD) void NAME(const u16* in, const u16* w, const int* nbr, int ld, u16* out, const int* n_out_dev, \
                                                  int n_out_cap, int cin, int cout, int co0, const u16* addend, double* stats) {   \
    igemm_direct_body<KS, WN, KM, NW, P>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, co0, addend, stats);                      \
  }
// name: k_igemm_direct_<cin_pad>x<cout>_<k|n>: k = weights [27][cin][cout] (forward), n = [27][cout][cin] (dgrad)
U3D_DIRECT_KERNEL(k_igemm_direct_32x16_k, 1, 1, true, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_32x16_n, 1, 1, false, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_32x32_k, 1, 2, true, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_32x32_n, 1, 2, false, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_32x64_k, 1, 4, true, DIR_NW_B, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_32x64_n, 1, 4, false, DIR_NW_B, DIR_P1)
U3D_DIRECT_KERNEL(k_igemm_direct_64x16_k, 2, 1, true, DIR_NW_A, DIR_P2)
U3D_DIRECT_KERNEL(k_igemm_direct_64x16_n, 2, 1, false, DIR_NW_A, DIR_P2)
U3D_DIRECT_KERNEL(k_igemm_direct_64x32_k, 2, 2, true, DIR_NW_B, DIR_P2)
U3D_DIRECT_KERNEL(k_igemm_direct_64x32_n, 2, 2, false, DIR_NW_B, DIR_P2)

#define U3D_DIRECT_KERNEL_S(NAME, KS, WN, NW, P)                                                                                     \
  __global__ __launch_bounds__(NW * 64, DIR_WAVES_PER_SIMD) void NAME(const u16* in, const u16* w, const int* nbr, int ld, u16* out, const int* n_out_dev, \
                                                  int n_out_cap, int cin, int cout, int co0, const u16* addend, double* stats) {   \
    igemm_direct_body<KS, WN, false, NW, P, true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, co0, addend, stats);              \
  }
U3D_DIRECT_KERNEL_S(k_igemm_direct_32x16_ns, 1, 1, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL_S(k_igemm_direct_32x32_ns, 1, 2, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL_S(k_igemm_direct_64x16_ns, 2, 1, DIR_NW_A, DIR_P2)
U3D_DIRECT_KERNEL_S(k_igemm_direct_64x32_ns, 2, 2, DIR_NW_B, DIR_P2)

#define U3D_DIRECT_KERNEL_F(NAME, KS, WN, NW, P)                                                                                     \
  __global__ __launch_bounds__(NW * 64, DIR_WAVES_PER_SIMD) void NAME(const u16* in, const u16* w, const int* nbr, int ld, u16* out, const int* n_out_dev, \
                                                  int n_out_cap, int cin, int cout, int co0, const u16* addend, double* stats) {   \
    igemm_direct_body<KS, WN, false, NW, P, false, true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, co0, addend, stats);       \
  }
U3D_DIRECT_KERNEL_F(k_igemm_direct_32x16_nf, 1, 1, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL_F(k_igemm_direct_32x32_nf, 1, 2, DIR_NW_A, DIR_P1)
U3D_DIRECT_KERNEL_F(k_igemm_direct_32x64_nf, 1, 4, DIR_NW_B, DIR_P1)
U3D_DIRECT_KERNEL_F(k_igemm_direct_64x16_nf, 2, 1, DIR_NW_A, DIR_P2)
U3D_DIRECT_KERNEL_F(k_igemm_direct_64x32_nf, 2, 2, DIR_NW_B, DIR_P2)

typedef void (*direct_kernel_t)(const u16*, const u16*, const int*, int, u16*, const int*, int, int, int, int, const u16*, double*);

// 0 = launched, U3D_ERR_UNSUPPORTED = shape not served here (caller falls through to the tiled kernels)
int u3d_launch_igemm_direct(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev, int n_out_cap,
                            int cin, int cout, int kvol, int transpose_w, hipStream_t s, const void* addend, double* stats,
                            int* stats_blocks, int f32acc) {
  // f32acc: 0 = bf16 output (+ bf16 addend); 1 / 2 = F32 output of an n-major launch, written (1) or accumulated into (2)
  // stats_blocks != nullptr: a QUERY - nothing is launched, *stats_blocks = number of per-wave statistics partials a launch writes
  if (kvol != DIR_K || !nbr || (cin != 16 && cin != 32 && cin != 64) || (cout != 16 && cout != 32 && cout != 64))
    return U3D_ERR_UNSUPPORTED;
#if !DIR_SPLIT_64
  if (cin == 64 && cout == 64) return U3D_ERR_UNSUPPORTED;
#endif
  const int ks = cin > 32 ? 2 : 1, kp = ks * 32;
  const int cout_total = cout;
  const int halves = (cin == 64 && cout == 64) ? 2 : 1;          // 64 -> 64: two launches over 32 output columns each
  if (halves == 2) cout = 32;
  const size_t lds = (size_t)DIR_K * cout * (kp + 8) * 2;
  if (lds > 160 * 1024) return U3D_ERR_UNSUPPORTED;
  direct_kernel_t kern = nullptr;
  int nw = 4;
#define DIR_PICK(KSV, CO, NAMEK, NAMEN, NWV)                              \
  if (ks == KSV && cout == CO) {                                          \
    kern = transpose_w ? NAMEN : NAMEK;                                   \
    nw = NWV;                                                             \
    if (transpose_w) U3D_ALLOW_LDS(NAMEN, lds); else U3D_ALLOW_LDS(NAMEK, lds); \
  }
  DIR_PICK(1, 16, k_igemm_direct_32x16_k, k_igemm_direct_32x16_n, DIR_NW_A)
  DIR_PICK(1, 32, k_igemm_direct_32x32_k, k_igemm_direct_32x32_n, DIR_NW_A)
  DIR_PICK(1, 64, k_igemm_direct_32x64_k, k_igemm_direct_32x64_n, DIR_NW_B)
  DIR_PICK(2, 16, k_igemm_direct_64x16_k, k_igemm_direct_64x16_n, DIR_NW_A)
  DIR_PICK(2, 32, k_igemm_direct_64x32_k, k_igemm_direct_64x32_n, DIR_NW_B)
#undef DIR_PICK
  if (!kern) return U3D_ERR_UNSUPPORTED;
  if (f32acc) {
    if (!transpose_w || stats || stats_blocks || halves == 2) return U3D_ERR_UNSUPPORTED;
    if (ks == 1 && cout == 16) { kern = k_igemm_direct_32x16_nf; U3D_ALLOW_LDS(k_igemm_direct_32x16_nf, lds); }
    if (ks == 1 && cout == 32) { kern = k_igemm_direct_32x32_nf; U3D_ALLOW_LDS(k_igemm_direct_32x32_nf, lds); }
    if (ks == 1 && cout == 64) { kern = k_igemm_direct_32x64_nf; U3D_ALLOW_LDS(k_igemm_direct_32x64_nf, lds); }
    if (ks == 2 && cout == 16) { kern = k_igemm_direct_64x16_nf; U3D_ALLOW_LDS(k_igemm_direct_64x16_nf, lds); }
    if (ks == 2 && cout == 32) { kern = k_igemm_direct_64x32_nf; U3D_ALLOW_LDS(k_igemm_direct_64x32_nf, lds); }
    addend = f32acc == 2 ? out : nullptr;                        // the F32ACC kernels read `addend` as the accumulate flag
  }
  if (stats || stats_blocks) {                                   // statistics epilogue: n-major kernels with <= 32 output columns
    if (halves == 2 || !transpose_w || cout > 32) return U3D_ERR_UNSUPPORTED;
    if (ks == 1 && cout == 16) { kern = k_igemm_direct_32x16_ns; U3D_ALLOW_LDS(k_igemm_direct_32x16_ns, lds); }
    if (ks == 1 && cout == 32) { kern = k_igemm_direct_32x32_ns; U3D_ALLOW_LDS(k_igemm_direct_32x32_ns, lds); }
    if (ks == 2 && cout == 16) { kern = k_igemm_direct_64x16_ns; U3D_ALLOW_LDS(k_igemm_direct_64x16_ns, lds); }
    if (ks == 2 && cout == 32) { kern = k_igemm_direct_64x32_ns; U3D_ALLOW_LDS(k_igemm_direct_64x32_ns, lds); }
  }
  if (n_out_cap <= 0) { if (stats_blocks) *stats_blocks = 0; return U3D_OK; }
  // persistent grid: as many workgroups as fit on the chip at once (LDS-limited), a multiple of 8 (one share per XCD), and no
  // more than there are tiles
  static int cu_count[64] = {0};                           // per device, read once (plain host query, legal during stream capture)
  int dev = 0;
  (void)hipGetDevice(&dev);
  int cus = (dev >= 0 && dev < 64) ? cu_count[dev] : 0;
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev >= 0 && dev < 64) cu_count[dev] = cus;
  }
  int per_cu = (int)((160 * 1024) / lds);
  const int by_waves = (4 * DIR_WAVES_PER_SIMD) / nw > 0 ? (4 * DIR_WAVES_PER_SIMD) / nw : 1;   // waves per SIMD: the register budget of these kernels
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  int grid = cus * per_cu;
  const int ntiles = u3d_cdiv(n_out_cap, 64);
  const int need = u3d_cdiv(ntiles, nw);
  if (grid > need) grid = need;
  grid = (grid + 7) / 8 * 8;
  if (stats_blocks) { *stats_blocks = grid * nw; return U3D_OK; }
  for (int h = 0; h < halves; ++h)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, s, (const u16*)in, (const u16*)w, nbr, ld, (u16*)out, n_out_dev, n_out_cap, cin,
                       cout_total, h * 32, (const u16*)addend, stats);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

// Split-bf16 product on the narrow 27-offset levels (cin, cout in {16, 32, 64}, not 64 -> 64): the three products hi.wh + hi.wl +
// lo.wh as three launches of the direct-operand kernel accumulating into ONE f32 output.  in: bf16 planes [2 * n_in_cap][cin]
// (u3d_split_rows_f32), w3: bf16 [3 * 27][cout][cin] = (wh, wl, wh) n-major (u3d_split3_weights), nbr / ld as u3d_igemm_fwd_bf16
// (ld < 0: reversed table), out f32 [n_out_cap][cout].
extern "C" int32_t u3d_igemm_direct_split_bf16(const void* in, const void* w3, const int32_t* nbr, int32_t ld, float* out,
                                               const int32_t* n_out_dev, int32_t n_out_cap, int32_t n_in_cap, int32_t cin, int32_t cout,
                                               u3d_stream s) {
  U3D_REQUIRE(in && w3 && nbr && out && n_out_dev && n_in_cap >= 0, U3D_ERR_ARG);
  const u16* hi = (const u16*)in;
  const u16* lo = hi + (long long)n_in_cap * cin;
  const u16* wh = (const u16*)w3;
  const u16* wl = wh + (long long)DIR_K * cout * cin;
  int rc = u3d_launch_igemm_direct(hi, wh, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, DIR_K, 1, (hipStream_t)s, nullptr, nullptr, nullptr, 1);
  if (rc != U3D_OK) return rc;
  rc = u3d_launch_igemm_direct(hi, wl, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, DIR_K, 1, (hipStream_t)s, nullptr, nullptr, nullptr, 2);
  if (rc != U3D_OK) return rc;
  return u3d_launch_igemm_direct(lo, wh, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, DIR_K, 1, (hipStream_t)s, nullptr, nullptr, nullptr, 2);
}
