// Output-stationary "halo" kernels for the 64- and 128-channel, 27-offset submanifold levels (ref: sparse_encoder_hd.py:106-138, the
// SparseBasicBlock convs of the stride-4 / stride-8 stages; spconv's SubMConv3d gathers each of the 27 offsets' rows again).
//
// The LDS-DMA implicit-GEMM kernels (igemm_bf16.hip) fetch, per 128-row output tile, 27 x 128 operand rows through 27 x 128 one-row
// DMA pieces, and the fill rate of that path is what bounds them.  But rows are numbered in 4x4x4-block-major
// order, so the 27 x 128 table entries of a tile name only ~210-420 DISTINCT rows (the tile's cells plus a one-cell shell).  This file:
//   1. k_halo_build  (once per level and step, shared by all of the level's convs and their gradients): per tile, the sorted
//      distinct table entries (LDS bitmap) -> tile_rows[tile][slot] (slot 0 = the all-zero row), and the table rewritten as 16-bit
//      LDS slots loc[tile][offset][row];
//   2. k_subm_halo64 (forward / input gradient, 64 channels): stage the tile's distinct rows ONCE (coalesced 16 B loads, 144 B LDS
//      row stride), then run all 27 offsets out of LDS without a barrier: wave w takes offsets w, w+4, ... for the whole 128 x 64
//      tile (64 MFMAs per 16 LDS reads and 8 weight-fragment loads), and the four partial tiles are summed by a two-round
//      reduce-scatter through the (by then dead) stage buffer;
//   3. k_subm_halo128 (forward / input gradient, 128 channels): waves split the output columns, rows staged one 64-channel half at
//      a time;
//   4. k_subm_halo_wgrad64 (weight gradient, 64 channels): both MFMA operands by transpose reads out of the staged rows / dy tile,
//      persistent 8-wave workgroups, LDS-DMA double-buffered staging.
// The transposed table of a SubM layer is the forward one with the offsets reversed, so the same loc[] serves the input gradient
// (krev: offset k reads loc[26 - k]).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define HL_T 128                 /* output rows per tile */
#define HL_K 27
#define HL_C 64
#define HL_TRC 3460              /* tile_rows row length: 1 + 27 * 128 slots at most, padded to 16 B */
#define HL_RS 72                 /* LDS row stride in elements: 144 B - consecutive slots land on distinct 4-bank groups */
#ifndef HL_MAXS
#define HL_MAXS 552              /* staged slots per tile (79.5 KB: two workgroups per CU); slots beyond are read from global memory */
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// table build
// ---------------------------------------------------------------------------------------------------------------------------
#ifdef HL_PHASE_TIMING       /* tools/halo_phase.py: shader-clock stamps of a few workgroups at the phase boundaries */
__device__ unsigned long long hl_dbg[8 * 8];
#define HL_MARK(id) do { if ((blockIdx.x & 255) == 7 && blockIdx.x < 8 * 256 && threadIdx.x == 0) hl_dbg[(blockIdx.x >> 8) * 8 + id] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t u3d_debug_halo_times(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(hl_dbg), 64 * 8) == hipSuccess ? 0 : -1; }
__device__ unsigned long long hb_dbg[8 * 8];
#define HB_MARK(id) do { if ((blockIdx.x & 255) == 7 && blockIdx.x < 8 * 256 && threadIdx.x == 0) hb_dbg[(blockIdx.x >> 8) * 8 + id] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t u3d_debug_halo_build_times(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(hb_dbg), 64 * 8) == hipSuccess ? 0 : -1; }
__device__ unsigned long long hw_dbg[64];
#define HW_MARK(id) do { if (blockIdx.x == 7 && threadIdx.x == 0 && (id) < 64) hw_dbg[id] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t u3d_debug_halo_wgrad_times(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(hw_dbg), 64 * 8) == hipSuccess ? 0 : -1; }
#else
#define HL_MARK(id)
#define HB_MARK(id)
#define HW_MARK(id)
#endif
// Distinct rows of a tile, sorted, without sorting: the keys are row numbers < n_cap, so the tile marks them in an LDS BITMAP over
// all rows (n_cap / 8 bytes; fire-and-forget ds_or), a popcount prefix over the bitmap words numbers the set bits in ascending
// order, and an entry's slot is prefix[word] + popcount(bits below) + 1 - two LDS reads.  (A hash set + bitonic sort of the keys
// took 44 us per level, a sort of the raw 27 x 128 entries 180 us; sorted slots keep the neighbours of consecutive rows on
// consecutive LDS rows - conflict-free fragment reads in k_subm_halo64.)
// dynamic LDS: bitmap u32 [W] | prefix u16 [W / 4] (one per GROUP of four words = 128 rows) | part int [256], W = words rounded up to a
// multiple of 1024.  (A prefix per word cost 6 B per 32 rows and capped a level at 869 k rows - ScanNet-large's 64-channel level has a
// capacity of 960 k; per group it is 4.5 B per 32 rows = 1.15 M rows, for up to three more popcounts per lookup.)
__global__ __launch_bounds__(256) void k_halo_build(const int32_t* __restrict__ nbr, int ld, int kvol, const int32_t* __restrict__ n_dev,
                                                    int n_cap, int wpt, int32_t* __restrict__ tile_rows, u16* __restrict__ loc,
                                                    int32_t* __restrict__ tile_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = wpt * 256;                        // wpt % 4 == 0: a thread owns whole groups
  unsigned* bits = (unsigned*)smem;
  u16* gpre = (u16*)(bits + W);                   // [W / 4]
  int* part = (int*)(gpre + W / 4);
  const int tid = threadIdx.x, tile = blockIdx.x;
  const int m0 = tile * HL_T;
  const int n = min(*n_dev, n_cap);
  HB_MARK(0);
  // entry e of the tile in OUTPUT order (k, r16, mt): e = k * 128 + r16 * 8 + mt <-> row mt * 16 + r16; kept in registers
  int gs[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int e = tid + i * 256;
    const int k = e >> 7, r = (e & 7) * 16 + ((e >> 3) & 15);
    gs[i] = (e < kvol * HL_T && m0 + r < n) ? nbr[(long long)k * ld + m0 + r] : -1;      // (kvol <= 27 offsets; loc keeps 27 slots per tile)
  }
  for (int i = tid; i < W; i += 256) bits[i] = 0u;
  __syncthreads();
  HB_MARK(1);
#pragma unroll
  for (int i = 0; i < 14; ++i)
    if (gs[i] >= 0) atomicOr(&bits[gs[i] >> 5], 1u << (gs[i] & 31));
  __syncthreads();
  HB_MARK(2);
  // thread t owns words [t * wpt, (t + 1) * wpt): popcount, block scan, per-group exclusive prefix
  int c = 0;
  for (int j = 0; j < wpt; ++j) c += __popc(bits[tid * wpt + j]);
  part[tid] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int total = part[255];
  HB_MARK(3);
  int32_t* tr = tile_rows + (long long)tile * HL_TRC;
  if (tid == 0) { tile_cnt[tile] = total + 1; tr[0] = -1; }
  int p = part[tid] - c;
  for (int j = 0; j < wpt; j += 4) {
    const int wd = tid * wpt + j;
    gpre[wd >> 2] = (u16)p;
    p += __popc(bits[wd]) + __popc(bits[wd + 1]) + __popc(bits[wd + 2]) + __popc(bits[wd + 3]);
  }
  __syncthreads();
  // slot-parallel: the i-th set bit = (last group whose prefix is <= i - a non-empty group is the last of its run of equal
  // prefixes -, then the word inside it, then the bit of that rank).  A word-parallel loop over set bits ran 26 words x up to 32 bits
  // with one lane live.
  const int NG = W >> 2;
  for (int i = tid; i < total; i += 256) {
    int lo = 0, hi = NG;                          // gpre[lo] <= i < gpre[hi] (gpre[NG] = total)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)gpre[mid] <= i) lo = mid; else hi = mid;
    }
    int k = i - (int)gpre[lo], wd = lo << 2;
#pragma unroll
    for (int q = 0; q < 3; ++q) {                 // the word of the group that holds rank k
      const int cw = __popc(bits[wd]);
      if (k >= cw) { k -= cw; ++wd; }
    }
    unsigned v = bits[wd];
    int pos = 0;
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
      const int cnt = __popc(v & ((1u << sh) - 1u));
      if (k >= cnt) { k -= cnt; pos += sh; v >>= sh; }
    }
    tr[1 + i] = wd * 32 + pos;
  }
  __syncthreads();
  HB_MARK(4);
  u16* lp = loc + (long long)tile * HL_K * HL_T;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int e = tid + i * 256;
    const int g = gs[i];
    int slot = 0;
    if (g >= 0) {
      const int wd = g >> 5, w0 = wd & ~3;
      slot = gpre[wd >> 2] + __popc(bits[wd] & ((1u << (g & 31)) - 1u)) + 1;
      if (wd > w0) slot += __popc(bits[w0]);
      if (wd > w0 + 1) slot += __popc(bits[w0 + 1]);
      if (wd > w0 + 2) slot += __popc(bits[w0 + 2]);
    }
    if (e < HL_K * HL_T) lp[e] = (u16)slot;      // entry (r % 16) * 8 + r / 16 of (tile, k): a lane's 8 row blocks in one 16 B word
  }
  HB_MARK(5);
}

// weights n-major [27][64 n][64 k] -> MFMA fragment order [27][nt 4][ks 2][lane 64][8]: a wave's fragment load is 1 KB contiguous
// (row-major fragments touch 16 half-used cache lines per load: the L1 tag rate, not the MFMAs, then bounds the kernel)
// blockIdx.y: weight of a batch (srcs / dsts device pointer arrays), or src / dst themselves when the arrays are null
__global__ __launch_bounds__(256) void k_halo_wpack(const u16* __restrict__ src, u16* __restrict__ dst, const u16* const* __restrict__ srcs,
                                                    u16* const* __restrict__ dsts) {
  if (srcs) { src = srcs[blockIdx.y]; dst = dsts[blockIdx.y]; }
  const int c = blockIdx.x * 256 + threadIdx.x;   // 16 B chunk of dst
  if (c >= HL_K * 512) return;
  const int lane = c & 63, ks = (c >> 6) & 1, nt = (c >> 7) & 3, k = c >> 9;
  *(u32x4*)(dst + (long long)c * 8) = *(const u32x4*)(src + k * 4096 + (nt * 16 + (lane & 15)) * 64 + ks * 32 + (lane >> 4) * 8);
}

// ---------------------------------------------------------------------------------------------------------------------------
// convolution
// ---------------------------------------------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (every lane gets it): 4 VALU ops - the statistics epilogue reduces 32 values per lane, which as
// ds_bpermute shuffles cost the launch 9 us
__device__ __forceinline__ float hl_row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));     // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));     // row_mirror
  return v;
}

__device__ __forceinline__ f32x4 hl_sel(bool c, f32x4 a, f32x4 b) {
  f32x4 r;
  r[0] = c ? a[0] : b[0]; r[1] = c ? a[1] : b[1]; r[2] = c ? a[2] : b[2]; r[3] = c ? a[3] : b[3];
  return r;
}

// BatchNorm-backward statistics mode of the epilogue (== u3d_bn_epi; see glds_epilogue.inc / BnEpi in igemm_bf16.hip)
struct HlBn {
  const u16* x = nullptr; const u16* y = nullptr;
  const float *mean = nullptr, *invstd = nullptr, *gamma = nullptr, *beta = nullptr;
  int relu = 0, pad = 0;
};

// in/out/addend bf16 [n][64]; wgt: k_halo_wpack of bf16 [27][64 (n)][64 (reduction)]; stats f64 [tiles][2][64] or null
__global__ __launch_bounds__(256, 2) void k_subm_halo64(const u16* __restrict__ in, const u16* __restrict__ wgt,
                                                        const int32_t* __restrict__ tile_rows, const u16* __restrict__ loc,
                                                        const int32_t* __restrict__ tile_cnt, const int32_t* __restrict__ n_dev, int n_cap,
                                                        int krev, const u16* __restrict__ addend, u16* __restrict__ out,
                                                        double* __restrict__ stats, const HlBn bn, int maxs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* xs = (u16*)smem;                           // [HL_MAXS][HL_RS]
  const int tid = threadIdx.x;
  // workgroups go round-robin over the 8 XCDs: give each XCD a CONTIGUOUS range of the LIVE tiles - neighbouring tiles share most of
  // their halo rows, which then hit that XCD's L2 instead of being fetched once per XCD (u3d_xcd_tile: the grid is capacity-sized)
  const int n = min(*n_dev, n_cap);
  const int live = (n + HL_T - 1) / HL_T;
  if (stats && live + (int)blockIdx.x < (int)gridDim.x && tid < 2 * HL_C)      // statistics rows of the tiles past the count: zeros
    stats[(long long)(live + blockIdx.x) * 2 * HL_C + tid] = 0.0;
#ifdef HL_NO_XCD_MAP
  const int tile = (int)blockIdx.x < live ? (int)blockIdx.x : -1;
#else
  const int tile = u3d_xcd_tile(blockIdx.x, live);
#endif
  if (tile < 0) return;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = tile * HL_T;
  HL_MARK(0);
  const int32_t* rows_p = tile_rows + (long long)tile * HL_TRC;
  const int nl = min(tile_cnt[tile], maxs);      // staged slots (maxs <= HL_MAXS: the stage buffer; lower only as a test hook)
  // stage the distinct rows: 8 lanes x 16 B per row.  All of a thread's row indices are requested at once, then all of its rows:
  // two memory round trips per 288 slots (a loop of index -> row -> store iterations spent 15 us per launch waiting in turn)
#ifndef HL_ABL_NOSTAGE
  {
    const int part = tid & 7, sb = tid >> 3;
    for (int base = 0; base < nl; base += 288) {
      int idx[9];
      u32x4 v[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int slot = base + sb + 32 * i;
        idx[i] = (slot > 0 && slot < nl) ? rows_p[slot] : -1;
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (idx[i] >= 0) v[i] = *(const u32x4*)(in + (long long)idx[i] * HL_C + part * 8);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int slot = base + sb + 32 * i;
        if (slot < nl) *(u32x4*)(xs + slot * HL_RS + part * 8) = v[i];
      }
    }
  }
#endif
  HL_MARK(1);
  __syncthreads();
  HL_MARK(2);

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u16* locp = loc + (long long)tile * HL_K * HL_T + r16 * 8;
  const u16* wl = wgt + lane * 8;                 // fragment-packed (k_halo_wpack)
  const u16* xl = xs + kq * 8;

  u16x8 sl = *(const u16x8*)(locp + (krev ? 26 - w : w) * HL_T);
  bf16x8 wf[4][2];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    wf[b][0] = *(const bf16x8*)(wl + ((w * 4 + b) * 2 + 0) * 512);
    wf[b][1] = *(const bf16x8*)(wl + ((w * 4 + b) * 2 + 1) * 512);
  }
  // FAST (every slot of the tile is staged - the normal case): straight-line LDS reads and MFMAs, no branch in the offset loop
  // (a per-row-block "staged or global" branch made the compiler drain every outstanding load at each of the 8 row blocks)
#ifdef HL_ABL_NOW
#define HL_KN(k) w
#else
#define HL_KN(k) ((k) + 4 < HL_K ? (k) + 4 : (k))      /* next offset of this wave (the last one re-reads its own: hits L1) */
#endif
#ifdef HL_ABL_NOMFMA
#define HL_MFMA(a, x0, x1) _Pragma("unroll") for (int b = 0; b < 4; ++b) { acc[a][b][0] += (float)x0[b] * (float)wf[b][0][0]; acc[a][b][1] += (float)x1[b] * (float)wf[b][1][1]; }
#else
#define HL_MFMA(a, x0, x1)                                                                              \
  _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][0], x0, acc[a][b], 0, 0, 0);              \
    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][1], x1, acc[a][b], 0, 0, 0);              \
  }
#endif
#ifdef HL_ABL_NOLDS
#define HL_LOADX(FAST, s, x0, x1) { x0 = wf[0][0]; x1 = wf[1][1]; x0[0] = (__bf16)(float)(s); }
#else
#define HL_LOADX(FAST, s, x0, x1)                                                                       \
  if (FAST || (s) < nl) {                                                                               \
    x0 = *(const bf16x8*)(xl + (s) * HL_RS);                                                            \
    x1 = *(const bf16x8*)(xl + (s) * HL_RS + 32);                                                       \
  } else {                                                                                              \
    const u16* g = in + (long long)rows_p[s] * HL_C + kq * 8;                                           \
    x0 = *(const bf16x8*)g;                                                                             \
    x1 = *(const bf16x8*)(g + 32);                                                                      \
  }
#endif
// the next offset's weights and slots are requested at the top of the iteration (an L2 round trip is about one offset's MFMA
// time), the operand rows of block a + 1 before the MFMAs of block a; scheduling barriers keep the compiler from sinking the
// prefetches to where their registers are free (it then waited for them at the loop end) and from hoisting all 16 LDS reads
#define HL_OFFSET_LOOP(FAST)                                                                            \
  for (int k = w; k < HL_K; k += 4) {                                                                   \
    const int kn = HL_KN(k);                                                                            \
    const u16x8 sln = *(const u16x8*)(locp + (krev ? 26 - kn : kn) * HL_T);                             \
    bf16x8 wn[4][2];                                                                                    \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                      \
      wn[b][0] = *(const bf16x8*)(wl + ((kn * 4 + b) * 2 + 0) * 512);                                   \
      wn[b][1] = *(const bf16x8*)(wl + ((kn * 4 + b) * 2 + 1) * 512);                                   \
    }                                                                                                   \
    bf16x8 x0, x1, y0, y1;                                                                              \
    HL_LOADX(FAST, (int)sl[0], x0, x1)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    _Pragma("unroll") for (int a = 0; a < 8; a += 2) {                                                   \
      HL_LOADX(FAST, (int)sl[a + 1], y0, y1)                                                            \
      HL_MFMA(a, x0, x1)                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      if (a + 2 < 8) { HL_LOADX(FAST, (int)sl[a + 2 < 8 ? a + 2 : 0], x0, x1) }                         \
      HL_MFMA(a + 1, y0, y1)                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
    sl = sln;                                                                                           \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) { wf[b][0] = wn[b][0]; wf[b][1] = wn[b][1]; }          \
  }
  if (tile_cnt[tile] <= maxs) { HL_OFFSET_LOOP(1) } else { HL_OFFSET_LOOP(0) }

  // sum the four waves' partial tiles: reduce-scatter in two rounds through the stage buffer
  f32x4* xb = (f32x4*)smem;
  const bool h = w & 1, q = (w >> 1) & 1;
  // BatchNorm-backward statistics mode: this wave's x (and y) fragments of the rows it will own are requested before the
  // reduce-scatter (8 B loads with a row stride: their latency hides behind the exchange)
  bf16x4 xr[2][4], yr[2][4];
  if (bn.x) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + (4 * (int)h + 2 * (int)q + j) * 16 + r16;
      if (m >= n) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const long long o_el = (long long)m * HL_C + b * 16 + 4 * kq;
        xr[j][b] = *(const bf16x4*)(bn.x + o_el);
        if (bn.y) yr[j][b] = *(const bf16x4*)(bn.y + o_el);
      }
    }
  }
  HL_MARK(3);
  __syncthreads();                                // all waves are done reading the staged rows
  f32x4 r4[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      xb[(w * 16 + i * 4 + b) * 64 + lane] = hl_sel(h, acc[i][b], acc[4 + i][b]);       // the half this wave gives away
      r4[i][b] = hl_sel(h, acc[4 + i][b], acc[i][b]);
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) r4[i][b] += xb[((w ^ 1) * 16 + i * 4 + b) * 64 + lane];
  __syncthreads();
  f32x4 fin[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      xb[(w * 8 + j * 4 + b) * 64 + lane] = hl_sel(q, r4[j][b], r4[2 + j][b]);
      fin[j][b] = hl_sel(q, r4[2 + j][b], r4[j][b]);
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b) fin[j][b] += xb[((w ^ 2) * 8 + j * 4 + b) * 64 + lane];

  HL_MARK(4);
  // epilogue: this wave owns row blocks mt = 4 h + 2 q + j, all 64 columns; fin[j][b][r] = C[row mt * 16 + r16][col b * 16 + 4 kq + r]
  f32x4 cs[4], cq[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (4 * (int)h + 2 * (int)q + j) * 16 + r16;
    if (m >= n) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = b * 16 + 4 * kq;
      f32x4 v = fin[j][b];
      if (addend) v += __builtin_convertvector(*(const bf16x4*)(addend + (long long)m * HL_C + col), f32x4);
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      *(bf16x4*)(out + (long long)m * HL_C + col) = o;
      const f32x4 vr = __builtin_convertvector(o, f32x4);      // statistics of the ROUNDED values: what the BatchNorm behind reads
      if (bn.x) {                                              // input gradient: the BatchNorm-backward sums of the producing layer
        const f32x4 xh = (__builtin_convertvector(xr[j][b], f32x4) - *(const f32x4*)(bn.mean + col)) * *(const f32x4*)(bn.invstd + col);
        f32x4 gm = vr;
        if (bn.relu) {
          f32x4 yv;
          if (bn.y) yv = __builtin_convertvector(yr[j][b], f32x4);
          else yv = xh * *(const f32x4*)(bn.gamma + col) + *(const f32x4*)(bn.beta + col);
#pragma unroll
          for (int r = 0; r < 4; ++r) if (!(yv[r] > 0.f)) gm[r] = 0.f;
        }
        cs[b] += gm;
        cq[b] += gm * xh;
      } else {
        cs[b] += vr;
        cq[b] += vr * vr;
      }
    }
  }
  HL_MARK(5);
  if (stats) {
    __syncthreads();                              // round-two buffers are dead
    float* red = (float*)smem;                    // [4 waves][2][64]
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = hl_row_sum(cs[b][r]), s2 = hl_row_sum(cq[b][r]);
        if (r16 == 0) {
          red[(w * 2 + 0) * HL_C + b * 16 + 4 * kq + r] = s1;
          red[(w * 2 + 1) * HL_C + b * 16 + 4 * kq + r] = s2;
        }
      }
    __syncthreads();
    if (tid < 2 * HL_C) {
      const int which = tid >> 6, cl = tid & 63;
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += (double)red[(k * 2 + which) * HL_C + cl];
      stats[((long long)tile * 2 + which) * HL_C + cl] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 128 -> 128 channels (the stride-8 stage's SparseBasicBlocks, ~54 k rows): same tables, same staging, different split
// ---------------------------------------------------------------------------------------------------------------------------
// 128 x 128 accumulators do not fit a wave, so the four waves split the output COLUMNS (32 each: 64 accumulator registers) and every
// wave walks all 27 offsets - no reduce-scatter; the price is that all four read the same operand rows from LDS (16 ds_read_b128
// per 32 MFMAs).  The 256-byte rows are staged one 64-channel HALF at a time (the stage buffer and the 144 B row stride of the
// 64-channel kernel: two workgroups per CU), and the accumulators run through both halves.  What this saves over the LDS-DMA tiled
// kernel is the operand stream: per 128-row tile that kernel fills 27 x 128 rows x 256 B = 884 KB of rows next to the 884 KB of
// weights, at the ~17 B/clk/CU fill rate that bounds it; here the rows are ~270 x 256 B = 69 KB.
// weights: k_halo_wpack128 = [27][half][nt 8][ks 2][lane 64][8]
__global__ __launch_bounds__(256) void k_halo_wpack128(const u16* __restrict__ src, u16* __restrict__ dst, const u16* const* __restrict__ srcs,
                                                       u16* const* __restrict__ dsts, int kvol) {
  if (srcs) { src = srcs[blockIdx.y]; dst = dsts[blockIdx.y]; }
  const int c = blockIdx.x * 256 + threadIdx.x;   // 16 B chunk of dst
  if (c >= kvol * 2048) return;
  const int lane = c & 63, ks = (c >> 6) & 1, nt = (c >> 7) & 7, half = (c >> 10) & 1, k = c >> 11;
  *(u32x4*)(dst + (long long)c * 8) = *(const u32x4*)(src + k * 16384 + (nt * 16 + (lane & 15)) * 128 + half * 64 + ks * 32 + (lane >> 4) * 8);
}

#define HL_C2 128
__global__ __launch_bounds__(256, 2) void k_subm_halo128(const u16* __restrict__ in, const u16* __restrict__ wgt,
                                                         const int32_t* __restrict__ tile_rows, const u16* __restrict__ loc,
                                                         const int32_t* __restrict__ tile_cnt, const int32_t* __restrict__ n_dev, int n_cap,
                                                         int krev, const u16* __restrict__ addend, u16* __restrict__ out,
                                                         double* __restrict__ stats, int maxs, int kvol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* xs = (u16*)smem;                           // [HL_MAXS][HL_RS]: one 64-channel half of the distinct rows
  const int tid = threadIdx.x;
  const int n = min(*n_dev, n_cap);
  const int live = (n + HL_T - 1) / HL_T;
  if (stats && live + (int)blockIdx.x < (int)gridDim.x && tid < 2 * HL_C2)     // statistics rows of the tiles past the count: zeros
    stats[(long long)(live + blockIdx.x) * 2 * HL_C2 + tid] = 0.0;
  const int tile = u3d_xcd_tile(blockIdx.x, live);      // XCD-contiguous ranges of the live tiles (k_subm_halo64)
  if (tile < 0) return;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = tile * HL_T;
  const int32_t* rows_p = tile_rows + (long long)tile * HL_TRC;
  const int cnt = tile_cnt[tile];
  const int nl = min(cnt, maxs);
  f32x4 acc[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a) { acc[a][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const u16* locp = loc + (long long)tile * HL_K * HL_T + r16 * 8;
  const u16* xl = xs + kq * 8;

#define H2_LOADX(FAST, s, x0, x1)                                                                       \
  if (FAST || (s) < nl) {                                                                               \
    x0 = *(const bf16x8*)(xl + (s) * HL_RS);                                                            \
    x1 = *(const bf16x8*)(xl + (s) * HL_RS + 32);                                                       \
  } else {                                                                                              \
    const u16* g_ = in + (long long)rows_p[s] * HL_C2 + half * 64 + kq * 8;                             \
    x0 = *(const bf16x8*)g_;                                                                            \
    x1 = *(const bf16x8*)(g_ + 32);                                                                     \
  }
#define H2_MFMA(a, x0, x1)                                                                              \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][0], x0, acc[a][b], 0, 0, 0);              \
    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][1], x1, acc[a][b], 0, 0, 0);              \
  }
#define H2_WLOAD(dst, k)                                                                                \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
    dst[b][0] = *(const bf16x8*)(wl + ((long long)(k) * 32 + b * 2 + 0) * 512);                          \
    dst[b][1] = *(const bf16x8*)(wl + ((long long)(k) * 32 + b * 2 + 1) * 512);                          \
  }
#define H2_OFFSET_LOOP(FAST)                                                                            \
  for (int k = 0; k < kvol; ++k) {                                                                      \
    const int kn = k + 1 < kvol ? k + 1 : k;                                                            \
    const u16x8 sln = *(const u16x8*)(locp + (krev ? kvol - 1 - kn : kn) * HL_T);                       \
    bf16x8 wn[2][2];                                                                                    \
    H2_WLOAD(wn, kn)                                                                                    \
    bf16x8 x0, x1, y0, y1;                                                                              \
    H2_LOADX(FAST, (int)sl[0], x0, x1)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    _Pragma("unroll") for (int a = 0; a < 8; a += 2) {                                                   \
      H2_LOADX(FAST, (int)sl[a + 1], y0, y1)                                                            \
      H2_MFMA(a, x0, x1)                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      if (a + 2 < 8) { H2_LOADX(FAST, (int)sl[a + 2 < 8 ? a + 2 : 0], x0, x1) }                         \
      H2_MFMA(a + 1, y0, y1)                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
    sl = sln;                                                                                           \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) { wf[b][0] = wn[b][0]; wf[b][1] = wn[b][1]; }          \
  }

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                    // the first half's reads are done
    {                                             // stage this 64-channel half of the distinct rows (k_subm_halo64's scheme)
      const int part = tid & 7, sb = tid >> 3;
      for (int base = 0; base < nl; base += 288) {
        int idx[9];
        u32x4 v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int slot = base + sb + 32 * i;
          idx[i] = (slot > 0 && slot < nl) ? rows_p[slot] : -1;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          v[i] = (u32x4){0u, 0u, 0u, 0u};
          if (idx[i] >= 0) v[i] = *(const u32x4*)(in + (long long)idx[i] * HL_C2 + half * 64 + part * 8);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int slot = base + sb + 32 * i;
          if (slot < nl) *(u32x4*)(xs + slot * HL_RS + part * 8) = v[i];
        }
      }
    }
    __syncthreads();
    // weights of (offset k, this half, n-tiles 2w and 2w + 1): chunk ((k * 2 + half) * 8 + nt) * 2 + ks
    const u16* wl = wgt + ((long long)(half * 8 + 2 * w) * 2) * 512 + lane * 8;
    u16x8 sl = *(const u16x8*)(locp + (krev ? kvol - 1 : 0) * HL_T);
    bf16x8 wf[2][2];
    H2_WLOAD(wf, 0)
    if (cnt <= maxs) { H2_OFFSET_LOOP(1) } else { H2_OFFSET_LOOP(0) }
  }
#undef H2_LOADX
#undef H2_MFMA
#undef H2_WLOAD
#undef H2_OFFSET_LOOP

  // epilogue: acc[a][b][r] = C[row a * 16 + r16][col (2w + b) * 16 + 4 kq + r]; this wave owns its 32 columns of all 128 rows
  f32x4 cs[2], cq[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int m = m0 + a * 16 + r16;
    if (m >= n) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = (2 * w + b) * 16 + 4 * kq;
      f32x4 v = acc[a][b];
      if (addend) v += __builtin_convertvector(*(const bf16x4*)(addend + (long long)m * HL_C2 + col), f32x4);
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      *(bf16x4*)(out + (long long)m * HL_C2 + col) = o;
      const f32x4 vr = __builtin_convertvector(o, f32x4);      // statistics of the ROUNDED values
      cs[b] += vr;
      cq[b] += vr * vr;
    }
  }
  if (stats) {                                    // a wave owns its columns: row sums by DPP, one f64 per (tile, column) straight out
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = hl_row_sum(cs[b][r]), s2 = hl_row_sum(cq[b][r]);
        if (r16 == 0) {
          const int col = (2 * w + b) * 16 + 4 * kq + r;
          stats[((long long)tile * 2 + 0) * HL_C2 + col] = (double)s1;
          stats[((long long)tile * 2 + 1) * HL_C2 + col] = (double)s2;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient:  dW[k][ci][co] = sum over rows m of  x[nbr_k(m)][ci] * dy[m][co]
// ---------------------------------------------------------------------------------------------------------------------------
// The tiled kernel (k_igemm_wgrad_glds_64, 118 us) gathers 27 x 128 row pieces per tile through LDS-DMA like the forward did.  Here
// the tile's distinct rows are staged once (same tables as the forward), the dy tile once, and BOTH MFMA operands come out of LDS
// by transpose reads (ds_read_b64_tr_b16: the reduction index - the row - is the LDS row).  A transpose read takes a per-lane
// address, so the operand of offset k is read straight from the halo rows at loc[k][m]: no gathered copy, no barrier per offset.
//   * One persistent workgroup of EIGHT waves per CU; wave w owns input-channel block w & 3 (16 of the 64), all four output-channel
//     blocks and 7 offsets (112 accumulator registers): the dy fragments (offset-independent) stay in registers for the tile, an
//     offset costs one 16 B slot load, 8 transpose reads and 16 MFMAs.  A workgroup covers 14 offsets, so two GROUPS of workgroups
//     walk the tiles (0-13 / 14-26: the rows are staged twice per launch, the second time from L2).
//   * Staging is asynchronous: LDS-DMA (buffer_load ... lds, no staging registers - the accumulators leave none) into the OTHER of
//     two stage buffers while the current tile is multiplied; the row indices of the tile after that are already in registers.  A
//     first version that staged through registers between two barriers spent 20 k clocks per tile waiting for 4-5 dependent round
//     trips against 1.8 k of MFMAs (108 us per launch: no better than the tiled kernel).
//   * LDS-DMA writes lane-linearly (8 rows of 128 B per wave instruction), so rows are unpadded; the 16-byte pieces of a row are
//     permuted on the SOURCE side (piece p of LDS row r lands in slot p ^ 2 * ((r >> 1) & 3), as wgrad_narrow.hip) so that the 16
//     rows x 32 B of a transpose read spread over the banks.
//   * One f32 partial [14][64][64] per workgroup, summed in workgroup order by k_halo_wgrad_reduce - deterministic.
//   * Tiles whose distinct rows exceed a stage buffer (424 slots) gather each offset's 128 rows into it instead (plain loads, two
//     barriers per offset; rare: a 128-row tile of block-major rows has 212-419 distinct rows on the bench scenes).
#define HW_MAXS 424
#define HW_G 2                   /* workgroup groups */
#define HW_OPG 14                /* offsets per workgroup (two wave sets of 7) */
#define HW_OPW 7                 /* offsets per wave */
#define HW_XS (HW_MAXS * HL_C)   /* elements of the distinct-row stage */
#define HW_BUF (HW_XS + HL_T * HL_C)
#define HW_WGS 256               /* persistent workgroups: one per CU */
typedef __attribute__((address_space(3))) void* hw_lds_ptr;

__device__ __forceinline__ int hw_sw(int r) { return (r >> 1) & 3; }
// transpose-read fragment of rows r0 + 4g + j and r1 + 4g + j (given as LDS row numbers ra, rb) at 16-column block `blk`
__device__ __forceinline__ bf16x8 hw_trf(const u16* tile, int ra, int rb, int blk, int q) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const u16* p0 = tile + ra * HL_C + (((2 * blk + (q >> 1)) ^ (2 * hw_sw(ra))) * 8) + (q & 1) * 4;
  const u16* p1 = tile + rb * HL_C + (((2 * blk + (q >> 1)) ^ (2 * hw_sw(rb))) * 8) + (q & 1) * 4;
  const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
  const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
  const s16x8 v = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void k_subm_halo_wgrad64(const u16* __restrict__ x, const u16* __restrict__ dy,
                                                           const int32_t* __restrict__ tile_rows, const u16* __restrict__ loc,
                                                           const int32_t* __restrict__ tile_cnt, const int32_t* __restrict__ n_dev,
                                                           int n_cap, float* __restrict__ partial, int maxs) {
  extern __shared__ __attribute__((aligned(16))) u16 hsm[];          // [2][HW_BUF]: distinct rows [HW_MAXS][64] | dy tile [128][64]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, L = lane & 15, j = L >> 2, q = L & 3;
  const int cib = w & 3;                           // input-channel block of this wave
  // the two workgroups that walk the same tiles (offset groups 0 / 1) sit 8 apart in the grid = on the SAME XCD (workgroups go
  // round-robin over the 8 XCDs): the second one's fetch of a tile's rows hits that XCD's L2 instead of going out again
  // (giving every XCD a contiguous slab of tiles on top of that measured slower: 71.5 vs 67.9 us)
  const int n = min(*n_dev, n_cap);
  const int ntiles = (n + HL_T - 1) / HL_T;
  int grp = blockIdx.x % HW_G, slotw = blockIdx.x / HW_G;
  if (gridDim.x % 16 == 0) { grp = (blockIdx.x >> 3) & 1; slotw = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); }
  const int nslot = ((int)gridDim.x - grp + HW_G - 1) / HW_G;
  const int kbase = grp * HW_OPG + (w >> 2) * HW_OPW;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, 0x80000000u, 0x00020000);

  f32x4 acc[HW_OPW][4];
#pragma unroll
  for (int o = 0; o < HW_OPW; ++o)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[o][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int r16p = 4 * g + j;                      // this lane's row within a 16-row block of the transpose reads

  // DMA roles: instruction a of this wave covers LDS rows (a * 8 + w) * 8 .. + 7 (8 waves interleave), lane -> row + lane / 8, piece slot lane % 8
  constexpr int XI = (HW_MAXS / 8 + 7) / 8;        // distinct-row instructions per wave (424 rows = 53 groups of 8 rows: 7 per wave)
  const int lrow = lane >> 3, lpc = lane & 7;
  int idx[XI];                                     // row numbers of the NEXT tile to stage (-1: zero row / beyond the tile's count)
  int cnt_nn = 0;                                  // ... and its number of distinct rows
  auto load_idx = [&](int tile) {
    const bool live = tile < ntiles;
    const int cnt = live ? tile_cnt[tile] : 0;
    cnt_nn = cnt;
    const int32_t* rows_p = tile_rows + (long long)(live ? tile : 0) * HL_TRC;
#pragma unroll
    for (int a = 0; a < XI; ++a) {
      const int slot = (a * 8 + w) * 8 + lrow;
      idx[a] = (cnt <= maxs && slot > 0 && slot < cnt) ? rows_p[slot] : -1;
    }
  };
  auto issue = [&](int tile, int buf, int cnt) {   // distinct rows (from idx[]) and the dy tile of `tile` -> stage buffer `buf`, asynchronously
    u16* base = hsm + buf * HW_BUF;
    const int nrow = cnt <= maxs ? cnt : 0;        // rows past the tile's count are never read: their 8-row groups are not fetched
#pragma unroll                                     // (an LDS-DMA instruction costs the CU ~100-170 clocks whatever it carries)
    for (int a = 0; a < XI; ++a) {
      const int grp8 = a * 8 + w;                  // 8-row group
      if (grp8 * 8 < nrow) {
        const int slot = grp8 * 8 + lrow;
        const unsigned voff = idx[a] >= 0 ? (unsigned)idx[a] * (unsigned)(HL_C * 2) + (unsigned)((lpc ^ (2 * hw_sw(slot))) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (hw_lds_ptr)(base + grp8 * 512), 16, voff, 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {                  // dy: 16 groups of 8 rows, two per wave; rows past n read as zeros
      const int grp8 = a * 8 + w, r = grp8 * 8 + lrow, m = tile * HL_T + r;
      const unsigned voff = (tile < ntiles && m < n) ? (unsigned)m * (unsigned)(HL_C * 2) + (unsigned)((lpc ^ (2 * hw_sw(r))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (hw_lds_ptr)(base + HW_XS + grp8 * 512), 16, voff, 0, 0, 0);
    }
  };

  int cnt_next = 0;
  HW_MARK(0);
  if (slotw < ntiles) {
    load_idx(slotw);
    cnt_next = cnt_nn;
    issue(slotw, 0, cnt_next);
    load_idx(slotw + nslot);
  }
  int it = 0;
  for (int tile = slotw; tile < ntiles; tile += nslot, ++it) {
    const int buf = it & 1;
    u16* xs = hsm + buf * HW_BUF;
    const u16* dys = xs + HW_XS;
    HW_MARK(1 + it * 5);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's share of `tile` has landed, the next tile's indices are here
    __syncthreads();                               // ... everybody's; and the other buffer's readers (previous tile) are done
    HW_MARK(2 + it * 5);
    const int cnt = cnt_next;                      // (read a tile ahead: a load here would stall the first transpose reads)
    cnt_next = cnt_nn;
    issue(tile + nslot, buf ^ 1, cnt_next);
    load_idx(tile + 2 * nslot);
    HW_MARK(3 + it * 5);
    const bool fast = cnt <= maxs;                 // (maxs <= HW_MAXS: the stage buffer; lower only as a test hook)
    const int32_t* rows_p = tile_rows + (long long)tile * HL_TRC;
    bf16x8 bfr[4][4];                              // dy fragments [output-channel block][32-row block]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) bfr[c][mb] = hw_trf(dys, mb * 32 + r16p, mb * 32 + 16 + r16p, c, q);
    __builtin_amdgcn_sched_barrier(0);
    HW_MARK(4 + it * 5);
    const u16* locp = loc + (long long)tile * HL_K * HL_T + r16p * 8;      // this lane's 8 slots of an offset: rows mt * 16 + r16p
    u16x8 sl = *(const u16x8*)(locp + min(kbase, HL_K - 1) * HL_T);
#pragma unroll
    for (int o = 0; o < HW_OPW; ++o) {
      const int k = kbase + o;
      const bool live = k < HL_K;                  // (the last wave set has 6 offsets: its seventh step multiplies zero rows)
      const u16x8 sln = *(const u16x8*)(locp + min(k + 1, HL_K - 1) * HL_T);
      if (!fast) {                                 // rows of offsets (k of wave set 0, k of wave set 1) -> stage rows 0..127 / 128..255
        __syncthreads();                           // the previous offset's reads are done
        const int set = tid >> 8, t2 = tid & 255, m = t2 >> 1, half = t2 & 1;
        const int kk = grp * HW_OPG + set * HW_OPW + o;
        const int s = kk < HL_K ? loc[(long long)tile * HL_K * HL_T + kk * HL_T + (m & 15) * 8 + (m >> 4)] : 0;
        const int row = set * 128 + m;
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (s > 0) {
          const u16* src = x + (long long)rows_p[s] * HL_C + half * 32;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = *(const u32x4*)(src + i * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(xs + row * HL_C + (((half * 4 + i) ^ (2 * hw_sw(row))) * 8)) = v[i];
        __syncthreads();
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int sbase = (w >> 2) * 128;
        const int s0 = fast ? (live ? (int)sl[2 * mb] : 0) : sbase + mb * 32 + r16p;
        const int s1 = fast ? (live ? (int)sl[2 * mb + 1] : 0) : sbase + mb * 32 + 16 + r16p;
        const bf16x8 a = hw_trf(xs, s0, s1, cib, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[c][mb], a, acc[o][c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);           // (or every offset's transpose reads are hoisted to the top: 112 registers of operands)
      sl = sln;
    }
    HW_MARK(5 + it * 5);
  }
  HW_MARK(1 + it * 5);
  __builtin_amdgcn_s_waitcnt(0x0F70);              // (nothing of a DMA issued past the last tile may be in flight at exit)
  // one partial per workgroup: acc[o][c][r] = dW[k = kbase + o][ci = cib * 16 + (lane & 15)][co = c * 16 + 4 * (lane >> 4) + r]
  float* pp = partial + (long long)blockIdx.x * HW_OPG * HL_C * HL_C + (long long)((w >> 2) * HW_OPW) * HL_C * HL_C;
#pragma unroll
  for (int o = 0; o < HW_OPW; ++o)
#pragma unroll
    for (int c = 0; c < 4; ++c) *(f32x4*)(pp + o * HL_C * HL_C + (cib * 16 + L) * HL_C + c * 16 + 4 * g) = acc[o][c];
  HW_MARK(2 + it * 5);
}

// dw[k][ci][co] = sum of the partials of the workgroups of group k / 14 in a FIXED order: thread (chunk c, element group) adds the
// workgroups c, c + 8, ... of the group (independent loads, all in flight), the 8 chunk sums are then added in order through LDS
// (one thread per element group walking 128 partials took 33 us for 59 MB)
__global__ __launch_bounds__(256) void k_halo_wgrad_reduce(const float* __restrict__ partial, int nwg, float* __restrict__ dw) {
  __shared__ f32x4 red[8][32];
  const int el = threadIdx.x & 31, ch = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;                // f32x4 element group; 27 * 1024 of them, a multiple of 32
  const int k = e / (HL_C * HL_C / 4), rest = e % (HL_C * HL_C / 4);
  const int grp = k / HW_OPG, o = k % HW_OPG;
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  // workgroup ids of group `grp` in ascending order (the mapping of k_subm_halo_wgrad64): slot s -> id
  const int nsl = (nwg - grp + HW_G - 1) / HW_G;
  for (int sidx = ch; sidx < nsl; sidx += 8) {
    const int wg = (nwg % 16 == 0) ? ((sidx & 7) + 16 * (sidx >> 3) + 8 * grp) : (sidx * HW_G + grp);
    a += *(const f32x4*)(partial + ((long long)wg * HW_OPG + o) * HL_C * HL_C + rest * 4);
  }
  red[ch][el] = a;
  __syncthreads();
  if (ch == 0) {
#pragma unroll
    for (int c = 1; c < 8; ++c) a += red[c][el];
    *(f32x4*)(dw + (long long)e * 4) = a;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// C ABI (include/u3d_hip.h)
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int32_t u3d_subm_halo_sizes(int32_t n_cap, int64_t* tile_rows_elems, int64_t* loc_elems, int32_t* tiles) {
  U3D_REQUIRE(n_cap > 0 && tile_rows_elems && loc_elems && tiles, U3D_ERR_ARG);
  const int t = u3d_cdiv(n_cap, HL_T);
  *tiles = t;
  *tile_rows_elems = (int64_t)t * HL_TRC;
  *loc_elems = (int64_t)t * HL_K * HL_T;
  return U3D_OK;
}

extern "C" int32_t u3d_subm_halo_build(const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t* tile_rows,
                                       uint16_t* loc, int32_t* tile_cnt, int32_t kvol, u3d_stream s) {
  U3D_REQUIRE(nbr && n_dev && tile_rows && loc && tile_cnt && n_cap > 0 && ld >= n_cap && kvol >= 1 && kvol <= HL_K, U3D_ERR_ARG);
  const int wpt = u3d_cdiv(u3d_cdiv(n_cap, 32), 1024) * 4;      // words per thread, a multiple of 4 (whole prefix groups)
  const int lds = wpt * 256 * 4 + wpt * 64 * 2 + 1024;
  if (lds > 160 * 1024) return U3D_ERR_UNSUPPORTED;      // > 1.15 M rows: the row bitmap does not fit the LDS
  U3D_ALLOW_LDS(k_halo_build, 160 * 1024);                // set once per device: the maximum, the launch asks for what n_cap needs
  k_halo_build<<<u3d_cdiv(n_cap, HL_T), 256, lds, (hipStream_t)s>>>(nbr, ld, kvol, n_dev, n_cap, wpt, tile_rows, loc, tile_cnt);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_subm_halo_wpack(const void* w_nmajor, void* w_packed, u3d_stream s) {
  U3D_REQUIRE(w_nmajor && w_packed, U3D_ERR_ARG);
  k_halo_wpack<<<u3d_cdiv(HL_K * 512, 256), 256, 0, (hipStream_t)s>>>((const u16*)w_nmajor, (u16*)w_packed, nullptr, nullptr);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_subm_halo_wpack_batched(const void* const* srcs_dev, void* const* dsts_dev, int32_t n, u3d_stream s) {
  U3D_REQUIRE(srcs_dev && dsts_dev && n >= 0, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  k_halo_wpack<<<dim3(u3d_cdiv(HL_K * 512, 256), n), 256, 0, (hipStream_t)s>>>(nullptr, nullptr, (const u16* const*)srcs_dev, (u16* const*)dsts_dev);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_subm_halo_conv64_bf16(const void* in, const void* w_packed, const int32_t* tile_rows, const uint16_t* loc,
                                             const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, int32_t krev,
                                             const void* addend, void* out, double* stats, const u3d_bn_epi* bn, int32_t max_slots,
                                             u3d_stream s) {
  U3D_REQUIRE(in && w_packed && tile_rows && loc && tile_cnt && n_dev && out && n_cap > 0, U3D_ERR_ARG);
  U3D_REQUIRE(!bn || (stats && bn->x && bn->mean && bn->invstd && (!bn->relu || bn->y || (bn->gamma && bn->beta))), U3D_ERR_ARG);
  HlBn e;
  if (bn) { e.x = (const u16*)bn->x; e.y = (const u16*)bn->y; e.mean = bn->mean; e.invstd = bn->invstd; e.gamma = bn->gamma; e.beta = bn->beta; e.relu = bn->relu; }
  const int lds = HL_MAXS * HL_RS * 2;
  static_assert(HL_MAXS * HL_RS * 2 >= 4 * 16 * 64 * 16, "stage buffer holds the first reduce-scatter round");
  U3D_ALLOW_LDS(k_subm_halo64, lds);
  k_subm_halo64<<<u3d_cdiv(n_cap, HL_T), 256, lds, (hipStream_t)s>>>((const u16*)in, (const u16*)w_packed, tile_rows, loc, tile_cnt, n_dev,
                                                                    n_cap, krev, (const u16*)addend, (u16*)out, stats, e,
                                                                    (max_slots > 0 && max_slots < HL_MAXS) ? max_slots : HL_MAXS);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int64_t u3d_subm_halo_wgrad64_workspace(void) { return (int64_t)HW_WGS * HW_OPG * HL_C * HL_C * 4; }

extern "C" int32_t u3d_subm_halo_wgrad64_bf16(const void* x, const void* dy, const int32_t* tile_rows, const uint16_t* loc,
                                              const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, float* dw, void* workspace,
                                              int64_t workspace_bytes, int32_t max_slots, u3d_stream s) {
  U3D_REQUIRE(x && dy && tile_rows && loc && tile_cnt && n_dev && dw && workspace && n_cap > 0, U3D_ERR_ARG);
  U3D_REQUIRE(workspace_bytes >= u3d_subm_halo_wgrad64_workspace(), U3D_ERR_WORKSPACE);
  if ((long long)n_cap * HL_C * 2 >= 0x7fffffffll) return U3D_ERR_UNSUPPORTED;      // 32-bit buffer offsets of the LDS-DMA
  constexpr int lds = 2 * HW_BUF * 2;
  static_assert(HW_MAXS >= 2 * HL_T && HW_MAXS % 8 == 0, "the per-offset fall-back stages 2 x 128 rows; whole 8-row DMA groups");
  static_assert(lds <= 160 * 1024, "two stage buffers must fit the LDS");
  U3D_ALLOW_LDS(k_subm_halo_wgrad64, lds);
  int nwg = HW_G * u3d_cdiv(n_cap, HL_T);
  if (nwg > HW_WGS) nwg = HW_WGS;
  k_subm_halo_wgrad64<<<nwg, 512, lds, (hipStream_t)s>>>((const u16*)x, (const u16*)dy, tile_rows, loc, tile_cnt, n_dev, n_cap, (float*)workspace,
                                                          (max_slots > 0 && max_slots < HW_MAXS) ? max_slots : HW_MAXS);
  k_halo_wgrad_reduce<<<HL_K * HL_C * HL_C / 4 / 32, 256, 0, (hipStream_t)s>>>((const float*)workspace, nwg, dw);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_subm_halo_wpack128(const void* w_nmajor, void* w_packed, int32_t kvol, u3d_stream s) {
  U3D_REQUIRE(w_nmajor && w_packed && kvol >= 1 && kvol <= HL_K, U3D_ERR_ARG);
  k_halo_wpack128<<<u3d_cdiv(kvol * 2048, 256), 256, 0, (hipStream_t)s>>>((const u16*)w_nmajor, (u16*)w_packed, nullptr, nullptr, kvol);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_subm_halo_wpack128_batched(const void* const* srcs_dev, void* const* dsts_dev, int32_t n, int32_t kvol, u3d_stream s) {
  U3D_REQUIRE(srcs_dev && dsts_dev && n >= 0 && kvol >= 1 && kvol <= HL_K, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  k_halo_wpack128<<<dim3(u3d_cdiv(kvol * 2048, 256), n), 256, 0, (hipStream_t)s>>>(nullptr, nullptr, (const u16* const*)srcs_dev, (u16* const*)dsts_dev, kvol);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_subm_halo_conv128_bf16(const void* in, const void* w_packed, const int32_t* tile_rows, const uint16_t* loc,
                                              const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, int32_t krev,
                                              const void* addend, void* out, double* stats, int32_t max_slots, int32_t kvol,
                                              u3d_stream s) {
  U3D_REQUIRE(in && w_packed && tile_rows && loc && tile_cnt && n_dev && out && n_cap > 0 && kvol >= 1 && kvol <= HL_K, U3D_ERR_ARG);
  const int lds = HL_MAXS * HL_RS * 2;
  U3D_ALLOW_LDS(k_subm_halo128, lds);
  k_subm_halo128<<<u3d_cdiv(n_cap, HL_T), 256, lds, (hipStream_t)s>>>((const u16*)in, (const u16*)w_packed, tile_rows, loc, tile_cnt, n_dev, n_cap,
                                                                     krev, (const u16*)addend, (u16*)out, stats,
                                                                     (max_slots > 0 && max_slots < HL_MAXS) ? max_slots : HL_MAXS, kvol);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

