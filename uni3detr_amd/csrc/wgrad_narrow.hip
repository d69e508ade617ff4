// bf16 weight gradient of the NARROW sparse levels (cin in {16, 32}, cout in {16, 32, 64}, 27 offsets):
//     dW[t][ci][co] = sum over output rows m of  in[nbr[t][m]][ci] * dout[m][co]
// (ref: the weight half of spconv's indice_conv_backward for the SubMConv3d / SparseConv3d layers of
//  models/pts_encoder/sparse_encoder_hd.py:80-138 on the 16- and 32-channel levels).
//
// The tiled kernel these shapes ran on (k_igemm_wgrad<2,2,1,1>) gives every offset its own workgroups: each of them streams `dout`
// again (PMC: 690-720 MB per launch on the 338 k-row level, 11 x the algorithmic bytes, profiles/r02_pmc_traffic_sparse_levels.csv)
// and does two MFMAs per wave between two workgroup barriers.  Here a workgroup of NINE waves owns a 64-row tile at a time and
// wave w owns the offsets 3w .. 3w+2 for it:
//   * the tile of `dout` goes global -> LDS once per tile for all 27 offsets (LDS-DMA, double buffered, ONE barrier per tile);
//   * every wave gathers the input rows of its own three offsets global -> LDS (LDS-DMA into a wave-private ring: no staging
//     registers, no ds_write, a missing neighbour is an out-of-range offset = zero fill), one tile ahead, and transposes them on the
//     way to the MFMA with ds_read_b64_tr_b16 (the reduction index - the row - is the LDS row for both operands);
//   * three offsets x (cin/16) x (cout/16) accumulators stay in registers over the workgroup's whole (strided, XCD-local) list of
//     tiles: `dout` is read once, the partial sums are written once per workgroup ([workgroups][27][cin][cout] f32) and summed in a
//     fixed order by k_wgrad_narrow_reduce - deterministic.
// LDS-DMA writes lane-linearly (1 KB per wave instruction), so rows are unpadded; bank conflicts of the transpose reads (16 rows
// x 32 B per instruction) are avoided by permuting the 16-byte pieces of a row on the SOURCE side: piece p of tile row r lands in
// piece slot p ^ 2*sw(r), sw(r) = (r>>2)&1 for 64-byte rows, (r>>1)&3 for 128-byte rows (each 16-byte bank column is then hit
// exactly twice by an instruction: the minimum for 512 B), nothing for 32-byte rows.
#include "common.h"

typedef unsigned short u16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define WN_LDS_PTR(p) ((s16x4 __attribute__((address_space(3)))*)(p))

#define WN_K 27
#define WN_WAVES 9
#define WN_TPW 3              /* offsets per wave */
#define WN_MAX_WG 256         /* persistent workgroups = partial sets (one per CU on MI355X) */

template <int C>              // C channels per row (16 / 32 / 64): permutation of the 16-byte pieces of tile row r
__device__ __forceinline__ int wn_sw(int r) { return C == 32 ? ((r >> 2) & 1) : (C == 64 ? ((r >> 1) & 3) : 0); }

// transpose-read fragment: rows k0 + 4g + j and + 16 of column tile T (16 columns) of a [64][C] tile -> 8 reduction values / lane
template <int C>
__device__ __forceinline__ bf16x8 wn_trf(const u16* tile, int k0, int T, int lane) {
  const int g = lane >> 4, L = lane & 15, j = L >> 2, q = L & 3;
  const int row = k0 + 4 * g + j;
  const int piece = (2 * T + (q >> 1)) ^ (2 * wn_sw<C>(row));
  const u16* p0 = tile + row * C + piece * 8 + (q & 1) * 4;
  s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(WN_LDS_PTR(p0));
  s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(WN_LDS_PTR(p0 + 16 * C));      // sw(row + 16) == sw(row)
  s16x8 v = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int CIN, int COUT>
__device__ __forceinline__ void wgrad_narrow_body(const u16* __restrict__ in, const u16* __restrict__ dout, const int* __restrict__ nbr, int ld,
                                                  float* __restrict__ partial, const int* __restrict__ n_out_dev, int n_out_cap) {
  constexpr int CI = CIN / 16, CO = COUT / 16;
  constexpr int XLPR = CIN / 8, XRPI = 64 / XLPR, XNI = 64 / XRPI;        // 16-byte lanes per row, rows per DMA instruction, instructions per tile
  constexpr int DLPR = COUT / 8, DRPI = 64 / DLPR, DNI = 64 / DRPI;
  constexpr int XT = 64 * CIN, DT = 64 * COUT;                            // elements of one staged tile
  static_assert(DNI <= WN_WAVES, "one dout DMA instruction per wave");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* const dbuf = smem;                                                 // [2][64][COUT]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  u16* const xring = smem + 2 * DT + wv * (WN_TPW * XT);                   // this wave's three [64][CIN] slots
  const int n_out = min(*n_out_dev, n_out_cap);

  // tiles of this workgroup: XCD x owns a contiguous slab, its workgroups interleave inside it (gathered rows stay in one L2)
  const int ntiles = (n_out + 63) >> 6;
  const int slab = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const int slab_end = min(ntiles, (xcd + 1) * slab);
  const int first = xcd * slab + slot;

  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rs = __builtin_amdgcn_make_buffer_rsrc((void*)nbr, 0, 0x7FFFFFFC, 0x00020000);
  const unsigned ld4 = (unsigned)ld * 4u;
  constexpr int XSHIFT = CIN == 16 ? 5 : 6;                               // log2(bytes per input row)

  // DMA lane roles.  Gathered tile, instruction a: tile row a*XRPI + lane/XLPR, piece slot lane%XLPR <- source piece slot ^ 2*sw(row)
  int xperm[XNI];
  unsigned xpiece[XNI];
#pragma unroll
  for (int a = 0; a < XNI; ++a) {
    const int r = a * XRPI + lane / XLPR;
    xperm[a] = r * 4;                                                     // ds_bpermute source lane (the lane that loaded row r's index)
    xpiece[a] = (unsigned)(((lane % XLPR) ^ (2 * wn_sw<CIN>(r))) * 16);
  }
  const int drow = wv * DRPI + lane / DLPR;                               // dout tile: wave wv < DNI loads rows wv*DRPI ...
  const unsigned dpiece = (unsigned)(((lane % DLPR) ^ (2 * wn_sw<COUT>(drow))) * 16);

  f32x4 acc[WN_TPW][CI][CO];
#pragma unroll
  for (int j = 0; j < WN_TPW; ++j)
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
      for (int b = 0; b < CO; ++b) acc[j][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto row_mask = [&](int t) -> int { return (t < slab_end && t * 64 + lane < n_out) ? 0 : -1; };
  auto load_idx = [&](int t, int j) -> int {                              // neighbour of (tile row `lane`, offset 3*wv + j), raw
    const unsigned m = (unsigned)max(0, min(t * 64 + lane, n_out - 1));
    return __builtin_amdgcn_raw_buffer_load_b32(nbr_rs, m * 4u, (unsigned)(wv * WN_TPW + j) * ld4, 0);
  };
  auto issue_x = [&](int idx_masked, int j) {                             // gather of one offset's 64 rows into ring slot j
#pragma unroll
    for (int a = 0; a < XNI; ++a) {
      const int idx = __builtin_amdgcn_ds_bpermute(xperm[a], idx_masked);
      const unsigned voff = ((unsigned)idx << XSHIFT) | xpiece[a];        // idx = -1 -> beyond the 2 GB bound -> zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(xring + j * XT + a * 512), 16, voff, 0, 0, 0);
    }
  };
  auto issue_d = [&](int t, int buf) {                                    // rows past n_out (or past the slab) must read as zeros
    if (wv < DNI) {
      const int m = t * 64 + drow;
      const unsigned voff = (t < slab_end && m < n_out) ? (unsigned)m * (unsigned)(COUT * 2) + dpiece : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (lds_void_ptr)(dbuf + buf * DT + wv * 512), 16, voff, 0, 0, 0);
    }
  };

  // prologue: first tile's dout + gathers in flight, indices of the second tile requested
  int idx[WN_TPW];
  if (first < slab_end) {
#pragma unroll
    for (int j = 0; j < WN_TPW; ++j) idx[j] = load_idx(first, j);
    issue_d(first, 0);
    {
      const int mk = row_mask(first);
#pragma unroll
      for (int j = 0; j < WN_TPW; ++j) issue_x(idx[j] | mk, j);
    }
#pragma unroll
    for (int j = 0; j < WN_TPW; ++j) idx[j] = load_idx(first + stride, j);
  }
  int it = 0;
  for (int tile = first; tile < slab_end; tile += stride, ++it) {
    const int buf = it & 1;
    // everything this wave requested for `tile` has landed; the barrier makes the shared dout tile visible and tells the loader
    // waves that the other dout buffer is no longer read
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    __syncthreads();
    issue_d(tile + stride, buf ^ 1);
    const u16* D = dbuf + buf * DT;
    bf16x8 bfr[CO][2];
#pragma unroll
    for (int b = 0; b < CO; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[b][ks] = wn_trf<COUT>(D, ks * 32, b, lane);
    bf16x8 af[WN_TPW][CI][2];
#pragma unroll
    for (int j = 0; j < WN_TPW; ++j)
#pragma unroll
      for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[j][a][ks] = wn_trf<CIN>(xring + j * XT, ks * 32, a, lane);
    // all fragments of this tile are in registers (the MFMAs below wait for them): the ring slots are free for the next tile
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // lgkmcnt(0): the transpose reads have returned
    {
      const int mk = row_mask(tile + stride);
#pragma unroll
      for (int j = 0; j < WN_TPW; ++j) issue_x(idx[j] | mk, j);
#pragma unroll
      for (int j = 0; j < WN_TPW; ++j) idx[j] = load_idx(tile + 2 * stride, j);
    }
#pragma unroll
    for (int j = 0; j < WN_TPW; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < CI; ++a)
#pragma unroll
          for (int b = 0; b < CO; ++b)
            acc[j][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b][ks], af[j][a][ks], acc[j][a][b], 0, 0, 0);
  }
  // acc[j][a][b][r] = dW[offset 3*wv + j][ci a*16 + li][co b*16 + 4g + r]: one 16-byte store per block
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j = 0; j < WN_TPW; ++j) {
    float* p = partial + ((long long)blockIdx.x * WN_K + (wv * WN_TPW + j)) * (CIN * COUT);
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
      for (int b = 0; b < CO; ++b) *(f32x4*)(p + (a * 16 + li) * COUT + b * 16 + 4 * g) = acc[j][a][b];
  }
}

#define U3D_WGRAD_NARROW_KERNEL(NAME, CIN, COUT)                                                                                  \
  __global__ __launch_bounds__(WN_WAVES * 64) void NAME(const u16* in, const u16* dout, const int* nbr, int ld, float* partial,   \
                                                         const int* n_out_dev, int n_out_cap) {                                   \
    wgrad_narrow_body<CIN, COUT>(in, dout, nbr, ld, partial, n_out_dev, n_out_cap);                                                \
  }
U3D_WGRAD_NARROW_KERNEL(k_wgrad_narrow_16x16, 16, 16)
U3D_WGRAD_NARROW_KERNEL(k_wgrad_narrow_16x32, 16, 32)
U3D_WGRAD_NARROW_KERNEL(k_wgrad_narrow_32x32, 32, 32)
U3D_WGRAD_NARROW_KERNEL(k_wgrad_narrow_32x64, 32, 64)

// sum of the `nsplit` partial sets in a FIXED order: a workgroup owns 16 consecutive groups of four outputs; thread (s, gi) adds the
// sets s, s+16, ... of group gi (its loads independent of each other: all in flight), the 16 per-s sums of a group are then added
// in order through LDS.  out_oik: result written as [cout][cin][27] (nn.Conv3d's layout)
__global__ __launch_bounds__(256) void k_wgrad_narrow_reduce(const float* __restrict__ partial, float* __restrict__ dw, int n4, int nsplit,
                                                            int cin, int cout, int out_oik) {
  __shared__ f32x4 red[16][16];
  const int gl = threadIdx.x & 15, s = threadIdx.x >> 4;
  const int gi = blockIdx.x * 16 + gl;
  const long long n = (long long)n4 * 4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (gi < n4) {
#pragma unroll 8
    for (int k = s; k < nsplit; k += 16) a += *(const f32x4*)(partial + (long long)k * n + (long long)gi * 4);
  }
  red[s][gl] = a;
  __syncthreads();
  if (s != 0 || gi >= n4) return;
  f32x4 t4 = red[0][gl];
#pragma unroll
  for (int k = 1; k < 16; ++k) t4 += red[k][gl];
  if (!out_oik) { *(f32x4*)(dw + (long long)gi * 4) = t4; return; }
  const int e = gi * 4, t = e / (cin * cout), ci = (e / cout) % cin, co = e % cout;           // four consecutive co of (t, ci)
#pragma unroll
  for (int r = 0; r < 4; ++r) dw[((long long)(co + r) * cin + ci) * WN_K + t] = t4[r];
}

typedef void (*wgrad_narrow_kernel_t)(const u16*, const u16*, const int*, int, float*, const int*, int);

static int wn_grid(int n_out_cap) {
  const int ntiles = u3d_cdiv(n_out_cap > 0 ? n_out_cap : 1, 64);
  int g = ntiles < WN_MAX_WG ? ntiles : WN_MAX_WG;
  return (g + 7) / 8 * 8;                        // one share per XCD
}
bool u3d_wgrad_narrow_shape(int cin, int cout, int kvol) {
  return kvol == WN_K && (cin == 16 || cin == 32) && (cout == 16 || cout == 32 || (cout == 64 && cin == 32));
}
int64_t u3d_wgrad_narrow_workspace(int n_out_cap, int cin, int cout) { return (int64_t)wn_grid(n_out_cap) * WN_K * cin * cout * 4; }

// 0 = done (dw written), U3D_ERR_UNSUPPORTED = shape not served here
int u3d_launch_wgrad_narrow(const void* in, const void* dout, const int32_t* nbr, int ld, float* dw, const int32_t* n_out_dev, int n_out_cap,
                            int cin, int cout, int kvol, int out_oik, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  if (!nbr || !u3d_wgrad_narrow_shape(cin, cout, kvol)) return U3D_ERR_UNSUPPORTED;
  const int grid = wn_grid(n_out_cap);
  if (workspace_bytes < (int64_t)grid * WN_K * cin * cout * 4) return U3D_ERR_WORKSPACE;
  wgrad_narrow_kernel_t kern = nullptr;
  const size_t lds = (size_t)(2 * 64 * cout + WN_WAVES * WN_TPW * 64 * cin) * 2;
#define WN_PICK(CI_, CO_, NAME) if (cin == CI_ && cout == CO_) { kern = NAME; if (lds > 64 * 1024) U3D_ALLOW_LDS(NAME, lds); }
  WN_PICK(16, 16, k_wgrad_narrow_16x16)
  WN_PICK(16, 32, k_wgrad_narrow_16x32)
  WN_PICK(32, 32, k_wgrad_narrow_32x32)
  WN_PICK(32, 64, k_wgrad_narrow_32x64)
#undef WN_PICK
  if (!kern) return U3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WN_WAVES * 64), lds, s, (const u16*)in, (const u16*)dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap);
  const int n4 = WN_K * cin * cout / 4;
  hipLaunchKernelGGL(k_wgrad_narrow_reduce, dim3(u3d_cdiv(n4, 16)), dim3(256), 0, s, (const float*)workspace, dw, n4, grid, cin, cout, out_oik);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}
