// Row-wise ops on sparse feature matrices [n, C]: BatchNorm1d statistics / apply / backward,
// densification.  All HBM-bound streaming kernels: 16-byte accesses, one pass per tensor.
#include "common.h"
#include <type_traits>

typedef unsigned short u16;
__device__ __forceinline__ float ld_elem(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float ld_elem(const u16* p, long long i) { return __uint_as_float(((unsigned)p[i]) << 16); }
__device__ __forceinline__ void st_elem(float* p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void st_elem(u16* p, long long i, float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) { p[i] = (u16)((u >> 16) | 0x40u); return; }
  u += 0x7fffu + ((u >> 16) & 1u);
  p[i] = (u16)(u >> 16);
}

// ---------------------------------------------------------------------------------------------
// column statistics, HBM-streaming: every thread loads 16-byte vectors (4 f32 / 8 bf16 columns of one row); a
// workgroup (256 threads) owns ST_ROWS rows; per-thread f32 partials over <= ST_ROWS/row_lanes rows, then f64 across
// row lanes (LDS) -> partial[block][2][C] (f64); stage 2 sums the blocks in order (deterministic).
// MODE 0: (x, x^2).  MODE 1: g = dy*(relu? y>0), (g, g*xhat).   Requires C % VEC == 0 (else the scalar kernel).
// ---------------------------------------------------------------------------------------------
// rows per workgroup of the statistics kernels: by tensor height and width - 16 for the smallest tensors (128 left the 12 000-row
// dense level with 94 workgroups: 17 us for 25 MB, 11 us at 16), 256 / 512 for the large ones
static inline int st_rows_per_block(int n_cap, int c) {
  if (n_cap >= 131072) return c >= 128 ? 256 : 512;     // measured (kernel trace): 192000x256 35 -> 34 us, x128 22 -> 18, 338532x32 28 -> 14,
  return n_cap >= 32768 ? 128 : 16;                     // and the second stage 9 -> 5 us (half / a quarter of the f64 partials)
}

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<u16> { static constexpr int N = 8; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float* out);
template <>
__device__ __forceinline__ void load_vec<float>(const float* p, float* out) {
  float4 v = *(const float4*)p;
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load_vec<u16>(const u16* p, float* out) {
  uint4 v = *(const uint4*)p;
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { out[2 * i] = __uint_as_float(w[i] << 16); out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// per-thread column parameters (V consecutive columns) as 16-byte loads: one load per 4 columns instead of V strided dword loads
// per array - with ~50 scalar loads per thread the address unit spent ~30 us per launch on parameter fetch alone, whatever N
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }      // column parameters are fetched with 16-byte loads
template <int V>
__device__ __forceinline__ void load_cols(const float* __restrict__ p, int col0, float* out) {
#pragma unroll
  for (int i = 0; i < V; i += 4) {
    const float4 t = *(const float4*)(p + col0 + i);
    out[i] = t.x; out[i + 1] = t.y; out[i + 2] = t.z; out[i + 3] = t.w;
  }
}
template <int V>
__device__ __forceinline__ void load_cols_d(const double* __restrict__ p, int col0, float scale, float* out) {
#pragma unroll
  for (int i = 0; i < V; i += 2) {
    const double2 t = *(const double2*)(p + col0 + i);
    out[i] = (float)t.x * scale; out[i + 1] = (float)t.y * scale;
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_col_stats_vec(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                       const int* __restrict__ n_dev, int n_cap, int c, double* __restrict__ partial,
                                                       const int* __restrict__ row_map, int rows_per_block) {
  constexpr int V = VecOf<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* red = (double*)smem_raw;                 // [2][rl][cw*V] laid out as [which][rowlane][col]
  const int n = min(*n_dev, n_cap);
  const int cv = c / V;                            // vectors per row
  const int cw = cv < 256 ? cv : 256;              // vector-columns handled concurrently
  const int rl = 256 / cw;                         // row lanes
  const int tcol = threadIdx.x % cw, trow = threadIdx.x / cw;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  for (int cb = 0; cb < cv; cb += cw) {
    const int vc = cb + tcol;
    float s0[V], s1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    const bool active = vc < cv && trow < rl;
    if (active) {
      float mu[V], is[V], ga[V], be[V];
      const bool remask = MODE == 1 && relu && y == nullptr;     // ReLU mask recomputed from x (no residual): one tensor less to read
      if (MODE == 1) {
        load_cols<V>(mean, vc * V, mu);
        load_cols<V>(invstd, vc * V, is);
        if (remask) { load_cols<V>(gamma, vc * V, ga); load_cols<V>(beta, vc * V, be); }
      }
#pragma unroll 4
      for (int r = r0 + trow; r < r1; r += rl) {
        const long long o = (long long)r * c + (long long)vc * V;
        float xv[V];
        load_vec<T>(x + o, xv);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < V; ++e) { s0[e] += xv[e]; s1[e] += xv[e] * xv[e]; }
        } else {
          float gv[V], yv[V];
          const long long oy = row_map ? (long long)row_map[r] * c + (long long)vc * V : o;   // y / dy live in the mapped row order
          load_vec<T>(dy + oy, gv);
          if (relu && !remask) load_vec<T>(y + oy, yv);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            float g = gv[e];
            if (remask) yv[e] = (xv[e] - mu[e]) * is[e] * ga[e] + be[e];
            if (relu && !(yv[e] > 0.f)) g = 0.f;
            s0[e] += g;
            s1[e] += g * ((xv[e] - mu[e]) * is[e]);
          }
        }
      }
    }
    const int ncol = cw * V;
    if (trow < rl) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        red[(0 * rl + trow) * ncol + tcol * V + e] = (double)s0[e];
        red[(1 * rl + trow) * ncol + tcol * V + e] = (double)s1[e];
      }
    }
    __syncthreads();
    // sum over the rl row lanes in a FIXED order.  Narrow tensors (16 / 32 channels: rl = 64 ... 128) used to leave this to 2*ncol
    // threads walking rl values each (~8 k clk for 32 outputs: these launches took 12-17 us whatever their bytes); now NP threads
    // per output take every NP-th row lane, and the NP sums are added in order
    const int NP = rl >= 32 ? 8 : 1;                  // wide tensors (rl <= 16): the short walk is cheaper than a second LDS stage + its footprint
    if (NP > 1) {
      double* red2 = red + 2 * rl * ncol;             // [NP][2*ncol]
      for (int i = threadIdx.x; i < 2 * ncol * NP; i += 256) {
        const int o = i % (2 * ncol), pp = i / (2 * ncol), which = o / ncol, col = o % ncol;
        double a = 0.0;
        for (int j = pp; j < rl; j += NP) a += red[(which * rl + j) * ncol + col];
        red2[pp * 2 * ncol + o] = a;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 2 * ncol; i += 256) {
        const int which = i / ncol, col = i % ncol;
        if (cb * V + col < c) {
          double a = 0.0;
          for (int pp = 0; pp < NP; ++pp) a += red2[pp * 2 * ncol + i];
          partial[((long long)blockIdx.x * 2 + which) * c + cb * V + col] = a;
        }
      }
    } else {
      for (int i = threadIdx.x; i < 2 * ncol; i += 256) {
        int which = i / ncol, col = i % ncol;
        if (cb * V + col < c) {
          double a = 0.0;
          for (int j = 0; j < rl; ++j) a += red[(which * rl + j) * ncol + col];
          partial[((long long)blockIdx.x * 2 + which) * c + cb * V + col] = a;
        }
      }
    }
    __syncthreads();
  }
}

// scalar fallback for channel counts that are not a multiple of the vector width
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_col_stats(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                   const int* __restrict__ n_dev, int n_cap, int c, double* __restrict__ partial,
                                                   int rows_per_block) {
  __shared__ double red[2][256];
  const int n = min(*n_dev, n_cap);
  const int cw = c < 256 ? c : 256;
  const int rl = 256 / cw;
  const int tcol = threadIdx.x % cw, trow = threadIdx.x / cw;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  for (int cb = 0; cb < c; cb += cw) {
    int col = cb + tcol;
    double s0 = 0.0, s1 = 0.0;
    if (col < c && trow < rl) {
      float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
      const bool remask = MODE == 1 && relu && y == nullptr;
      if (MODE == 1) { mu = mean[col]; is = invstd[col]; }
      if (remask) { ga = gamma[col]; be = beta[col]; }
      for (int r = r0 + trow; r < r1; r += rl) {
        long long o = (long long)r * c + col;
        if (MODE == 0) {
          float v = ld_elem(x, o);
          s0 += (double)v;
          s1 += (double)v * (double)v;
        } else {
          float g = ld_elem(dy, o);
          float yv = remask ? (ld_elem(x, o) - mu) * is * ga + be : (relu ? ld_elem(y, o) : 1.f);
          if (relu && !(yv > 0.f)) g = 0.f;
          float xh = (ld_elem(x, o) - mu) * is;
          s0 += (double)g;
          s1 += (double)g * (double)xh;
        }
      }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (trow == 0 && col < c) {
      double a0 = 0.0, a1 = 0.0;
      for (int j = 0; j < rl; ++j) { a0 += red[0][j * cw + tcol]; a1 += red[1][j * cw + tcol]; }
      partial[((long long)blockIdx.x * 2 + 0) * c + col] = a0;
      partial[((long long)blockIdx.x * 2 + 1) * c + col] = a1;
    }
    __syncthreads();
  }
}

// stage 2 of both statistics reductions.  partial f64 [blocks][2][C]: a workgroup owns 8 adjacent columns, i.e. the two 64-byte
// lines {sum, sum of squares} x 8 columns of every partial row; 16 threads read one row's two lines, 64 rows are in flight per
// pass, the 64 row-lane sums are combined in a fixed order (deterministic).  (One wave per column with lanes over rows touched a
// different line per lane and used 8 of its 64 bytes: 8x the L2 traffic, ~9 us per launch x 90 launches per step.)
#define FIN_THREADS 1024
#define FIN_LANES (FIN_THREADS / 16)
__device__ __forceinline__ void fin_reduce(const double* __restrict__ partial, int used, int c, int col0, double (*red)[16]) {
  const int v = threadIdx.x & 15, bl = threadIdx.x >> 4;            // v = which * 8 + column in group; bl = row lane
  const int which = v >> 3, col = col0 + (v & 7);
  double a = 0.0;
  if (col < c) {
#pragma unroll 8
    for (int b = bl; b < used; b += FIN_LANES) a += partial[((long long)b * 2 + which) * c + col];
  }
  red[bl][v] = a;
  __syncthreads();
  if (threadIdx.x < 16) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int j = 0; j < FIN_LANES; j += 4) { s0 += red[j][threadIdx.x]; s1 += red[j + 1][threadIdx.x]; s2 += red[j + 2][threadIdx.x]; s3 += red[j + 3][threadIdx.x]; }
    red[0][threadIdx.x] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
}

__global__ __launch_bounds__(FIN_THREADS) void k_col_stats_final(const double* __restrict__ partial, int nblocks, const int* __restrict__ n_dev,
                                                                 int n_cap, int c, double* __restrict__ sums, float* __restrict__ sums_f32,
                                                                 int rows_per_block) {
  __shared__ double red[FIN_LANES][16];
  const int n = min(*n_dev, n_cap);
  int used = rows_per_block > 0 ? (n + rows_per_block - 1) / rows_per_block : nblocks;     // 0: every partial counts (per-wave partials)
  if (used > nblocks) used = nblocks;
  const int col0 = blockIdx.x * 8;
  fin_reduce(partial, used, c, col0, red);
  if (threadIdx.x < 16) {
    const int which = threadIdx.x >> 3, col = col0 + (threadIdx.x & 7);
    if (col < c) {
      const double s = red[0][threadIdx.x];
      sums[which * c + col] = s;
      if (sums_f32) sums_f32[which * c + col] = (float)s;
    }
  }
}

extern "C" int64_t u3d_bn_stats_workspace(int32_t n_cap, int32_t c) {
  return (int64_t)u3d_cdiv(n_cap > 0 ? n_cap : 1, st_rows_per_block(n_cap, c)) * 2 * c * 8;
}

template <int MODE>
static int run_stats(const void* x, const void* dy, const void* y, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int relu, const int32_t* n_dev, int n_cap, int c, int dtype, double* sums, void* ws, int64_t ws_bytes, hipStream_t s,
                     const int32_t* row_map = nullptr, float* sums_f32 = nullptr) {
  U3D_REQUIRE(x && n_dev && sums && ws && c > 0, U3D_ERR_ARG);
  U3D_REQUIRE(!row_map || (dtype == U3D_F32 ? c % 4 == 0 : c % 8 == 0), U3D_ERR_UNSUPPORTED);   // mapped rows: vector kernels only
  if (n_cap <= 0) { hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s); return U3D_OK; }
  U3D_REQUIRE(ws_bytes >= u3d_bn_stats_workspace(n_cap, c), U3D_ERR_WORKSPACE);
  const int rpb = st_rows_per_block(n_cap, c);
  int nb = u3d_cdiv(n_cap, rpb);
  const bool pal = al16(mean) && al16(invstd) && al16(gamma) && al16(beta);      // nullptr counts as aligned
  U3D_REQUIRE(!row_map || pal, U3D_ERR_ARG);
  if (dtype == U3D_F32) {
    if (c % 4 == 0 && pal) {
      int cw = (c / 4) < 256 ? (c / 4) : 256; int rl = 256 / cw;
      size_t lds = (size_t)2 * (rl + (rl >= 32 ? 8 : 0)) * cw * 4 * sizeof(double);
      hipLaunchKernelGGL((k_col_stats_vec<float, MODE>), dim3(nb), dim3(256), lds, s, (const float*)x, (const float*)dy, (const float*)y, mean, invstd, gamma, beta, relu, n_dev, n_cap, c, (double*)ws, row_map, rpb);
    } else {
      hipLaunchKernelGGL((k_col_stats<float, MODE>), dim3(nb), dim3(256), 0, s, (const float*)x, (const float*)dy, (const float*)y, mean, invstd, gamma, beta, relu, n_dev, n_cap, c, (double*)ws, rpb);
    }
  } else if (dtype == U3D_BF16) {
    if (c % 8 == 0 && pal) {
      int cw = (c / 8) < 256 ? (c / 8) : 256; int rl = 256 / cw;
      size_t lds = (size_t)2 * (rl + (rl >= 32 ? 8 : 0)) * cw * 8 * sizeof(double);
      hipLaunchKernelGGL((k_col_stats_vec<u16, MODE>), dim3(nb), dim3(256), lds, s, (const u16*)x, (const u16*)dy, (const u16*)y, mean, invstd, gamma, beta, relu, n_dev, n_cap, c, (double*)ws, row_map, rpb);
    } else {
      hipLaunchKernelGGL((k_col_stats<u16, MODE>), dim3(nb), dim3(256), 0, s, (const u16*)x, (const u16*)dy, (const u16*)y, mean, invstd, gamma, beta, relu, n_dev, n_cap, c, (double*)ws, rpb);
    }
  } else return U3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_col_stats_final, dim3(u3d_cdiv(c, 8)), dim3(FIN_THREADS), 0, s, (const double*)ws, nb, n_dev, n_cap, c, sums, sums_f32, rpb);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_bn_stats(const void* x, const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, double* sums,
                                void* workspace, int64_t workspace_bytes, u3d_stream s) {
  return run_stats<0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_dev, n_cap, c, dtype, sums, workspace, workspace_bytes, s);
}
extern "C" int32_t u3d_bn_bwd_stats(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, int32_t relu, const int32_t* n_dev, int32_t n_cap,
                                    int32_t c, int32_t dtype, double* sums, void* workspace, int64_t workspace_bytes,
                                    const int32_t* row_map, float* sums_f32, u3d_stream s) {
  U3D_REQUIRE(dy && mean && invstd && (!relu || y || (gamma && beta)), U3D_ERR_ARG);
  return run_stats<1>(x, dy, y, mean, invstd, gamma, beta, relu, n_dev, n_cap, c, dtype, sums, workspace, workspace_bytes, s, row_map, sums_f32);
}

// stage 2 alone, for partial sums an input-gradient launch left in its epilogue (u3d_igemm_dgrad_bnstats_bf16 / the halo kernel)
extern "C" int32_t u3d_bn_bwd_finalize_partials(const double* partial, int32_t nblocks, int32_t rows_per_block, const int32_t* n_dev,
                                                int32_t n_cap, int32_t c, double* sums, float* sums_f32, u3d_stream s) {
  U3D_REQUIRE(partial && n_dev && sums && nblocks > 0 && c > 0, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_col_stats_final, dim3(u3d_cdiv(c, 8)), dim3(FIN_THREADS), 0, (hipStream_t)s, partial, nblocks, n_dev, n_cap, c, sums, sums_f32,
                     rows_per_block);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums of a dense [n, C] matrix (bias gradients of the query decoder / head linears), f32 accumulation in a fixed
// order: stage 1 = one workgroup per CS_ROWS rows -> partial[block][C]; stage 2 = one thread per column over the blocks.
// ---------------------------------------------------------------------------------------------
#define CS_ROWS 32
template <typename T, int V>
__global__ __launch_bounds__(256) void k_colsum_partial(const T* __restrict__ x, int n, int c, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int cv = (c + V - 1) / V;
  const int cw = cv < 256 ? cv : 256;
  const int rl = 256 / cw;
  const int tcol = threadIdx.x % cw, trow = threadIdx.x / cw;
  const int r0 = blockIdx.x * CS_ROWS, r1 = min(n, r0 + CS_ROWS);
  for (int cb = 0; cb < cv; cb += cw) {
    const int vc = cb + tcol;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (vc < cv && trow < rl) {
      for (int r = r0 + trow; r < r1; r += rl) {
        const long long o = (long long)r * c + (long long)vc * V;
        float xv[V];
        if constexpr (V == 1) xv[0] = ld_elem(x, o); else load_vec<T>(x + o, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += xv[e];
      }
    }
    if (trow < rl) {
#pragma unroll
      for (int e = 0; e < V; ++e) red[(trow * cw + tcol) * V + e] = s[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cw * V; i += 256) {
      if (cb * V + i < c) {
        float a = 0.f;
        for (int j = 0; j < rl; ++j) a += red[j * cw * V + i];
        partial[(long long)blockIdx.x * c + cb * V + i] = a;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int nb, int c, float* __restrict__ out) {
  int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= c) return;
  float s = 0.f;
  for (int b = lane; b < nb; b += 64) s += partial[(long long)b * c + col];
  s = u3d_wave_sum(s);
  if (lane == 0) out[col] = s;
}
#define U3D_COLSUM_BATCH_MAX 64
struct ColsumBatch { const void* x[U3D_COLSUM_BATCH_MAX]; float* out[U3D_COLSUM_BATCH_MAX]; };
// consecutive batch slots naming the SAME output are summed into it by the group's first slot (mult = group size; 0 = no output of its
// own): the bias of a linear shared by several decoder layers
struct ColsumGroups { unsigned char mult[U3D_COLSUM_BATCH_MAX]; };
template <typename T, int V>
__global__ __launch_bounds__(256) void k_colsum_partial_b(ColsumBatch bt, int n, int c, float* __restrict__ partial, long long pstride) {
  // same body as k_colsum_partial, batch index = blockIdx.y
  const T* x = (const T*)bt.x[blockIdx.y];
  partial += (long long)blockIdx.y * pstride;
  __shared__ float red[256 * 8];
  const int cv = (c + V - 1) / V;
  const int cw = cv < 256 ? cv : 256;
  const int rl = 256 / cw;
  const int tcol = threadIdx.x % cw, trow = threadIdx.x / cw;
  const int r0 = blockIdx.x * CS_ROWS, r1 = min(n, r0 + CS_ROWS);
  for (int cb = 0; cb < cv; cb += cw) {
    const int vc = cb + tcol;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (vc < cv && trow < rl) {
      for (int r = r0 + trow; r < r1; r += rl) {
        const long long o = (long long)r * c + (long long)vc * V;
        float xv[V];
        if constexpr (V == 1) xv[0] = ld_elem(x, o); else load_vec<T>(x + o, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += xv[e];
      }
    }
    if (trow < rl) {
#pragma unroll
      for (int e = 0; e < V; ++e) red[(trow * cw + tcol) * V + e] = s[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cw * V; i += 256) {
      if (cb * V + i < c) {
        float a = 0.f;
        for (int j = 0; j < rl; ++j) a += red[j * cw * V + i];
        partial[(long long)blockIdx.x * c + cb * V + i] = a;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_colsum_final_b(ColsumBatch bt, ColsumGroups gr, const float* __restrict__ partial, int nb, int c,
                                                        long long pstride) {
  int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = gr.mult[blockIdx.y];
  if (col >= c || g == 0) return;
  const float* p = partial + (long long)blockIdx.y * pstride;      // (pstride = nb * c: the members' partial rows follow each other)
  nb *= g;
  float s = 0.f;
  for (int b = lane; b < nb; b += 64) s += p[(long long)b * c + col];
  s = u3d_wave_sum(s);
  if (lane == 0) bt.out[blockIdx.y][col] = s;
}
extern "C" int64_t u3d_colsum_batched_workspace(int32_t count, int32_t n, int32_t c) {
  return (int64_t)count * u3d_cdiv(n > 0 ? n : 1, CS_ROWS) * c * 4;
}
extern "C" int32_t u3d_colsum_batched(const void* const* x, float* const* out, int32_t count, int32_t n, int32_t c, int32_t dtype,
                                      void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(x && out && workspace && count >= 0 && count <= U3D_COLSUM_BATCH_MAX && n > 0 && c > 0, U3D_ERR_ARG);
  if (count == 0) return U3D_OK;
  U3D_REQUIRE(workspace_bytes >= u3d_colsum_batched_workspace(count, n, c), U3D_ERR_WORKSPACE);
  ColsumBatch bt;
  for (int i = 0; i < U3D_COLSUM_BATCH_MAX; ++i) { bt.x[i] = i < count ? x[i] : nullptr; bt.out[i] = i < count ? out[i] : nullptr; }
  const int nb = u3d_cdiv(n, CS_ROWS);
  const long long pstride = (long long)nb * c;
  float* ws = (float*)workspace;
  dim3 grid(nb, count);
  if (dtype == U3D_F32) {
    if (c % 4 == 0) hipLaunchKernelGGL((k_colsum_partial_b<float, 4>), grid, dim3(256), 0, s, bt, n, c, ws, pstride);
    else hipLaunchKernelGGL((k_colsum_partial_b<float, 1>), grid, dim3(256), 0, s, bt, n, c, ws, pstride);
  } else if (dtype == U3D_BF16) {
    if (c % 8 == 0) hipLaunchKernelGGL((k_colsum_partial_b<u16, 8>), grid, dim3(256), 0, s, bt, n, c, ws, pstride);
    else hipLaunchKernelGGL((k_colsum_partial_b<u16, 1>), grid, dim3(256), 0, s, bt, n, c, ws, pstride);
  } else return U3D_ERR_UNSUPPORTED;
  ColsumGroups gr;
  for (int i = 0; i < U3D_COLSUM_BATCH_MAX; ++i) gr.mult[i] = 0;
  for (int i = 0, lead = 0; i < count; ++i) {
    if (i > 0 && out[i] == out[i - 1]) { gr.mult[lead]++; } else { lead = i; gr.mult[i] = 1; }
  }
  hipLaunchKernelGGL(k_colsum_final_b, dim3(u3d_cdiv(c, 4), count), dim3(256), 0, s, bt, gr, (const float*)ws, nb, c, pstride);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int64_t u3d_colsum_workspace(int32_t n, int32_t c) { return (int64_t)u3d_cdiv(n > 0 ? n : 1, CS_ROWS) * c * 4; }
extern "C" int32_t u3d_colsum(const void* x, int32_t n, int32_t c, int32_t dtype, float* out, void* workspace,
                              int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(out && c > 0 && n >= 0, U3D_ERR_ARG);
  if (n == 0) { hipMemsetAsync(out, 0, sizeof(float) * c, s); return U3D_OK; }
  U3D_REQUIRE(x && workspace && workspace_bytes >= u3d_colsum_workspace(n, c), U3D_ERR_WORKSPACE);
  const int nb = u3d_cdiv(n, CS_ROWS);
  float* ws = (float*)workspace;
  if (dtype == U3D_F32) {
    if (c % 4 == 0) hipLaunchKernelGGL((k_colsum_partial<float, 4>), dim3(nb), dim3(256), 0, s, (const float*)x, n, c, ws);
    else hipLaunchKernelGGL((k_colsum_partial<float, 1>), dim3(nb), dim3(256), 0, s, (const float*)x, n, c, ws);
  } else if (dtype == U3D_BF16) {
    if (c % 8 == 0) hipLaunchKernelGGL((k_colsum_partial<u16, 8>), dim3(nb), dim3(256), 0, s, (const u16*)x, n, c, ws);
    else hipLaunchKernelGGL((k_colsum_partial<u16, 1>), dim3(nb), dim3(256), 0, s, (const u16*)x, n, c, ws);
  } else return U3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_colsum_final, dim3(u3d_cdiv(c, 4)), dim3(256), 0, s, (const float*)ws, nb, c, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// sums -> mean / invstd (biased variance) and the running-stat update of nn.BatchNorm1d (unbiased variance), one launch
__global__ void k_bn_finalize(const double* __restrict__ sums, const int* __restrict__ n_dev, int n_cap, int c, float eps,
                              float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                              long long* __restrict__ num_batches, float* __restrict__ mean, float* __restrict__ invstd) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = min(*n_dev, n_cap);
  if (i == 0 && num_batches) *num_batches += 1;
  if (i >= c) return;
  double nn = n > 0 ? (double)n : 1.0;
  double mu = sums[i] / nn;
  double var = sums[c + i] / nn - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)mu;
  invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unbiased = n > 1 ? var * nn / (nn - 1.0) : var;
    running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)mu;
    running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unbiased;
  }
}

extern "C" int32_t u3d_bn_finalize(const double* sums, const int32_t* n_dev, int32_t n_cap, int32_t c, float eps,
                                   float momentum, float* running_mean, float* running_var, int64_t* num_batches,
                                   float* mean, float* invstd, u3d_stream s) {
  U3D_REQUIRE(sums && n_dev && mean && invstd && c > 0 && (!running_mean || running_var), U3D_ERR_ARG);
  hipLaunchKernelGGL(k_bn_finalize, dim3(u3d_cdiv(c, 256)), dim3(256), 0, s, sums, n_dev, n_cap, c, eps, momentum,
                     running_mean, running_var, (long long*)num_batches, mean, invstd);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------
// y = relu?((x-mean)*invstd*gamma + beta (+res))      /     backward apply
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_bn_apply(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res, int relu,
                           T* __restrict__ y, const int* __restrict__ n_dev, int n_cap, int c) {
  long long total = (long long)min(*n_dev, n_cap) * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int col = (int)(i % c);
    float v = (ld_elem(x, i) - mean[col]) * invstd[col] * gamma[col] + beta[col];
    if (res) v += ld_elem(res, i);
    if (relu) v = v > 0.f ? v : 0.f;
    st_elem(y, i, v);
  }
}

template <typename T>
__global__ void k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                               const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const double* __restrict__ sums, int relu, T* __restrict__ dx,
                               T* __restrict__ dres, const int* __restrict__ n_dev, int n_cap, int c) {
  int n = min(*n_dev, n_cap);
  long long total = (long long)n * c;
  float inv_n = n > 0 ? 1.f / (float)n : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int col = (int)(i % c);
    float g = ld_elem(dy, i);
    if (relu) {
      float yv = y ? ld_elem(y, i) : (ld_elem(x, i) - mean[col]) * invstd[col] * gamma[col] + beta[col];
      if (!(yv > 0.f)) g = 0.f;
    }
    float is = invstd[col];
    float xh = (ld_elem(x, i) - mean[col]) * is;
    float mg = (float)sums[col] * inv_n, mgx = (float)sums[c + col] * inv_n;
    st_elem(dx, i, gamma[col] * is * (g - mg - xh * mgx));
    if (dres) st_elem(dres, i, g);
  }
}

// forward statistics in two launches instead of three: stage 2 of the reduction (fin_reduce) and the mean / invstd / running-stat
// update in one kernel
__global__ __launch_bounds__(FIN_THREADS) void k_bn_final_finalize(const double* __restrict__ partial, int nblocks, int rows_per_block,
                                                                   const int* __restrict__ n_dev,
                                                                   int n_cap, int c, float eps, float momentum, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                                   float* __restrict__ mean, float* __restrict__ invstd) {
  __shared__ double red[FIN_LANES][16];
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  const int n = min(*n_dev, n_cap);
  int used = rows_per_block > 0 ? (n + rows_per_block - 1) / rows_per_block : nblocks;     // 0: every partial counts (per-wave partials)
  if (used > nblocks) used = nblocks;
  const int col0 = blockIdx.x * 8;
  fin_reduce(partial, used, c, col0, red);
  const int col = col0 + threadIdx.x;
  if (threadIdx.x >= 8 || col >= c) return;
  const double s0 = red[0][threadIdx.x], s1 = red[0][8 + threadIdx.x];
  const double nn = n > 0 ? (double)n : 1.0;
  const double mu = s0 / nn;
  double var = s1 / nn - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[col] = (float)mu;
  invstd[col] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = n > 1 ? var * nn / (nn - 1.0) : var;
    running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * (float)mu;
    running_var[col] = (1.f - momentum) * running_var[col] + momentum * (float)unbiased;
  }
}
extern "C" int32_t u3d_bn_forward_stats(const void* x, const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, float eps,
                                        float momentum, float* running_mean, float* running_var, int64_t* num_batches, float* mean,
                                        float* invstd, void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(x && n_dev && mean && invstd && workspace && c > 0 && (!running_mean || running_var), U3D_ERR_ARG);
  U3D_REQUIRE(workspace_bytes >= u3d_bn_stats_workspace(n_cap, c), U3D_ERR_WORKSPACE);
  const int rpb = st_rows_per_block(n_cap, c);
  const int nb = n_cap > 0 ? u3d_cdiv(n_cap, rpb) : 0;
  if (nb > 0) {
    const bool f32 = dtype == U3D_F32;
    if (!f32 && dtype != U3D_BF16) return U3D_ERR_UNSUPPORTED;
    const int v = f32 ? 4 : 8;
    if (c % v == 0) {
      int cw = (c / v) < 256 ? (c / v) : 256; int rl = 256 / cw;
      size_t lds = (size_t)2 * (rl + (rl >= 32 ? 8 : 0)) * cw * v * sizeof(double);
      if (f32) hipLaunchKernelGGL((k_col_stats_vec<float, 0>), dim3(nb), dim3(256), lds, s, (const float*)x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_dev, n_cap, c, (double*)workspace, (const int*)nullptr, rpb);
      else hipLaunchKernelGGL((k_col_stats_vec<u16, 0>), dim3(nb), dim3(256), lds, s, (const u16*)x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_dev, n_cap, c, (double*)workspace, (const int*)nullptr, rpb);
    } else {
      if (f32) hipLaunchKernelGGL((k_col_stats<float, 0>), dim3(nb), dim3(256), 0, s, (const float*)x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_dev, n_cap, c, (double*)workspace, rpb);
      else hipLaunchKernelGGL((k_col_stats<u16, 0>), dim3(nb), dim3(256), 0, s, (const u16*)x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_dev, n_cap, c, (double*)workspace, rpb);
    }
  }
  hipLaunchKernelGGL(k_bn_final_finalize, dim3(u3d_cdiv(c, 8)), dim3(FIN_THREADS), 0, s, (const double*)workspace, nb, rpb, n_dev, n_cap, c,
                     eps, momentum, running_mean, running_var, (long long*)num_batches, mean, invstd);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
// statistics already reduced per row block by the producer (u3d_igemm_fwd_stats_bf16): partial f64 [nblocks][2][C], block b covering
// rows [b*rows_per_block, (b+1)*rows_per_block) - only the blocks below the device-side row count are summed
extern "C" int32_t u3d_bn_finalize_partials(const double* partial, int32_t nblocks, int32_t rows_per_block, const int32_t* n_dev,
                                            int32_t n_cap, int32_t c, float eps, float momentum, float* running_mean,
                                            float* running_var, int64_t* num_batches, float* mean, float* invstd, u3d_stream s) {
  U3D_REQUIRE(partial && n_dev && mean && invstd && c > 0 && nblocks >= 0 && rows_per_block >= 0 && (!running_mean || running_var), U3D_ERR_ARG);
  hipLaunchKernelGGL(k_bn_final_finalize, dim3(u3d_cdiv(c, 8)), dim3(FIN_THREADS), 0, s, partial, nblocks, rows_per_block, n_dev, n_cap, c, eps, momentum,
                     running_mean, running_var, (long long*)num_batches, mean, invstd);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// Vectorized forms (C a multiple of the 16-byte vector width and C/V a divisor of 256): a thread keeps ONE column group, so the
// per-column parameters live in registers; 16-byte loads/stores; rows strided over the grid.
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float* v);
template <>
__device__ __forceinline__ void store_vec<float>(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <>
__device__ __forceinline__ void store_vec<u16>(u16* p, const float* v) {
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  f32x8_t f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  *(bf16x8_t*)p = __builtin_convertvector(f, bf16x8_t);        // 4 x v_cvt_pk_bf16_f32 (round to nearest even), one 16-byte store
}

// hi / lo bf16 planes of four f32 values (u3d_split_rows_f32's arithmetic): p[0..3] = bf16(v), p[plane + 0..3] = bf16(v - hi)
__device__ __forceinline__ void store_planes4(u16* p, long long plane, const float* v) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t f = {v[0], v[1], v[2], v[3]};
  const bf16x4_t h = __builtin_convertvector(f, bf16x4_t);
  const bf16x4_t l = __builtin_convertvector(f - __builtin_convertvector(h, f32x4_t), bf16x4_t);
  *(bf16x4_t*)p = h;
  *(bf16x4_t*)(p + plane) = l;
}
// PLANES (f32 rows only, u3d_bn_apply_planes / u3d_bn_bwd_apply_planes): the kernel ALSO leaves its output as the two bf16 planes a
// split-bf16 convolution reads ([2 * n_cap][c]: hi | lo, padding rows n .. n_cap zero) - the separate u3d_split_rows_f32 pass over the
// tensor (one more read + write of every element, 186 launches per `parity` step) disappears for every tensor a BatchNorm produces
template <typename T, bool PLANES = false>
__global__ __launch_bounds__(256) void k_bn_apply_vec(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const T* __restrict__ res, int relu, T* __restrict__ y,
                                                      const int* __restrict__ n_dev, int n_cap, int c, const int* __restrict__ row_map,
                                                      const T* __restrict__ post_add, u16* __restrict__ planes = nullptr) {
  constexpr int V = VecOf<T>::N;
  static_assert(!PLANES || V == 4, "planes are an f32 feature");
  const int n = min(*n_dev, n_cap), cv = c / V, rpb = 256 / cv;
  const int vc = threadIdx.x % cv, rl = threadIdx.x / cv;
  float mu[V], is[V], ga[V], be[V];
  load_cols<V>(mean, vc * V, mu); load_cols<V>(invstd, vc * V, is); load_cols<V>(gamma, vc * V, ga); load_cols<V>(beta, vc * V, be);
  for (int r = blockIdx.x * rpb + rl; r < n; r += gridDim.x * rpb) {
    const long long o = (long long)r * c + (long long)vc * V;
    float xv[V], rv[V], out[V];
    load_vec<T>(x + o, xv);
    if (res) load_vec<T>(res + o, rv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = (xv[e] - mu[e]) * is[e] * ga[e] + be[e];
      if (res) v += rv[e];
      if (relu) v = v > 0.f ? v : 0.f;
      out[e] = v;
    }
    const long long oy = row_map ? (long long)row_map[r] * c + (long long)vc * V : o;
    if (post_add) {                                   // y = act(bn(x)) + post_add (added AFTER the ReLU, in the output's row order)
      float pv[V];
      load_vec<T>(post_add + oy, pv);
#pragma unroll
      for (int e = 0; e < V; ++e) out[e] += pv[e];
    }
    store_vec<T>(y + oy, out);
    if constexpr (PLANES) store_planes4(planes + oy, (long long)n_cap * c, out);
  }
  if constexpr (PLANES) {                             // capacity padding: zero rows of both planes (never a stale NaN pattern)
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = n + blockIdx.x * rpb + rl; r < n_cap; r += gridDim.x * rpb)
      store_planes4(planes + (long long)r * c + (long long)vc * V, (long long)n_cap * c, z);
  }
}

template <typename T, bool PLANES = false>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_vec(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const double* __restrict__ sums, int relu, T* __restrict__ dx,
                                                          T* __restrict__ dres, const int* __restrict__ n_dev, int n_cap, int c,
                                                          const int* __restrict__ row_map, u16* __restrict__ planes = nullptr) {
  constexpr int V = VecOf<T>::N;
  static_assert(!PLANES || V == 4, "planes are an f32 feature");
  const int n = min(*n_dev, n_cap), cv = c / V, rpb = 256 / cv;
  const int vc = threadIdx.x % cv, rl = threadIdx.x / cv;
  const float inv_n = n > 0 ? 1.f / (float)n : 0.f;               // f32: a per-thread f64 divide + 16 f64 multiplies cost as much as the stream
  const bool remask = relu && y == nullptr;
  float mu[V], is[V], ga[V], be[V], mg[V], mgx[V];
  load_cols<V>(mean, vc * V, mu); load_cols<V>(invstd, vc * V, is); load_cols<V>(gamma, vc * V, ga);
  if (remask) load_cols<V>(beta, vc * V, be);
  else {
#pragma unroll
    for (int e = 0; e < V; ++e) be[e] = 0.f;
  }
  load_cols_d<V>(sums, vc * V, inv_n, mg);
  load_cols_d<V>(sums + c, vc * V, inv_n, mgx);
#pragma unroll 2
  for (int r = blockIdx.x * rpb + rl; r < n; r += gridDim.x * rpb) {
    const long long o = (long long)r * c + (long long)vc * V;
    float gv[V], xv[V], yv[V], dxv[V];
    const long long oy = row_map ? (long long)row_map[r] * c + (long long)vc * V : o;
    load_vec<T>(dy + oy, gv);
    load_vec<T>(x + o, xv);
    if (relu && !remask) load_vec<T>(y + oy, yv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float g = gv[e];
      if (remask) yv[e] = (xv[e] - mu[e]) * is[e] * ga[e] + be[e];
      if (relu && !(yv[e] > 0.f)) g = 0.f;
      const float xh = (xv[e] - mu[e]) * is[e];
      dxv[e] = ga[e] * is[e] * (g - mg[e] - xh * mgx[e]);
      gv[e] = g;
    }
    store_vec<T>(dx + o, dxv);
    if constexpr (PLANES) store_planes4(planes + o, (long long)n_cap * c, dxv);      // the planes of dx: what the conv in front reads as dy
    if (dres) store_vec<T>(dres + o, gv);
  }
  if constexpr (PLANES) {
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = n + blockIdx.x * rpb + rl; r < n_cap; r += gridDim.x * rpb)
      store_planes4(planes + (long long)r * c + (long long)vc * V, (long long)n_cap * c, z);
  }
}

static inline bool bn_vec_ok(int c, int v) { return c % v == 0 && (c / v) <= 256 && 256 % (c / v) == 0; }
static inline int bn_vec_grid(int n_cap, int c, int v) {
  int rpb = 256 / (c / v);
  long long b = ((long long)n_cap + rpb - 1) / rpb;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));                // >= 8 waves per SIMD of loads in flight, per-thread parameter setup amortised
}

static inline int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int32_t u3d_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const void* residual, int32_t relu, void* y, const int32_t* n_dev, int32_t n_cap, int32_t c,
                                int32_t dtype, const int32_t* row_map, const void* post_add, u3d_stream s) {
  U3D_REQUIRE(x && mean && invstd && gamma && beta && y && n_dev && c > 0, U3D_ERR_ARG);
  U3D_REQUIRE(!row_map || (!residual && bn_vec_ok(c, dtype == U3D_F32 ? 4 : 8)), U3D_ERR_UNSUPPORTED);
  U3D_REQUIRE(!post_add || bn_vec_ok(c, dtype == U3D_F32 ? 4 : 8), U3D_ERR_UNSUPPORTED);
  const bool pal = al16(mean) && al16(invstd) && al16(gamma) && al16(beta);
  U3D_REQUIRE(!(row_map || post_add) || pal, U3D_ERR_ARG);
  if (n_cap <= 0) return U3D_OK;
  int g = ew_grid((long long)n_cap * c);
  if (dtype == U3D_F32 && bn_vec_ok(c, 4) && pal)
    hipLaunchKernelGGL(k_bn_apply_vec<float>, dim3(bn_vec_grid(n_cap, c, 4)), dim3(256), 0, s, (const float*)x, mean, invstd, gamma, beta, (const float*)residual, relu, (float*)y, n_dev, n_cap, c, row_map, (const float*)post_add);
  else if (dtype == U3D_BF16 && bn_vec_ok(c, 8) && pal)
    hipLaunchKernelGGL(k_bn_apply_vec<u16>, dim3(bn_vec_grid(n_cap, c, 8)), dim3(256), 0, s, (const u16*)x, mean, invstd, gamma, beta, (const u16*)residual, relu, (u16*)y, n_dev, n_cap, c, row_map, (const u16*)post_add);
  else if (dtype == U3D_F32)
    hipLaunchKernelGGL(k_bn_apply<float>, dim3(g), dim3(256), 0, s, (const float*)x, mean, invstd, gamma, beta, (const float*)residual, relu, (float*)y, n_dev, n_cap, c);
  else if (dtype == U3D_BF16)
    hipLaunchKernelGGL(k_bn_apply<u16>, dim3(g), dim3(256), 0, s, (const u16*)x, mean, invstd, gamma, beta, (const u16*)residual, relu, (u16*)y, n_dev, n_cap, c);
  else return U3D_ERR_UNSUPPORTED;
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const double* sums, int32_t relu, void* dx, void* dres,
                                    const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, const int32_t* row_map, u3d_stream s) {
  U3D_REQUIRE(dy && x && mean && invstd && gamma && sums && dx && n_dev && c > 0 && (!relu || y || beta), U3D_ERR_ARG);
  U3D_REQUIRE(!row_map || (!dres && bn_vec_ok(c, dtype == U3D_F32 ? 4 : 8)), U3D_ERR_UNSUPPORTED);
  const bool pal = al16(mean) && al16(invstd) && al16(gamma) && al16(beta) && al16(sums);
  U3D_REQUIRE(!row_map || pal, U3D_ERR_ARG);
  if (n_cap <= 0) return U3D_OK;
  int g = ew_grid((long long)n_cap * c);
  if (dtype == U3D_F32 && bn_vec_ok(c, 4) && pal)
    hipLaunchKernelGGL(k_bn_bwd_apply_vec<float>, dim3(bn_vec_grid(n_cap, c, 4)), dim3(256), 0, s, (const float*)dy, (const float*)y, (const float*)x, mean, invstd, gamma, beta, sums, relu, (float*)dx, (float*)dres, n_dev, n_cap, c, row_map);
  else if (dtype == U3D_BF16 && bn_vec_ok(c, 8) && pal)
    hipLaunchKernelGGL(k_bn_bwd_apply_vec<u16>, dim3(bn_vec_grid(n_cap, c, 8)), dim3(256), 0, s, (const u16*)dy, (const u16*)y, (const u16*)x, mean, invstd, gamma, beta, sums, relu, (u16*)dx, (u16*)dres, n_dev, n_cap, c, row_map);
  else if (dtype == U3D_F32)
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)y, (const float*)x, mean, invstd, gamma, beta, sums, relu, (float*)dx, (float*)dres, n_dev, n_cap, c);
  else if (dtype == U3D_BF16)
    hipLaunchKernelGGL(k_bn_bwd_apply<u16>, dim3(g), dim3(256), 0, s, (const u16*)dy, (const u16*)y, (const u16*)x, mean, invstd, gamma, beta, sums, relu, (u16*)dx, (u16*)dres, n_dev, n_cap, c);
  else return U3D_ERR_UNSUPPORTED;
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// f32 rows only: y (dx) AND its hi / lo bf16 planes [2 * n_cap][c] in one pass (see k_bn_apply_vec<., PLANES>).  U3D_ERR_UNSUPPORTED for
// shapes the vector kernels do not take (the caller then runs u3d_bn_apply + u3d_split_rows_f32).
extern "C" int32_t u3d_bn_apply_planes(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                       const float* residual, int32_t relu, float* y, void* planes, const int32_t* n_dev, int32_t n_cap,
                                       int32_t c, const int32_t* row_map, const float* post_add, u3d_stream s) {
  U3D_REQUIRE(x && mean && invstd && gamma && beta && y && planes && n_dev && c > 0, U3D_ERR_ARG);
  if (!bn_vec_ok(c, 4) || !(al16(mean) && al16(invstd) && al16(gamma) && al16(beta)) || (row_map && residual)) return U3D_ERR_UNSUPPORTED;
  if (n_cap <= 0) return U3D_OK;
  hipLaunchKernelGGL((k_bn_apply_vec<float, true>), dim3(bn_vec_grid(n_cap, c, 4)), dim3(256), 0, s, x, mean, invstd, gamma, beta, residual, relu, y,
                     n_dev, n_cap, c, row_map, post_add, (u16*)planes);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_bn_bwd_apply_planes(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, const double* sums, int32_t relu, float* dx, float* dres,
                                           void* planes, const int32_t* n_dev, int32_t n_cap, int32_t c, const int32_t* row_map, u3d_stream s) {
  U3D_REQUIRE(dy && x && mean && invstd && gamma && sums && dx && planes && n_dev && c > 0 && (!relu || y || beta), U3D_ERR_ARG);
  if (!bn_vec_ok(c, 4) || !(al16(mean) && al16(invstd) && al16(gamma) && al16(beta) && al16(sums)) || (row_map && dres)) return U3D_ERR_UNSUPPORTED;
  if (n_cap <= 0) return U3D_OK;
  hipLaunchKernelGGL((k_bn_bwd_apply_vec<float, true>), dim3(bn_vec_grid(n_cap, c, 4)), dim3(256), 0, s, dy, y, x, mean, invstd, gamma, beta, sums, relu,
                     dx, dres, n_dev, n_cap, c, row_map, (u16*)planes);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------
// dense() / its adjoint: rows <-> channels-last volume [B, Dz, Dy, Dx, C]
// ---------------------------------------------------------------------------------------------
template <bool TO_DENSE>
__global__ void k_dense_rows(const uint32_t* __restrict__ src, const int4* __restrict__ coors, const int* __restrict__ n_dev,
                             int n_cap, int row_words, uint32_t* __restrict__ dst, int dz, int dy, int dx) {
  long long total = (long long)min(*n_dev, n_cap) * row_words;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int r = (int)(t / row_words), w = (int)(t % row_words);
    int4 c = coors[r];
    long long cell = (((long long)c.x * dz + c.y) * dy + c.z) * dx + c.w;
    if (TO_DENSE) dst[cell * row_words + w] = src[t];
    else dst[t] = src[cell * row_words + w];
  }
}

static int run_dense(bool to_dense, const void* a, const int32_t* coors, const int32_t* n_dev, int n_cap, int c, void* b, int dz,
                     int dy, int dx, int dtype, hipStream_t s) {
  U3D_REQUIRE(a && coors && n_dev && b && c > 0, U3D_ERR_ARG);
  int row_bytes = c * (dtype == U3D_BF16 ? 2 : 4);
  U3D_REQUIRE((row_bytes & 3) == 0, U3D_ERR_UNSUPPORTED);
  if (n_cap <= 0) return U3D_OK;
  int rw = row_bytes / 4;
  int g = ew_grid((long long)n_cap * rw);
  if (to_dense) hipLaunchKernelGGL(k_dense_rows<true>, dim3(g), dim3(256), 0, s, (const uint32_t*)a, (const int4*)coors, n_dev, n_cap, rw, (uint32_t*)b, dz, dy, dx);
  else hipLaunchKernelGGL(k_dense_rows<false>, dim3(g), dim3(256), 0, s, (const uint32_t*)a, (const int4*)coors, n_dev, n_cap, rw, (uint32_t*)b, dz, dy, dx);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_to_dense(const void* feat, const int32_t* coors, const int32_t* n_dev, int32_t n_cap, int32_t c,
                                void* dense, int32_t dz, int32_t dy, int32_t dx, int32_t dtype, u3d_stream s) {
  return run_dense(true, feat, coors, n_dev, n_cap, c, dense, dz, dy, dx, dtype, s);
}
extern "C" int32_t u3d_from_dense(const void* dense, const int32_t* coors, const int32_t* n_dev, int32_t n_cap, int32_t c,
                                  void* feat, int32_t dz, int32_t dy, int32_t dx, int32_t dtype, u3d_stream s) {
  return run_dense(false, dense, coors, n_dev, n_cap, c, feat, dz, dy, dx, dtype, s);
}

// ---------------------------------------------------------------------------------------------
// Input gradient of a STRIDED convolution as "per-offset products, then gather":  P = dout @ [W_0^T | W_1^T | ...]  is one plain
// GEMM over the (few) output rows, P[o][kappa*C + c]; then  din[i][c] = sum_kappa P[nbr[kappa][i]][kappa*C + c]  over the offsets
// that reach input row i (nbr = the transposed table, -1 = none).  The output-stationary dgrad kernel instead runs every offset for
// every input row, and with stride s only ~1/s^d of those (row, offset) pairs exist: at stride 4 it spent 15/16 of its MFMAs on
// zero rows.  f32 accumulation over the (<= 2^d) contributing offsets.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_tap_gather_sum(const T* __restrict__ p, const int* __restrict__ nbr, int ld,
                                                        const int* __restrict__ n_dev, int n_cap, int c, int kvol, T* __restrict__ out,
                                                        const T* __restrict__ addend) {
  constexpr int V = VecOf<T>::N;
  const int n = min(*n_dev, n_cap), cv = c / V;
  const long long total = (long long)n * cv;
  const long long prow = (long long)kvol * c;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t / cv), vc = (int)(t % cv);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (addend) load_vec<T>(addend + (long long)i * c + (long long)vc * V, acc);      // e.g. the other branches' input gradients (FanoutToken)
    for (int k = 0; k < kvol; ++k) {
      const int o = nbr[(long long)k * ld + i];
      if (o >= 0) {
        float v[V];
        load_vec<T>(p + (long long)o * prow + (long long)k * c + (long long)vc * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += v[e];
      }
    }
    store_vec<T>(out + (long long)i * c + (long long)vc * V, acc);
  }
}
extern "C" int32_t u3d_tap_gather_sum(const void* p, const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t c,
                                      int32_t kvol, int32_t dtype, void* out, u3d_stream s) {
  return u3d_tap_gather_sum_add(p, nbr, ld, n_dev, n_cap, c, kvol, dtype, nullptr, out, s);
}
extern "C" int32_t u3d_tap_gather_sum_add(const void* p, const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t c,
                                          int32_t kvol, int32_t dtype, const void* addend, void* out, u3d_stream s) {
  U3D_REQUIRE(p && nbr && n_dev && out && c > 0 && kvol > 0, U3D_ERR_ARG);
  if (n_cap <= 0) return U3D_OK;
  if (dtype == U3D_BF16 && c % 8 == 0) {
    int g = ew_grid((long long)n_cap * (c / 8));
    hipLaunchKernelGGL(k_tap_gather_sum<u16>, dim3(g), dim3(256), 0, s, (const u16*)p, nbr, ld, n_dev, n_cap, c, kvol, (u16*)out, (const u16*)addend);
  } else if (dtype == U3D_F32 && c % 4 == 0) {
    int g = ew_grid((long long)n_cap * (c / 4));
    hipLaunchKernelGGL(k_tap_gather_sum<float>, dim3(g), dim3(256), 0, s, (const float*)p, nbr, ld, n_dev, n_cap, c, kvol, (float*)out, (const float*)addend);
  } else return U3D_ERR_UNSUPPORTED;
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}


// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension of a row matrix (+ optional ReLU), forward and backward, for the decoder / head
// (ref: nn.LayerNorm in uni3detr_transformer.py:232-236, uni3detr_head.py:95-125 and the mmcv BaseTransformerLayer norms).
// One wavefront per row (lanes stride over the columns, C <= 1024), 32 rows per workgroup.  Input and output dtypes are independent
// (f32 or bf16): the bf16 output feeds the next GEMM without a cast launch.  Backward emits per-workgroup partial sums of
// dgamma / dbeta ([2][nblocks][C] f32); the caller reduces them (u3d_colsum / u3d_colsum_batched).
// ---------------------------------------------------------------------------------------------
#define LN_ROWS_PER_BLOCK 8        /* 2 rows per wave: 900 workgroups for the decoder's 7200 rows (32 rows/block left 225 on 256 CUs) */
#define LN_MAXJ 16

// NJ > 0: C == 64 * NJ exactly, lane owns NJ CONSECUTIVE columns (vector loads/stores); NJ == 0: any C <= 1024, lanes stride.
template <typename T, int NJ>
__device__ __forceinline__ void ln_load_row(const T* __restrict__ p, int c, int lane, float* v) {
  if constexpr (NJ == 4 && std::is_same<T, float>::value) {
    float4 t = *(const float4*)(p + lane * 4);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (NJ == 4) {
    uint2 t = *(const uint2*)(p + lane * 4);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  } else if constexpr (NJ > 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = ld_elem(p, lane * NJ + j);
  } else {
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int col = lane + 64 * j; v[j] = col < c ? ld_elem(p, col) : 0.f; }
  }
}
template <typename T, int NJ>
__device__ __forceinline__ void ln_store_row(T* __restrict__ p, int c, int lane, const float* v) {
  if constexpr (NJ == 4 && std::is_same<T, float>::value) {
    *(float4*)(p + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (NJ == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    f4 f = {v[0], v[1], v[2], v[3]};
    *(b4*)(p + lane * 4) = __builtin_convertvector(f, b4);
  } else if constexpr (NJ > 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) st_elem(p, lane * NJ + j, v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int col = lane + 64 * j; if (col < c) st_elem(p, col, v[j]); }
  }
}
template <int NJ>
__device__ __forceinline__ int ln_col(int lane, int j) { return NJ > 0 ? lane * NJ + j : lane + 64 * j; }

template <typename TX, typename TY, int NJ>
__global__ __launch_bounds__(256) void k_layernorm_fwd(const TX* __restrict__ x, int n, int c, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int relu, TY* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int J = NJ > 0 ? NJ : LN_MAXJ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ga[J], be[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int col = ln_col<NJ>(lane, j);
    ga[j] = col < c ? gamma[col] : 0.f;
    be[j] = col < c ? beta[col] : 0.f;
  }
  const float inv_c = 1.f / (float)c;
#pragma unroll
  for (int rr = 0; rr < LN_ROWS_PER_BLOCK / 4; ++rr) {
    const int r = blockIdx.x * LN_ROWS_PER_BLOCK + wv * (LN_ROWS_PER_BLOCK / 4) + rr;
    if (r >= n) break;
    float v[J];
    ln_load_row<TX, NJ>(x + (long long)r * c, c, lane, v);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) s += v[j];
    const float mu = u3d_wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float d = ln_col<NJ>(lane, j) < c ? v[j] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(u3d_wave_sum(q) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      float o = (v[j] - mu) * rs * ga[j] + be[j];
      v[j] = relu ? (o > 0.f ? o : 0.f) : o;
    }
    ln_store_row<TY, NJ>(y + (long long)r * c, c, lane, v);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

template <typename TX, typename TY, int NJ>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const TY* __restrict__ dy, const TX* __restrict__ x, int n, int c,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                                       TX* __restrict__ dx, float* __restrict__ partial, int nblocks) {
  constexpr int J = NJ > 0 ? NJ : LN_MAXJ;
  __shared__ float red[2][4][64 * J];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ga[J], be[J], dg[J], db[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int col = ln_col<NJ>(lane, j);
    ga[j] = col < c ? gamma[col] : 0.f;
    be[j] = col < c ? beta[col] : 0.f;
    dg[j] = 0.f; db[j] = 0.f;
  }
  const float inv_c = 1.f / (float)c;
#pragma unroll
  for (int rr = 0; rr < LN_ROWS_PER_BLOCK / 4; ++rr) {
    const int r = blockIdx.x * LN_ROWS_PER_BLOCK + wv * (LN_ROWS_PER_BLOCK / 4) + rr;
    if (r >= n) break;
    const float mu = mean[r], rs = rstd[r];
    float xh[J], gg[J];
    ln_load_row<TX, NJ>(x + (long long)r * c, c, lane, xh);
    ln_load_row<TY, NJ>(dy + (long long)r * c, c, lane, gg);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool in = ln_col<NJ>(lane, j) < c;
      xh[j] = in ? (xh[j] - mu) * rs : 0.f;
      float g = in ? gg[j] : 0.f;
      if (relu && !(xh[j] * ga[j] + be[j] > 0.f)) g = 0.f;
      dg[j] += g * xh[j];
      db[j] += g;
      gg[j] = g * ga[j];
      a += gg[j];
      b += gg[j] * xh[j];
    }
    a = u3d_wave_sum(a) * inv_c;
    b = u3d_wave_sum(b) * inv_c;
#pragma unroll
    for (int j = 0; j < J; ++j) gg[j] = rs * (gg[j] - a - xh[j] * b);
    ln_store_row<TX, NJ>(dx + (long long)r * c, c, lane, gg);
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int col = ln_col<NJ>(lane, j);
    red[0][wv][col < 64 * J ? col : 0] = dg[j];
    red[1][wv][col < 64 * J ? col : 0] = db[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += 256) {
    partial[((long long)0 * nblocks + blockIdx.x) * c + i] = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    partial[((long long)1 * nblocks + blockIdx.x) * c + i] = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
  }
}

extern "C" int32_t u3d_layernorm_blocks(int32_t n) { return u3d_cdiv(n > 0 ? n : 1, LN_ROWS_PER_BLOCK); }

#define LN_FWD_CASE(TX, TY, NJ) hipLaunchKernelGGL((k_layernorm_fwd<TX, TY, NJ>), grid, dim3(256), 0, s, (const TX*)x, n, c, gamma, beta, eps, relu, (TY*)y, mean, rstd)
#define LN_BWD_CASE(TX, TY, NJ) hipLaunchKernelGGL((k_layernorm_bwd<TX, TY, NJ>), grid, dim3(256), 0, s, (const TY*)dy, (const TX*)x, n, c, gamma, beta, mean, rstd, relu, (TX*)dx, partial, nb)
#define LN_DISPATCH(CASE)                                                                                        \
  do {                                                                                                           \
    const bool xf = x_dtype == U3D_F32, yf = y_dtype == U3D_F32;                                                 \
    if ((!xf && x_dtype != U3D_BF16) || (!yf && y_dtype != U3D_BF16)) return U3D_ERR_UNSUPPORTED;                \
    if (c == 256) {                                                                                              \
      if (xf && yf) CASE(float, float, 4); else if (xf) CASE(float, u16, 4); else if (yf) CASE(u16, float, 4); else CASE(u16, u16, 4); \
    } else {                                                                                                     \
      if (xf && yf) CASE(float, float, 0); else if (xf) CASE(float, u16, 0); else if (yf) CASE(u16, float, 0); else CASE(u16, u16, 0); \
    }                                                                                                            \
  } while (0)

extern "C" int32_t u3d_layernorm_fwd(const void* x, int32_t x_dtype, int32_t n, int32_t c, const float* gamma, const float* beta,
                                     float eps, int32_t relu, void* y, int32_t y_dtype, float* mean, float* rstd, u3d_stream s) {
  U3D_REQUIRE(x && gamma && beta && y && mean && rstd && c > 0 && c <= 64 * LN_MAXJ && n >= 0, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  dim3 grid(u3d_layernorm_blocks(n));
  LN_DISPATCH(LN_FWD_CASE);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_layernorm_bwd(const void* dy, int32_t y_dtype, const void* x, int32_t x_dtype, int32_t n, int32_t c,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd, int32_t relu,
                                     void* dx, float* partial, u3d_stream s) {
  U3D_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && partial && c > 0 && c <= 64 * LN_MAXJ && n > 0, U3D_ERR_ARG);
  const int nb = u3d_layernorm_blocks(n);
  dim3 grid(nb);
  LN_DISPATCH(LN_BWD_CASE);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------
// "Skinny" weight gradient of a linear layer whose input OR output has <= 16 features (the heads' final 256 -> 10 / 8 / 1 layers,
// the position encoder's 3 -> 256): dW[n][k] = sum_m dy[m][n] * x[m][k].  hipBLASLt runs these [n x 7200] x [7200 x k] products
// at 20-50 us; here one thread owns one element of the WIDE dimension and keeps the <= 16 accumulators of the skinny one, a
// workgroup covers SK_ROWS rows -> partial f32 [chunks][n*k]; the caller column-sums the chunks (u3d_colsum / u3d_colsum_batched).
// ---------------------------------------------------------------------------------------------
#define SK_ROWS 32              /* rows per workgroup: 7200 rows -> 225 workgroups per column block (128 rows left 3/4 of the CUs idle) */
#define SK_MAX 32
template <bool DY_SKINNY>
__device__ __forceinline__ void skinny_wgrad_body(const u16* __restrict__ dy, const u16* __restrict__ x, int m, int n, int k,
                                                  float* __restrict__ partial) {
  // DY_SKINNY: n <= SK_MAX, thread = column of x (k wide);  else k <= SK_MAX, thread = column of dy (n wide)
  const int wide = DY_SKINNY ? k : n, small = DY_SKINNY ? n : k;
  const int col = blockIdx.y * 256 + threadIdx.x;
  const int r0 = blockIdx.x * SK_ROWS, r1 = min(m, r0 + SK_ROWS);
  __shared__ float sk[SK_ROWS][SK_MAX];                 // the skinny operand of this row chunk (zero rows past the end)
  const u16* skp = DY_SKINNY ? dy : x;
  for (int i = threadIdx.x; i < SK_ROWS * small; i += 256) {
    const int rr = i / small, s = i % small;
    sk[rr][s] = r0 + rr < r1 ? ld_elem(skp, (long long)(r0 + rr) * small + s) : 0.f;
  }
  __syncthreads();
  float acc[SK_MAX];
#pragma unroll
  for (int s = 0; s < SK_MAX; ++s) acc[s] = 0.f;
  if (col < wide) {
    const u16* wp = DY_SKINNY ? x : dy;
    float v[SK_ROWS];
#pragma unroll
    for (int rr = 0; rr < SK_ROWS; ++rr) v[rr] = ld_elem(wp, (long long)min(r0 + rr, r1 - 1) * wide + col);     // all loads in flight at once
#pragma unroll
    for (int rr = 0; rr < SK_ROWS; ++rr)
#pragma unroll
      for (int s = 0; s < SK_MAX; ++s)
        if (s < small) acc[s] += sk[rr][s] * v[rr];
    float* p = partial + (long long)blockIdx.x * n * k;
#pragma unroll
    for (int s = 0; s < SK_MAX; ++s)
      if (s < small) {
        if (DY_SKINNY) p[(long long)s * k + col] = acc[s];      // dW[n = s][k = col]
        else p[(long long)col * k + s] = acc[s];                // dW[n = col][k = s]
      }
  }
}
template <bool DY_SKINNY>
__global__ __launch_bounds__(256) void k_skinny_wgrad(const u16* __restrict__ dy, const u16* __restrict__ x, int m, int n, int k,
                                                      float* __restrict__ partial) {
  skinny_wgrad_body<DY_SKINNY>(dy, x, m, n, k, partial);
}
// `count` independent skinny products over the same m rows in ONE launch (blockIdx.z = product; operands arrive by value in the
// kernel arguments: capturable as is).  The decoder's backward has 15 of them per step, 20-27 us each when launched one by one.
#define SKB_MAX 32
struct SkinnyBatch { const u16* dy[SKB_MAX]; const u16* x[SKB_MAX]; float* partial[SKB_MAX]; int n[SKB_MAX]; int k[SKB_MAX]; };
template <bool DY_SKINNY>
__global__ __launch_bounds__(256) void k_skinny_wgrad_b(SkinnyBatch bt, int m) {
  const int b = blockIdx.z;
  const int n = bt.n[b], k = bt.k[b];
  if ((int)blockIdx.y * 256 >= (DY_SKINNY ? k : n)) return;          // the grid is sized for the widest product (uniform exit)
  skinny_wgrad_body<DY_SKINNY>(bt.dy[b], bt.x[b], m, n, k, bt.partial[b]);
}
extern "C" int32_t u3d_skinny_wgrad_chunks(int32_t m) { return u3d_cdiv(m > 0 ? m : 1, SK_ROWS); }
extern "C" int32_t u3d_skinny_wgrad_batched(const void* const* dy, const void* const* x, float* const* partial, const int32_t* n,
                                            const int32_t* k, int32_t count, int32_t m, int32_t dy_skinny, u3d_stream s) {
  U3D_REQUIRE(dy && x && partial && n && k && count > 0 && count <= SKB_MAX && m > 0, U3D_ERR_ARG);
  SkinnyBatch bt;
  int wide = 0;
  for (int i = 0; i < count; ++i) {
    U3D_REQUIRE(dy[i] && x[i] && partial[i] && n[i] > 0 && k[i] > 0 && (dy_skinny ? n[i] : k[i]) <= SK_MAX, U3D_ERR_ARG);
    bt.dy[i] = (const u16*)dy[i]; bt.x[i] = (const u16*)x[i]; bt.partial[i] = partial[i]; bt.n[i] = n[i]; bt.k[i] = k[i];
    const int w = dy_skinny ? k[i] : n[i];
    if (w > wide) wide = w;
  }
  const dim3 grid(u3d_skinny_wgrad_chunks(m), u3d_cdiv(wide, 256), count);
  if (dy_skinny) hipLaunchKernelGGL(k_skinny_wgrad_b<true>, grid, dim3(256), 0, s, bt, m);
  else hipLaunchKernelGGL(k_skinny_wgrad_b<false>, grid, dim3(256), 0, s, bt, m);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_skinny_wgrad_bf16(const void* dy, const void* x, int32_t m, int32_t n, int32_t k, float* partial, u3d_stream s) {
  U3D_REQUIRE(dy && x && partial && m > 0 && n > 0 && k > 0, U3D_ERR_ARG);
  if (n > SK_MAX && k > SK_MAX) return U3D_ERR_UNSUPPORTED;
  const int chunks = u3d_skinny_wgrad_chunks(m);
  if (n <= SK_MAX) hipLaunchKernelGGL(k_skinny_wgrad<true>, dim3(chunks, u3d_cdiv(k, 256)), dim3(256), 0, s, (const u16*)dy, (const u16*)x, m, n, k, partial);
  else hipLaunchKernelGGL(k_skinny_wgrad<false>, dim3(chunks, u3d_cdiv(n, 256)), dim3(256), 0, s, (const u16*)dy, (const u16*)x, m, n, k, partial);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
