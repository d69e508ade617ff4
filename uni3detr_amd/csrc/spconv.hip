// Sparse convolution on the neighbour table: output-stationary implicit GEMM (forward / dgrad) and
// split-row weight gradient.  gfx950 MFMA: exact-f32 v_mfma_f32_16x16x4_f32, or bf16 16x16x32 with f32
// accumulation.  One output row is written exactly once (no atomics, deterministic).
//
// Algorithmic HBM bytes per call: N_in*Cin*s + N_out*Cout*s + 4*K*ld (table) + K*Cin*Cout*s.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40u);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

#define SP_TILE_M 128   // output rows per workgroup (4 waves x 2 m-tiles of 16)

// ---------------------------------------------------------------------------------------------
// forward / dgrad
//   NT  : 16-column tiles per workgroup column block (cout block = NT*16)
//   KC  : reduction chunk staged per step (16 or 32)
//   BF16: operand type
// LDS: idx[128] | A[128][KC+PAD] | Wt[NT*16][KC+PAD]   (Wt is "n-major": k contiguous per output column)
// ---------------------------------------------------------------------------------------------
template <int NT, int KC, bool BF16>
__global__ __launch_bounds__(256) void k_spconv_fwd(const void* __restrict__ in_, const void* __restrict__ w_,
                                                    const int* __restrict__ nbr, int ld, void* __restrict__ out_,
                                                    const int* __restrict__ n_out_dev, int n_out_cap, int cin, int cout,
                                                    int kvol, int transpose_w) {
  using elem_t = typename std::conditional<BF16, u16, float>::type;
  constexpr int VEC = BF16 ? 8 : 4;          // elements per 16-byte vector
  constexpr int PAD = BF16 ? 8 : 4;
  constexpr int LDA = KC + PAD;              // LDS row stride (elements)
  constexpr int NCOL = NT * 16;
  __shared__ int idx_s[SP_TILE_M];
  __shared__ __attribute__((aligned(16))) elem_t As[SP_TILE_M * LDA];
  __shared__ __attribute__((aligned(16))) elem_t Ws[NCOL * LDA];

  const elem_t* in = (const elem_t*)in_;
  const elem_t* w = (const elem_t*)w_;
  elem_t* out = (elem_t*)out_;

  const int n_out = min(*n_out_dev, n_out_cap);
  const int m0 = blockIdx.x * SP_TILE_M;
  if (m0 >= n_out) return;
  const int col0 = blockIdx.y * NCOL;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;

  f32x4 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kap = 0; kap < kvol; ++kap) {
    // ---- neighbour indices of this tile for offset kap
    int any = 0;
    if (tid < SP_TILE_M) {
      int m = m0 + tid;
      int src = -1;
      if (m < n_out) src = nbr ? nbr[(long long)kap * ld + m] : m;
      idx_s[tid] = src;
      any = src >= 0;
    }
    if (!__syncthreads_or(any)) continue;   // tile has no partner at this offset (also orders idx_s)

    const elem_t* wk = w + (long long)kap * cin * cout;
    for (int c0 = 0; c0 < cin; c0 += KC) {
      // ---- stage A chunk (gathered rows)
      constexpr int AV = SP_TILE_M * (KC / VEC);  // 16-byte vectors in the A chunk
#pragma unroll
      for (int f = tid; f < AV; f += 256) {
        int row = f / (KC / VEC), cv = f % (KC / VEC);
        int src = idx_s[row];
        int c = c0 + cv * VEC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src >= 0 && c < cin) v = *(const uint4*)(in + (long long)src * cin + c);
        *(uint4*)(&As[row * LDA + cv * VEC]) = v;
      }
      // ---- stage W chunk as Wt[n][k]
      if (transpose_w) {
        // global W[kap][n][k] (k contiguous): dgrad
        constexpr int WV = NCOL * (KC / VEC);
        for (int f = tid; f < WV; f += 256) {
          int n = f / (KC / VEC), cv = f % (KC / VEC);
          int c = c0 + cv * VEC;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (col0 + n < cout && c < cin) v = *(const uint4*)(wk + (long long)(col0 + n) * cin + c);
          *(uint4*)(&Ws[n * LDA + cv * VEC]) = v;
        }
      } else {
        // global W[kap][k][n] (n contiguous): read vectors along n, scatter-transpose into LDS
        constexpr int WV = KC * (NCOL / VEC);
        for (int f = tid; f < WV; f += 256) {
          int k = f / (NCOL / VEC), nv = f % (NCOL / VEC);
          int c = c0 + k;
          int n = nv * VEC;
          elem_t tmp[VEC];
          if (c < cin && col0 + n < cout) {
            *(uint4*)tmp = *(const uint4*)(wk + (long long)c * cout + col0 + n);
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) tmp[e] = 0;
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) Ws[(n + e) * LDA + k] = tmp[e];
        }
      }
      __syncthreads();
      // ---- MFMA
      if constexpr (!BF16) {
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
          f32x4 a[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            a[mt] = *(const f32x4*)(&As[(wv * 32 + mt * 16 + li) * LDA + ks * 16 + kq * 4]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4 b = *(const f32x4*)(&Ws[(nt * 16 + li) * LDA + ks * 16 + kq * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][t], b[t], acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][t], b[t], acc[1][nt], 0, 0, 0);
            }
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
          bf16x8 a[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            a[mt] = *(const bf16x8*)(&As[(wv * 32 + mt * 16 + li) * LDA + ks * 32 + kq * 8]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bf16x8 b = *(const bf16x8*)(&Ws[(nt * 16 + li) * LDA + ks * 32 + kq * 8]);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b, acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b, acc[1][nt], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int m = m0 + wv * 32 + mt * 16 + kq * 4 + r;
      if (m >= n_out) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int col = col0 + nt * 16 + li;
        if (col < cout) {
          if constexpr (BF16) out[(long long)m * cout + col] = f32_to_bf16_rne(acc[mt][nt][r]);
          else out[(long long)m * cout + col] = acc[mt][nt][r];
        }
      }
    }
}

template <int NT, int KC, bool BF16>
static int launch_fwd(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev,
                      int n_out_cap, int cin, int cout, int kvol, int tw, hipStream_t s) {
  dim3 grid(u3d_cdiv(n_out_cap, SP_TILE_M), u3d_cdiv(cout, NT * 16));
  hipLaunchKernelGGL((k_spconv_fwd<NT, KC, BF16>), grid, dim3(256), 0, s, in, w, nbr, ld, out, n_out_dev, n_out_cap, cin,
                     cout, kvol, tw);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

template <bool BF16>
static int dispatch_fwd(const void* in, const void* w, const int32_t* nbr, int ld, void* out, const int32_t* n_out_dev,
                        int n_out_cap, int cin, int cout, int kvol, int tw, hipStream_t s) {
  constexpr int KMIN = BF16 ? 32 : 16;
  int nt = cout >= 128 ? 8 : (cout + 15) / 16;   // column block: up to 128 columns
  if (nt > 4 && nt < 8) nt = 8;
  if (nt == 3) nt = 4;
  bool small_k = (!BF16) && cin <= 16;
#define U3D_FWD_CASE(NTv)                                                                                               \
  case NTv:                                                                                                             \
    if (small_k) return launch_fwd<NTv, KMIN, BF16>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, tw, s); \
    return launch_fwd<NTv, 32, BF16>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, tw, s);
  switch (nt) {
    U3D_FWD_CASE(1)
    U3D_FWD_CASE(2)
    U3D_FWD_CASE(4)
    U3D_FWD_CASE(8)
  }
#undef U3D_FWD_CASE
  return U3D_ERR_UNSUPPORTED;
}

extern "C" int32_t u3d_spconv_fwd(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                                  const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                  int32_t transpose_w, int32_t dtype, u3d_stream s) {
  U3D_REQUIRE(in && w && out && n_out_dev, U3D_ERR_ARG);
  U3D_REQUIRE(cin > 0 && cout > 0 && kvol > 0 && (nbr || kvol == 1), U3D_ERR_ARG);
  if (n_out_cap <= 0) return U3D_OK;
  if (dtype == U3D_F32) {
    U3D_REQUIRE(cin % 4 == 0 && cout % 4 == 0, U3D_ERR_UNSUPPORTED);
    return dispatch_fwd<false>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, transpose_w, s);
  } else if (dtype == U3D_BF16) {
    U3D_REQUIRE(cin % 8 == 0 && cout % 8 == 0, U3D_ERR_UNSUPPORTED);
    return dispatch_fwd<true>(in, w, nbr, ld, out, n_out_dev, n_out_cap, cin, cout, kvol, transpose_w, s);
  }
  return U3D_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[kap][ci][co] = sum_m A_kap[m][ci] * dout[m][co]
// grid = (nsplit, kvol, ci_blocks*co_blocks); each workgroup owns a [<=128 ci] x [<=256 co] block of dW[kap]
// and a contiguous range of 32-row tiles; accumulates in registers, writes one partial; a second kernel sums
// the partials in split order (deterministic).
//   wave w: all CIT ci-tiles x COT co-tiles starting at co-tile w*COT.
// ---------------------------------------------------------------------------------------------
#define WG_ROWS 32
template <int CIT, int COT, bool BF16>
__global__ __launch_bounds__(256) void k_spconv_wgrad(const void* __restrict__ in_, const void* __restrict__ dout_,
                                                      const int* __restrict__ nbr, int ld, float* __restrict__ partial,
                                                      const int* __restrict__ n_out_dev, int n_out_cap, int cin, int cout,
                                                      int kvol, int co_blocks) {
  using elem_t = typename std::conditional<BF16, u16, float>::type;
  constexpr int CIB = CIT * 16;        // ci block
  constexpr int COB = COT * 16 * 4;    // co block (4 waves)
  constexpr int LDA = CIB + 4, LDD = COB + 4;
  __shared__ int idx_s[WG_ROWS];
  __shared__ __attribute__((aligned(16))) float As[WG_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float Ds[WG_ROWS * LDD];

  const elem_t* in = (const elem_t*)in_;
  const elem_t* dout = (const elem_t*)dout_;
  const int n_out = min(*n_out_dev, n_out_cap);
  const int nsplit = gridDim.x, split = blockIdx.x, kap = blockIdx.y;
  const int ci0 = (blockIdx.z / co_blocks) * CIB, co0 = (blockIdx.z % co_blocks) * COB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;

  f32x4 acc[CIT][COT];
#pragma unroll
  for (int a = 0; a < CIT; ++a)
#pragma unroll
    for (int b = 0; b < COT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (n_out + WG_ROWS - 1) / WG_ROWS;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

  for (int t = t_begin; t < t_end; ++t) {
    const int m0 = t * WG_ROWS;
    int any = 0;
    if (tid < WG_ROWS) {
      int m = m0 + tid, src = -1;
      if (m < n_out) src = nbr ? nbr[(long long)kap * ld + m] : m;
      idx_s[tid] = src;
      any = src >= 0;
    }
    if (!__syncthreads_or(any)) continue;
    // stage gathered input rows (ci block) and dout rows (co block), converted to f32
    for (int f = tid; f < WG_ROWS * (CIB / 4); f += 256) {
      int row = f / (CIB / 4), c = (f % (CIB / 4)) * 4;
      int src = idx_s[row];
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (src >= 0 && ci0 + c < cin) {
        if constexpr (BF16) {
          uint2 r = *(const uint2*)(in + (long long)src * cin + ci0 + c);
          v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
          v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        } else {
          v = *(const f32x4*)(in + (long long)src * cin + ci0 + c);
        }
      }
      *(f32x4*)(&As[row * LDA + c]) = v;
    }
    for (int f = tid; f < WG_ROWS * (COB / 4); f += 256) {
      int row = f / (COB / 4), c = (f % (COB / 4)) * 4;
      int m = m0 + row;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < n_out && idx_s[row] >= 0 && co0 + c < cout) {
        if constexpr (BF16) {
          uint2 r = *(const uint2*)(dout + (long long)m * cout + co0 + c);
          v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
          v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        } else {
          v = *(const f32x4*)(dout + (long long)m * cout + co0 + c);
        }
      }
      *(f32x4*)(&Ds[row * LDD + c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < WG_ROWS / 4; ++ks) {
      float a[CIT], b[COT];
#pragma unroll
      for (int ct = 0; ct < CIT; ++ct) a[ct] = As[(ks * 4 + kq) * LDA + ct * 16 + li];
#pragma unroll
      for (int ot = 0; ot < COT; ++ot) b[ot] = Ds[(ks * 4 + kq) * LDD + (wv * COT + ot) * 16 + li];
#pragma unroll
      for (int ct = 0; ct < CIT; ++ct)
#pragma unroll
        for (int ot = 0; ot < COT; ++ot)
          acc[ct][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct], b[ot], acc[ct][ot], 0, 0, 0);
    }
    __syncthreads();
  }
  // partial[split][kap][ci][co]
  float* p = partial + ((long long)split * kvol + kap) * cin * cout;
#pragma unroll
  for (int ct = 0; ct < CIT; ++ct)
#pragma unroll
    for (int ot = 0; ot < COT; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = ci0 + ct * 16 + kq * 4 + r;
        int co = co0 + (wv * COT + ot) * 16 + li;
        if (ci < cin && co < cout) p[(long long)ci * cout + co] = acc[ct][ot][r];
      }
}

__global__ void k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, long long n, int nsplit) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += partial[(long long)k * n + i];
  dw[i] = s;
}

static int wgrad_nsplit(int n_out_cap, int kvol, int zblocks) {
  int ntiles = u3d_cdiv(n_out_cap > 0 ? n_out_cap : 1, WG_ROWS);
  int target = 2048 / (kvol * zblocks);
  if (target < 1) target = 1;
  int ns = ntiles < target ? ntiles : target;
  if (ns > 64) ns = 64;
  if (ns < 1) ns = 1;
  return ns;
}
static void wgrad_blocks(int cin, int cout, int* cit, int* cot, int* zblocks, int* co_blocks) {
  int c = (cin + 15) / 16;
  *cit = c >= 8 ? 8 : (c > 4 ? 8 : (c > 2 ? 4 : (c > 1 ? 2 : 1)));
  int o = (cout + 63) / 64;   // co tiles per wave needed to cover cout with 4 waves
  *cot = o >= 4 ? 4 : (o > 2 ? 4 : (o > 1 ? 2 : 1));
  int ci_blocks = u3d_cdiv(cin, *cit * 16);
  *co_blocks = u3d_cdiv(cout, *cot * 64);
  *zblocks = ci_blocks * *co_blocks;
}

extern "C" int64_t u3d_spconv_wgrad_workspace(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol) {
  int cit, cot, zb, cob;
  wgrad_blocks(cin, cout, &cit, &cot, &zb, &cob);
  return (int64_t)wgrad_nsplit(n_out_cap, kvol, zb) * kvol * cin * cout * 4;
}

template <bool BF16>
static int dispatch_wgrad(const void* in, const void* dout, const int32_t* nbr, int ld, float* partial,
                          const int32_t* n_out_dev, int n_out_cap, int cin, int cout, int kvol, int cit, int cot, int ns,
                          int zb, int cob, hipStream_t s) {
  dim3 grid(ns, kvol, zb);
#define U3D_WG_CASE(A, B)                                                                                             \
  if (cit == A && cot == B) {                                                                                         \
    hipLaunchKernelGGL((k_spconv_wgrad<A, B, BF16>), grid, dim3(256), 0, s, in, dout, nbr, ld, partial, n_out_dev,    \
                       n_out_cap, cin, cout, kvol, cob);                                                              \
    return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;                                                 \
  }
  U3D_WG_CASE(1, 1) U3D_WG_CASE(1, 2) U3D_WG_CASE(1, 4)
  U3D_WG_CASE(2, 1) U3D_WG_CASE(2, 2) U3D_WG_CASE(2, 4)
  U3D_WG_CASE(4, 1) U3D_WG_CASE(4, 2) U3D_WG_CASE(4, 4)
  U3D_WG_CASE(8, 1) U3D_WG_CASE(8, 2) U3D_WG_CASE(8, 4)
#undef U3D_WG_CASE
  return U3D_ERR_UNSUPPORTED;
}

extern "C" int32_t u3d_spconv_wgrad(const void* in, const void* dout, const int32_t* nbr, int32_t ld, float* dw,
                                    const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                    int32_t dtype, void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(in && dout && dw && n_out_dev && workspace, U3D_ERR_ARG);
  U3D_REQUIRE(cin > 0 && cout > 0 && kvol > 0 && (nbr || kvol == 1), U3D_ERR_ARG);
  U3D_REQUIRE(cin % 4 == 0 && cout % 4 == 0, U3D_ERR_UNSUPPORTED);
  int cit, cot, zb, cob;
  wgrad_blocks(cin, cout, &cit, &cot, &zb, &cob);
  int ns = wgrad_nsplit(n_out_cap, kvol, zb);
  long long n = (long long)kvol * cin * cout;
  U3D_REQUIRE(workspace_bytes >= (int64_t)ns * n * 4, U3D_ERR_WORKSPACE);
  int rc;
  if (dtype == U3D_F32) rc = dispatch_wgrad<false>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, cit, cot, ns, zb, cob, s);
  else if (dtype == U3D_BF16) rc = dispatch_wgrad<true>(in, dout, nbr, ld, (float*)workspace, n_out_dev, n_out_cap, cin, cout, kvol, cit, cot, ns, zb, cob, s);
  else return U3D_ERR_UNSUPPORTED;
  if (rc != U3D_OK) return rc;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(u3d_cdiv(n, 256)), dim3(256), 0, s, (const float*)workspace, dw, n, ns);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
