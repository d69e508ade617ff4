// Geometry of the sparse levels: occupancy lattice (BitGrid), neighbour tables, hard voxelization.
// All integer work; results are bit-exact against oracle/geometry.py.
#include "common.h"

// ============================================================================================
// BitGrid
// ============================================================================================
extern "C" int64_t u3d_bitgrid_nwords(int32_t batch, int32_t dz, int32_t dy, int32_t dx) {
  return (int64_t)batch * ((dz + 3) / 4) * ((dy + 3) / 4) * ((dx + 3) / 4);
}
extern "C" int64_t u3d_bitgrid_nwords_layout(int32_t batch, int32_t dz, int32_t dy, int32_t dx, int32_t layout) {
  if (layout == 1) return ((int64_t)batch * dz * dy * dx + 63) / 64;
  return u3d_bitgrid_nwords(batch, dz, dy, dx);
}
static inline long long grid_nwords(const u3d_bitgrid* g) { return u3d_bitgrid_nwords_layout(g->batch, g->dz, g->dy, g->dx, g->layout); }

__global__ void k_bitgrid_mark(BitGridDev g, unsigned long long* words, const int4* __restrict__ coors, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coors[i];  // (b,z,y,x)
  if (c.x < 0 || c.x >= g.B) return;
  if ((unsigned)c.y >= (unsigned)g.Dz || (unsigned)c.z >= (unsigned)g.Dy || (unsigned)c.w >= (unsigned)g.Dx) return;
  atomicOr(&words[u3d_word_index(g, c.x, c.y, c.z, c.w)], 1ull << u3d_bit_index(g, c.x, c.y, c.z, c.w));
}

extern "C" int32_t u3d_bitgrid_mark(const u3d_bitgrid* g, const int32_t* coors, int32_t n, u3d_stream s) {
  U3D_REQUIRE(g && g->words && coors, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  BitGridDev d = u3d_make_grid(g);
  hipLaunchKernelGGL(k_bitgrid_mark, dim3(u3d_cdiv(n, 256)), dim3(256), 0, s, d, (unsigned long long*)g->words,
                     (const int4*)coors, n);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

struct Conv3 { int k[3], s[3], p[3]; };

__global__ void k_bitgrid_mark_strided(BitGridDev g, unsigned long long* words, const int4* __restrict__ coors,
                                       const int* __restrict__ n_dev, int n_cap, Conv3 cv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = min(*n_dev, n_cap);
  if (i >= n) return;
  int4 c = coors[i];
  for (int kz = 0; kz < cv.k[0]; ++kz) {
    int tz = c.y + cv.p[0] - kz;
    if (tz < 0 || tz % cv.s[0]) continue;
    int oz = tz / cv.s[0];
    if (oz >= g.Dz) continue;
    for (int ky = 0; ky < cv.k[1]; ++ky) {
      int ty = c.z + cv.p[1] - ky;
      if (ty < 0 || ty % cv.s[1]) continue;
      int oy = ty / cv.s[1];
      if (oy >= g.Dy) continue;
      for (int kx = 0; kx < cv.k[2]; ++kx) {
        int tx = c.w + cv.p[2] - kx;
        if (tx < 0 || tx % cv.s[2]) continue;
        int ox = tx / cv.s[2];
        if (ox >= g.Dx) continue;
        atomicOr(&words[u3d_word_index(g, c.x, oz, oy, ox)], 1ull << u3d_bit_index(g, c.x, oz, oy, ox));
      }
    }
  }
}

// The same marking with the atomics aggregated per workgroup.  Rows are in block-major order, so the 256 voxels of a workgroup
// come from a handful of 4x4x4 input blocks and their (up to 27 each) targets fall into a few dozen output WORDS: the plain kernel
// sends 0.4-1.1 M global atomics per launch, most of them onto words another lane is hitting too (60-99 us, the cost IS the L2
// atomic unit).  Here every target first goes into a small open-addressing table in LDS (key = word index: ds_cmpst; bits: ds_or),
// and only the table's occupied slots go to global memory - one atomic per distinct word and workgroup.  A full table (never seen
// on real levels) falls back to the direct global atomic.
#define MK_SLOTS 512
__global__ __launch_bounds__(256) void k_bitgrid_mark_strided_agg(BitGridDev g, unsigned long long* words, const int4* __restrict__ coors,
                                                                  const int* __restrict__ n_dev, int n_cap, Conv3 cv) {
  __shared__ unsigned long long keys[MK_SLOTS];        // word index + 1 (0 = empty)
  __shared__ unsigned long long bits[MK_SLOTS];
  for (int i = threadIdx.x; i < MK_SLOTS; i += 256) { keys[i] = 0ull; bits[i] = 0ull; }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(*n_dev, n_cap);
  if (i < n) {
    const int4 c = coors[i];
    for (int kz = 0; kz < cv.k[0]; ++kz) {
      const int tz = c.y + cv.p[0] - kz;
      if (tz < 0 || tz % cv.s[0]) continue;
      const int oz = tz / cv.s[0];
      if (oz >= g.Dz) continue;
      for (int ky = 0; ky < cv.k[1]; ++ky) {
        const int ty = c.z + cv.p[1] - ky;
        if (ty < 0 || ty % cv.s[1]) continue;
        const int oy = ty / cv.s[1];
        if (oy >= g.Dy) continue;
        for (int kx = 0; kx < cv.k[2]; ++kx) {
          const int tx = c.w + cv.p[2] - kx;
          if (tx < 0 || tx % cv.s[2]) continue;
          const int ox = tx / cv.s[2];
          if (ox >= g.Dx) continue;
          const unsigned long long w = (unsigned long long)u3d_word_index(g, c.x, oz, oy, ox);
          const unsigned long long bit = 1ull << u3d_bit_index(g, c.x, oz, oy, ox);
          unsigned h = (unsigned)((w * 0x9E3779B97F4A7C15ull) >> 55) & (MK_SLOTS - 1);
          bool done = false;
          for (int probe = 0; probe < 16 && !done; ++probe) {
            const unsigned long long prev = atomicCAS(&keys[h], 0ull, w + 1ull);
            if (prev == 0ull || prev == w + 1ull) { atomicOr(&bits[h], bit); done = true; }
            else h = (h + 1) & (MK_SLOTS - 1);
          }
          if (!done) atomicOr(&words[w], bit);
        }
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < MK_SLOTS; j += 256) {
    const unsigned long long k = keys[j];
    if (k) atomicOr(&words[k - 1ull], bits[j]);
  }
}

extern "C" int32_t u3d_bitgrid_mark_strided(const u3d_bitgrid* g, const int32_t* in_coors, const int32_t* n_dev,
                                            int32_t n_cap, const int32_t ksize[3], const int32_t stride[3],
                                            const int32_t pad[3], u3d_stream s) {
  U3D_REQUIRE(g && g->words && in_coors && n_dev, U3D_ERR_ARG);
  if (n_cap <= 0) return U3D_OK;
  Conv3 cv;
  for (int i = 0; i < 3; ++i) { cv.k[i] = ksize[i]; cv.s[i] = stride[i]; cv.p[i] = pad[i]; U3D_REQUIRE(stride[i] > 0 && ksize[i] > 0, U3D_ERR_ARG); }
  BitGridDev d = u3d_make_grid(g);
#ifndef BITGRID_MARK_AGG
#define BITGRID_MARK_AGG 1
#endif
#if BITGRID_MARK_AGG
  hipLaunchKernelGGL(k_bitgrid_mark_strided_agg, dim3(u3d_cdiv(n_cap, 256)), dim3(256), 0, s, d,
                     (unsigned long long*)g->words, (const int4*)in_coors, n_dev, n_cap, cv);
#else
  hipLaunchKernelGGL(k_bitgrid_mark_strided, dim3(u3d_cdiv(n_cap, 256)), dim3(256), 0, s, d,
                     (unsigned long long*)g->words, (const int4*)in_coors, n_dev, n_cap, cv);
#endif
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---- exclusive popcount scan: 3 launches (chunk sums, scan of chunk sums, rescan + offset) -------
#define SCAN_TPB 256
#define SCAN_ITEMS 16
#define SCAN_CHUNK (SCAN_TPB * SCAN_ITEMS)

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* lds, unsigned* total) {
  // 256 threads = 4 waves
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) lds[wid] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wid; ++w) base += lds[w];
  if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + incl - v;
}

__global__ void k_scan_chunksum(const unsigned long long* __restrict__ words, long long nwords, unsigned* chunk_sum) {
  __shared__ unsigned lds[4];
  long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    long long w = base + j;
    if (w < nwords) c += __popcll(words[w]);
  }
  unsigned tot;
  block_exclusive_scan(c, lds, &tot);
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = tot;
}

__global__ void k_scan_top(unsigned* chunk_sum, int nchunks) {
  // single workgroup, sequential over tiles of 256
  __shared__ unsigned lds[4];
  __shared__ unsigned carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nchunks; base += SCAN_TPB) {
    int i = base + threadIdx.x;
    unsigned v = i < nchunks ? chunk_sum[i] : 0u;
    unsigned tot;
    unsigned ex = block_exclusive_scan(v, lds, &tot);
    unsigned carry = carry_s;
    if (i < nchunks) chunk_sum[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_sum[nchunks] = carry_s;
}

__global__ void k_scan_final(const unsigned long long* __restrict__ words, long long nwords,
                             const unsigned* __restrict__ chunk_off, unsigned* __restrict__ prefix, int nchunks) {
  __shared__ unsigned lds[4];
  long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned pc[SCAN_ITEMS];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    long long w = base + j;
    pc[j] = (w < nwords) ? (unsigned)__popcll(words[w]) : 0u;
    c += pc[j];
  }
  unsigned ex = block_exclusive_scan(c, lds, nullptr) + chunk_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    long long w = base + j;
    if (w < nwords) prefix[w] = ex;
    ex += pc[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) prefix[nwords] = chunk_off[nchunks];
}

extern "C" int64_t u3d_bitgrid_scan_scratch(int64_t nwords) { return (nwords + SCAN_CHUNK - 1) / SCAN_CHUNK + 1; }

extern "C" int32_t u3d_bitgrid_scan(const u3d_bitgrid* g, void* scratch, u3d_stream s) {
  U3D_REQUIRE(g && g->words && g->prefix && scratch, U3D_ERR_ARG);
  long long nwords = grid_nwords(g);
  int nchunks = (int)((nwords + SCAN_CHUNK - 1) / SCAN_CHUNK);
  unsigned* cs = (unsigned*)scratch;
  hipLaunchKernelGGL(k_scan_chunksum, dim3(nchunks), dim3(SCAN_TPB), 0, s, (const unsigned long long*)g->words, nwords, cs);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_TPB), 0, s, cs, nchunks);
  hipLaunchKernelGGL(k_scan_final, dim3(nchunks), dim3(SCAN_TPB), 0, s, (const unsigned long long*)g->words, nwords,
                     (const unsigned*)cs, (unsigned*)g->prefix, nchunks);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

__global__ void k_bitgrid_rank(BitGridDev g, const int4* __restrict__ coors, int n, int* __restrict__ rank) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coors[i];
  rank[i] = (c.x < 0 || c.x >= g.B) ? -1 : u3d_grid_lookup(g, c.x, c.y, c.z, c.w);
}

extern "C" int32_t u3d_bitgrid_rank(const u3d_bitgrid* g, const int32_t* coors, int32_t n, int32_t* rank, u3d_stream s) {
  U3D_REQUIRE(g && g->words && g->prefix && coors && rank, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  hipLaunchKernelGGL(k_bitgrid_rank, dim3(u3d_cdiv(n, 256)), dim3(256), 0, s, u3d_make_grid(g), (const int4*)coors, n, rank);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

__global__ void k_bitgrid_coords(BitGridDev g, long long nwords, int4* __restrict__ out, int cap) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // rows past the occupied count (capacity-sized lists of a captured step): (-1, -1, -1, -1), written here instead of by a fill launch
  {
    const int count = g.prefix[nwords];
    for (long long i = w; i < cap; i += (long long)gridDim.x * blockDim.x)
      if (i >= count) out[i] = make_int4(-1, -1, -1, -1);
  }
  if (w >= nwords) return;
  unsigned long long bits = g.words[w];
  if (!bits) return;
  unsigned r = g.prefix[w];
  if (g.linear) {
    while (bits) {
      int bit = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      long long t = w * 64 + bit;
      int x = (int)(t % g.Dx); t /= g.Dx;
      int y = (int)(t % g.Dy); t /= g.Dy;
      int z = (int)(t % g.Dz); t /= g.Dz;
      if ((int)r < cap) out[r] = make_int4((int)t, z, y, x);
      ++r;
    }
    return;
  }
  long long t = w;
  int bx = (int)(t % g.bx); t /= g.bx;
  int by = (int)(t % g.by); t /= g.by;
  int bz = (int)(t % g.bz); t /= g.bz;
  int b = (int)t;
  while (bits) {
    int bit = __ffsll((long long)bits) - 1;
    bits &= bits - 1;
    if ((int)r < cap) out[r] = make_int4(b, bz * 4 + (bit >> 4), by * 4 + ((bit >> 2) & 3), bx * 4 + (bit & 3));
    ++r;
  }
}

extern "C" int32_t u3d_bitgrid_coords(const u3d_bitgrid* g, int32_t* coors_out, int32_t cap, u3d_stream s) {
  U3D_REQUIRE(g && g->words && g->prefix && coors_out, U3D_ERR_ARG);
  long long nwords = grid_nwords(g);
  hipLaunchKernelGGL(k_bitgrid_coords, dim3(u3d_cdiv(nwords, 256)), dim3(256), 0, s, u3d_make_grid(g), nwords,
                     (int4*)coors_out, cap);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Neighbour table
// ============================================================================================
__global__ void k_nbr_table(BitGridDev g, const int4* __restrict__ q, const int* __restrict__ n_dev, int n_cap,
                            Conv3 cv, int mode, int* __restrict__ nbr, int ld) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ld) return;
  int n = min(*n_dev, n_cap);
  int kvol = cv.k[0] * cv.k[1] * cv.k[2];
  if (i >= n) {
    for (int k = 0; k < kvol; ++k) nbr[(long long)k * ld + i] = -1;
    return;
  }
  int4 c = q[i];
  int k = 0;
  for (int kz = 0; kz < cv.k[0]; ++kz)
    for (int ky = 0; ky < cv.k[1]; ++ky)
      for (int kx = 0; kx < cv.k[2]; ++kx, ++k) {
        int r = -1;
        if (mode == 0) {
          r = u3d_grid_lookup(g, c.x, c.y * cv.s[0] - cv.p[0] + kz, c.z * cv.s[1] - cv.p[1] + ky, c.w * cv.s[2] - cv.p[2] + kx);
        } else {
          int tz = c.y + cv.p[0] - kz, ty = c.z + cv.p[1] - ky, tx = c.w + cv.p[2] - kx;
          if (tz >= 0 && ty >= 0 && tx >= 0 && tz % cv.s[0] == 0 && ty % cv.s[1] == 0 && tx % cv.s[2] == 0)
            r = u3d_grid_lookup(g, c.x, tz / cv.s[0], ty / cv.s[1], tx / cv.s[2]);
        }
        nbr[(long long)k * ld + i] = r;
      }
}

extern "C" int32_t u3d_nbr_table(const u3d_bitgrid* target, const int32_t* q_coors, const int32_t* n_dev, int32_t n_cap,
                                 const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3], int32_t mode,
                                 int32_t* nbr, int32_t ld, u3d_stream s) {
  U3D_REQUIRE(target && target->words && target->prefix && q_coors && n_dev && nbr, U3D_ERR_ARG);
  U3D_REQUIRE(ld >= n_cap && (mode == 0 || mode == 1), U3D_ERR_ARG);
  if (ld <= 0) return U3D_OK;
  Conv3 cv;
  for (int i = 0; i < 3; ++i) { cv.k[i] = ksize[i]; cv.s[i] = stride[i]; cv.p[i] = pad[i]; U3D_REQUIRE(stride[i] > 0 && ksize[i] > 0, U3D_ERR_ARG); }
  hipLaunchKernelGGL(k_nbr_table, dim3(u3d_cdiv(ld, 256)), dim3(256), 0, s, u3d_make_grid(target), (const int4*)q_coors,
                     n_dev, n_cap, cv, mode, nbr, ld);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// Dense lattice variant: every cell of [B, dims] is a row, row id = lexicographic (b,z,y,x) index (== the memory order of
// a channels-last volume).  q_dims: lattice of the query rows, t_dims: lattice of the partner rows.
__global__ void k_dense_nbr_table(int B, int qz, int qy, int qx, int tz, int ty, int tx, Conv3 cv, int mode,
                                  int* __restrict__ nbr, int ld) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ld) return;
  int kvol = cv.k[0] * cv.k[1] * cv.k[2];
  long long nq = (long long)B * qz * qy * qx;
  if (i >= nq) {
    for (int k = 0; k < kvol; ++k) nbr[(long long)k * ld + i] = -1;
    return;
  }
  int x = i % qx; int t = i / qx;
  int y = t % qy; t /= qy;
  int z = t % qz; int b = t / qz;
  int k = 0;
  for (int kz = 0; kz < cv.k[0]; ++kz)
    for (int ky = 0; ky < cv.k[1]; ++ky)
      for (int kx = 0; kx < cv.k[2]; ++kx, ++k) {
        int pz, py, px;
        bool ok = true;
        if (mode == 0) {
          pz = z * cv.s[0] - cv.p[0] + kz; py = y * cv.s[1] - cv.p[1] + ky; px = x * cv.s[2] - cv.p[2] + kx;
        } else {
          int uz = z + cv.p[0] - kz, uy = y + cv.p[1] - ky, ux = x + cv.p[2] - kx;
          ok = uz >= 0 && uy >= 0 && ux >= 0 && uz % cv.s[0] == 0 && uy % cv.s[1] == 0 && ux % cv.s[2] == 0;
          pz = uz / cv.s[0]; py = uy / cv.s[1]; px = ux / cv.s[2];
        }
        ok = ok && (unsigned)pz < (unsigned)tz && (unsigned)py < (unsigned)ty && (unsigned)px < (unsigned)tx;
        nbr[(long long)k * ld + i] = ok ? (int)((((long long)b * tz + pz) * ty + py) * tx + px) : -1;
      }
}

extern "C" int32_t u3d_dense_nbr_table(int32_t batch, const int32_t q_dims[3], const int32_t t_dims[3], const int32_t ksize[3],
                                       const int32_t stride[3], const int32_t pad[3], int32_t mode, int32_t* nbr, int32_t ld,
                                       u3d_stream s) {
  U3D_REQUIRE(nbr && batch > 0 && (mode == 0 || mode == 1), U3D_ERR_ARG);
  long long nq = (long long)batch * q_dims[0] * q_dims[1] * q_dims[2];
  U3D_REQUIRE(ld >= nq && nq < 0x7fffffffll, U3D_ERR_ARG);
  Conv3 cv;
  for (int i = 0; i < 3; ++i) { cv.k[i] = ksize[i]; cv.s[i] = stride[i]; cv.p[i] = pad[i]; U3D_REQUIRE(stride[i] > 0 && ksize[i] > 0, U3D_ERR_ARG); }
  hipLaunchKernelGGL(k_dense_nbr_table, dim3(u3d_cdiv(ld, 256)), dim3(256), 0, s, batch, q_dims[0], q_dims[1], q_dims[2],
                     t_dims[0], t_dims[1], t_dims[2], cv, mode, nbr, ld);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Row gather / scatter (4-byte granularity, one wave per row segment)
// ============================================================================================
__global__ void k_gather_rows(const uint32_t* __restrict__ in, const int* __restrict__ idx, int n, int row_words,
                              uint32_t* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)n * row_words;
  if (t >= total) return;
  int r = (int)(t / row_words), w = (int)(t % row_words);
  int src = idx[r];
  out[t] = src >= 0 ? in[(long long)src * row_words + w] : 0u;
}
__global__ void k_scatter_rows(const uint32_t* __restrict__ in, const int* __restrict__ idx, int n, int row_words,
                               uint32_t* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)n * row_words;
  if (t >= total) return;
  int r = (int)(t / row_words), w = (int)(t % row_words);
  int dst = idx[r];
  if (dst >= 0) out[(long long)dst * row_words + w] = in[t];
}

extern "C" int32_t u3d_gather_rows(const void* in, const int32_t* idx, int32_t n, int32_t row_bytes, void* out, u3d_stream s) {
  U3D_REQUIRE(in && idx && out && row_bytes > 0 && (row_bytes & 3) == 0, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  int rw = row_bytes / 4;
  hipLaunchKernelGGL(k_gather_rows, dim3(u3d_cdiv((long long)n * rw, 256)), dim3(256), 0, s, (const uint32_t*)in, idx, n, rw, (uint32_t*)out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_scatter_rows(const void* in, const int32_t* idx, int32_t n, int32_t row_bytes, void* out, u3d_stream s) {
  U3D_REQUIRE(in && idx && out && row_bytes > 0 && (row_bytes & 3) == 0, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  int rw = row_bytes / 4;
  hipLaunchKernelGGL(k_scatter_rows, dim3(u3d_cdiv((long long)n * rw, 256)), dim3(256), 0, s, (const uint32_t*)in, idx, n, rw, (uint32_t*)out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Hard voxelization + mean VFE
//   pass A: per point -> cell key; open-addressing insert (per-scene table); atomicMin(first point);
//           push the point on the cell's list (atomicExch head).
//   pass B: one workgroup per scene: exclusive scan of "creator" flags in point order -> voxel id
//           (first-appearance rank), truncated at max_voxels; scene counts.
//   pass C: scene offsets (tiny), then per creator point: walk the cell list picking the max_points
//           smallest point indices in increasing order; write voxels / coors / count / mean.
// ============================================================================================
struct VoxCfg {
  float vs[3], lo[3];
  int grid[3];  // x,y,z
  int nfeat, max_points, max_voxels, hsize;  // hsize: per-scene table slots (pow2)
};

__device__ __forceinline__ unsigned hash_u32(unsigned k) {
  k ^= k >> 16; k *= 0x7feb352dU; k ^= k >> 15; k *= 0x846ca68bU; k ^= k >> 16;
  return k;
}

__global__ void k_vox_insert(const float* __restrict__ pts, const int* __restrict__ scene_off, int B, int n_total, VoxCfg cfg,
                             unsigned* __restrict__ keys, unsigned* __restrict__ first, int* __restrict__ head,
                             int* __restrict__ next, int* __restrict__ slot_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  // scene of point i (B small: linear search)
  int b = 0;
  while (b + 1 < B && i >= scene_off[b + 1]) ++b;
  int il = i - scene_off[b];
  const float* p = pts + (long long)i * cfg.nfeat;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // fp32 subtract, IEEE divide, floor — the upstream formula; NaN fails both tests
    float q = __fdiv_rn(__fsub_rn(p[j], cfg.lo[j]), cfg.vs[j]);
    bool in = (q >= 0.f) && (q < (float)cfg.grid[j]);
    c[j] = in ? (int)floorf(q) : 0;
    ok = ok && in;
  }
  if (!ok) { slot_of[i] = -1; next[i] = -1; return; }
  unsigned key = ((unsigned)c[2] * (unsigned)cfg.grid[1] + (unsigned)c[1]) * (unsigned)cfg.grid[0] + (unsigned)c[0];
  unsigned mask = (unsigned)cfg.hsize - 1u;
  unsigned h = hash_u32(key) & mask;
  long long tb = (long long)b * cfg.hsize;
  while (true) {
    unsigned prev = atomicCAS(&keys[tb + h], 0xFFFFFFFFu, key);
    if (prev == 0xFFFFFFFFu || prev == key) break;
    h = (h + 1) & mask;
  }
  atomicMin(&first[tb + h], (unsigned)il);
  next[i] = atomicExch(&head[tb + h], il);
  slot_of[i] = (int)h;
}

__global__ void k_vox_rank(const int* __restrict__ scene_off, VoxCfg cfg, const unsigned* __restrict__ first,
                           const int* __restrict__ slot_of, int* __restrict__ vid, int* __restrict__ scene_cnt) {
  // one workgroup (1024 threads) per scene
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry_s;
  int b = blockIdx.x;
  int p0 = scene_off[b], n = scene_off[b + 1] - p0;
  long long tb = (long long)b * cfg.hsize;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    int il = base + threadIdx.x;
    int sl = -1;
    unsigned flag = 0;
    if (il < n) {
      sl = slot_of[p0 + il];
      if (sl >= 0 && first[tb + sl] == (unsigned)il) flag = 1;
    }
    unsigned long long bal = __ballot(flag);
    unsigned in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    unsigned woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { unsigned v = wsum[w]; if (w < wid) woff += v; tot += v; }
    unsigned carry = carry_s;
    if (flag) {
      unsigned r = carry + woff + in_wave;
      vid[tb + sl] = (r < (unsigned)cfg.max_voxels) ? (int)r : -1;
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) scene_cnt[b] = (int)min(carry_s, (unsigned)cfg.max_voxels);
}

__global__ void k_vox_offsets(const int* __restrict__ scene_cnt, int B, int* __restrict__ voxel_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { voxel_off[b] = acc; acc += scene_cnt[b]; }
    voxel_off[B] = acc;
  }
}

__global__ void k_vox_write(const float* __restrict__ pts, const int* __restrict__ scene_off, int B, int n_total, VoxCfg cfg,
                            const unsigned* __restrict__ keys, const unsigned* __restrict__ first, const int* __restrict__ head,
                            const int* __restrict__ next, const int* __restrict__ slot_of, const int* __restrict__ vid,
                            const int* __restrict__ voxel_off, float* __restrict__ voxels, int4* __restrict__ coors,
                            int* __restrict__ num_points, float* __restrict__ mean) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int sl = slot_of[i];
  if (sl < 0) return;
  int b = 0;
  while (b + 1 < B && i >= scene_off[b + 1]) ++b;
  int p0 = scene_off[b];
  int il = i - p0;
  long long tb = (long long)b * cfg.hsize;
  if (first[tb + sl] != (unsigned)il) return;   // only the creator point writes the voxel
  int v = vid[tb + sl];
  if (v < 0) return;
  long long row = (long long)voxel_off[b] + v;
  unsigned key = keys[tb + sl];
  int cx = (int)(key % (unsigned)cfg.grid[0]);
  unsigned t = key / (unsigned)cfg.grid[0];
  int cy = (int)(t % (unsigned)cfg.grid[1]);
  int cz = (int)(t / (unsigned)cfg.grid[1]);
  coors[row] = make_int4(b, cz, cy, cx);
  // selection of the max_points smallest indices, ascending
  int last = -1, cnt = 0;
  float acc[8];
  const int nf = cfg.nfeat;
  for (int f = 0; f < 8; ++f) acc[f] = 0.f;
  for (int r = 0; r < cfg.max_points; ++r) {
    int best = 0x7fffffff;
    for (int j = head[tb + sl]; j >= 0; j = next[p0 + j])
      if (j > last && j < best) best = j;
    if (best == 0x7fffffff) break;
    last = best;
    const float* p = pts + (long long)(p0 + best) * nf;
    for (int f = 0; f < nf; ++f) {
      float val = p[f];
      if (voxels) voxels[(row * cfg.max_points + r) * nf + f] = val;
      if (f < 8) acc[f] += val;
    }
    ++cnt;
  }
  if (voxels)
    for (int r = cnt; r < cfg.max_points; ++r)
      for (int f = 0; f < nf; ++f) voxels[(row * cfg.max_points + r) * nf + f] = 0.f;
  num_points[row] = cnt;
  if (mean) {
    float inv = (float)cnt;
    for (int f = 0; f < nf && f < 8; ++f) mean[row * nf + f] = acc[f] / inv;
  }
}

__global__ void k_fill_u32(unsigned* __restrict__ p, long long n, unsigned v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct VoxWs { unsigned* keys; unsigned* first; int* head; int* vid; int* next; int* slot_of; int* scene_cnt; int hsize; int64_t bytes; };

static VoxWs vox_ws_layout(void* base, int n_total, int B, int max_pts_per_scene) {
  VoxWs w;
  w.hsize = next_pow2(2 * (max_pts_per_scene > 1 ? max_pts_per_scene : 1));
  char* p = (char*)base;
  size_t tab = (size_t)B * w.hsize * 4;
  w.keys = (unsigned*)p; p += tab;      // 0xFF-filled
  w.first = (unsigned*)p; p += tab;     // 0xFF-filled
  w.head = (int*)p; p += tab;           // 0xFF-filled (-1)
  w.vid = (int*)p; p += tab;            // 0xFF-filled (-1)
  w.next = (int*)p; p += (size_t)n_total * 4;
  w.slot_of = (int*)p; p += (size_t)n_total * 4;
  w.scene_cnt = (int*)p; p += (size_t)(B + 1) * 4;
  w.bytes = (int64_t)(p - (char*)base);
  return w;
}

extern "C" int64_t u3d_voxelize_hard_workspace(int32_t n_total, int32_t batch, int32_t max_pts_per_scene) {
  return vox_ws_layout(nullptr, n_total, batch, max_pts_per_scene).bytes;
}

extern "C" int32_t u3d_voxelize_hard(const float* points, const int32_t* scene_off, int32_t batch, int32_t n_total,
                                     int32_t max_pts_per_scene, int32_t nfeat, const float voxel_size[3],
                                     const float pc_range[6], int32_t max_points, int32_t max_voxels, float* voxels,
                                     int32_t* coors, int32_t* num_points, float* mean, int32_t* voxel_off,
                                     void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(points && scene_off && coors && num_points && voxel_off && workspace, U3D_ERR_ARG);
  U3D_REQUIRE(batch > 0 && n_total >= 0 && nfeat >= 3 && nfeat <= 8 && max_points > 0 && max_voxels > 0, U3D_ERR_ARG);
  VoxWs w = vox_ws_layout(workspace, n_total, batch, max_pts_per_scene);
  U3D_REQUIRE(workspace_bytes >= w.bytes, U3D_ERR_WORKSPACE);
  VoxCfg cfg;
  for (int j = 0; j < 3; ++j) {
    cfg.vs[j] = voxel_size[j];
    cfg.lo[j] = pc_range[j];
    cfg.grid[j] = (int)lroundf((pc_range[3 + j] - pc_range[j]) / voxel_size[j]);
  }
  U3D_REQUIRE((long long)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] < 0xFFFFFFFFll, U3D_ERR_ARG);
  cfg.nfeat = nfeat; cfg.max_points = max_points; cfg.max_voxels = max_voxels; cfg.hsize = w.hsize;
  {  // keys / first / head / vid tables <- 0xFFFFFFFF.  A kernel, not hipMemsetAsync: memset nodes of re-launched HIP graphs
     // that share a memory pool were observed to leave the tables stale (list cycles -> hang) on ROCm 7.2.
    long long nfill = (long long)batch * w.hsize * 4;
    int blocks = (int)((nfill + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, s, w.keys, nfill, 0xFFFFFFFFu);
  }
  if (n_total > 0) {
    hipLaunchKernelGGL(k_vox_insert, dim3(u3d_cdiv(n_total, 256)), dim3(256), 0, s, points, scene_off, batch, n_total, cfg,
                       w.keys, w.first, w.head, w.next, w.slot_of);
  }
  hipLaunchKernelGGL(k_vox_rank, dim3(batch), dim3(1024), 0, s, scene_off, cfg, (const unsigned*)w.first,
                     (const int*)w.slot_of, w.vid, w.scene_cnt);
  hipLaunchKernelGGL(k_vox_offsets, dim3(1), dim3(64), 0, s, (const int*)w.scene_cnt, batch, voxel_off);
  if (n_total > 0) {
    hipLaunchKernelGGL(k_vox_write, dim3(u3d_cdiv(n_total, 256)), dim3(256), 0, s, points, scene_off, batch, n_total, cfg,
                       (const unsigned*)w.keys, (const unsigned*)w.first, (const int*)w.head, (const int*)w.next,
                       (const int*)w.slot_of, (const int*)w.vid, (const int*)voxel_off, voxels, (int4*)coors, num_points, mean);
  }
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}


// ============================================================================================
// Dynamic voxelization + scatter-mean (DynamicSimpleVFE)
// ============================================================================================
__global__ void k_vox_dynamic(const float* __restrict__ pts, const int* __restrict__ scene_off, int B, int n_total, VoxCfg cfg,
                              int4* __restrict__ coors) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int b = 0;
  while (b + 1 < B && i >= scene_off[b + 1]) ++b;
  const float* p = pts + (long long)i * cfg.nfeat;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float q = __fdiv_rn(__fsub_rn(p[j], cfg.lo[j]), cfg.vs[j]);
    bool in = (q >= 0.f) && (q < (float)cfg.grid[j]);
    c[j] = in ? (int)floorf(q) : 0;
    ok = ok && in;
  }
  coors[i] = ok ? make_int4(b, c[2], c[1], c[0]) : make_int4(b, -1, -1, -1);
}

extern "C" int32_t u3d_voxelize_dynamic(const float* points, const int32_t* scene_off, int32_t batch, int32_t n_total, int32_t nfeat,
                                        const float voxel_size[3], const float pc_range[6], int32_t* coors, u3d_stream s) {
  U3D_REQUIRE(points && scene_off && coors && batch > 0 && nfeat >= 3, U3D_ERR_ARG);
  if (n_total <= 0) return U3D_OK;
  VoxCfg cfg;
  for (int j = 0; j < 3; ++j) {
    cfg.vs[j] = voxel_size[j];
    cfg.lo[j] = pc_range[j];
    cfg.grid[j] = (int)lroundf((pc_range[3 + j] - pc_range[j]) / voxel_size[j]);
  }
  cfg.nfeat = nfeat; cfg.max_points = 0; cfg.max_voxels = 0; cfg.hsize = 0;
  hipLaunchKernelGGL(k_vox_dynamic, dim3(u3d_cdiv(n_total, 256)), dim3(256), 0, s, points, scene_off, batch, n_total, cfg, (int4*)coors);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

__global__ void k_scatter_accum(const float* __restrict__ pts, const int* __restrict__ rank, int n_total, int nfeat,
                                float* __restrict__ sums, int* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int r = rank[i];
  if (r < 0) return;
  for (int f = 0; f < nfeat; ++f) atomicAdd(&sums[(long long)r * nfeat + f], pts[(long long)i * nfeat + f]);
  atomicAdd(&counts[r], 1);
}
__global__ void k_scatter_divide(float* __restrict__ sums, const int* __restrict__ counts, int n_voxels, int nfeat) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_voxels * nfeat) return;
  int c = counts[t / nfeat];
  sums[t] = sums[t] / (float)(c > 0 ? c : 1);
}

extern "C" int32_t u3d_scatter_mean(const float* points, const int32_t* rank, int32_t n_total, int32_t nfeat, float* sums,
                                    int32_t* counts, int32_t n_voxels, u3d_stream s) {
  U3D_REQUIRE(points && rank && sums && counts && nfeat > 0, U3D_ERR_ARG);
  if (n_total > 0) hipLaunchKernelGGL(k_scatter_accum, dim3(u3d_cdiv(n_total, 256)), dim3(256), 0, s, points, rank, n_total, nfeat, sums, counts);
  if (n_voxels > 0) hipLaunchKernelGGL(k_scatter_divide, dim3(u3d_cdiv((long long)n_voxels * nfeat, 256)), dim3(256), 0, s, sums, counts, n_voxels, nfeat);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
