// Shared device/host helpers for libu3d_hip.so (gfx950 / MI355X only).
// No torch types anywhere below this line: raw device pointers, sizes, hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/u3d_hip.h"

#define U3D_WAVE 64

#define U3D_CHECK_LAUNCH()                                        \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) return U3D_ERR_LAUNCH;                 \
  } while (0)

#define U3D_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

static inline int u3d_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; a process-wide flag would leave a second GPU
// of the process at the 64 KiB default).  `mask`: one word per call site (bit = device id; setting the attribute twice is harmless,
// so the relaxed read-modify-write needs no lock).  Devices >= 64 simply set it every time.
static inline void u3d_allow_lds_impl(const void* kernel, int bytes, unsigned long long* mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && ((__atomic_load_n(mask, __ATOMIC_RELAXED) >> dev) & 1ull)) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0 && dev < 64) __atomic_fetch_or(mask, 1ull << dev, __ATOMIC_RELAXED);
}
#define U3D_ALLOW_LDS(kernel, bytes)                                        \
  do {                                                                      \
    static unsigned long long u3d_lds_mask__ = 0;                           \
    u3d_allow_lds_impl((const void*)(kernel), (int)(bytes), &u3d_lds_mask__); \
  } while (0)

// ---------------------------------------------------------------------------------------
// BitGrid: occupancy of a [B, Dz, Dy, Dx] voxel lattice, one 64-bit word per 4x4x4 block.
// Row id of an occupied cell = prefix[word] + popcount(bits below it) ("block-major rank").
// The rank order is the internal row order of every sparse level: rows that are close in
// space are close in memory, so a 64-row tile of the conv kernel gathers from few lines.
// ---------------------------------------------------------------------------------------
struct BitGridDev {
  const unsigned long long* words;
  const unsigned int* prefix;  // exclusive popcount scan, nwords + 1 entries
  int B, Dz, Dy, Dx;           // logical extent (cells)
  int bz, by, bx;              // extent in 4x4x4 blocks
  int linear;                  // 1: one bit per cell in lexicographic (b,z,y,x) order (rank == torch.unique(dim=0) order)
  int cap;                     // > 0: row capacity of the level; lookups never return a row id >= cap (static-shape overflow guard)
};

__host__ __device__ inline BitGridDev u3d_make_grid(const u3d_bitgrid* g) {
  BitGridDev d;
  d.words = (const unsigned long long*)g->words;
  d.prefix = (const unsigned int*)g->prefix;
  d.B = g->batch; d.Dz = g->dz; d.Dy = g->dy; d.Dx = g->dx;
  d.bz = (g->dz + 3) >> 2; d.by = (g->dy + 3) >> 2; d.bx = (g->dx + 3) >> 2;
  d.linear = g->layout;
  d.cap = g->row_capacity;
  return d;
}

__device__ __forceinline__ long long u3d_cell_linear(const BitGridDev& g, int b, int z, int y, int x) {
  return (((long long)b * g.Dz + z) * g.Dy + y) * g.Dx + x;
}
__device__ __forceinline__ long long u3d_word_index(const BitGridDev& g, int b, int z, int y, int x) {
  if (g.linear) return u3d_cell_linear(g, b, z, y, x) >> 6;
  return (((long long)b * g.bz + (z >> 2)) * g.by + (y >> 2)) * g.bx + (x >> 2);
}
__device__ __forceinline__ int u3d_bit_index(const BitGridDev& g, int b, int z, int y, int x) {
  if (g.linear) return (int)(u3d_cell_linear(g, b, z, y, x) & 63);
  return ((z & 3) << 4) | ((y & 3) << 2) | (x & 3);
}
// row id of (b,z,y,x) or -1 when out of range / unoccupied
__device__ __forceinline__ int u3d_grid_lookup(const BitGridDev& g, int b, int z, int y, int x) {
  if ((unsigned)z >= (unsigned)g.Dz || (unsigned)y >= (unsigned)g.Dy || (unsigned)x >= (unsigned)g.Dx) return -1;
  long long w = u3d_word_index(g, b, z, y, x);
  unsigned long long bits = g.words[w];
  int bit = u3d_bit_index(g, b, z, y, x);
  if (!((bits >> bit) & 1ull)) return -1;
  int r = (int)(g.prefix[w] + __popcll(bits & ((1ull << bit) - 1ull)));
  return (g.cap > 0 && r >= g.cap) ? -1 : r;
}

// XCD-aware tile order over the LIVE tiles of a grid sized for a CAPACITY (captured steps launch ceil(capacity / tile) workgroups and
// read the row count on the device): workgroup b runs on XCD b & 7; each XCD gets a CONTIGUOUS range of the live tiles (neighbouring
// tiles share rows -> that XCD's L2), and the surplus workgroups exit.  (Partitioning gridDim instead left the XCDs at the end of the
// range with nothing but padding: with 33 % capacity margin a quarter of the chip idled - 52 -> 73 us per k_subm_halo128 launch.)
// -> tile index, or -1 for a surplus workgroup.  Needs gridDim.x >= live_tiles.
__device__ __forceinline__ int u3d_xcd_tile(int block, int live_tiles) {
  const int xcd = block & 7, slot = block >> 3;
  const int q = live_tiles >> 3, r = live_tiles & 7;
  if (slot >= q + (xcd < r ? 1 : 0)) return -1;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// wave-level helpers (wave = 64 lanes)
__device__ __forceinline__ float u3d_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double u3d_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
