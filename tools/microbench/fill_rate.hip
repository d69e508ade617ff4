// tools/microbench/fill_rate.hip - what bounds k_igemm_glds8_256x256 (uni3detr_amd/csrc/igemm_bf16.hip)?
//
// The dominant layer of the benched step (SECOND3DFPN's 256 -> 256 3x3x3 convolutions, ref models/necks/second3d_fpn.py:73-104) is an
// implicit GEMM over 192 000 lattice rows x 27 offsets: 750 workgroups (8 waves) walk 108 k-tiles of 64 KiB (256 gathered activation
// row slices of 128 B + 256 weight row slices of 128 B) through two 64 KiB LDS stage buffers.  DESIGN.md 3.1 claimed that the bytes
// staged / time of every such kernel is the same ~17 B/clk/CU and called it the L2 -> CU fabric's ceiling.  This program measures
// that claim in isolation: the SAME address stream, tile order, staging instruction and buffer discipline, with the other
// consumers of the loop (fragment reads, MFMAs) switched on and off one at a time.
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/fill_rate.hip -o tools/microbench/fill_rate
//   run:   tools/microbench/fill_rate [iters]          (prints one table; gpurun_out/ copy -> profiles/r06_fill_rate.txt)
//
// Variants (template parameters):
//   SRC    0 LDS-DMA (raw_ptr_buffer_load_lds, 16 B / lane), 1 global_load_dwordx4 into registers (no LDS write), 2 no loads,
//          3 activations by LDS-DMA, WEIGHTS by global loads straight into the MFMA operand registers (fragment-packed: no LDS traffic for B)
//   WIN    activation rows folded into a 1024-row window (L2-resident: 512 KiB) instead of the layer's real stream (98 MB)
//   READS  the 24 ds_read_b128 fragment reads per wave and k-tile of the real loop
//   MFMA   0 none, 1: 64 x v_mfma_f32_16x16x32_bf16 per wave and k-tile, 2: 32 x v_mfma_f32_32x32x16_bf16 (same flops)
//   DEPTH  k-tiles in flight while the wave waits (1 = the product kernel's discipline: two stage buffers)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_tile(int block, int live_tiles) {       // = u3d_xcd_tile (csrc/common.h)
  const int xcd = block & 7, slot = block >> 3;
  const int q = live_tiles >> 3, r = live_tiles & 7;
  if (slot >= q + (xcd < r ? 1 : 0)) return -1;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <int SRC, bool WIN, bool READS, int MFMA, int SPREAD = 0, bool DEEP = false>
__global__ __launch_bounds__(512) void k_fill(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr, int ld,
                                              int n_out, int cin, int cout, int kvol, float* __restrict__ sink,
                                              unsigned long long* __restrict__ clk) {
  constexpr int BM = 256, BK = 64, PIECE = 128 * BK, STAGE_ELEMS = 4 * PIECE;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int tile = xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);
  if (tile < 0) return;
  const unsigned long long t0 = clock64();
  const int m0 = tile * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int nstage = kvol * (cin / BK);
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  constexpr unsigned row_bytes = 512u;                      // cin = 256 (a run-time cin made hipcc use v_mad_u64_u32 whose 64-bit addend
                                                            // pair aliased a pending index load's register: s_waitcnt vmcnt(3) before the first request)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_part16[2], w_voff[2][2];
  int mrow[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (wv * 2 + u) * 8 + lrow;
    a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const int ar = (r >> 6) * 128 + sp * 64 + (r & 63);
      mrow[sp][u] = min(m0 + ar, n_out - 1);
      const int tc = (r >> 5) * 64 + sp * 32 + (r & 31);
      w_voff[sp][u] = (unsigned)(tc * cin + (lslot ^ ((r >> 1) & 7)) * 8) * 2u;
    }
  }
  // gather indices exactly as the product kernel handles them: requested at the top of a trip for the k-tile after the next,
  // turned into the "current" set by a VALU select at the end of the SAME trip (hipcc's own wait for them is then the vmcnt(8)
  // the schedule wants; a plain copy would be coalesced into a register rotation and a pending load across the back edge).
  // (An inline-asm load + explicit waits was tried first: under register pressure hipcc spilled the asm's output register
  // straight after the request, i.e. before the value had landed - memory faults in the MFMA + reads variants.)
  int idx_cur[2][2], idx_nxt[2][2];
  bool live[2][2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp)
#pragma unroll
    for (int u = 0; u < 2; ++u) live[sp][u] = m0 + (((wv * 2 + u) * 8 + lrow) >> 6) * 128 + sp * 64 + (((wv * 2 + u) * 8 + lrow) & 63) < n_out;
  auto load_idx_next = [&](int stage) {
    const int* row = nbr + (long long)(stage % kvol) * ld;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_nxt[sp][u] = row[mrow[sp][u]];
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 2; ++u) idx_cur[sp][u] = live[sp][u] ? idx_nxt[sp][u] : -1;
  };
  f32x4 junk = {0.f, 0.f, 0.f, 0.f};
  f32x4 hold[2][8];                                         // SRC 1: the loads of a k-tile stay in flight until the NEXT k-tile's are issued
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) hold[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto issue_a = [&](int st, int buf, int sp) {
    const unsigned soff = (unsigned)((st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + sp * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int iv = idx_cur[sp][u];
      const unsigned voff = iv >= 0 ? (unsigned)(WIN ? (iv & 1023) : iv) * row_bytes + a_part16[u] : 0xFFFFFFFFu;
      if constexpr (SRC == 0 || SRC == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(dst + u * 512), 16, voff, soff, 0, 0);
      else if constexpr (SRC == 1) {
        hold[buf][sp * 2 + u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, voff, soff, 0));
      }
    }
  };
  auto issue_b = [&](int st, int buf, int sp) {
    if constexpr (SRC == 3) {
      // B operand straight into registers: fragment-packed weights, this wave's 64 columns x 64 k of the k-tile = 8 blocks of 1 KiB
      // (wave columns share their blocks through L1 / L2; the two calls per k-tile fetch four blocks each)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned off = (unsigned)((((st * 4 + wn) * 8) + sp * 4 + j) * 1024 + lane * 16);
        hold[buf][sp * 4 + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, off, 0, 0));
      }
      return;
    }
    const unsigned soff = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + wv * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (SRC == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)(dst + u * 512), 16, w_voff[sp][u], soff, 0, 0);
      else if constexpr (SRC == 1) {
        hold[buf][4 + sp * 2 + u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_voff[sp][u], soff, 0));
      }
    }
  };
  auto issue_a1 = [&](int st, int buf, int sp, int u) {
    const unsigned soff = (unsigned)((st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + sp * PIECE + wv * 1024;
    const int iv = idx_cur[sp][u];
    const unsigned voff = iv >= 0 ? (unsigned)(WIN ? (iv & 1023) : iv) * row_bytes + a_part16[u] : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)(dst + u * 512), 16, voff, soff, 0, 0);
  };
  auto issue_b1 = [&](int st, int buf, int sp, int u) {
    const unsigned soff = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
    u16* dst = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + wv * 1024;
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
  f32x4 acc[8][4];
  f32x16 acc2[4][2];
  if constexpr (MFMA == 1 || MFMA == 3) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (MFMA == 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
  }
  // operands of the MFMA-only variants: registers (no LDS dependence), non-trivial values
  bf16x8 ra[8], rb[4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) ra[a][e] = (__bf16)(0.01f * (float)((lane * 7 + a * 3 + e) % 17 - 8));
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) rb[b][e] = (__bf16)(0.02f * (float)((lane * 5 + b * 11 + e) % 13 - 6));

  if (SRC != 2) {
    load_idx_next(0);
    advance_idx();
    issue_a(0, 0, 0); issue_b(0, 0, 0); issue_b(0, 0, 1); issue_a(0, 0, 1);
    load_idx_next(1 < nstage ? 1 : 0);
    advance_idx();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int st2 = 0; st2 < nstage; st2 += 2) {               // (nstage is even: 27 x 4)
#pragma unroll
   for (int par = 0; par < 2; ++par) {
    const int st = st2 + par;
    const int buf = par;
    const int nx = st + 1 < nstage ? st + 1 : st;           // past the end: a harmless re-fetch
    if (SRC != 2) {
      load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPREAD == 0 || SPREAD >= 4) { issue_a(nx, buf ^ 1, 0); issue_b(nx, buf ^ 1, 0); issue_b(nx, buf ^ 1, 1); issue_a(nx, buf ^ 1, 1); }
    }
    if constexpr (MFMA == 3) {
      // software-pipelined fragment reads (hipcc's own order for the plain loop is read -> s_waitcnt lgkmcnt(0) -> 8 MFMAs, i.e. every
      // group of 8 MFMAs waits out an LDS round trip): the B fragment of group g + 1 and, in the last groups of a k-step, the A fragments
      // of the next k-step are requested BEFORE the MFMAs of group g, pinned by scheduling barriers
      const u16* A = smem + buf * STAGE_ELEMS + (wm * 64 + li) * BK;
      const u16* B = smem + buf * STAGE_ELEMS + 2 * PIECE + (wn * 32 + li) * BK;
      bf16x8 afp[2][8], bcur, bnxt, bnn;
#pragma unroll
      for (int a = 0; a < 8; ++a) afp[0][a] = frag(A + (a >> 2) * PIECE + (a & 3) * 16 * BK, 0);
      bcur = frag(B, 0);
      if constexpr (SPREAD == 5) bnxt = frag(B + 16 * BK, 0);
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) {                      // group = (k-step, column block)
        const int ks = g8 >> 2, b = g8 & 3;
        if constexpr (SPREAD == 5) {                        // B fragments TWO groups ahead
          if (g8 + 2 < 8) {
            const int ks1 = (g8 + 2) >> 2, b1 = (g8 + 2) & 3;
            bnn = frag(B + (b1 >> 1) * PIECE + (b1 & 1) * 16 * BK, ks1);
          }
        } else
        if (g8 + 1 < 8) {
          const int ks1 = (g8 + 1) >> 2, b1 = (g8 + 1) & 3;
          bnxt = frag(B + (b1 >> 1) * PIECE + (b1 & 1) * 16 * BK, ks1);
        }
        if (ks == 0) {                                      // two A fragments of k-step 1 per group of k-step 0
          afp[1][2 * b] = frag(A + ((2 * b) >> 2) * PIECE + ((2 * b) & 3) * 16 * BK, 1);
          afp[1][2 * b + 1] = frag(A + ((2 * b + 1) >> 2) * PIECE + ((2 * b + 1) & 3) * 16 * BK, 1);
        }
        if constexpr (SPREAD != 0 && SPREAD < 4 && SRC == 0) {
          // the next k-tile's 8 LDS-DMA requests dealt out behind the MFMA groups instead of one burst at the top of the trip (a wave that
          // issues a request is blocked ~100 clocks; in a burst both waves of a SIMD are blocked together and the matrix pipe idles):
          // 1: one per group; 2: two per group in the first four groups; 3: as 2, the second wave row two groups later
          const int slot = SPREAD == 1 ? g8 : (SPREAD == 2 ? (g8 < 4 ? g8 : -1) : ((g8 - 2 * wm >= 0 && g8 - 2 * wm < 4) ? g8 - 2 * wm : -1));
          if (SPREAD == 1) {
            if (slot == 0) issue_a1(nx, buf ^ 1, 0, 0); if (slot == 1) issue_a1(nx, buf ^ 1, 0, 1);
            if (slot == 2) issue_b1(nx, buf ^ 1, 0, 0); if (slot == 3) issue_b1(nx, buf ^ 1, 0, 1);
            if (slot == 4) issue_b1(nx, buf ^ 1, 1, 0); if (slot == 5) issue_b1(nx, buf ^ 1, 1, 1);
            if (slot == 6) issue_a1(nx, buf ^ 1, 1, 0); if (slot == 7) issue_a1(nx, buf ^ 1, 1, 1);
          } else {
            if (slot == 0) issue_a(nx, buf ^ 1, 0);
            if (slot == 1) issue_b(nx, buf ^ 1, 0);
            if (slot == 2) issue_b(nx, buf ^ 1, 1);
            if (slot == 3) issue_a(nx, buf ^ 1, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPREAD == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur, afp[ks][a], acc[a][b], 0, 0, 0);
        if constexpr (SPREAD == 4) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        bcur = bnxt;
        if constexpr (SPREAD == 5) bnxt = bnn;
      }
    } else
    if constexpr (READS || MFMA) {
      const u16* A = smem + buf * STAGE_ELEMS + (wm * 64 + li) * BK;
      const u16* B = smem + buf * STAGE_ELEMS + 2 * PIECE + (wn * 32 + li) * BK;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[8], bfr[4];
        if constexpr (READS) {
#pragma unroll
          for (int a = 0; a < 8; ++a) af[a] = frag(A + (a >> 2) * PIECE + (a & 3) * 16 * BK, ks);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if constexpr (SRC == 3) bfr[b] = __builtin_bit_cast(bf16x8, hold[buf][ks * 4 + b]);
            else bfr[b] = frag(B + (b >> 1) * PIECE + (b & 1) * 16 * BK, ks);
          }
        } else {
#pragma unroll
          for (int a = 0; a < 8; ++a) af[a] = ra[a];
#pragma unroll
          for (int b = 0; b < 4; ++b) bfr[b] = rb[b];
        }
        if constexpr (MFMA == 1) {
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
        } else if constexpr (MFMA == 2) {
          // same flops with 32x32x16: 4 x 2 blocks x 4 k-steps per k-tile = 2 k-steps per ks; operands reuse the 16-B fragments
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int a = 0; a < 4; ++a)
                acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b * 2 + k2], af[a * 2 + k2], acc2[a][b], 0, 0, 0);
        } else {
#pragma unroll
          for (int a = 0; a < 8; ++a) junk += __builtin_bit_cast(f32x4, af[a]);
#pragma unroll
          for (int b = 0; b < 4; ++b) junk += __builtin_bit_cast(f32x4, bfr[b]);
        }
      }
    }
    if (SRC != 2) {
      __builtin_amdgcn_sched_barrier(0);
      advance_idx();                                        // (hipcc waits for the indices here: vmcnt(8))
      // k-tile st + 1 was requested during THIS trip and is read right after the barrier: with two stage buffers nothing may stay in
      // flight across it (the first version of this program waited vmcnt(8) here - one k-tile too few: timing-only, but a trip more of
      // latency tolerance than a correct kernel has; the DEEP variants keep that discipline on purpose to price it)
      if constexpr (SRC == 0 && !DEEP) __builtin_amdgcn_s_waitcnt(0x0F70);
      else if constexpr (SRC == 3) __builtin_amdgcn_s_waitcnt(0x0F7C);      // vmcnt(12): 4 LDS-DMA + 8 register loads of k-tile st + 1 stay in flight
      else __builtin_amdgcn_s_waitcnt(0x0F78);
      if constexpr (SRC == 1) {                             // consume k-tile st (requested one trip ago): k-tile st + 1 stays in flight
#pragma unroll
        for (int j = 0; j < 8; ++j) junk += hold[buf][j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
   }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  float s = junk[0] + junk[1] + junk[2] + junk[3];
  if constexpr (MFMA == 1 || MFMA == 3) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  }
  if constexpr (MFMA == 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc2[a][b][r];
  }
  if (s == 12345.678f) sink[tid] = s;                       // never true: keeps the reads / MFMAs alive
  if (tid == 0) clk[tile] = clock64() - t0;
}


// Role split: does the matrix pipe overlap with the LDS-side work of ANOTHER wave on the same SIMD when nothing ties the two
// together but one barrier per k-tile?  Waves 0-3 (one per SIMD) only multiply (all 128 MFMAs of the CU-quarter's k-tile, register
// operands); waves 4-7 only load (all 16 LDS-DMA requests of a quarter of the tile + 48 fragment reads, the CU's whole LDS-side work).
template <bool READS, bool MFMAS, bool LOADS>
__global__ __launch_bounds__(512) void k_roles(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr, int ld,
                                               int n_out, int cin, int cout, int kvol, float* __restrict__ sink,
                                               unsigned long long* __restrict__ clk) {
  constexpr int BM = 256, BK = 64, PIECE = 128 * BK, STAGE_ELEMS = 4 * PIECE;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int tile = xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);
  if (tile < 0) return;
  const unsigned long long t0 = clock64();
  const int m0 = tile * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nstage = kvol * (cin / BK);
  float s = 0.f;
  if (wv < 4) {
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ra[8], rb[4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) ra[a][e] = (__bf16)(0.01f * (float)((lane * 7 + a * 3 + e) % 17 - 8));
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) rb[b][e] = (__bf16)(0.02f * (float)((lane * 5 + b * 11 + e) % 13 - 6));
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
      if (MFMAS) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[b], ra[a], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  } else {
    const int lw = wv - 4;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
    constexpr unsigned row_bytes = 512u;
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned a_part16[4], w_voff[2][4];
    int mrow[2][4];
    bool live[2][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = (lw * 4 + u) * 8 + lrow;
      a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const int ar = (r >> 6) * 128 + sp * 64 + (r & 63);
        mrow[sp][u] = min(m0 + ar, n_out - 1);
        live[sp][u] = m0 + ar < n_out;
        const int tc = (r >> 5) * 64 + sp * 32 + (r & 31);
        w_voff[sp][u] = (unsigned)(tc * cin + (lslot ^ ((r >> 1) & 7)) * 8) * 2u;
      }
    }
    int idx_cur[2][4], idx_nxt[2][4];
    auto load_idx_next = [&](int stage) {
      const int* row = nbr + (long long)(stage % kvol) * ld;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int u = 0; u < 4; ++u) idx_nxt[sp][u] = row[mrow[sp][u]];
    };
    auto advance_idx = [&]() {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int u = 0; u < 4; ++u) idx_cur[sp][u] = live[sp][u] ? idx_nxt[sp][u] : -1;
    };
    auto issue = [&](int st, int buf) {
      const unsigned soff_a = (unsigned)((st / kvol) * BK) * 2u;
      const unsigned soff_b = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          u16* da = smem + buf * STAGE_ELEMS + sp * PIECE + (lw * 4 + u) * 512;
          const unsigned voff = idx_cur[sp][u] >= 0 ? (unsigned)idx_cur[sp][u] * row_bytes + a_part16[u] : 0xFFFFFFFFu;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)da, 16, voff, soff_a, 0, 0);
          u16* db = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + (lw * 4 + u) * 512;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)db, 16, w_voff[sp][u], soff_b, 0, 0);
        }
    };
    const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = ((ks * 4 + g) ^ fsw) << 3;
    typedef const volatile s16x8 __attribute__((address_space(3))) * lds_vptr;
    f32x4 junk = {0.f, 0.f, 0.f, 0.f};
    if (LOADS) {
      load_idx_next(0);
      advance_idx();
      issue(0, 0);
      load_idx_next(1);
      advance_idx();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int st2 = 0; st2 < nstage; st2 += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int st = st2 + par, buf = par;
        const int nx = st + 1 < nstage ? st + 1 : st;
        if (LOADS) {
          load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
          __builtin_amdgcn_sched_barrier(0);
          issue(nx, buf ^ 1);
        }
        if (READS) {
          // the fragment reads of TWO compute waves (wave columns lw and the two wave rows): 48 ds_read_b128
#pragma unroll
          for (int wm = 0; wm < 2; ++wm) {
            const u16* A = smem + buf * STAGE_ELEMS + (wm * 64 + li) * BK;
            const u16* B = smem + buf * STAGE_ELEMS + 2 * PIECE + (lw * 32 + li) * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
              for (int a = 0; a < 8; ++a) {
                s16x8 v = *(lds_vptr)(A + (a >> 2) * PIECE + (a & 3) * 16 * BK + foff[ks]);
                junk += __builtin_bit_cast(f32x4, v);
              }
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                s16x8 v = *(lds_vptr)(B + (b >> 1) * PIECE + (b & 1) * 16 * BK + foff[ks]);
                junk += __builtin_bit_cast(f32x4, v);
              }
            }
          }
        }
        if (LOADS) {
          __builtin_amdgcn_sched_barrier(0);
          advance_idx();
          __builtin_amdgcn_s_waitcnt(0x0F70 | 0x4000);        // vmcnt(16): k-tile st has landed, k-tile st + 1 (16 requests) stays in flight
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    s = junk[0] + junk[1] + junk[2] + junk[3];
  }
  __syncthreads();
  if (s == 12345.678f) sink[tid] = s;
  if (tid == 0) clk[tile] = clock64() - t0;
}


// One wave per SIMD (4 waves, up to 512 registers each): a wave owns a 128 x 128 block of the 256 x 256 tile (2 x 2 wave grid) -
// 32 fragment reads per wave and k-tile instead of 24 x 2 waves (128 KiB of LDS reads per k-tile and CU instead of 192), 128 MFMAs,
// 16 LDS-DMA requests; nothing but the wave's own instruction stream overlaps the three.
template <bool LOADS, bool READS>
__global__ __launch_bounds__(256) void k_wide(const u16* __restrict__ in, const u16* __restrict__ w, const int* __restrict__ nbr, int ld,
                                              int n_out, int cin, int cout, int kvol, float* __restrict__ sink,
                                              unsigned long long* __restrict__ clk) {
  constexpr int BM = 256, BK = 64, PIECE = 128 * BK, STAGE_ELEMS = 4 * PIECE;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int tile = xcd_tile(blockIdx.x, (n_out + BM - 1) / BM);
  if (tile < 0) return;
  const unsigned long long t0 = clock64();
  const int m0 = tile * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int nstage = kvol * (cin / BK);
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);
  constexpr unsigned row_bytes = 512u;
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_part16[4], w_voff[2][4];
  int mrow[2][4];
  bool live[2][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = (wv * 4 + u) * 8 + lrow;
    a_part16[u] = (unsigned)(lslot ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const int ar = sp * 128 + r;                           // piece A_sp = tile rows sp*128 .. +127 (the wave row sp's rows)
      mrow[sp][u] = min(m0 + ar, n_out - 1);
      live[sp][u] = m0 + ar < n_out;
      const int tc = sp * 128 + r;
      w_voff[sp][u] = (unsigned)(tc * cin + (lslot ^ ((r >> 1) & 7)) * 8) * 2u;
    }
  }
  int idx_cur[2][4], idx_nxt[2][4];
  auto load_idx_next = [&](int stage) {
    const int* row = nbr + (long long)(stage % kvol) * ld;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 4; ++u) idx_nxt[sp][u] = row[mrow[sp][u]];
  };
  auto advance_idx = [&]() {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 4; ++u) idx_cur[sp][u] = live[sp][u] ? idx_nxt[sp][u] : -1;
  };
  auto issue = [&](int st, int buf) {
    const unsigned soff_a = (unsigned)((st / kvol) * BK) * 2u;
    const unsigned soff_b = (unsigned)((st % kvol) * cin * cout + (st / kvol) * BK) * 2u;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u16* da = smem + buf * STAGE_ELEMS + sp * PIECE + (wv * 4 + u) * 512;
        const unsigned voff = idx_cur[sp][u] >= 0 ? (unsigned)idx_cur[sp][u] * row_bytes + a_part16[u] : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void_ptr)da, 16, voff, soff_a, 0, 0);
        u16* db = smem + buf * STAGE_ELEMS + (2 + sp) * PIECE + (wv * 4 + u) * 512;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_ptr)db, 16, w_voff[sp][u], soff_b, 0, 0);
      }
  };
  const int g = lane >> 4, li = lane & 15, fsw = (lane >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = ((ks * 4 + g) ^ fsw) << 3;
  typedef const volatile s16x8 __attribute__((address_space(3))) * lds_vptr;
  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ra[8], rb[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ra[a][e] = (__bf16)(0.01f * (float)((lane * 7 + a * 3 + e) % 17 - 8)); rb[a][e] = (__bf16)(0.02f * (float)((lane * 5 + a * 11 + e) % 13 - 6)); }
  if (LOADS) {
    load_idx_next(0);
    advance_idx();
    issue(0, 0);
    load_idx_next(1);
    advance_idx();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int st2 = 0; st2 < nstage; st2 += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int st = st2 + par, buf = par;
      const int nx = st + 1 < nstage ? st + 1 : st;
      if (LOADS) {
        load_idx_next(st + 2 < nstage ? st + 2 : nstage - 1);
        issue(nx, buf ^ 1);
      }
      const u16* A = smem + buf * STAGE_ELEMS + wm * PIECE + li * BK;
      const u16* B = smem + buf * STAGE_ELEMS + (2 + wn) * PIECE + li * BK;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[8], bfr[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          if (READS) { s16x8 v = *(lds_vptr)(A + a * 16 * BK + foff[ks]); af[a] = __builtin_bit_cast(bf16x8, v); } else af[a] = ra[a];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (READS) { s16x8 v = *(lds_vptr)(B + b * 16 * BK + foff[ks]); bfr[b] = __builtin_bit_cast(bf16x8, v); } else bfr[b] = rb[b];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int a = 0; a < 8; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
      }
      if (LOADS) {
        advance_idx();
        __builtin_amdgcn_s_waitcnt(0x0F70 | 0x4000);          // vmcnt(16)
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  __syncthreads();
  if (s == 12345.678f) sink[tid] = s;
  if (tid == 0) clk[tile] = clock64() - t0;
}

struct Variant { const char* name; void (*kern)(const u16*, const u16*, const int*, int, int, int, int, int, float*, unsigned long long*); bool loads; bool mfma; int threads = 512; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int B = 8, Z = 15, Y = 40, X = 40, cin = 256, cout = 256, kvol = 27;
  const int n = B * Z * Y * X;
  // dense 3x3x3 neighbour table of the lattice (what u3d_dense_nbr_table builds): nbr[k][m] = row of the input cell, -1 outside
  std::vector<int> h_nbr((size_t)kvol * n);
  for (int k = 0; k < kvol; ++k) {
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    for (int m = 0; m < n; ++m) {
      const int x = m % X, y = (m / X) % Y, z = (m / (X * Y)) % Z, b = m / (X * Y * Z);
      const int xx = x + dx, yy = y + dy, zz = z + dz;
      h_nbr[(size_t)k * n + m] = (xx < 0 || xx >= X || yy < 0 || yy >= Y || zz < 0 || zz >= Z) ? -1 : ((b * Z + zz) * Y + yy) * X + xx;
    }
  }
  std::vector<u16> h_in((size_t)n * cin), h_w((size_t)kvol * cin * cout);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (u16)(0x3C00u + ((s >> 9) & 0x3FFu) + ((s >> 31) << 15)); };   // bf16 of magnitude ~1, random sign
  for (auto& v : h_in) v = rnd();
  for (auto& v : h_w) v = rnd();
  u16 *d_in, *d_w; int* d_nbr; float* d_sink; unsigned long long* d_clk;
  const int tiles = (n + 255) / 256;
  CK(hipMalloc(&d_in, h_in.size() * 2)); CK(hipMalloc(&d_w, h_w.size() * 2)); CK(hipMalloc(&d_nbr, h_nbr.size() * 4));
  CK(hipMalloc(&d_sink, 4096)); CK(hipMalloc(&d_clk, tiles * 8));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, h_w.data(), h_w.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_nbr, h_nbr.data(), h_nbr.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int grid = ((tiles + 7) / 8) * 8;
  const size_t lds = 2 * 4 * 128 * 64 * 2;                 // two 64 KiB stages
  Variant vs[] = {
      {"dma_stream                    (LDS-DMA only, the layer's stream)", k_fill<0, false, false, 0>, true, false},
      {"dma_window1024                (LDS-DMA only, L2-resident rows)", k_fill<0, true, false, 0>, true, false},
      {"gload_stream                  (global_load_dwordx4 -> VGPR, no LDS)", k_fill<1, false, false, 0>, true, false},
      {"gload_window1024", k_fill<1, true, false, 0>, true, false},
      {"dma_stream + frag reads", k_fill<0, false, true, 0>, true, false},
      {"mfma16x16x32 only             (register operands, no loads)", k_fill<2, false, false, 1>, false, true},
      {"mfma32x32x16 only", k_fill<2, false, false, 2>, false, true},
      {"mfma16x16x32 + frag reads     (no loads)", k_fill<2, false, true, 1>, false, true},
      {"mfma32x32x16 + frag reads     (no loads)", k_fill<2, false, true, 2>, false, true},
      {"dma_stream + mfma16 (reg operands)", k_fill<0, false, false, 1>, true, true},
      {"dma_stream + mfma32 (reg operands)", k_fill<0, false, false, 2>, true, true},
      {"dma_stream + reads + mfma16   (= the product loop, unscheduled)", k_fill<0, false, true, 1>, true, true},
      {"dma_stream + reads + mfma32", k_fill<0, false, true, 2>, true, true},
      {"dma_window + reads + mfma16", k_fill<0, true, true, 1>, true, true},
      {"dma_stream + PIPELINED reads + mfma16 (fragments one group ahead)", k_fill<0, false, true, 3>, true, true},
      {"PIPELINED reads + mfma16 (no loads)", k_fill<2, false, true, 3>, false, true},
      {"PIPELINED + LDS-DMA spread: one request per MFMA group", k_fill<0, false, true, 3, 1>, true, true},
      {"PIPELINED + LDS-DMA spread: two per group, first four groups", k_fill<0, false, true, 3, 2>, true, true},
      {"PIPELINED, one k-tile MORE in flight than two buffers allow (timing only)", k_fill<0, false, true, 3, 0, true>, true, true},
      {"unscheduled loop, one k-tile more in flight (timing only)", k_fill<0, false, true, 1, 0, true>, true, true},
      {"PIPELINED + s_setprio(1) around the MFMA groups", k_fill<0, false, true, 3, 4>, true, true},
      {"PIPELINED, B fragments two groups ahead", k_fill<0, false, true, 3, 5>, true, true},
      {"B FROM REGISTERS: A by LDS-DMA, B by global loads, A reads + mfma16", k_fill<3, false, true, 1>, true, true},
      {"B FROM REGISTERS, no MFMA (A dma + B loads + A reads)", k_fill<3, false, true, 0>, true, false},
      {"ROLES: 4 waves MFMA only | 4 waves idle", k_roles<false, true, false>, false, true},
      {"ROLES: 4 waves MFMA | 4 waves DMA", k_roles<false, true, true>, true, true},
      {"ROLES: 4 waves MFMA | 4 waves DMA + all frag reads", k_roles<true, true, true>, true, true},
      {"ROLES: 4 waves idle | 4 waves DMA + all frag reads", k_roles<true, false, true>, true, false},
      {"WIDE (4 waves x 128x128): mfma only", k_wide<false, false>, false, true, 256},
      {"WIDE: mfma + frag reads", k_wide<false, true>, false, true, 256},
      {"WIDE: dma + mfma", k_wide<true, false>, true, true, 256},
      {"WIDE: dma + frag reads + mfma", k_wide<true, true>, true, true, 256},
  };
  const double staged = (double)tiles * kvol * (cin / 64) * 65536.0;      // bytes moved into LDS (or registers) per launch
  const double flops = 2.0 * (double)tiles * 256 * 256 * kvol * cin;     // all tiles full (192 000 = 750 x 256)
  printf("# fill_rate: %d CUs, %d tiles x %d k-tiles x 64 KiB = %.2f GB staged per launch, %.1f GFLOP where MFMAs run; iters %d\n", cus, tiles,
         kvol * (cin / 64), staged / 1e9, flops / 1e9, iters);
  printf("# %-66s %9s %9s %11s %11s %9s %9s\n", "variant", "us", "TB/s", "B/clk/CU@2.4", "B/clk/WG", "TFLOP/s", "eff GHz");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> h_clk(tiles);
  for (auto& v : vs) {
    CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), lds, 0, d_in, d_w, d_nbr, n, n, cin, cout, kvol, d_sink, d_clk);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), lds, 0, d_in, d_w, d_nbr, n, n, cin, cout, kvol, d_sink, d_clk);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2] * 1e3;
    CK(hipMemcpy(h_clk.data(), d_clk, tiles * 8, hipMemcpyDeviceToHost));
    // effective shader clock: a workgroup's s_memtime span against its share of the wall time (tiles / CUs waves of workgroups)
    double csum = 0; for (auto c : h_clk) csum += (double)c;
    const double eff_ghz = csum / cus / (us * 1e3);        // cycles every CU spent inside workgroups / wall ns (upper bound on busy clock)
    const double tbs = v.loads ? staged / us / 1e6 : 0.0;
    const double bpc = v.loads ? staged / (us * 1e-6) / 2.4e9 / cus : 0.0;
    const double bpce = v.loads ? (double)kvol * (cin / 64) * 65536.0 / (csum / tiles) : 0.0;      // per workgroup: bytes / its own s_memtime span
    printf("  %-66s %9.1f %9.2f %11.1f %11.1f %9.0f %9.2f\n", v.name, us, tbs, bpc, bpce, v.mfma ? flops / us / 1e6 : 0.0, eff_ghz);
    fflush(stdout);
  }
  return 0;
}
