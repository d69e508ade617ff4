// Query-side kernels: farthest point sampling, match cost, rectangular assignment (one wavefront per
// problem), aligned rotated 3-D IoU.  Latency-bound work: one launch each for the whole batch, no host syncs.
#include "common.h"
#include <math.h>

// ============================================================================================
// D-FPS (ref: models/detectors/uni3detr.py:138,178-187; upstream mmcv furthest_point_sample).
// One workgroup (1024 threads x 20 register-resident points) per point set.  Point k of set s = triple base[3k..3k+2] (the upstream
// kernel's packed-triple view of the buffer).  Start index 0, running min of squared L2, arg-max with the
// upstream block-reduction tie rule: smallest (k mod T, k), T = min(1024, 2^floor(log2 n)).
// ============================================================================================
#define FPS_THREADS 1024
#define FPS_MAXJ 20   // register-resident points per thread (n <= 20480); larger sets stream min-dist via `temp`

// candidate (d, k) beats (bd, bk): larger distance, or equal distance and smaller (k mod T, k); T is a power of two
__device__ __forceinline__ bool fps_better(float d, int k, float bd, int bk, unsigned tmask, unsigned n) {
  if (d != bd) return d > bd;
  unsigned t1 = ((unsigned)k & tmask) * n + (unsigned)k, t2 = ((unsigned)bk & tmask) * n + (unsigned)bk;
  return t1 < t2;
}

// One round = distance update + arg-max.  In the streaming path the winner's COORDINATES travel with (distance, index) through the
// reductions.  When T == FPS_THREADS (every set with n >= 1024) all points of a thread share k mod T = tid and come in ascending k, so inside a
// thread the tie rule degenerates to "first maximum wins": a strict compare, no tie keys (-6 VALU per point).
struct FpsBest { float d; int k; float x, y, z; };
__device__ __forceinline__ void fps_take(FpsBest& b, bool c, float d, int k, float x, float y, float z) {
  b.d = c ? d : b.d; b.k = c ? k : b.k; b.x = c ? x : b.x; b.y = c ? y : b.y; b.z = c ? z : b.z;
}
__device__ __forceinline__ void fps_merge_shfl(FpsBest& b, int o, unsigned tmask, unsigned n) {
  const float od = __shfl_xor(b.d, o, 64); const int ok = __shfl_xor(b.k, o, 64);
  const float ox = __shfl_xor(b.x, o, 64), oy = __shfl_xor(b.y, o, 64), oz = __shfl_xor(b.z, o, 64);
  fps_take(b, fps_better(od, ok, b.d, b.k, tmask, n), od, ok, ox, oy, oz);
}

template <bool REG, bool FIRST_WINS>
__device__ __forceinline__ void fps_rounds(const float* __restrict__ p, int n, int m, int* __restrict__ out, float* __restrict__ tmp,
                                           unsigned tmask, float* s_d, int* s_k, float* s_x, float* s_y, float* s_z, int* s_win) {
  constexpr int NW = FPS_THREADS / 64;
  const unsigned un = (unsigned)n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float px[REG ? FPS_MAXJ : 1], py[REG ? FPS_MAXJ : 1], pz[REG ? FPS_MAXJ : 1], md[REG ? FPS_MAXJ : 1];
  if (REG) {
#pragma unroll
    for (int j = 0; j < FPS_MAXJ; ++j) {
      int k = tid + j * FPS_THREADS;
      px[j] = py[j] = pz[j] = 0.f;
      if (k < n) { px[j] = p[3 * k]; py[j] = p[3 * k + 1]; pz[j] = p[3 * k + 2]; }
      md[j] = k < n ? 1e10f : -1.f;           // padding never wins (all real distances are >= 0)
    }
  } else {
    for (int k = tid; k < n; k += FPS_THREADS) tmp[k] = 1e10f;
  }
  if (tid == 0) { out[0] = 0; s_win[0] = 0; s_x[0] = p[0]; s_y[0] = p[1]; s_z[0] = p[2]; }
  __syncthreads();
  for (int r = 1; r < m; ++r) {
    const int wprev = s_win[0];                   // the wave whose candidate won the previous round
    const float cx = s_x[wprev], cy = s_y[wprev], cz = s_z[wprev];
    FpsBest b = {-2.f, 0x7fffffff, 0.f, 0.f, 0.f};
    int tl = tid;
    asm volatile("" : "+v"(tl));      // launder: stops LICM from hoisting 2 x FPS_MAXJ index/tie registers out of the round loop
    if (REG) {
#pragma unroll
      for (int j = 0; j < FPS_MAXJ; ++j) {
        float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        float v = fminf(md[j], d);
        md[j] = v;
        int k = tl + j * FPS_THREADS;
        const bool c = FIRST_WINS ? (v > b.d) : fps_better(v, k, b.d, b.k, tmask, un);
        b.d = c ? v : b.d; b.k = c ? k : b.k;
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound live temporaries: the points stay register-resident
      }
    } else {
      for (int k = tid; k < n; k += FPS_THREADS) {
        const float x = p[3 * k], y = p[3 * k + 1], z = p[3 * k + 2];
        float dx = x - cx, dy = y - cy, dz = z - cz;
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        float v = fminf(tmp[k], d);
        tmp[k] = v;
        fps_take(b, fps_better(v, k, b.d, b.k, tmask, un), v, k, x, y, z);
      }
    }
    if (REG) {
      // wave arg-max of (distance, index) alone: the register-resident path re-reads the winner's coordinates from global memory (L2)
      // below - every way of taking them out of the owner lane's registers (selects in the point loop, a uniform switch, scalar-
      // condition selects after the butterfly) tipped the 128-VGPR allocation of this 1024-thread kernel into thousands of spills
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float od = __shfl_xor(b.d, o, 64); const int ok = __shfl_xor(b.k, o, 64);
        const bool c = fps_better(od, ok, b.d, b.k, tmask, un);
        b.d = c ? od : b.d; b.k = c ? ok : b.k;
      }
      const int kw = __builtin_amdgcn_readfirstlane(b.k);
      // every wave's candidate re-reads ITS coordinates from global memory (L2) now: 16 loads in flight underneath the barrier and wave
      // 0's reduction, instead of one dependent load of the winner after them
      float wx = 0.f, wy = 0.f, wz = 0.f;
      if (lane == 0) { wx = p[3 * kw]; wy = p[3 * kw + 1]; wz = p[3 * kw + 2]; s_d[wid] = b.d; s_k[wid] = kw; }
      __syncthreads();
      if (lane == 0) { s_x[wid] = wx; s_y[wid] = wy; s_z[wid] = wz; }
    } else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) fps_merge_shfl(b, o, tmask, un);
      if (lane == 0) { s_d[wid] = b.d; s_k[wid] = b.k; }
      __syncthreads();
      if (lane == 0) { s_x[wid] = b.x; s_y[wid] = b.y; s_z[wid] = b.z; }      // behind the barrier: a slow wave may still be reading s_x[wprev]
    }
    if (wid == 0) {
      float d2 = lane < NW ? s_d[lane] : -3.f; int k2 = lane < NW ? s_k[lane] : 0x7fffffff; int w2 = lane;
#pragma unroll
      for (int o = NW / 2; o > 0; o >>= 1) {
        const float od = __shfl_xor(d2, o, 64); const int ok = __shfl_xor(k2, o, 64), ow = __shfl_xor(w2, o, 64);
        const bool c = fps_better(od, ok, d2, k2, tmask, un);
        d2 = c ? od : d2; k2 = c ? ok : k2; w2 = c ? ow : w2;
      }
      if (lane == 0) { out[r] = k2; s_win[0] = w2; }
    }
    __syncthreads();      // (s_x / s_win are rewritten only behind the NEXT round's first barrier: every thread has read them by then)
  }
}

template <bool REG>
__global__ __launch_bounds__(FPS_THREADS) void k_fps(const float* __restrict__ base, const float* __restrict__ base2, int split,
                                                    const long long* __restrict__ set_off,
                                                    const int* __restrict__ set_n, int m, int* __restrict__ out_idx,
                                                    float* __restrict__ temp, long long temp_stride) {
  constexpr int NW = FPS_THREADS / 64;
  __shared__ float s_d[NW], s_x[NW], s_y[NW], s_z[NW];
  __shared__ int s_k[NW], s_win[1];
  const int s = blockIdx.x;
  const float* p = (s < split ? base : base2) + set_off[s];      // sets [split, nsets) live in a second buffer (u3d_fps2)
  const int n = set_n[s];
  int* out = out_idx + (long long)s * m;
  if (n <= 0) { for (int j = threadIdx.x; j < m; j += FPS_THREADS) out[j] = 0; return; }
  int T = 1;
  while ((T << 1) <= n && (T << 1) <= 1024) T <<= 1;
  float* tmp = temp + (long long)s * temp_stride;
  // first-maximum-wins for sets of >= 1024 points, the general tie rule for smaller ones (uniform branch: one path per workgroup); both
  // register-resident in the REG kernel (`temp` may be null there), both streaming through `temp` in the other
  if (REG && T == FPS_THREADS) fps_rounds<true, true>(p, n, m, out, tmp, (unsigned)T - 1u, s_d, s_k, s_x, s_y, s_z, s_win);
  else fps_rounds<REG, false>(p, n, m, out, tmp, (unsigned)T - 1u, s_d, s_k, s_x, s_y, s_z, s_win);
}

// --------------------------------------------------------------------------------------------
// Sets of more than FPS_THREADS * FPS_MAXJ = 20 480 points (nuScenes: 250 000 raw points / up to 90 000 voxels per scene, 900 samples;
// ScanNet-large: 100 000): the streaming path above walks such a set with ONE workgroup - 72 us per round, 65 ms of the 85 ms nuScenes
// step.  Here a set is split over W = ceil(n / 20 480) workgroups (<= FPS_MULTI_MAXW), each keeps its 20 480-point slice REGISTER-resident
// exactly like the single-workgroup kernel (point k of slice w: k = w * 20 480 + tid + j * 1024, so k mod T = tid as before and the
// in-thread "first maximum wins" rule holds), and the W local winners meet once per round through data-tagged 8-byte granules
// ({value, round} written by ONE agent-scope store, polled with agent-scope loads: no fence, no counter; MI355X_MICROARCH.md
// "handoff-1to1"): slots[round & 1][w] = {distance bits | round} {index | round}.  Two slot sets by round parity: a workgroup can be at
// most one round ahead of the slowest one (it needs everybody's round-r candidate before it can produce round r + 1).  Every
// workgroup reduces the W candidates with the same total order (fps_better) and fetches the winner's coordinates from the (read-only)
// input itself.  All workgroups of a launch must be resident together (one 1024-thread workgroup owns a CU): the launcher sends the sets
// out in groups of at most 3/4 of the DEVICE's CUs (hipDeviceAttributeMultiprocessorCount; the rest is room for the kernels of another
// stream) and takes the single-workgroup streaming kernel when one set alone would need more.
// Time-out: the wait for the siblings is bounded in WALL TIME (s_memrealtime, 100 MHz; FPS_POLL_TICKS = 0.5 s unless the caller passes
// its own limit).  The first workgroup that gives up sets err[0] = 1 (and counts the call in err[1]); every poll loop also watches
// err[0], so all other workgroups of the call leave within one poll instead of waiting out their own limit round after round; rounds
// that did not complete write index 0 (a valid point).  The caller must not use the samples of a call that left err[0] != 0: the
// training step feeds err[0] into its collective HOLD flag (uni3detr_amd/trainer.py), so no rank applies an update computed from them.
// --------------------------------------------------------------------------------------------
#ifndef FPS_MULTI
#define FPS_MULTI 1
#endif
#define FPS_MULTI_MAXW 16
#define FPS_CHUNK (FPS_THREADS * FPS_MAXJ)
#define FPS_POLL_TICKS 50000000ll      /* 0.5 s of the 100 MHz constant clock */
__device__ __forceinline__ void fps_rounds_multi(const float* __restrict__ p, int n, int m, int* __restrict__ out, int w, int W,
                                                 unsigned long long* __restrict__ slots, int* __restrict__ err, long long poll_ticks,
                                                 float* s_d, int* s_k, float* s_x, float* s_y, float* s_z, int* s_abort) {
  constexpr int NW = FPS_THREADS / 64;
  const unsigned un = (unsigned)n, tmask = FPS_THREADS - 1u;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kbase = w * FPS_CHUNK;
  float px[FPS_MAXJ], py[FPS_MAXJ], pz[FPS_MAXJ], md[FPS_MAXJ];
#pragma unroll
  for (int j = 0; j < FPS_MAXJ; ++j) {
    const int k = kbase + tid + j * FPS_THREADS;
    px[j] = py[j] = pz[j] = 0.f;
    if (k < n) { px[j] = p[3 * (long long)k]; py[j] = p[3 * (long long)k + 1]; pz[j] = p[3 * (long long)k + 2]; }
    md[j] = k < n ? 1e10f : -1.f;
  }
  if (w == 0 && tid == 0) out[0] = 0;
  if (tid == 0) { s_x[0] = p[0]; s_y[0] = p[1]; s_z[0] = p[2]; s_abort[0] = 0; }
  __syncthreads();
  for (int r = 1; r < m; ++r) {
    if (s_abort[0]) {                                         // a workgroup of this call timed out (uniform: written before the round's last barrier)
      if (w == 0) for (int j = r + tid; j < m; j += FPS_THREADS) out[j] = 0;
      return;
    }
    const float cx = s_x[0], cy = s_y[0], cz = s_z[0];
    float bd = -2.f; int bk = 0x7fffffff;
    int tl = kbase + tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int j = 0; j < FPS_MAXJ; ++j) {
      float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
      float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float v = fminf(md[j], d);
      md[j] = v;
      const bool c = v > bd;                                  // first maximum wins inside a thread (ascending k, one k mod T)
      bd = c ? v : bd; bk = c ? tl + j * FPS_THREADS : bk;
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 64); const int ok = __shfl_xor(bk, o, 64);
      const bool c = fps_better(od, ok, bd, bk, tmask, un);
      bd = c ? od : bd; bk = c ? ok : bk;
    }
    if (lane == 0) { s_d[wid] = bd; s_k[wid] = bk; }
    __syncthreads();                                          // (everybody has read s_x[0] of the previous round by now)
    if (wid == 0) {
      float d2 = lane < NW ? s_d[lane] : -3.f; int k2 = lane < NW ? s_k[lane] : 0x7fffffff;
#pragma unroll
      for (int o = NW / 2; o > 0; o >>= 1) {
        const float od = __shfl_xor(d2, o, 64); const int ok = __shfl_xor(k2, o, 64);
        const bool c = fps_better(od, ok, d2, k2, tmask, un);
        d2 = c ? od : d2; k2 = c ? ok : k2;
      }
      d2 = __shfl(d2, 0, 64); k2 = __shfl(k2, 0, 64);         // this workgroup's candidate of round r
      unsigned long long* sl = slots + (size_t)(r & 1) * FPS_MULTI_MAXW * 2;
      const unsigned long long tag = (unsigned long long)(unsigned)r << 32;
      if (lane == 0) {
        __hip_atomic_store(sl + w * 2, tag | (unsigned long long)__float_as_uint(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sl + w * 2 + 1, tag | (unsigned long long)(unsigned)k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // lane l < W collects workgroup l's candidate (its own included: one code path)
      float dd = -3.f; int kk = 0x7fffffff;
      bool ready = lane >= W;
      bool failed = false;
      const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
      while (!__all(ready)) {
        if (!ready) {
          const unsigned long long g0 = __hip_atomic_load(sl + lane * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long g1 = __hip_atomic_load(sl + lane * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((g0 >> 32) == (unsigned)r && (g1 >> 32) == (unsigned)r) {
            dd = __uint_as_float((unsigned)g0); kk = (int)(unsigned)g1; ready = true;
          }
        }
        if (__all(ready)) break;
        // somebody else of this call gave up (leave with them), or a sibling has not arrived within the wall-time limit (flag it)
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { failed = true; break; }
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > poll_ticks) {
          if (lane == 0 && atomicExch(err, 1) == 0) atomicAdd(err + 1, 1);
          failed = true; break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (failed) {
        if (lane == 0) { s_abort[0] = 1; if (w == 0) out[r] = 0; }
        dd = -3.f; kk = 0;
      }
#pragma unroll
      for (int o = FPS_MULTI_MAXW / 2; o > 0; o >>= 1) {
        const float od = __shfl_xor(dd, o, 64); const int ok = __shfl_xor(kk, o, 64);
        const bool c = fps_better(od, ok, dd, kk, tmask, un);
        dd = c ? od : dd; kk = c ? ok : kk;
      }
      kk = __shfl(kk, 0, 64);
      if (lane == 0) {
        const int kw = ((unsigned)kk < un && !failed) ? kk : 0;
        s_x[0] = p[3 * (long long)kw]; s_y[0] = p[3 * (long long)kw + 1]; s_z[0] = p[3 * (long long)kw + 2];
        if (w == 0) out[r] = kw;
      }
    }
    __syncthreads();
  }
}

__global__ void k_fps_zero(unsigned long long* __restrict__ p, int n, int* __restrict__ err) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0ull;
  if (threadIdx.x == 0 && err) err[0] = 0;                  // err[1] (calls that timed out so far) is the caller's to clear
}

__global__ __launch_bounds__(FPS_THREADS) void k_fps_multi(const float* __restrict__ base, const float* __restrict__ base2, int split,
                                                          const long long* __restrict__ set_off, const int* __restrict__ set_n, int m,
                                                          int* __restrict__ out_idx, unsigned long long* __restrict__ slots,
                                                          int* __restrict__ err, long long poll_ticks, int s0) {
  constexpr int NW = FPS_THREADS / 64;
  __shared__ float s_d[NW], s_x[NW], s_y[NW], s_z[NW];
  __shared__ int s_k[NW], s_win[1], s_abort[1];
  const int s = s0 + blockIdx.y, w = blockIdx.x;      // (s0: first set of this launch - many sets go out in groups that fit the chip)
  const float* p = (s < split ? base : base2) + set_off[s];
  const int n = set_n[s];
  int* out = out_idx + (long long)s * m;
  const int W = n > 0 ? (n + FPS_CHUNK - 1) / FPS_CHUNK : 1;
  if (w >= W) return;
  if (n <= 0) { for (int j = threadIdx.x; j < m; j += FPS_THREADS) out[j] = 0; return; }
  if (W == 1) {                                     // a small set next to large ones: the single-workgroup rounds, no exchange
    int T = 1;
    while ((T << 1) <= n && (T << 1) <= 1024) T <<= 1;
    if (T == FPS_THREADS) fps_rounds<true, true>(p, n, m, out, nullptr, (unsigned)T - 1u, s_d, s_k, s_x, s_y, s_z, s_win);
    else fps_rounds<true, false>(p, n, m, out, nullptr, (unsigned)T - 1u, s_d, s_k, s_x, s_y, s_z, s_win);
    return;
  }
  fps_rounds_multi(p, n, m, out, w, W, slots + (size_t)s * 2 * FPS_MULTI_MAXW * 2, err, poll_ticks, s_d, s_k, s_x, s_y, s_z, s_abort);
}

static int fps_launch(const float* base, const float* base2, int split, const int64_t* set_off, const int32_t* set_n, int32_t nsets,
                      int32_t max_n, int32_t m, int32_t* out_idx, float* temp, int64_t temp_stride, int32_t* err, int64_t poll_ticks,
                      int32_t max_wg, hipStream_t s) {
  U3D_REQUIRE(base && base2 && set_off && set_n && out_idx && nsets > 0 && m > 0 && poll_ticks >= 0 && max_wg >= 0, U3D_ERR_ARG);
  const int W = u3d_cdiv(max_n, FPS_CHUNK);
  int budget = max_wg;                                 // workgroups that may be asked to be resident together
  if (FPS_MULTI && W > 1 && W <= FPS_MULTI_MAXW && budget == 0) {
    // one 1024-thread workgroup of k_fps_multi owns a CU: 3/4 of THIS device's CUs, the rest is room for another stream's kernels
    int dev = 0, cus = 0;
    U3D_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess,
                U3D_ERR_LAUNCH);
    budget = cus * 3 / 4;
  }
  if (max_n <= FPS_CHUNK) {
    hipLaunchKernelGGL(k_fps<true>, dim3(nsets), dim3(FPS_THREADS), 0, s, base, base2, split, (const long long*)set_off, set_n, m, out_idx, temp, (long long)temp_stride);
    if (err) hipLaunchKernelGGL(k_fps_zero, dim3(1), dim3(64), 0, s, (unsigned long long*)nullptr, 0, err);
  } else if (FPS_MULTI && W <= FPS_MULTI_MAXW && W <= budget) {
    // large sets split over several resident workgroups (k_fps_multi); the head of `temp` carries the per-set candidate slots
    // (zeroed here: round tags start at 1), so the workspace contract of the streaming path covers it; err is the caller's buffer
    U3D_REQUIRE(temp && temp_stride >= max_n && err, U3D_ERR_WORKSPACE);
    const size_t slot_bytes = (size_t)nsets * 2 * FPS_MULTI_MAXW * 2 * 8;
    U3D_REQUIRE((size_t)nsets * (size_t)temp_stride * 4 >= slot_bytes, U3D_ERR_WORKSPACE);
    // (a kernel, not hipMemsetAsync: the memset NODE of a captured graph left a repeating 16-byte pattern of two device pointers
    //  in this range on replay instead of zeros - ROCm 7.2, HISTORY.md; stale round tags of the previous replay must go)
    hipLaunchKernelGGL(k_fps_zero, dim3(1), dim3(256), 0, s, (unsigned long long*)temp, (int)(slot_bytes / 8), err);
    const int per = budget / W;                        // sets per launch: all their workgroups must be resident together
    for (int s0 = 0; s0 < nsets; s0 += per) {
      const int ns = nsets - s0 < per ? nsets - s0 : per;
      hipLaunchKernelGGL(k_fps_multi, dim3(W, ns), dim3(FPS_THREADS), 0, s, base, base2, split, (const long long*)set_off, set_n, m, out_idx,
                         (unsigned long long*)temp, err, (long long)(poll_ticks > 0 ? poll_ticks : FPS_POLL_TICKS), s0);
    }
  } else {
    U3D_REQUIRE(temp && temp_stride >= max_n, U3D_ERR_WORKSPACE);
    hipLaunchKernelGGL(k_fps<false>, dim3(nsets), dim3(FPS_THREADS), 0, s, base, base2, split, (const long long*)set_off, set_n, m, out_idx, temp, (long long)temp_stride);
    if (err) hipLaunchKernelGGL(k_fps_zero, dim3(1), dim3(64), 0, s, (unsigned long long*)nullptr, 0, err);
  }
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_fps(const float* base, const int64_t* set_off, const int32_t* set_n, int32_t nsets, int32_t max_n,
                           int32_t m, int32_t* out_idx, float* temp, int64_t temp_stride, int32_t* err, int64_t poll_ticks,
                           int32_t max_wg, u3d_stream s) {
  return fps_launch(base, base, nsets, set_off, set_n, nsets, max_n, m, out_idx, temp, temp_stride, err, poll_ticks, max_wg, (hipStream_t)s);
}
// the same over TWO buffers: sets [0, split) are offsets into `base`, sets [split, nsets) into `base2` (the detector's raw-point sets
// and voxel-coordinate sets without a concatenated copy of both)
extern "C" int32_t u3d_fps2(const float* base, const float* base2, int32_t split, const int64_t* set_off, const int32_t* set_n,
                            int32_t nsets, int32_t max_n, int32_t m, int32_t* out_idx, float* temp, int64_t temp_stride, int32_t* err,
                            int64_t poll_ticks, int32_t max_wg, u3d_stream s) {
  U3D_REQUIRE(split >= 0 && split <= nsets, U3D_ERR_ARG);
  return fps_launch(base, base2, split, set_off, set_n, nsets, max_n, m, out_idx, temp, temp_stride, err, poll_ticks, max_wg, (hipStream_t)s);
}

// ---------------------------------------------------------------------------------------------
// The detector's glue around the two FPS passes of a batch (ref: models/detectors/uni3detr.py:178-189), one launch each side:
//   k_fps_prep:   float-cast (z,y,x) voxel coordinates + the 2B set descriptors (B raw-point sets in the packed-triple view of the
//                 [N,F] point buffer, B voxel-coordinate sets)
//   k_fps_points: gather the sampled raw points (x,y,z) / voxel coordinates ((z,y,x) -> (x,y,z)), per-scene min / max over the m
//                 samples, affine map to the unit cube (shift_scale_points, ref :18-46), both groups concatenated: out [B, 2m, 3]
// ---------------------------------------------------------------------------------------------
__global__ void k_fps_prep(const int* __restrict__ coors, int v_rows, const int* __restrict__ scene_off, const int* __restrict__ voxel_off,
                           int B, int F, float* __restrict__ vox, long long* __restrict__ set_off, int* __restrict__ set_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < v_rows) {
    const int4 c = ((const int4*)coors)[i];
    vox[3 * i] = (float)c.y; vox[3 * i + 1] = (float)c.z; vox[3 * i + 2] = (float)c.w;
  }
  if (i < B) {
    set_off[i] = (long long)scene_off[i] * F;
    set_n[i] = scene_off[i + 1] - scene_off[i];
    set_off[B + i] = (long long)voxel_off[i] * 3;
    set_n[B + i] = voxel_off[i + 1] - voxel_off[i];
  }
}
extern "C" int32_t u3d_fps_prep(const int32_t* coors, int32_t v_rows, const int32_t* scene_off, const int32_t* voxel_off, int32_t batch,
                                int32_t nfeat, float* vox, int64_t* set_off, int32_t* set_n, u3d_stream s) {
  U3D_REQUIRE(scene_off && voxel_off && vox && set_off && set_n && batch > 0 && nfeat >= 3 && v_rows >= 0 && (coors || v_rows == 0), U3D_ERR_ARG);
  const int n = v_rows > batch ? v_rows : batch;
  hipLaunchKernelGGL(k_fps_prep, dim3(u3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, coors, v_rows, scene_off, voxel_off, batch, nfeat, vox,
                     (long long*)set_off, set_n);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

__global__ __launch_bounds__(256) void k_fps_points(const float* __restrict__ pts, int F, const float* __restrict__ vox,
                                                    const int* __restrict__ idx, const int* __restrict__ scene_off,
                                                    const int* __restrict__ voxel_off, int B, int m, float* __restrict__ out) {
  __shared__ float s_lo[4][3], s_hi[4][3];
  const int b = blockIdx.x, which = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int* ix = idx + ((long long)which * B + b) * m;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto fetch = [&](int j, float* v) {
    if (which == 0) {
      const float* p = pts + ((long long)scene_off[b] + ix[j]) * F;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
      const float* p = vox + ((long long)voxel_off[b] + ix[j]) * 3;
      v[0] = p[2]; v[1] = p[1]; v[2] = p[0];                        // (z,y,x) voxel coordinates -> (x,y,z)
    }
  };
  for (int j = tid; j < m; j += 256) {
    float v[3];
    fetch(j, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[c]); hi[c] = fmaxf(hi[c], v[c]); }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64)); }
    if (lane == 0) { s_lo[wid][c] = lo[c]; s_hi[wid][c] = hi[c]; }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = fminf(fminf(s_lo[0][c], s_lo[1][c]), fminf(s_lo[2][c], s_lo[3][c]));
    hi[c] = fmaxf(fmaxf(s_hi[0][c], s_hi[1][c]), fmaxf(s_hi[2][c], s_hi[3][c]));
  }
  float* o = out + ((long long)b * 2 * m + (long long)which * m) * 3;
  for (int j = tid; j < m; j += 256) {
    float v[3];
    fetch(j, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[3 * j + c] = __fdiv_rn(__fsub_rn(v[c], lo[c]), __fsub_rn(hi[c], lo[c]));      // ((x - lo) * (1 - 0)) / (hi - lo) + 0
  }
}
extern "C" int32_t u3d_fps_points(const float* pts, int32_t nfeat, const float* vox, const int32_t* idx, const int32_t* scene_off,
                                  const int32_t* voxel_off, int32_t batch, int32_t m, float* out, u3d_stream s) {
  U3D_REQUIRE(pts && vox && idx && scene_off && voxel_off && out && batch > 0 && m > 0 && nfeat >= 3, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_fps_points, dim3(batch, 2), dim3(256), 0, (hipStream_t)s, pts, nfeat, vox, idx, scene_off, voxel_off, batch, m, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Query assembly of Uni3DETRHead.forward (ref: dense_heads/uni3detr_head.py:436-455): G query groups of nq queries each,
//   group 0: (tgt_embed[:nq], refpoint_embed), groups g >= 1: (tgt_embed[nq:], inverse_sigmoid(points of group g)) -
//   query [B, G*nq, 256] | ref_logits [B, G*nq, 3] and their concatenation query_embeds [B, G*nq, 259] in ONE launch (the reference:
//   3 cats, 2 expands, an inverse_sigmoid of 6 element-wise ops); backward: the batch / group sums of the two embeddings' gradients.
//   points: fps [B, 2*nq, 3] (groups 1, 2) and, in the 4-group eval layout, rand [B, nq, 3] (group 3)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float inv_sigmoid_ref(float x) {      // torch: x.clamp(0, 1); log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
  const float eps = 1e-5f;
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(__fdiv_rn(fmaxf(x, eps), fmaxf(__fsub_rn(1.f, x), eps)));
}
__global__ __launch_bounds__(256) void k_query_embed(const float* __restrict__ tgt, const float* __restrict__ anchor,
                                                     const float* __restrict__ fps, const float* __restrict__ rnd, int B, int nq, int G,
                                                     int C, float* __restrict__ qe, float* __restrict__ query, float* __restrict__ ref,
                                                     float* __restrict__ ref_sig) {
  const long long row = blockIdx.x;                        // b * G*nq + g * nq + j
  const int N = G * nq, b = (int)(row / N), r = (int)(row % N), g = r / nq, j = r % nq;
  const float* t = tgt + (long long)(g == 0 ? j : nq + j) * C;
  float* qrow = qe + row * (C + 3);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = t[c];
    qrow[c] = v;
    query[row * C + c] = v;
  }
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    float v;
    if (g == 0) v = anchor[j * 3 + c];
    else if (g <= 2) v = inv_sigmoid_ref(fps[((long long)b * 2 * nq + (long long)(g - 1) * nq + j) * 3 + c]);
    else v = inv_sigmoid_ref(rnd[((long long)b * nq + j) * 3 + c]);
    qrow[C + c] = v;
    ref[row * 3 + c] = v;
    if (ref_sig) ref_sig[row * 3 + c] = 1.f / (1.f + expf(-v));      // init_reference of the transformer's return value
  }
}
extern "C" int32_t u3d_query_embed_fwd(const float* tgt, const float* anchor, const float* fps, const float* rnd, int32_t batch,
                                       int32_t nq, int32_t groups, int32_t c, float* query_embeds, float* query, float* ref, float* ref_sig, u3d_stream s) {
  U3D_REQUIRE(tgt && anchor && fps && query_embeds && query && ref && batch > 0 && nq > 0 && groups >= 1 && groups <= 4 && c > 0, U3D_ERR_ARG);
  U3D_REQUIRE(groups <= 3 || rnd, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_query_embed, dim3((unsigned)((long long)batch * groups * nq)), dim3(256), 0, (hipStream_t)s, tgt, anchor, fps, rnd, batch, nq,
                     groups, c, query_embeds, query, ref, ref_sig);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
// d_tgt [2*nq, C], d_anchor [nq, 3] from the gradients of the three outputs (any of them may be null): sums over scenes (and, for
// tgt_embed[nq:], over the groups that share it) in a fixed order
__global__ __launch_bounds__(256) void k_query_embed_bwd(const float* __restrict__ dqe, const float* __restrict__ dquery,
                                                         const float* __restrict__ dref, const float* __restrict__ dref_sig,
                                                         const float* __restrict__ anchor, int B, int nq, int G, int C,
                                                         float* __restrict__ dtgt, float* __restrict__ danchor) {
  const int N = G * nq;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long nt = (long long)2 * nq * C;
  if (i < nt) {
    const int tr = (int)(i / C), c = (int)(i % C);
    const int j = tr < nq ? tr : tr - nq, g0 = tr < nq ? 0 : 1, g1 = tr < nq ? 1 : G;
    float a = 0.f;
    for (int b = 0; b < B; ++b)
      for (int g = g0; g < g1; ++g) {
        const long long row = (long long)b * N + (long long)g * nq + j;
        if (dquery) a += dquery[row * C + c];
        if (dqe) a += dqe[row * (C + 3) + c];
      }
    dtgt[i] = a;
  } else if (i < nt + (long long)nq * 3) {
    const int k = (int)(i - nt), j = k / 3, c = k % 3;
    float a = 0.f, as = 0.f;
    for (int b = 0; b < B; ++b) {
      const long long row = (long long)b * N + j;
      if (dref) a += dref[row * 3 + c];
      if (dqe) a += dqe[row * (C + 3) + C + c];
      if (dref_sig) as += dref_sig[row * 3 + c];
    }
    if (dref_sig) {                                        // through ref_sig = sigmoid(anchor) (the first layer's box decode reads it)
      const float sg = 1.f / (1.f + expf(-anchor[k]));
      a += as * sg * (1.f - sg);
    }
    danchor[k] = a;
  }
}
extern "C" int32_t u3d_query_embed_bwd(const float* d_query_embeds, const float* d_query, const float* d_ref, const float* d_ref_sig,
                                       const float* anchor, int32_t batch, int32_t nq, int32_t groups, int32_t c, float* d_tgt,
                                       float* d_anchor, u3d_stream s) {
  U3D_REQUIRE(d_tgt && d_anchor && batch > 0 && nq > 0 && groups >= 1 && c > 0 && (!d_ref_sig || anchor), U3D_ERR_ARG);
  const long long n = (long long)2 * nq * c + (long long)nq * 3;
  hipLaunchKernelGGL(k_query_embed_bwd, dim3((unsigned)u3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, d_query_embeds, d_query, d_ref, d_ref_sig, anchor,
                     batch, nq, groups, c, d_tgt, d_anchor);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// box code helpers (ref: core/bbox/util.py:8-80)
// ============================================================================================
struct Box7 { float x, y, z, dx, dy, dz, yaw; };

__device__ __forceinline__ Box7 denorm_box(const float* c) {
  Box7 b;
  b.x = c[0]; b.y = c[1]; b.z = c[4];
  b.dx = expf(c[3]); b.dy = expf(c[2]); b.dz = expf(c[5]);      // (l, w, h) = exp(code 3, 2, 5)
  b.yaw = -atan2f(c[6], c[7]) - 1.57079632679489662f;
  return b;
}
__device__ __forceinline__ void norm_box(const float* g, float* c) {
  float rot = -g[6] - 1.57079632679489662f;
  c[0] = g[0]; c[1] = g[1]; c[2] = logf(g[4] + 1e-5f); c[3] = logf(g[3] + 1e-5f); c[4] = g[2]; c[5] = logf(g[5] + 1e-5f);
  c[6] = sinf(rot); c[7] = cosf(rot);
}
// nearest-BEV axis-aligned box (mmdet3d nearest_bev)
__device__ __forceinline__ void nearest_bev(float x, float y, float dx, float dy, float yaw, float* o) {
  const float PI = 3.14159265358979323846f;
  float r = yaw - floorf(yaw / PI + 0.5f) * PI;
  r = fabsf(r);
  float w = dx, h = dy;
  if (r > PI / 4) { w = dy; h = dx; }
  o[0] = x - w / 2; o[1] = y - h / 2; o[2] = x + w / 2; o[3] = y + h / 2;
}
__device__ __forceinline__ float iou_aa(const float* a, const float* b) {
  float a1 = (a[2] - a[0]) * (a[3] - a[1]), a2 = (b[2] - b[0]) * (b[3] - b[1]);
  float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  float ov = w * h;
  return ov / fmaxf(a1 + a2 - ov, 1e-6f);
}

// ============================================================================================
// match cost (ref: core/bbox/assigners/hungarian_assigner_3d.py:110-121; match_costs/match_cost.py:19-30,91-97;
// upstream FocalLossCost).  Output is GT-major: cost[p][g][q], p = (layer*B + b), q over ALL queries of the scene,
// so that the assignment kernel streams contiguous rows.  gt: gravity-centre boxes [sumG,7]; gt_off [B+1].
// ============================================================================================
__global__ void k_match_cost(const float* __restrict__ cls, const float* __restrict__ box, const float* __restrict__ gt,
                             const int* __restrict__ labels, const int* __restrict__ gt_off, int L, int B, int Q, int C,
                             int code, int gmax, float w_cls, float w_reg, float w_iou, float alpha, float gamma,
                             float* __restrict__ cost) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  int p = blockIdx.y;             // layer*B + b
  int b = p % B;
  if (q >= Q) return;
  int g0 = gt_off[b], G = gt_off[b + 1] - g0;
  if (G <= 0) return;
  const float* c = cls + ((long long)p * Q + q) * C;
  const float* bx = box + ((long long)p * Q + q) * code;
  float code8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) code8[j] = bx[j];
  Box7 pb = denorm_box(code8);
  float pbev[4];
  nearest_bev(pb.x, pb.y, pb.dx, pb.dy, pb.yaw, pbev);
  for (int g = 0; g < G; ++g) {
    const float* gb = gt + (long long)(g0 + g) * 7;
    float logit = c[labels[g0 + g]];
    float pr = 1.f / (1.f + expf(-logit));
    float neg = -logf(1.f - pr + 1e-12f) * (1.f - alpha) * powf(pr, gamma);
    float pos = -logf(pr + 1e-12f) * alpha * powf(1.f - pr, gamma);
    float ccls = (pos - neg) * w_cls;
    float gc[8];
    norm_box(gb, gc);
    float l1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) l1 += fabsf(code8[j] - gc[j]);
    float gbev[4];
    nearest_bev(gb[0], gb[1], gb[3], gb[4], gb[6], gbev);
    float ciou = (1.f - iou_aa(pbev, gbev)) * w_iou;
    cost[((long long)p * gmax + g) * Q + q] = ccls + l1 * w_reg + ciou;
  }
}

extern "C" int32_t u3d_match_cost(const float* cls, const float* box, const float* gt, const int32_t* labels,
                                  const int32_t* gt_off, int32_t nlayers, int32_t batch, int32_t nq_total, int32_t ncls,
                                  int32_t code_size, int32_t gmax, float w_cls, float w_reg, float w_iou, float alpha,
                                  float gamma, float* cost, u3d_stream s) {
  U3D_REQUIRE(cls && box && gt && labels && gt_off && cost && code_size >= 8 && gmax > 0, U3D_ERR_ARG);
  dim3 grid(u3d_cdiv(nq_total, 128), nlayers * batch);
  hipLaunchKernelGGL(k_match_cost, grid, dim3(128), 0, s, cls, box, gt, labels, gt_off, nlayers, batch, nq_total, ncls,
                     code_size, gmax, w_cls, w_reg, w_iou, alpha, gamma, cost);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Rectangular linear-sum assignment, one wavefront per problem, float64, all state in LDS — the
// shortest-augmenting-path algorithm of scipy.optimize.linear_sum_assignment (Crouse 2016) including its scan
// order over `remaining` and its tie rule (value < lowest, or == lowest and column unassigned), so that the
// result equals scipy's also on degenerate costs (ref: hungarian_assigner_3d.py:129-139).
// Problem (p, grp): rows = the G GTs of scene b (scipy transposes when rows > cols), cols = the nq queries of
// group grp; cost row i = cost[p][i][grp*nq .. +nq].   Output assigned[p][q] = 1-based gt index or 0.
// ============================================================================================
__global__ __launch_bounds__(64) void k_lsa(const float* __restrict__ cost, const int* __restrict__ gt_off, int B, int Q,
                                            int nq, int gmax, int* __restrict__ assigned) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ngrp = Q / nq;
  const int p = blockIdx.x / ngrp, grp = blockIdx.x % ngrp;
  const int b = p % B;
  const int G = gt_off[b + 1] - gt_off[b];
  const int lane = threadIdx.x;
  int* asg = assigned + (long long)p * Q + grp * nq;
  for (int j = lane; j < nq; j += 64) asg[j] = 0;
  if (G <= 0) return;
  const int nr = G < gmax ? G : gmax, nc = nq;
  const float* cm = cost + (long long)p * gmax * Q + grp * nq;   // row i at cm + i*Q
  double* v = (double*)smem;          // [nc]
  double* spc = v + nc;               // [nc] shortestPathCosts
  double* u = spc + nc;               // [gmax]
  int* row4col = (int*)(u + gmax);    // [nc]
  int* path = row4col + nc;           // [nc]
  int* remaining = path + nc;         // [nc]
  int* SC = remaining + nc;           // [nc]
  int* col4row = SC + nc;             // [gmax]
  int* SR = col4row + gmax;           // [gmax]
  for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; }
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  __syncthreads();
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int num_remaining = nc;
    for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; spc[it] = INF; SC[it] = 0; }
    for (int i = lane; i < nr; i += 64) SR[i] = 0;
    __syncthreads();
    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      const float* crow = cm + (long long)i * Q;
      double lo = INF; int idx = -1; int lo_un = 0;
      for (int it = lane; it < num_remaining; it += 64) {
        int j = remaining[it];
        double r = minVal + (double)crow[j] - ui - v[j];
        double sp = spc[j];
        if (r < sp) { path[j] = i; spc[j] = r; sp = r; }
        int un = row4col[j] == -1;
        if (sp < lo || (sp == lo && un)) { lo = sp; idx = it; lo_un = un; }
      }
      // combine lanes exactly like one sequential scan over it = 0..num_remaining-1 would: minimal value; among equal
      // values the LAST unassigned position if any is unassigned, else the FIRST position.
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        double olo = __shfl_xor(lo, o, 64); int oidx = __shfl_xor(idx, o, 64); int oun = __shfl_xor(lo_un, o, 64);
        bool take = false;
        if (oidx >= 0) {
          if (idx < 0 || olo < lo) take = true;
          else if (olo == lo) {
            if (oun && lo_un) take = oidx > idx;
            else if (oun) take = true;
            else if (!lo_un) take = oidx < idx;
          }
        }
        if (take) { lo = olo; idx = oidx; lo_un = oun; }
      }
      minVal = lo;
      if (idx < 0 || !(minVal < INF)) { sink = -2; break; }   // infeasible (inf / NaN costs): scene stays background
      int j = remaining[idx];
      int r4 = row4col[j];
      __syncthreads();
      if (lane == 0) { SC[j] = 1; remaining[idx] = remaining[num_remaining - 1]; }
      --num_remaining;
      if (r4 == -1) sink = j; else i = r4;
      __syncthreads();
    }
    if (sink < 0) break;
    if (lane == 0) u[cur] += minVal;
    for (int r = lane; r < nr; r += 64)
      if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
    for (int j = lane; j < nc; j += 64)
      if (SC[j]) v[j] -= minVal - spc[j];
    __syncthreads();
    if (lane == 0) {
      int j = sink;
      while (true) {
        int ii = path[j];
        row4col[j] = ii;
        int t = col4row[ii]; col4row[ii] = j; j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }
  for (int i2 = lane; i2 < nr; i2 += 64) {
    int j = col4row[i2];
    if (j >= 0) asg[j] = i2 + 1;
  }
}

static inline size_t lsa_lds_bytes(int nq, int gmax) { return (size_t)(2 * nq + gmax) * 8 + (size_t)(4 * nq + 2 * gmax) * 4; }

extern "C" int32_t u3d_lsa(const float* cost, const int32_t* gt_off, int32_t nlayers, int32_t batch, int32_t nq_total,
                           int32_t nq, int32_t gmax, int32_t* assigned, u3d_stream s) {
  U3D_REQUIRE(cost && gt_off && assigned && nq > 0 && nq_total % nq == 0 && gmax > 0, U3D_ERR_ARG);
  U3D_REQUIRE(gmax <= nq, U3D_ERR_UNSUPPORTED);
  size_t lds = lsa_lds_bytes(nq, gmax);
  U3D_REQUIRE(lds <= 160 * 1024, U3D_ERR_UNSUPPORTED);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_lsa, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nproblems = nlayers * batch * (nq_total / nq);
  hipLaunchKernelGGL(k_lsa, dim3(nproblems), dim3(64), lds, s, cost, gt_off, batch, nq_total, nq, gmax, assigned);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Aligned rotated 3-D IoU: diag(bbox_overlaps_3d(a, b)) without the O(N^2) matrix the reference builds
// (ref: models/dense_heads/uni3detr_head.py:695; upstream mmdet3d BaseInstance3DBoxes.overlaps + mmcv box_iou_rotated,
// z is treated as the BOTTOM face, BEV w/h clamped >= 1e-4).  a, b: [n,7] f32.
// ============================================================================================
struct P2 { float x, y; };

__device__ int clip_poly(const P2* in, int n, P2 a, P2 b, P2* out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    P2 p = in[i], q = in[(i + 1 == n) ? 0 : i + 1];
    float sp = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
    float sq = (b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x);
    if (sp >= 0.f) out[m++] = p;
    if ((sp >= 0.f) != (sq >= 0.f)) {
      float t = sp / (sp - sq);
      out[m++] = P2{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
    }
  }
  return m;
}

__device__ void rect_corners(float cx, float cy, float w, float h, float ang, P2* c) {
  float cs = cosf(ang), sn = sinf(ang);
  const float sx[4] = {-0.5f, 0.5f, 0.5f, -0.5f}, sy[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
  for (int i = 0; i < 4; ++i) {
    float x = sx[i] * w, y = sy[i] * h;
    c[i] = P2{cx + x * cs - y * sn, cy + x * sn + y * cs};
  }
}

__global__ void k_iou3d_rotated_aligned(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = a + (long long)i * 7;
  const float* q = b + (long long)i * 7;
  float w1 = fmaxf(p[3], 1e-4f), h1 = fmaxf(p[4], 1e-4f), w2 = fmaxf(q[3], 1e-4f), h2 = fmaxf(q[4], 1e-4f);
  float a1 = w1 * h1, a2 = w2 * h2;
  float iou2d = 0.f;
  if (a1 >= 1e-14f && a2 >= 1e-14f) {
    P2 poly[12], tmp[12], clipper[4];
    // work relative to the first centre: keeps f32 clipping accurate for far-away boxes
    rect_corners(0.f, 0.f, w1, h1, p[6], poly);
    rect_corners(q[0] - p[0], q[1] - p[1], w2, h2, q[6], clipper);
    int m = 4;
    for (int e = 0; e < 4 && m > 0; ++e) {
      m = clip_poly(poly, m, clipper[e], clipper[(e + 1) & 3], tmp);
      for (int t = 0; t < m; ++t) poly[t] = tmp[t];
    }
    float inter = 0.f;
    if (m >= 3) {
      for (int t = 0; t < m; ++t) {
        P2 u = poly[t], v = poly[(t + 1 == m) ? 0 : t + 1];
        inter += u.x * v.y - v.x * u.y;
      }
      inter = fabsf(inter) * 0.5f;
    }
    iou2d = inter / (a1 + a2 - inter);
  }
  float ov_bev = iou2d * (a1 + a2) / (1.f + iou2d);
  float top = fminf(p[2] + p[5], q[2] + q[5]), bot = fmaxf(p[2], q[2]);
  float ov_h = fmaxf(top - bot, 0.f);
  float ov = ov_bev * ov_h;
  float v1 = p[3] * p[4] * p[5], v2 = q[3] * q[4] * q[5];
  out[i] = ov / fmaxf(v1 + v2 - ov, 1e-8f);
}

extern "C" int32_t u3d_iou3d_rotated_aligned(const float* a, const float* b, int32_t n, float* out, u3d_stream s) {
  U3D_REQUIRE(a && b && out, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  hipLaunchKernelGGL(k_iou3d_rotated_aligned, dim3(u3d_cdiv(n, 128)), dim3(128), 0, s, a, b, n, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Trilinear sampling of a channels-last volume at query points — the "cross attention" of UniCrossAtten
// (ref: models/utils/uni3detr_transformer.py:342-345: F.grid_sample(value [B,C,D,H,W], grid [B,1,1,N,3]),
// mode bilinear(=trilinear), padding zeros, align_corners=False; grid (x,y,z) in [-1,1] indexes (W,H,D)).
// value: rows [B*D*H*W, C] (f32 or bf16), grid f32 [B,N,3], out [B,N,C] (value dtype).  One wavefront per query.
// backward: dvalue f32 [B*D*H*W, C] (+= atomics, caller zeroes), dgrid f32 [B,N,3].
// ============================================================================================
struct TriCorner { long long row[8]; float w[8]; float dwx[8], dwy[8], dwz[8]; };

__device__ __forceinline__ void tri_setup(const float* g, int b, int D, int H, int W, TriCorner& tc) {
  float ix = ((g[0] + 1.f) * W - 1.f) * 0.5f, iy = ((g[1] + 1.f) * H - 1.f) * 0.5f, iz = ((g[2] + 1.f) * D - 1.f) * 0.5f;
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  float tx = ix - fx, ty = iy - fy, tz = iz - fz;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
    int x = x0 + dx, y = y0 + dy, z = z0 + dz;
    float wx = dx ? tx : 1.f - tx, wy = dy ? ty : 1.f - ty, wz = dz ? tz : 1.f - tz;
    bool ok = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H && (unsigned)z < (unsigned)D;
    tc.row[c] = ok ? (((long long)b * D + z) * H + y) * W + x : -1;
    tc.w[c] = wx * wy * wz;
    tc.dwx[c] = (dx ? 1.f : -1.f) * wy * wz;
    tc.dwy[c] = (dy ? 1.f : -1.f) * wx * wz;
    tc.dwz[c] = (dz ? 1.f : -1.f) * wx * wy;
  }
}

__device__ __forceinline__ float ld_any(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float ld_any(const unsigned short* p, long long i) { return __uint_as_float(((unsigned)p[i]) << 16); }
__device__ __forceinline__ void st_any(float* p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void st_any(unsigned short* p, long long i, float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  p[i] = (unsigned short)(u >> 16);
}

template <typename T>
__global__ __launch_bounds__(256) void k_trilinear_fwd(const T* __restrict__ value, const float* __restrict__ grid, int B, int N,
                                                       int D, int H, int W, int C, T* __restrict__ out) {
  int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (q >= B * N) return;
  int b = q / N;
  TriCorner tc;
  tri_setup(grid + (long long)q * 3, b, D, H, W, tc);
  for (int ch = lane; ch < C; ch += 64) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (tc.row[c] >= 0) acc += tc.w[c] * ld_any(value, tc.row[c] * C + ch);
    st_any(out, (long long)q * C + ch, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_trilinear_bwd(const T* __restrict__ value, const float* __restrict__ grid,
                                                       const T* __restrict__ dout, int B, int N, int D, int H, int W, int C,
                                                       float* __restrict__ dvalue, float* __restrict__ dgrid) {
  int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (q >= B * N) return;
  int b = q / N;
  TriCorner tc;
  tri_setup(grid + (long long)q * 3, b, D, H, W, tc);
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int ch = lane; ch < C; ch += 64) {
    float go = ld_any(dout, (long long)q * C + ch);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (tc.row[c] >= 0) {
        if (dvalue) atomicAdd(&dvalue[tc.row[c] * C + ch], tc.w[c] * go);
        float v = ld_any(value, tc.row[c] * C + ch) * go;
        gx += tc.dwx[c] * v; gy += tc.dwy[c] * v; gz += tc.dwz[c] * v;
      }
  }
  if (dgrid) {
    gx = u3d_wave_sum(gx); gy = u3d_wave_sum(gy); gz = u3d_wave_sum(gz);
    if (lane == 0) {
      dgrid[(long long)q * 3 + 0] = gx * W * 0.5f;
      dgrid[(long long)q * 3 + 1] = gy * H * 0.5f;
      dgrid[(long long)q * 3 + 2] = gz * D * 0.5f;
    }
  }
}

extern "C" int32_t u3d_trilinear_fwd(const void* value, const float* grid, int32_t batch, int32_t nq, int32_t dz, int32_t dy,
                                     int32_t dx, int32_t c, void* out, int32_t dtype, u3d_stream s) {
  U3D_REQUIRE(value && grid && out && batch > 0 && nq > 0 && c > 0, U3D_ERR_ARG);
  dim3 g(u3d_cdiv((long long)batch * nq, 4));
  if (dtype == U3D_F32) hipLaunchKernelGGL(k_trilinear_fwd<float>, g, dim3(256), 0, s, (const float*)value, grid, batch, nq, dz, dy, dx, c, (float*)out);
  else if (dtype == U3D_BF16) hipLaunchKernelGGL(k_trilinear_fwd<unsigned short>, g, dim3(256), 0, s, (const unsigned short*)value, grid, batch, nq, dz, dy, dx, c, (unsigned short*)out);
  else return U3D_ERR_UNSUPPORTED;
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_trilinear_bwd(const void* value, const float* grid, const void* dout, int32_t batch, int32_t nq,
                                     int32_t dz, int32_t dy, int32_t dx, int32_t c, float* dvalue, float* dgrid, int32_t dtype,
                                     u3d_stream s) {
  U3D_REQUIRE(value && grid && dout && batch > 0 && nq > 0 && c > 0 && (dvalue || dgrid), U3D_ERR_ARG);
  dim3 g(u3d_cdiv((long long)batch * nq, 4));
  if (dtype == U3D_F32) hipLaunchKernelGGL(k_trilinear_bwd<float>, g, dim3(256), 0, s, (const float*)value, grid, (const float*)dout, batch, nq, dz, dy, dx, c, dvalue, dgrid);
  else if (dtype == U3D_BF16) hipLaunchKernelGGL(k_trilinear_bwd<unsigned short>, g, dim3(256), 0, s, (const unsigned short*)value, grid, (const unsigned short*)dout, batch, nq, dz, dy, dx, c, dvalue, dgrid);
  else return U3D_ERR_UNSUPPORTED;
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ============================================================================================
// Class-aware rotated BEV NMS (ref: models/dense_heads/uni3detr_head.py:849-865 -> upstream mmcv.ops.nms3d per class;
// boxes sorted by score, a box is suppressed by any kept higher-scored box of the same label whose
// rotated BEV IoU exceeds thr; height ignored).  Input must already be sorted by descending score.
//   pass 1: mask[i][w] bit j set iff j > i, label equal, iou_bev(i, j) > thr      (n x ceil(n/64) words)
//   pass 2: one wavefront sweeps i = 0..n-1 (kept rows OR their mask into `removed`), writes keep[i].
// ============================================================================================
__device__ float bev_iou_rot(const float* p, const float* q) {
  float a1 = p[3] * p[4], a2 = q[3] * q[4];
  if (a1 <= 0.f || a2 <= 0.f) return 0.f;
  P2 poly[12], tmp[12], clipper[4];
  rect_corners(0.f, 0.f, p[3], p[4], p[6], poly);
  rect_corners(q[0] - p[0], q[1] - p[1], q[3], q[4], q[6], clipper);
  int m = 4;
  for (int e = 0; e < 4 && m > 0; ++e) {
    m = clip_poly(poly, m, clipper[e], clipper[(e + 1) & 3], tmp);
    for (int t = 0; t < m; ++t) poly[t] = tmp[t];
  }
  float inter = 0.f;
  if (m >= 3) {
    for (int t = 0; t < m; ++t) {
      P2 u = poly[t], v = poly[(t + 1 == m) ? 0 : t + 1];
      inter += u.x * v.y - v.x * u.y;
    }
    inter = fabsf(inter) * 0.5f;
  }
  return inter / fmaxf(a1 + a2 - inter, 1e-8f);
}

__global__ void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ labels, int n, float thr,
                           unsigned long long* __restrict__ mask, int nw) {
  int i = blockIdx.x;
  int j = blockIdx.y * 64 + threadIdx.x;
  bool sup = false;
  if (j < n && j > i && labels[i] == labels[j]) sup = bev_iou_rot(boxes + (long long)i * 7, boxes + (long long)j * 7) > thr;
  unsigned long long b = __ballot(sup);
  if (threadIdx.x == 0) mask[(long long)i * nw + blockIdx.y] = b;
}

__global__ void k_nms_sweep(const unsigned long long* __restrict__ mask, int n, int nw, unsigned char* __restrict__ keep) {
  extern __shared__ unsigned long long removed[];
  for (int w = threadIdx.x; w < nw; w += 64) removed[w] = 0ull;
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    bool k = !((removed[i >> 6] >> (i & 63)) & 1ull);
    if (threadIdx.x == 0) keep[i] = k ? 1 : 0;
    if (k)
      for (int w = threadIdx.x; w < nw; w += 64) removed[w] |= mask[(long long)i * nw + w];
    __syncthreads();
  }
}

extern "C" int64_t u3d_nms3d_workspace(int32_t n) { return (int64_t)n * ((n + 63) / 64) * 8; }

extern "C" int32_t u3d_nms3d(const float* boxes, const int32_t* labels, int32_t n, float thr, uint8_t* keep, void* workspace,
                             int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(boxes && labels && keep && workspace, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  int nw = (n + 63) / 64;
  U3D_REQUIRE(workspace_bytes >= u3d_nms3d_workspace(n), U3D_ERR_WORKSPACE);
  U3D_REQUIRE((size_t)nw * 8 <= 64 * 1024, U3D_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(k_nms_mask, dim3(n, nw), dim3(64), 0, s, boxes, labels, n, thr, (unsigned long long*)workspace, nw);
  hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), (size_t)nw * 8, s, (const unsigned long long*)workspace, n, nw, keep);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
