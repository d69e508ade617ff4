// Flat AdamW with global-norm gradient clipping: the parameter-update end of the training step
// (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:234-235 — AdamW(lr, weight_decay=0.01), grad_clip max_norm=10, norm_type=2;
//  upstream torch.optim.AdamW + torch.nn.utils.clip_grad_norm_).  Parameters, gradients and both moments live in four flat f32
// buffers, so the whole update is three launches (partial sum of squares -> coefficients -> element-wise update) that stream
// 7 x 4 bytes per parameter once, instead of ~30 multi-tensor launches.  Capturable: the step counter is device state.
#include "common.h"

#define OPT_BLOCK 256
#define OPT_ELEMS_PER_BLOCK (OPT_BLOCK * 4 * 8)          // 8 float4 per thread

__global__ __launch_bounds__(OPT_BLOCK) void k_sumsq_partial(const float* __restrict__ g, long long n, double* __restrict__ partial) {
  const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    long long o = base + ((long long)i * OPT_BLOCK + threadIdx.x) * 4;
    if (o + 3 < n) {
      float4 v = *(const float4*)(g + o);
      acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    } else {
      for (int e = 0; e < 4; ++e)
        if (o + e < n) acc += (double)g[o + e] * (double)g[o + e];
    }
  }
  acc = u3d_wave_sum_d(acc);
  __shared__ double red[OPT_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < OPT_BLOCK / 64; ++i) s += red[i];
    partial[blockIdx.x] = s;
  }
}

// state (16 floats): [0] step count (as float, like torch's capturable AdamW), [1] clip coefficient, [2] 1 - beta1^t, [3] 1 - beta2^t,
//        [4] total gradient norm (for logging); hyper-parameters, read by the kernels at run time so that a captured graph follows a
//        learning-rate / momentum schedule: [5] lr, [6] beta1, [7] beta2, [8] eps, [9] weight decay, [10] max_norm (<= 0: no clipping);
//        [11] hold flag of THIS step (copied from the caller's device flag: > 0 = leave parameters, moments and the step count alone),
//        [12] number of held steps so far (read by the host every N steps: a capacity overflow somewhere in the job)
__global__ void k_adamw_set_hyper(float* __restrict__ state, float lr, float beta1, float beta2, float eps, float wd, float max_norm) {
  if (threadIdx.x == 0) { state[5] = lr; state[6] = beta1; state[7] = beta2; state[8] = eps; state[9] = wd; state[10] = max_norm; }
}
__global__ __launch_bounds__(256) void k_adamw_prepare(const double* __restrict__ partial, int nb, float* __restrict__ state,
                                                       const float* __restrict__ hold) {
  const float max_norm = state[10], beta1 = state[6], beta2 = state[7];
  const bool held = hold != nullptr && *hold > 0.f;          // uniform
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = u3d_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = sqrt(red[0] + red[1] + red[2] + red[3]);
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / ((float)tot + 1e-6f);              // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
      coef = coef < 1.f ? coef : 1.f;
    }
    state[11] = held ? 1.f : 0.f;
    state[4] = (float)tot;
    if (held) {
      state[12] += 1.f;
      return;
    }
    float t = state[0] + 1.f;
    state[0] = t;
    state[1] = coef;
    // bias corrections as running products (exact for constant betas; with a momentum schedule they follow the schedule the way
    // torch's per-step beta^t would only for constant betas - cyclic momentum changes beta1 by < 1e-4 per iteration)
    state[2] = 1.f - powf(beta1, t);
    state[3] = 1.f - powf(beta2, t);
    state[4] = (float)tot;
  }
}

// skip: optional uint8 per 64-element chunk (1 = leave parameter and moments untouched): parameters that received no gradient are
// skipped the way torch.optim.AdamW skips parameters whose .grad is None (no decay, no moment decay)
__global__ __launch_bounds__(OPT_BLOCK) void k_adamw_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long long n, const float* __restrict__ state,
                                                          const unsigned char* __restrict__ skip) {
  if (state[11] > 0.f) return;                            // held step (uniform): see k_adamw_prepare
  const float coef = state[1], bc1 = state[2], bc2 = state[3];
  const float lr = state[5], beta1 = state[6], beta2 = state[7], eps = state[8], wd = state[9];
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2), decay = 1.f - lr * wd;
  const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
#pragma unroll 2
  for (int i = 0; i < 8; ++i) {
    long long o = base + ((long long)i * OPT_BLOCK + threadIdx.x) * 4;
    if (o >= n) break;
    if (skip && skip[o >> 6]) continue;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = o + 3 < n;
    if (full) {
      float4 a = *(const float4*)(p + o), b = *(const float4*)(g + o), c = *(const float4*)(m + o), d = *(const float4*)(v + o);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int e = 0; e < 4; ++e) {
        bool ok = o + e < n;
        pv[e] = ok ? p[o + e] : 0.f; gv[e] = ok ? g[o + e] : 0.f; mv[e] = ok ? m[o + e] : 0.f; vv[e] = ok ? v[o + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gv[e] * coef;
      float pe = pv[e] * decay;                            // decoupled weight decay
      mv[e] = mv[e] + (gr - mv[e]) * (1.f - beta1);        // lerp form, as torch's fused kernel
      vv[e] = beta2 * vv[e] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] = pe - step_size * (mv[e] / denom);
    }
    if (full) {
      *(float4*)(p + o) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *(float4*)(m + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *(float4*)(v + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (int e = 0; e < 4; ++e)
        if (o + e < n) { p[o + e] = pv[e]; m[o + e] = mv[e]; v[o + e] = vv[e]; }
    }
  }
}

extern "C" int64_t u3d_adamw_workspace(int64_t n) { return (int64_t)u3d_cdiv(n > 0 ? n : 1, OPT_ELEMS_PER_BLOCK) * 8; }

extern "C" int32_t u3d_adamw_set_hyper(float* state, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                       u3d_stream s) {
  U3D_REQUIRE(state, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_adamw_set_hyper, dim3(1), dim3(64), 0, s, state, lr, beta1, beta2, eps, weight_decay, max_norm);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_adamw_step_hold(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                       const uint8_t* skip, const float* hold, void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && workspace && n >= 0, U3D_ERR_ARG);
  U3D_REQUIRE(workspace_bytes >= u3d_adamw_workspace(n), U3D_ERR_WORKSPACE);
  U3D_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, U3D_ERR_ARG);
  const int nb = u3d_cdiv(n > 0 ? n : 1, OPT_ELEMS_PER_BLOCK);
  hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(OPT_BLOCK), 0, s, grad, (long long)n, (double*)workspace);
  hipLaunchKernelGGL(k_adamw_prepare, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, state, hold);
  if (n > 0)
    hipLaunchKernelGGL(k_adamw_flat, dim3(nb), dim3(OPT_BLOCK), 0, s, param, grad, exp_avg, exp_avg_sq, (long long)n, (const float*)state, skip);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_adamw_step_state(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                        const uint8_t* skip, void* workspace, int64_t workspace_bytes, u3d_stream s) {
  return u3d_adamw_step_hold(param, grad, exp_avg, exp_avg_sq, n, state, skip, nullptr, workspace, workspace_bytes, s);
}

// flag[0] = number of sparse levels whose device-side row count exceeds its capacity.  counts: HOST array of n (<= 8) device pointers
// to int32 counts, caps: HOST int32 [n] (both travel as kernel arguments: a captured launch keeps them).  The trainer all-reduces
// the flag together with the positive counts and hands it to u3d_adamw_step_hold.
struct CapArgs { const int* cnt[8]; int cap[8]; int n; };
__global__ void k_capacity_flag(CapArgs a, float* __restrict__ flag) {
  if (threadIdx.x == 0) {
    int over = 0;
    for (int i = 0; i < a.n; ++i) over += *a.cnt[i] > a.cap[i] ? 1 : 0;
    *flag = (float)over;
  }
}
extern "C" int32_t u3d_capacity_flag(const int32_t* const* counts, const int32_t* caps, int32_t n, float* flag, u3d_stream s) {
  U3D_REQUIRE(flag && n >= 0 && n <= 8 && (n == 0 || (counts && caps)), U3D_ERR_ARG);
  CapArgs a;
  a.n = n;
  for (int i = 0; i < 8; ++i) { a.cnt[i] = i < n ? (const int*)counts[i] : nullptr; a.cap[i] = i < n ? caps[i] : 0; }
  for (int i = 0; i < n; ++i) U3D_REQUIRE(a.cnt[i], U3D_ERR_ARG);
  hipLaunchKernelGGL(k_capacity_flag, dim3(1), dim3(64), 0, s, a, flag);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// host-scalar form: writes the hyper-parameters into the state vector, then steps (a captured graph of THIS entry bakes them in;
// schedules use u3d_adamw_set_hyper between replays + u3d_adamw_step_state inside the graph)
extern "C" int32_t u3d_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, float max_norm, float* state, void* workspace,
                                  int64_t workspace_bytes, u3d_stream s) {
  int32_t rc = u3d_adamw_set_hyper(state, lr, beta1, beta2, eps, weight_decay, max_norm, s);
  if (rc != U3D_OK) return rc;
  return u3d_adamw_step_state(param, grad, exp_avg, exp_avg_sq, n, state, nullptr, workspace, workspace_bytes, s);
}


// ---------------------------------------------------------------------------------------------
// bf16 shadows of the flat f32 parameter buffer (uni3detr_amd/shadow.py): one cast launch for every parameter in its own layout
// (linears, dhwio conv weights as [K,Cin,Cout]) + one batched re-layout launch for the conv-weight layouts the GEMM kernels read
// row-linearly (koi [K,Cout,Cin]; both layouts of nn.Conv3d's [Cout,Cin,K]).  Replaces ~140 per-tensor cast / strided-copy launches.
// ---------------------------------------------------------------------------------------------
typedef unsigned short u16;
#define SH_ELEMS_PER_BLOCK 2048                                  // 256 threads x 8 bf16 = one 16-byte store per thread

__global__ __launch_bounds__(256) void k_cast_bf16_flat(const float* __restrict__ src, u16* __restrict__ dst, long long n) {
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  for (long long o = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; o < n; o += (long long)gridDim.x * 2048) {
    if (o + 7 < n) {
      float4 a = *(const float4*)(src + o), b = *(const float4*)(src + o + 4);
      f32x8_t f = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      *(bf16x8_t*)(dst + o) = __builtin_convertvector(f, bf16x8_t);
    } else {
      for (int e = 0; e < 8 && o + e < n; ++e) { __bf16 h = (__bf16)src[o + e]; dst[o + e] = *(u16*)&h; }
    }
  }
}

__global__ __launch_bounds__(256) void k_permute_bf16(const u16* __restrict__ src, u16* __restrict__ dst,
                                                      const u3d_permute_desc* __restrict__ descs, const int2* __restrict__ blocks) {
  const int2 b = blocks[blockIdx.x];
  const u3d_permute_desc d = descs[b.x];
  const int e0 = b.y + threadIdx.x * 8;
  if (e0 >= d.n) return;
  const int rc = d.rows * d.cols;
  u16 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int e = e0 + j;
    const int k = e / rc, rem = e - k * rc, r = rem / d.cols, c = rem - r * d.cols;
    v[j] = e < d.n ? src[d.src_off + k * d.stride_k + r * d.stride_r + c * d.stride_c] : (u16)0;
  }
  u16* out = dst + d.dst_off + e0;
  if (e0 + 7 < d.n) {
    uint4 w = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
               (unsigned)v[6] | ((unsigned)v[7] << 16)};
    *(uint4*)out = w;
  } else {
    for (int j = 0; j < 8 && e0 + j < d.n; ++j) out[j] = v[j];
  }
}

// The same re-layout for descriptors whose SOURCE is contiguous along k (stride_k == 1: nn.Conv3d's [Cout][Cin][K], 50 of the 55 M
// shadow elements of the SUN RGB-D model): k_permute_bf16 gathers them 2 bytes at a time, 54 bytes apart.  Here a workgroup takes a
// 32 x 32 (row, column) tile with all K offsets: for each index of the slow source dimension that is ONE contiguous run of 32 * K
// elements - read coalesced into LDS as [outer][inner][k], written back as 16-byte pieces of dst[k][row][col ...].
#define PT_TILE 32
__global__ __launch_bounds__(256) void k_permute_k1_tiled(const u16* __restrict__ src, u16* __restrict__ dst,
                                                          const u3d_permute_desc* __restrict__ descs, const int4* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) u16 pt_lds[];
  const int4 t = tiles[blockIdx.x];
  const u3d_permute_desc d = descs[t.x];
  const int r0 = t.y, c0 = t.z, rc = d.rows * d.cols, K = d.n / rc;
  const bool c_inner = d.stride_c == K;                    // the column index is the fast source dimension (else the row index)
  const long long so = c_inner ? d.stride_r : d.stride_c;  // stride of the slow one
  const int outer0 = c_inner ? r0 : c0, inner0 = c_inner ? c0 : r0;
  const int run = PT_TILE * K, run2 = run >> 1;            // elements / dwords of one contiguous run
  const u16* base = src + d.src_off + (long long)outer0 * so + (long long)inner0 * K;
  unsigned* l32 = (unsigned*)pt_lds;
  for (int i = threadIdx.x; i < PT_TILE * run2; i += 256) {
    const int o = i / run2, j = i - o * run2;
    l32[i] = *(const unsigned*)(base + (long long)o * so + 2 * j);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * PT_TILE * (PT_TILE / 8); i += 256) {
    const int c8 = i & 3, r = (i >> 2) & (PT_TILE - 1), k = i >> 7;
    u16 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 * 8 + j;
      v[j] = pt_lds[(c_inner ? r : c) * run + (c_inner ? c : r) * K + k];
    }
    const uint4 w = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                     (unsigned)v[6] | ((unsigned)v[7] << 16)};
    *(uint4*)(dst + d.dst_off + (long long)k * rc + (long long)(r0 + r) * d.cols + c0 + c8 * 8) = w;
  }
}
// tiles_dev: int32 [ntiles][4] = (descriptor, first row, first column, 0); descriptors must have stride_k == 1, rows % 32 == 0,
// cols % 32 == 0, one of stride_r / stride_c == K (= n / (rows * cols)) <= max_k <= 32, even src_off, dst_off % 8 == 0
extern "C" int32_t u3d_permute_bf16_tiled(const void* src, void* dst, const u3d_permute_desc* descs_dev, const int32_t* tiles_dev,
                                          int32_t ntiles, int32_t max_k, u3d_stream s) {
  U3D_REQUIRE(src && dst && descs_dev && tiles_dev && ntiles >= 0 && max_k >= 1 && max_k <= 32 && ((uintptr_t)dst & 15) == 0
              && ((uintptr_t)src & 3) == 0, U3D_ERR_ARG);
  if (ntiles == 0) return U3D_OK;
  hipLaunchKernelGGL(k_permute_k1_tiled, dim3(ntiles), dim3(256), (size_t)PT_TILE * PT_TILE * max_k * 2, s, (const u16*)src, (u16*)dst,
                     descs_dev, (const int4*)tiles_dev);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_cast_bf16(const float* src, void* dst, int64_t n, u3d_stream s) {
  U3D_REQUIRE(src && dst && n >= 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  long long nb = (n + SH_ELEMS_PER_BLOCK - 1) / SH_ELEMS_PER_BLOCK;
  hipLaunchKernelGGL(k_cast_bf16_flat, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, s, src, (u16*)dst, (long long)n);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_permute_block_elems(void) { return SH_ELEMS_PER_BLOCK; }

extern "C" int32_t u3d_permute_bf16_batched(const void* src, void* dst, const u3d_permute_desc* descs_dev, const int32_t* blocks_dev,
                                            int32_t nblocks, u3d_stream s) {
  U3D_REQUIRE(src && dst && descs_dev && blocks_dev && nblocks >= 0 && ((uintptr_t)dst & 15) == 0, U3D_ERR_ARG);
  if (nblocks == 0) return U3D_OK;
  hipLaunchKernelGGL(k_permute_bf16, dim3(nblocks), dim3(256), 0, s, (const u16*)src, (u16*)dst, descs_dev, (const int2*)blocks_dev);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
