// Detection losses of Uni3DETRHead for all decoder layers in ONE forward and ONE backward launch
// (ref: projects/mmdet3d_plugin/models/dense_heads/uni3detr_head.py:579-720 loss_single / loss; models/losses/rdiouloss.py:93-223
//  SoftFocalLoss / IoU3DLoss; upstream mmdet L1Loss; core/bbox/util.py normalize_bbox / denormalize_bbox).
//
// The torch formulation is ~150 element-wise launches on [L,B,Q] tensors forward and ~250 backward, each 3-5 us of pure launch
// cost.  Here one thread owns one (layer, scene, query) element: it rebuilds the box from its code, evaluates the axis-snapped BEV
// IoU and the z IoU with FORWARD-MODE derivatives w.r.t. the six box parameters that reach them (cx, cy, dx, dy, cz, dz), and forms
//   loss_cls  = sum_c BCE(z_c, soft_c) * fw_c / (cls_avg + eps) * w_cls     soft_c = [c == label] * quality,  quality = (iou_bev + iou_z)/2
//   loss_bbox = sum_d |code_d - target_code_d| * bw_d / (n_pos + eps) * w_bbox                    (quality is NOT detached: D-6)
//   loss_iou  = (1 - iou_bev) * mean(bw) / (n_pos + eps) * w_iou + (1 - iou_z) * bw_0 / n_pos
//   loss_ioup = BCE(iou_logit, iou_true) * bw_0 / n_pos * 1.2
// exactly as Uni3DETRHead.loss_from_targets writes them.  Sub-gradient conventions are torch's (abs'(0) = 0, clamp passes at the
// bound, min/max take the first operand on ties - ties have measure zero here).
#include "common.h"

#define DL_NV 6          // derivative slots: cx, cy, dx, dy, cz, dz
struct Dual {
  float v;
  float d[DL_NV];
};
__device__ __forceinline__ Dual dl_const(float v) { Dual r; r.v = v; for (int i = 0; i < DL_NV; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ Dual dl_var(float v, int k) { Dual r = dl_const(v); r.d[k] = 1.f; return r; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < DL_NV; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < DL_NV; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < DL_NV; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, float s) { Dual r; r.v = a.v * s; for (int i = 0; i < DL_NV; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  Dual r; const float inv = 1.f / b.v; r.v = a.v * inv;
  for (int i = 0; i < DL_NV; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ Dual dl_min(const Dual& a, float b) { return a.v <= b ? a : dl_const(b); }
__device__ __forceinline__ Dual dl_max(const Dual& a, float b) { return a.v >= b ? a : dl_const(b); }
__device__ __forceinline__ Dual dl_clamp_min(const Dual& a, float lo) { return a.v >= lo ? a : dl_const(lo); }

struct DetLossCfg {
  int L, M, C, code, tdim;          // M = B*Q elements per layer; code = 8 or 10; tdim = 7 or 9
  float alpha, w_cls, w_box, w_iou, eps32;
};

// per element: the four un-reduced loss contributions (already scaled by 1/avg factors and loss weights) and, when GRAD, the
// gradients w.r.t. the class logits, the box code and the IoU logit, each scaled by the incoming gradient of its loss scalar
template <bool GRAD>
__device__ __forceinline__ void det_loss_elem(const DetLossCfg& cf, int l, long long e, const float* __restrict__ cls,
                                              const float* __restrict__ box, const float* __restrict__ iou_logit,
                                              const float* __restrict__ tgt, const long long* __restrict__ lab,
                                              const float* __restrict__ w, const float* __restrict__ iou_true,
                                              const float* __restrict__ cls_avg, const float* __restrict__ npos,
                                              const float* __restrict__ code_w, const float* __restrict__ gout, float* terms,
                                              float* __restrict__ dcls, float* __restrict__ dbox, float* __restrict__ diou) {
  const float* p = box + e * cf.code;
  const float* t = tgt + e * cf.tdim;
  const float we = w[e];
  const float inv_np_eps = 1.f / (npos[l] + cf.eps32), inv_np = 1.f / npos[l], inv_cls = 1.f / (cls_avg[l] + cf.eps32);
  // ---- box -> geometry (denormalize_bbox): sizes exp(code), yaw = -atan2(sin, cos) - pi/2 (yaw only selects the snap)
  const float PI = 3.14159265358979323846f;
  Dual cx = dl_var(p[0], 0), cy = dl_var(p[1], 1), cz = dl_var(p[4], 4);
  Dual dx = dl_var(__expf(p[3]), 2), dy = dl_var(__expf(p[2]), 3), dz = dl_var(__expf(p[5]), 5);
  dx.v = expf(p[3]); dy.v = expf(p[2]); dz.v = expf(p[5]);
  const float yaw = -atan2f(p[6], p[7]) - PI * 0.5f;
  const float ra = fabsf(yaw - floorf(yaw / PI + 0.5f) * PI);
  const bool swap_a = ra > PI * 0.25f;
  const Dual wa = swap_a ? dy : dx, ha = swap_a ? dx : dy;
  const float tyaw = t[6];
  const float rb = fabsf(tyaw - floorf(tyaw / PI + 0.5f) * PI);
  const bool swap_b = rb > PI * 0.25f;
  const float wb = swap_b ? t[4] : t[3], hb = swap_b ? t[3] : t[4];
  const float bx1 = t[0] - wb * 0.5f, by1 = t[1] - hb * 0.5f, bx2 = t[0] + wb * 0.5f, by2 = t[1] + hb * 0.5f;
  const Dual ax1 = cx - wa * 0.5f, ay1 = cy - ha * 0.5f, ax2 = cx + wa * 0.5f, ay2 = cy + ha * 0.5f;
  const Dual iw = dl_clamp_min(dl_min(ax2, bx2) - dl_max(ax1, bx1), 0.f), ih = dl_clamp_min(dl_min(ay2, by2) - dl_max(ay1, by1), 0.f);
  const Dual ov = iw * ih;
  const Dual uni = dl_clamp_min((ax2 - ax1) * (ay2 - ay1) + dl_const((bx2 - bx1) * (by2 - by1)) - ov, 1e-6f);
  const Dual iou_bev = ov / uni;
  const Dual z1 = cz - dz * 0.5f, z2 = cz + dz * 0.5f;
  const float z3 = t[2] - t[5] * 0.5f, z4 = t[2] + t[5] * 0.5f;
  const Dual iou_z = dl_clamp_min(dl_min(z2, z4) - dl_max(z1, z3), 0.f) / (dl_max(z2, z4) - dl_min(z1, z3));
  const Dual quality = (iou_bev + iou_z) * 0.5f;
  // ---- weights
  float cw_mean = 0.f;
  for (int d = 0; d < cf.code; ++d) cw_mean += code_w[d];
  cw_mean /= (float)cf.code;
  const float bw0 = we * code_w[0];
  const float c_iou_bev = we * cw_mean * inv_np_eps * cf.w_iou, c_iou_z = bw0 * inv_np;
  const float g_cls = GRAD ? gout[l * 4 + 0] : 0.f, g_box = GRAD ? gout[l * 4 + 1] : 0.f, g_iou = GRAD ? gout[l * 4 + 2] : 0.f,
              g_ioup = GRAD ? gout[l * 4 + 3] : 0.f;
  // ---- classification (quality focal): soft target only on the label's class
  const long long lb = lab[e];
  const float a = cf.alpha;
  float lcls = 0.f, dq = 0.f;                           // dq = d loss_cls / d quality
  for (int c = 0; c < cf.C; ++c) {
    const float z = cls[e * cf.C + c];
    const float ps = 1.f / (1.f + expf(-z));
    const float s = (c == lb) ? quality.v : 0.f;
    const float bce = fmaxf(z, 0.f) - z * s + log1pf(expf(-fabsf(z)));
    const float pt = s - ps;
    const float af = (1.f - a) + (2.f * a - 1.f) * s;
    const float fw = af * pt * pt;
    lcls += bce * fw;
    if (GRAD) {
      const float dfw_dz = af * 2.f * pt * (-ps * (1.f - ps));
      dcls[e * cf.C + c] = ((ps - s) * fw + bce * dfw_dz) * inv_cls * cf.w_cls * g_cls;
      if (c == lb) dq = (-z * fw + bce * ((2.f * a - 1.f) * pt * pt + af * 2.f * pt)) * inv_cls * cf.w_cls;
    }
  }
  terms[0] = lcls * inv_cls * cf.w_cls;
  // ---- box L1 on the codes: target code = normalize_bbox(target)
  float nt[10];
  nt[0] = t[0]; nt[1] = t[1]; nt[2] = logf(t[4] + 1e-5f); nt[3] = logf(t[3] + 1e-5f); nt[4] = t[2]; nt[5] = logf(t[5] + 1e-5f);
  const float rot_t = -t[6] - PI * 0.5f;
  nt[6] = sinf(rot_t); nt[7] = cosf(rot_t);
  if (cf.code > 8) { nt[8] = t[7]; nt[9] = t[8]; }
  float lbox = 0.f;
  float dp[10];
  for (int d = 0; d < cf.code; ++d) {
    const float diff = p[d] - nt[d];
    const float bwd = we * code_w[d] * inv_np_eps * cf.w_box;
    lbox += fabsf(diff) * bwd;
    dp[d] = GRAD ? ((diff > 0.f) - (diff < 0.f)) * bwd * g_box : 0.f;
  }
  terms[1] = lbox;
  terms[2] = (1.f - iou_bev.v) * c_iou_bev + (1.f - iou_z.v) * c_iou_z;
  // ---- IoU prediction branch
  const float s_l = iou_logit[e], it = iou_true[e];
  const float bce_i = fmaxf(s_l, 0.f) - s_l * it + log1pf(expf(-fabsf(s_l)));
  const float c_ioup = bw0 * inv_np * 1.2f;
  terms[3] = bce_i * c_ioup;
  if (GRAD) {
    diou[e] = (1.f / (1.f + expf(-s_l)) - it) * c_ioup * g_ioup;
    // chain the geometric terms back onto the code: slots (cx, cy, dx, dy, cz, dz) -> code (0, 1, 3, 2, 4, 5); sizes are exp(code)
    float gq[DL_NV];
    for (int k = 0; k < DL_NV; ++k)
      gq[k] = -iou_bev.d[k] * c_iou_bev * g_iou - iou_z.d[k] * c_iou_z * g_iou + quality.d[k] * dq * g_cls;
    dp[0] += gq[0]; dp[1] += gq[1]; dp[3] += gq[2] * dx.v; dp[2] += gq[3] * dy.v; dp[4] += gq[4]; dp[5] += gq[5] * dz.v;
    for (int d = 0; d < cf.code; ++d) dbox[e * cf.code + d] = dp[d];
  }
}

__global__ __launch_bounds__(256) void k_det_loss_fwd(DetLossCfg cf, const float* cls, const float* box, const float* iou_logit,
                                                      const float* tgt, const long long* lab, const float* w, const float* iou_true,
                                                      const float* cls_avg, const float* npos, const float* code_w,
                                                      float* __restrict__ partial) {
  // grid (blocks per layer, L); partial [L][gridDim.x][4]
  const int l = blockIdx.y;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cf.M; i += gridDim.x * 256) {
    float terms[4];
    det_loss_elem<false>(cf, l, (long long)l * cf.M + i, cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, nullptr,
                         terms, nullptr, nullptr, nullptr);
    for (int k = 0; k < 4; ++k) acc[k] += terms[k];
  }
  __shared__ float red[4][4];
  for (int k = 0; k < 4; ++k) {
    float s = u3d_wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) partial[((long long)l * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
      red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ void k_det_loss_reduce(const float* __restrict__ partial, int nb, int L, float* __restrict__ out) {
  // one wave per (layer, loss type): out [L][4]
  const int i = blockIdx.x, l = i / 4, k = i % 4, lane = threadIdx.x;
  float s = 0.f;
  for (int b = lane; b < nb; b += 64) s += partial[((long long)l * nb + b) * 4 + k];
  s = u3d_wave_sum(s);
  if (lane == 0) out[i] = s;
}
__global__ __launch_bounds__(256) void k_det_loss_bwd(DetLossCfg cf, const float* cls, const float* box, const float* iou_logit,
                                                      const float* tgt, const long long* lab, const float* w, const float* iou_true,
                                                      const float* cls_avg, const float* npos, const float* code_w, const float* gout,
                                                      float* dcls, float* dbox, float* diou) {
  const int l = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cf.M; i += gridDim.x * 256) {
    float terms[4];
    det_loss_elem<true>(cf, l, (long long)l * cf.M + i, cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, gout, terms,
                        dcls, dbox, diou);
  }
}

static inline int det_loss_blocks(int m) { int b = u3d_cdiv(m > 0 ? m : 1, 256); return b > 64 ? 64 : b; }

extern "C" int64_t u3d_det_loss_workspace(int32_t L, int32_t m) { return (int64_t)L * det_loss_blocks(m) * 4 * 4; }

extern "C" int32_t u3d_det_loss_fwd(const float* cls, const float* box, const float* iou_logit, const float* tgt, const int64_t* lab,
                                    const float* w, const float* iou_true, const float* cls_avg, const float* npos,
                                    const float* code_w, int32_t L, int32_t m, int32_t c, int32_t code, int32_t tdim, float alpha,
                                    float w_cls, float w_box, float w_iou, float eps, float* out, void* workspace,
                                    int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(cls && box && iou_logit && tgt && lab && w && iou_true && cls_avg && npos && code_w && out && workspace, U3D_ERR_ARG);
  U3D_REQUIRE(L > 0 && m > 0 && c > 0 && (code == 8 || code == 10) && tdim == code - 1, U3D_ERR_UNSUPPORTED);
  U3D_REQUIRE(workspace_bytes >= u3d_det_loss_workspace(L, m), U3D_ERR_WORKSPACE);
  DetLossCfg cf = {L, m, c, code, tdim, alpha, w_cls, w_box, w_iou, eps};
  const int nb = det_loss_blocks(m);
  hipLaunchKernelGGL(k_det_loss_fwd, dim3(nb, L), dim3(256), 0, s, cf, cls, box, iou_logit, tgt, (const long long*)lab, w, iou_true, cls_avg,
                     npos, code_w, (float*)workspace);
  hipLaunchKernelGGL(k_det_loss_reduce, dim3(L * 4), dim3(64), 0, s, (const float*)workspace, nb, L, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_det_loss_bwd(const float* cls, const float* box, const float* iou_logit, const float* tgt, const int64_t* lab,
                                    const float* w, const float* iou_true, const float* cls_avg, const float* npos,
                                    const float* code_w, const float* gout, int32_t L, int32_t m, int32_t c, int32_t code, int32_t tdim,
                                    float alpha, float w_cls, float w_box, float w_iou, float eps, float* dcls, float* dbox,
                                    float* diou, u3d_stream s) {
  U3D_REQUIRE(cls && box && iou_logit && tgt && lab && w && iou_true && cls_avg && npos && code_w && gout && dcls && dbox && diou, U3D_ERR_ARG);
  U3D_REQUIRE(L > 0 && m > 0 && c > 0 && (code == 8 || code == 10) && tdim == code - 1, U3D_ERR_UNSUPPORTED);
  DetLossCfg cf = {L, m, c, code, tdim, alpha, w_cls, w_box, w_iou, eps};
  hipLaunchKernelGGL(k_det_loss_bwd, dim3(u3d_cdiv(m, 256), L), dim3(256), 0, s, cf, cls, box, iou_logit, tgt, (const long long*)lab, w,
                     iou_true, cls_avg, npos, code_w, gout, dcls, dbox, diou);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// Target construction behind the assignment (ref: dense_heads/uni3detr_head.py:510-570 _get_target_single / get_targets), all layers
// and scenes in one launch: asg int32 [L,B,Q] (0 = background, else 1-based GT of the scene) ->
//   asg64 (the same as int64), w f32 [L,B,Q] (1 on matched queries), tgt f32 [L,B,Q,gd] (the matched GT row, zeros for background),
//   lab int64 [L,B,Q] (the GT's label, num_classes for background), num_pos f32 [L] (matched queries per layer: exact integer sums).
// One workgroup per layer.
__global__ __launch_bounds__(1024) void k_loss_targets(const int* __restrict__ asg, const float* __restrict__ gt, const int* __restrict__ labels,
                                                       const int* __restrict__ gt_off, int B, int Q, int gd, int ncls,
                                                       long long* __restrict__ asg64, float* __restrict__ w, float* __restrict__ tgt,
                                                       long long* __restrict__ lab, float* __restrict__ num_pos) {
  __shared__ int s_cnt[16];
  const int l = blockIdx.x, tid = threadIdx.x;
  const long long base = (long long)l * B * Q;
  int cnt = 0;
  for (int i = tid; i < B * Q; i += 1024) {
    const int b = i / Q, a = asg[base + i];
    const bool pos = a > 0;
    asg64[base + i] = a;
    w[base + i] = pos ? 1.f : 0.f;
    const int gi = gt_off[b] + (pos ? a - 1 : 0);
    lab[base + i] = pos ? (long long)labels[gi] : (long long)ncls;
    float* t = tgt + (base + i) * gd;
    for (int c = 0; c < gd; ++c) t[c] = pos ? gt[(long long)gi * gd + c] : 0.f;
    cnt += pos;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int k = 0; k < 16; ++k) a += s_cnt[k];
    num_pos[l] = (float)a;
  }
}
extern "C" int32_t u3d_loss_targets(const int32_t* asg, const float* gt, const int32_t* labels, const int32_t* gt_off, int32_t L, int32_t B,
                                    int32_t Q, int32_t gd, int32_t ncls, int64_t* asg64, float* w, float* tgt, int64_t* lab, float* num_pos,
                                    u3d_stream s) {
  U3D_REQUIRE(asg && gt && labels && gt_off && asg64 && w && tgt && lab && num_pos && L > 0 && B > 0 && Q > 0 && gd > 0, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_loss_targets, dim3(L), dim3(1024), 0, (hipStream_t)s, asg, gt, labels, gt_off, B, Q, gd, ncls, (long long*)asg64, w, tgt,
                     (long long*)lab, num_pos);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// codes [n, code] -> boxes [n, 7] (cx, cy, cz, dx, dy, dz, yaw): denormalize_bbox without the velocity columns (what the rotated
// IoU target of the IoU-prediction branch is evaluated on)
__global__ void k_denorm_boxes(const float* __restrict__ codes, int n, int code, float* __restrict__ boxes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = codes + (long long)i * code;
  float* b = boxes + (long long)i * 7;
  b[0] = p[0]; b[1] = p[1]; b[2] = p[4];
  b[3] = expf(p[3]); b[4] = expf(p[2]); b[5] = expf(p[5]);
  b[6] = -atan2f(p[6], p[7]) - 3.14159265358979323846f * 0.5f;
}
extern "C" int32_t u3d_denormalize_boxes(const float* codes, int32_t n, int32_t code, float* boxes, u3d_stream s) {
  U3D_REQUIRE(codes && boxes && n >= 0 && code >= 8, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_denorm_boxes, dim3(u3d_cdiv(n, 256)), dim3(256), 0, s, codes, n, code, boxes);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}


// ---------------------------------------------------------------------------------------------
// Box decode of the detection head (ref: dense_heads/uni3detr_head.py:475-490): the regression branch's raw code + the layer's
// reference point -> the normalised box code.  reference = inverse_sigmoid(ref) (ref in sigmoid space, eps 1e-5 clamps as
// mmdet's inverse_sigmoid); columns 0,1 (+ref x,y) and 4 (+ref z) go through a sigmoid and the point-cloud
// range; every other column passes through.  One thread per query row; the backward recomputes the sigmoids.
// In torch this is ~20 launches forward and ~25 backward per decoder layer.
// ---------------------------------------------------------------------------------------------
struct BoxRange { float lo[3], span[3]; };

__device__ __forceinline__ float bd_load(const void* p, long long i, int bf16) {
  return bf16 ? __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16) : ((const float*)p)[i];
}
__device__ __forceinline__ float bd_inv_sigmoid(float r, float eps) {
  r = fminf(fmaxf(r, 0.f), 1.f);
  return logf(fmaxf(r, eps) / fmaxf(1.f - r, eps));
}

__global__ void k_box_decode_fwd(const void* __restrict__ tmp, int bf16, const float* __restrict__ ref, int n, int code, BoxRange pr,
                                 float eps, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int col[3] = {0, 1, 4};
  float* o = out + (long long)i * code;
  for (int c = 0; c < code; ++c) o[c] = bd_load(tmp, (long long)i * code + c, bf16);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float v = o[col[j]] + bd_inv_sigmoid(ref[(long long)i * 3 + j], eps);
    o[col[j]] = (1.f / (1.f + expf(-v))) * pr.span[j] + pr.lo[j];
  }
}

// dtmp (tmp's dtype) and, when dref != NULL, the gradient w.r.t. the sigmoid-space reference point
__global__ void k_box_decode_bwd(const void* __restrict__ tmp, int bf16, const float* __restrict__ ref, const float* __restrict__ dout,
                                 int n, int code, BoxRange pr, float eps, void* __restrict__ dtmp, float* __restrict__ dref) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int col[3] = {0, 1, 4};
  float g[16];
  for (int c = 0; c < code; ++c) g[c] = dout[(long long)i * code + c];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float r = ref[(long long)i * 3 + j];
    const bool inside = r >= 0.f && r <= 1.f;                           // clamp(0,1) passes the gradient on [0,1] (torch semantics)
    r = fminf(fmaxf(r, 0.f), 1.f);
    const float a = fmaxf(r, eps), b = fmaxf(1.f - r, eps);
    const float v = bd_load(tmp, (long long)i * code + col[j], bf16) + logf(a / b);
    const float sg = 1.f / (1.f + expf(-v));
    const float gv = g[col[j]] * pr.span[j] * sg * (1.f - sg);
    g[col[j]] = gv;
    if (dref) {
      float d = 0.f;
      if (r >= eps) d += 1.f / a;                                       // clamp(min=eps) passes the gradient for x >= eps
      if (1.f - r >= eps) d += 1.f / b;
      dref[(long long)i * 3 + j] = inside ? gv * d : 0.f;
    }
  }
  if (bf16) {
    for (int c = 0; c < code; ++c) { __bf16 h = (__bf16)g[c]; ((unsigned short*)dtmp)[(long long)i * code + c] = *(unsigned short*)&h; }
  } else {
    for (int c = 0; c < code; ++c) ((float*)dtmp)[(long long)i * code + c] = g[c];
  }
}

static inline BoxRange box_range(const float* pc_range) {
  BoxRange r;
  for (int j = 0; j < 3; ++j) { r.lo[j] = pc_range[j]; r.span[j] = pc_range[3 + j] - pc_range[j]; }
  return r;
}

extern "C" int32_t u3d_box_decode_fwd(const void* tmp, int32_t dtype, const float* ref, int32_t n, int32_t code, const float* pc_range,
                                      float eps, float* out, u3d_stream s) {
  U3D_REQUIRE(tmp && ref && out && pc_range && n >= 0 && code >= 5 && code <= 16, U3D_ERR_ARG);
  U3D_REQUIRE(dtype == U3D_F32 || dtype == U3D_BF16, U3D_ERR_UNSUPPORTED);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_box_decode_fwd, dim3(u3d_cdiv(n, 128)), dim3(128), 0, s, tmp, dtype == U3D_BF16, ref, n, code, box_range(pc_range), eps, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// The per-layer tail of the decoder loop and the head's box decode in ONE launch (ref: models/utils/uni3detr_transformer.py:194-202
// reference-point refinement, dense_heads/uni3detr_head.py:463-490 box decode): from the regression branch's raw code `tmp` [n, code]
// (f32), the layer's INPUT reference point as logits `ref_in` [n, 3] and in sigmoid space `ref_s` [n, 3]:
//   out      [n, code] = the box decode of k_box_decode_fwd (same arithmetic, same operands)
//   ref_out  [n, 3]    = ref_in + tmp[:, (0, 1, 4)]        (the next layer's reference logits, detached in the reference)
//   ref_sig  [n, 3]    = sigmoid(ref_out)                  (inter_references of the transformer's return value)
// In torch: index_select + add per layer, a stack and a sigmoid behind the loop, and the decode launch.  Backward = k_box_decode_bwd.
__global__ void k_refine_decode_fwd(const float* __restrict__ tmp, const float* __restrict__ ref_in, const float* __restrict__ ref_s, int n,
                                    int code, BoxRange pr, float eps, float* __restrict__ out, float* __restrict__ ref_out,
                                    float* __restrict__ ref_sig) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int col[3] = {0, 1, 4};
  float* o = out + (long long)i * code;
  for (int c = 0; c < code; ++c) o[c] = tmp[(long long)i * code + c];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float t = o[col[j]];
    const float r = __fadd_rn(ref_in[(long long)i * 3 + j], t);
    ref_out[(long long)i * 3 + j] = r;
    ref_sig[(long long)i * 3 + j] = 1.f / (1.f + expf(-r));
    const float v = t + bd_inv_sigmoid(ref_s[(long long)i * 3 + j], eps);
    o[col[j]] = (1.f / (1.f + expf(-v))) * pr.span[j] + pr.lo[j];
  }
}
extern "C" int32_t u3d_refine_decode_fwd(const float* tmp, const float* ref_in, const float* ref_s, int32_t n, int32_t code,
                                         const float* pc_range, float eps, float* out, float* ref_out, float* ref_sig, u3d_stream s) {
  U3D_REQUIRE(tmp && ref_in && ref_s && out && ref_out && ref_sig && pc_range && n >= 0 && code >= 5 && code <= 16, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_refine_decode_fwd, dim3(u3d_cdiv(n, 128)), dim3(128), 0, (hipStream_t)s, tmp, ref_in, ref_s, n, code, box_range(pc_range), eps, out,
                     ref_out, ref_sig);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_box_decode_bwd(const void* tmp, int32_t dtype, const float* ref, const float* dout, int32_t n, int32_t code,
                                      const float* pc_range, float eps, void* dtmp, float* dref, u3d_stream s) {
  U3D_REQUIRE(tmp && ref && dout && dtmp && pc_range && n >= 0 && code >= 5 && code <= 16, U3D_ERR_ARG);
  U3D_REQUIRE(dtype == U3D_F32 || dtype == U3D_BF16, U3D_ERR_UNSUPPORTED);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_box_decode_bwd, dim3(u3d_cdiv(n, 128)), dim3(128), 0, s, tmp, dtype == U3D_BF16, ref, dout, n, code, box_range(pc_range), eps, dtmp, dref);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}


// ---------------------------------------------------------------------------------------------
// Sine position embedding of the decoder's reference points (ref: models/utils/uni3detr_transformer.py:33-65, called on
// reference_points.sigmoid() at :181): out[n][j*F + f] = (f even ? sin : cos)(sigmoid(logit[n][j]) * 2*pi / dim_t[f]),
// dim_t[f] = T^(2*(f/2)/F) (table computed on the host exactly as upstream does).  One launch instead of sigmoid + ~8 element-wise
// kernels; the backward (layer 0's reference points depend on learned anchors) folds the F features back onto each coordinate.
// ---------------------------------------------------------------------------------------------
__global__ void k_sine_embed_fwd(const float* __restrict__ logits, const float* __restrict__ dim_t, int n, int nc, int F, int bf16,
                                 void* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)n * nc * F;
  if (i >= total) return;
  const int f = (int)(i % F);
  const long long rj = i / F;                                         // row * nc + coordinate
  const float pos = 1.f / (1.f + expf(-logits[rj]));
  const float sarg = pos * 6.283185307179586f / dim_t[f];
  const float v = (f & 1) ? cosf(sarg) : sinf(sarg);
  if (bf16) { __bf16 h = (__bf16)v; ((unsigned short*)out)[i] = *(unsigned short*)&h; }
  else ((float*)out)[i] = v;
}

// one wave per (row, coordinate): dlogit = sigmoid' * sum_f dout * d(sin|cos)/dpos
__global__ __launch_bounds__(256) void k_sine_embed_bwd(const float* __restrict__ logits, const float* __restrict__ dim_t,
                                                        const void* __restrict__ dout, int bf16, int n, int nc, int F,
                                                        float* __restrict__ dlogits) {
  const long long rj = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (rj >= (long long)n * nc) return;
  const float pos = 1.f / (1.f + expf(-logits[rj]));
  float acc = 0.f;
  for (int f = lane; f < F; f += 64) {
    const float sc = 6.283185307179586f / dim_t[f];
    const float sarg = pos * 6.283185307179586f / dim_t[f];
    const long long o = rj * F + f;
    const float g = bf16 ? __uint_as_float((unsigned)((const unsigned short*)dout)[o] << 16) : ((const float*)dout)[o];
    acc += g * ((f & 1) ? -sinf(sarg) : cosf(sarg)) * sc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) dlogits[rj] = acc * pos * (1.f - pos);
}

extern "C" int32_t u3d_sine_embed_fwd(const float* logits, const float* dim_t, int32_t n, int32_t nc, int32_t nfeat, int32_t out_dtype,
                                      void* out, u3d_stream s) {
  U3D_REQUIRE(logits && dim_t && out && n >= 0 && nc > 0 && nfeat > 0, U3D_ERR_ARG);
  U3D_REQUIRE(out_dtype == U3D_F32 || out_dtype == U3D_BF16, U3D_ERR_UNSUPPORTED);
  if (n == 0) return U3D_OK;
  const long long total = (long long)n * nc * nfeat;
  hipLaunchKernelGGL(k_sine_embed_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, logits, dim_t, n, nc, nfeat, out_dtype == U3D_BF16, out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_sine_embed_bwd(const float* logits, const float* dim_t, const void* dout, int32_t dout_dtype, int32_t n, int32_t nc,
                                      int32_t nfeat, float* dlogits, u3d_stream s) {
  U3D_REQUIRE(logits && dim_t && dout && dlogits && n >= 0 && nc > 0 && nfeat > 0, U3D_ERR_ARG);
  U3D_REQUIRE(dout_dtype == U3D_F32 || dout_dtype == U3D_BF16, U3D_ERR_UNSUPPORTED);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_sine_embed_bwd, dim3((unsigned)(((long long)n * nc + 3) / 4)), dim3(256), 0, s, logits, dim_t, dout, dout_dtype == U3D_BF16, n,
                     nc, nfeat, dlogits);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
