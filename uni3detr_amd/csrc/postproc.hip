// Test-time post-processing on the device:
//   u3d_soft_nms   class-wise Gaussian soft-NMS with the rotated 3-D IoU (ref: models/dense_heads/uni3detr_head.py:796-823, called
//                  per class from get_bboxes :849-880) - the reference is a host loop with two .item() syncs per kept box
//   u3d_box_merge  KITTI "box_merging": score-sorted greedy merge, each kept box replaced by the per-coordinate MEDIAN of itself and
//                  the same-class boxes it absorbs (ref: core/bbox/bbox_merging.py:112-158 bboxes_nms_merge_only with
//                  overlapped_boxes_3d_fast_poly :67-92, called from uni3detr_head.py:881-891 with overlapped_thres 0.1) - the
//                  reference is numpy + shapely on the CPU
#include "common.h"

struct Q2 { float x, y; };

__device__ static int pp_clip(const Q2* in, int n, Q2 a, Q2 b, Q2* out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    Q2 p = in[i], q = in[(i + 1 == n) ? 0 : i + 1];
    float sp = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
    float sq = (b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x);
    if (sp >= 0.f) out[m++] = p;
    if ((sp >= 0.f) != (sq >= 0.f)) {
      float t = sp / (sp - sq);
      out[m++] = Q2{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
    }
  }
  return m;
}
__device__ static void pp_rect(float cx, float cy, float w, float h, float ang, Q2* c) {
  float cs = cosf(ang), sn = sinf(ang);
  const float sx[4] = {-0.5f, 0.5f, 0.5f, -0.5f}, sy[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
  for (int i = 0; i < 4; ++i) {
    float x = sx[i] * w, y = sy[i] * h;
    c[i] = Q2{cx + x * cs - y * sn, cy + x * sn + y * cs};
  }
}
// intersection area of two rectangles given as counter-clockwise corner lists (the second one relative to the first's frame)
__device__ static float pp_inter_area(const Q2* a, const Q2* b) {
  Q2 poly[12], tmp[12];
  for (int i = 0; i < 4; ++i) poly[i] = a[i];
  int m = 4;
  for (int e = 0; e < 4 && m > 0; ++e) {
    m = pp_clip(poly, m, b[e], b[(e + 1) & 3], tmp);
    for (int t = 0; t < m; ++t) poly[t] = tmp[t];
  }
  float inter = 0.f;
  if (m >= 3) {
    for (int t = 0; t < m; ++t) {
      Q2 u = poly[t], v = poly[(t + 1 == m) ? 0 : t + 1];
      inter += u.x * v.y - v.x * u.y;
    }
    inter = fabsf(inter) * 0.5f;
  }
  return inter;
}
// rotated 3-D IoU of bottom-centre LiDAR boxes (x, y, z_bottom, dx, dy, dz, yaw): the arithmetic of k_iou3d_rotated_aligned (query.hip)
__device__ static float pp_iou3d(const float* p, const float* q) {
  float w1 = fmaxf(p[3], 1e-4f), h1 = fmaxf(p[4], 1e-4f), w2 = fmaxf(q[3], 1e-4f), h2 = fmaxf(q[4], 1e-4f);
  float a1 = w1 * h1, a2 = w2 * h2, iou2d = 0.f;
  if (a1 >= 1e-14f && a2 >= 1e-14f) {
    Q2 ra[4], rb[4];
    pp_rect(0.f, 0.f, w1, h1, p[6], ra);
    pp_rect(q[0] - p[0], q[1] - p[1], w2, h2, q[6], rb);
    float inter = pp_inter_area(ra, rb);
    iou2d = inter / (a1 + a2 - inter);
  }
  float ov_bev = iou2d * (a1 + a2) / (1.f + iou2d);
  float top = fminf(p[2] + p[5], q[2] + q[5]), bot = fmaxf(p[2], q[2]);
  float ov = ov_bev * fmaxf(top - bot, 0.f);
  float v1 = p[3] * p[4] * p[5], v2 = q[3] * q[4] * q[5];
  return ov / fmaxf(v1 + v2 - ov, 1e-8f);
}

// ---------------------------------------------------------------------------------------------------------------------------
// soft-NMS: one workgroup per class.  out_idx / out_score: [num_classes][n] (selection order), out_cnt [num_classes].
// ---------------------------------------------------------------------------------------------------------------------------
#define SNMS_THREADS 256
__global__ __launch_bounds__(SNMS_THREADS) void k_soft_nms(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           const int* __restrict__ labels, int n, float sigma, float prune,
                                                           int* __restrict__ out_idx, float* __restrict__ out_score, int* __restrict__ out_cnt) {
  extern __shared__ unsigned char smem[];
  float* sc = (float*)smem;                 // [n] current score of member k (dead: -1)
  int* mem = (int*)(sc + n);                // [n] original index of member k
  __shared__ float red_v[SNMS_THREADS];
  __shared__ int red_i[SNMS_THREADS];
  __shared__ int m_sh, top_sh;
  __shared__ float topbox[7];
  const int cls = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (labels[i] == cls) { mem[m] = i; sc[m] = scores[i]; ++m; }
    m_sh = m;
  }
  __syncthreads();
  const int m = m_sh;
  int cnt = 0;
  while (true) {
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int k = tid; k < m; k += SNMS_THREADS) {
      const float v = sc[k];
      if (v >= 0.f && (v > bv || (v == bv && k < bi))) { bv = v; bi = k; }
    }
    red_v[tid] = bv; red_i[tid] = bi;
    __syncthreads();
    for (int s = SNMS_THREADS / 2; s > 0; s >>= 1) {
      if (tid < s) {
        const float v2 = red_v[tid + s];
        const int i2 = red_i[tid + s];
        if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
      }
      __syncthreads();
    }
    if (red_i[0] == 0x7fffffff) break;              // nothing alive
    if (tid == 0) {
      const int k = red_i[0];
      top_sh = k;
      out_idx[(long long)cls * n + cnt] = mem[k];
      out_score[(long long)cls * n + cnt] = red_v[0];
      for (int j = 0; j < 7; ++j) topbox[j] = boxes[(long long)mem[k] * 7 + j];
    }
    __syncthreads();
    const int top = top_sh;
    for (int k = tid; k < m; k += SNMS_THREADS) {
      float v = sc[k];
      if (v < 0.f) continue;
      const float iou = pp_iou3d(topbox, boxes + (long long)mem[k] * 7);
      v *= expf(-iou * iou / sigma);
      sc[k] = (k != top && v > prune) ? v : -1.f;
    }
    ++cnt;
    __syncthreads();
  }
  if (tid == 0) out_cnt[cls] = cnt;
}

extern "C" int32_t u3d_soft_nms(const float* boxes, const float* scores, const int32_t* labels, int32_t n, int32_t num_classes, float sigma,
                                float prune, int32_t* out_idx, float* out_score, int32_t* out_cnt, u3d_stream s) {
  U3D_REQUIRE(boxes && scores && labels && out_idx && out_score && out_cnt && num_classes > 0 && sigma > 0.f, U3D_ERR_ARG);
  if (n <= 0) {
    (void)hipMemsetAsync(out_cnt, 0, sizeof(int32_t) * num_classes, s);
    return U3D_OK;
  }
  const size_t lds = (size_t)n * 8;
  U3D_REQUIRE(lds <= 60 * 1024, U3D_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(k_soft_nms, dim3(num_classes), dim3(SNMS_THREADS), lds, s, boxes, scores, labels, n, sigma, prune, out_idx, out_score,
                     out_cnt);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// box merging.  Input boxes are ALREADY sorted by descending score.  The reference hands its LiDAR boxes (x, y, z, dx, dy, dz, yaw) to
// a routine written for (x3d, y3d, z3d, l, h, w, yaw) camera boxes, so the polygon lives in the (x, z) plane with extents (dx, dz),
// rotated by -yaw, and the "height" interval is [y - dy, y]: reproduced as is.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ static float pp_merge_overlap(const float* p, const float* q) {
  Q2 a[4], b[4];
  pp_rect(0.f, 0.f, p[3], p[5], -p[6], a);
  pp_rect(q[0] - p[0], q[2] - p[2], q[3], q[5], -q[6], b);
  float ax0 = a[0].x, ax1 = a[0].x, az0 = a[0].y, az1 = a[0].y, bx0 = b[0].x, bx1 = b[0].x, bz0 = b[0].y, bz1 = b[0].y;
  for (int i = 1; i < 4; ++i) {
    ax0 = fminf(ax0, a[i].x); ax1 = fmaxf(ax1, a[i].x); az0 = fminf(az0, a[i].y); az1 = fmaxf(az1, a[i].y);
    bx0 = fminf(bx0, b[i].x); bx1 = fmaxf(bx1, b[i].x); bz0 = fminf(bz0, b[i].y); bz1 = fmaxf(bz1, b[i].y);
  }
  const float ay1 = fmaxf(p[1], p[1] - p[4]), ay0 = fminf(p[1], p[1] - p[4]);
  const float by1 = fmaxf(q[1], q[1] - q[4]), by0 = fminf(q[1], q[1] - q[4]);
  if (ax1 < bx0 || ax0 > bx1 || az1 < bz0 || az0 > bz1 || ay1 < by0 || ay0 > by1) return 0.f;
  const float area1 = fabsf(p[3] * p[5]), area2 = fabsf(q[3] * q[5]);
  // clip in whichever orientation the corner lists have (negative extents flip it): use absolute areas
  Q2 bb[4] = {b[0], b[1], b[2], b[3]};
  float cross = (b[1].x - b[0].x) * (b[2].y - b[1].y) - (b[1].y - b[0].y) * (b[2].x - b[1].x);
  if (cross < 0.f) { bb[1] = b[3]; bb[3] = b[1]; }
  const float shared = pp_inter_area(a, bb);
  const float shared_y = fminf(by1, ay1) - fmaxf(by0, ay0);
  const float inter = shared_y * shared;
  const float uni = (by1 - by0) * area2 + (ay1 - ay0) * area1;
  return inter / (uni - inter);
}

__global__ void k_merge_mask(const float* __restrict__ boxes, const int* __restrict__ labels, int n, float thr,
                             unsigned long long* __restrict__ mask, int nw) {
  const int i = blockIdx.x, j = blockIdx.y * 64 + threadIdx.x;
  bool hit = false;
  if (j < n && j > i && labels[i] == labels[j]) hit = pp_merge_overlap(boxes + (long long)i * 7, boxes + (long long)j * 7) > thr;
  const unsigned long long b = __ballot(hit);
  if (threadIdx.x == 0) mask[(long long)i * nw + blockIdx.y] = b;
}
// one wavefront: keep[i] and, for kept i, its members (row & still-alive) written back into its mask row
__global__ void k_merge_sweep(unsigned long long* __restrict__ mask, int n, int nw, unsigned char* __restrict__ keep) {
  extern __shared__ unsigned long long removed[];
  for (int w = threadIdx.x; w < nw; w += 64) removed[w] = 0ull;
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    const bool alive = !((removed[i >> 6] >> (i & 63)) & 1ull);
    if (threadIdx.x == 0) keep[i] = alive ? 1 : 0;
    if (alive) {
      for (int w = threadIdx.x; w < nw; w += 64) {
        const unsigned long long r = mask[(long long)i * nw + w] & ~removed[w];
        mask[(long long)i * nw + w] = r;
        removed[w] |= r;
      }
    }
    __syncthreads();
  }
}
// one workgroup per box: kept boxes become the per-column median over {members, self}; others are copied
#define MERGE_THREADS 256
__global__ __launch_bounds__(MERGE_THREADS) void k_merge_median(const float* __restrict__ boxes, const unsigned long long* __restrict__ mask,
                                                                const unsigned char* __restrict__ keep, int n, int nw, float* __restrict__ out) {
  extern __shared__ int members[];           // [n]
  __shared__ int m_sh;
  const int i = blockIdx.x, tid = threadIdx.x;
  if (!keep[i]) {
    if (tid < 7) out[(long long)i * 7 + tid] = boxes[(long long)i * 7 + tid];
    return;
  }
  if (tid == 0) {
    int m = 0;
    for (int w = 0; w < nw; ++w) {
      unsigned long long b = mask[(long long)i * nw + w];
      while (b) {
        const int t = __ffsll((long long)b) - 1;
        members[m++] = w * 64 + t;
        b &= b - 1ull;
      }
    }
    members[m++] = i;
    m_sh = m;
  }
  __syncthreads();
  const int m = m_sh;
  __shared__ float lo[7], hi[7];
  for (int col = 0; col < 7; ++col) {
    // rank selection: the element whose rank (ties by position) is (m-1)/2 and the one of rank m/2
    for (int a = tid; a < m; a += MERGE_THREADS) {
      const float va = boxes[(long long)members[a] * 7 + col];
      int rank = 0;
      for (int b = 0; b < m; ++b) {
        const float vb = boxes[(long long)members[b] * 7 + col];
        rank += (vb < va || (vb == va && b < a)) ? 1 : 0;
      }
      if (rank == (m - 1) / 2) lo[col] = va;
      if (rank == m / 2) hi[col] = va;
    }
  }
  __syncthreads();
  if (tid < 7) out[(long long)i * 7 + tid] = (m & 1) ? lo[tid] : 0.5f * (lo[tid] + hi[tid]);      // numpy's median: mean of the two middle values
}

extern "C" int64_t u3d_box_merge_workspace(int32_t n) { return (int64_t)n * ((n + 63) / 64) * 8; }

extern "C" int32_t u3d_box_merge(const float* boxes, const int32_t* labels, int32_t n, float thr, float* merged, uint8_t* keep,
                                 void* workspace, int64_t workspace_bytes, u3d_stream s) {
  U3D_REQUIRE(boxes && labels && merged && keep && workspace, U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  const int nw = (n + 63) / 64;
  U3D_REQUIRE(workspace_bytes >= u3d_box_merge_workspace(n), U3D_ERR_WORKSPACE);
  U3D_REQUIRE((size_t)nw * 8 <= 48 * 1024 && (size_t)n * 4 <= 60 * 1024, U3D_ERR_UNSUPPORTED);
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(k_merge_mask, dim3(n, nw), dim3(64), 0, s, boxes, labels, n, thr, mask, nw);
  hipLaunchKernelGGL(k_merge_sweep, dim3(1), dim3(64), (size_t)nw * 8, s, mask, n, nw, keep);
  hipLaunchKernelGGL(k_merge_median, dim3(n), dim3(MERGE_THREADS), (size_t)n * 4, s, boxes, (const unsigned long long*)mask, keep, n, nw, merged);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
