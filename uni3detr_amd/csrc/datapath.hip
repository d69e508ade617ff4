// On-device training data path: the per-scene transforms of the shipped train pipelines
// (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:150-174 = RandomFlip3D -> GlobalRotScaleTrans -> PointsRangeFilter ->
// PointSample; the plugin's Unified* variants projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py:325-589 apply the same
// geometry and additionally publish the 3x3 matrix `uni_rot_aug`), run on the packed batch that already lives in HBM instead of in
// DataLoader worker processes.  The reference delegates the arithmetic to mmdet3d's box / points classes (not in /root/reference):
// semantics restated in oracle/datapath.py; the rotation / scale / flip MATRICES are the ones transform_3d.py writes out itself
// (:375-383, :429-431, :571-580), which is what the oracle is pinned to.
//
// Layout: points of all scenes packed [n_total, F] f32, scene b = rows scene_off[b] .. scene_off[b+1]); boxes packed [G, D]
// (x, y, z, dx, dy, dz, yaw [, vx, vy]), scene b = rows gt_off[b] .. gt_off[b+1]).  Per-scene parameters: f32 [B][U3D_AUG_NPARAM]
// = (flip_horizontal, flip_vertical, rot_sin, rot_cos, rot_angle, scale, tx, ty, tz): flip -> rotate -> scale -> translate, the order of
// mmdet3d's GlobalRotScaleTrans.  Everything is order-preserving and deterministic.
#include "common.h"

#define AUG_NP U3D_AUG_NPARAM

__device__ __forceinline__ int dp_scene_of(const int* __restrict__ off, int batch, int i) {
  int lo = 0, hi = batch;               // off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// coord: 0 = Depth (SUN RGB-D / ScanNet boxes), 1 = LiDAR (KITTI / nuScenes).  Flip axes follow mmdet3d v1.0 (recalled):
//   Depth : horizontal x -> -x (yaw -> pi - yaw), vertical y -> -y (yaw -> -yaw)
//   LiDAR : horizontal y -> -y (yaw -> -yaw),     vertical x -> -x (yaw -> pi - yaw)
__device__ __forceinline__ void dp_flip_xy(int coord, bool fh, bool fv, float& x, float& y) {
  if (coord == 0) { if (fh) x = -x; if (fv) y = -y; }
  else { if (fh) y = -y; if (fv) x = -x; }
}

__global__ void k_points_augment(float* __restrict__ pts, const int* __restrict__ scene_off, int batch, int n_total, int feat,
                                 const float* __restrict__ params, int coord, int height_dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  const int b = dp_scene_of(scene_off, batch, i);
  if (i >= scene_off[batch]) return;
  const float* p = params + b * AUG_NP;
  float* r = pts + (long long)i * feat;
  float x = r[0], y = r[1], z = r[2];
  dp_flip_xy(coord, p[0] != 0.f, p[1] != 0.f, x, y);
  // row vector times rot_mat_T = [[c, s, 0], [-s, c, 0], [0, 0, 1]] (transform_3d.py:380-383), then the uniform scale
  const float s = p[2], c = p[3], sc = p[5];
  const float xr = x * c - y * s, yr = x * s + y * c;
  r[0] = xr * sc + p[6]; r[1] = yr * sc + p[7]; r[2] = z * sc + p[8];      // translation last (GlobalRotScaleTrans translation_std, ScanNet configs)
  if (height_dim >= 3 && height_dim < feat) r[height_dim] *= sc;          // shift_height=True: the height attribute scales too (:411-415)
}

__global__ void k_boxes_augment(float* __restrict__ boxes, const int* __restrict__ gt_off, int batch, int n, int dim,
                                const float* __restrict__ params, int coord) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = dp_scene_of(gt_off, batch, i);
  if (i >= gt_off[batch]) return;
  const float* p = params + b * AUG_NP;
  float* r = boxes + (long long)i * dim;
  const bool fh = p[0] != 0.f, fv = p[1] != 0.f;
  float x = r[0], y = r[1], yaw = r[6];
  dp_flip_xy(coord, fh, fv, x, y);
  const float PI = 3.14159265358979323846f;
  if (coord == 0) { if (fh) yaw = -yaw + PI; if (fv) yaw = -yaw; }
  else { if (fh) yaw = -yaw; if (fv) yaw = -yaw + PI; }
  const float s = p[2], c = p[3], sc = p[5];
  r[0] = (x * c - y * s) * sc + p[6];
  r[1] = (x * s + y * c) * sc + p[7];
  r[2] = r[2] * sc + p[8];
  r[3] *= sc; r[4] *= sc; r[5] *= sc;
  r[6] = yaw + p[4];
  if (dim >= 9) {      // velocities flip and rotate with the frame and SCALE with it (mmdet3d BaseInstance3DBoxes.scale: tensor[:, 7:] *= s, recalled)
    float vx = r[7], vy = r[8];
    dp_flip_xy(coord, fh, fv, vx, vy);
    r[7] = (vx * c - vy * s) * sc;
    r[8] = (vx * s + vy * c) * sc;
  }
}

// PointsRangeFilter (mmdet3d, recalled: BasePoints.in_range_3d, strict inequalities on x, y, z): one workgroup per scene walks the
// scene's points in order, 1024 at a time, and compacts the survivors to the front of the scene's OWN segment of `out`
// (order kept: block prefix sum over the keep flags); count[b] = survivors.  out may alias the input.
#define DP_RF_THREADS 1024
__global__ __launch_bounds__(DP_RF_THREADS) void k_range_filter(const float* __restrict__ pts, const int* __restrict__ scene_off, int feat,
                                                               float lx, float ly, float lz, float hx, float hy, float hz,
                                                               float* __restrict__ out, int* __restrict__ count) {
  __shared__ int wsum[DP_RF_THREADS / 64];
  __shared__ int base_s;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int r0 = scene_off[b], r1 = scene_off[b + 1];
  if (t == 0) base_s = 0;
  __syncthreads();
  for (int c0 = r0; c0 < r1; c0 += DP_RF_THREADS) {
    const int i = c0 + t;
    float v[8];
    bool keep = false;
    if (i < r1) {
      const float* r = pts + (long long)i * feat;
      for (int f = 0; f < feat && f < 8; ++f) v[f] = r[f];
      keep = v[0] > lx && v[1] > ly && v[2] > lz && v[0] < hx && v[1] < hy && v[2] < hz;
    }
    const unsigned long long m = __ballot(keep);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();                                  // also orders this chunk's reads before any write below (out may alias pts)
    int before = 0, total = 0;
    for (int w = 0; w < DP_RF_THREADS / 64; ++w) { const int s = wsum[w]; if (w < wv) before += s; total += s; }
    const int base = base_s;
    if (keep) {
      float* o = out + (long long)(r0 + base + before + in_wave) * feat;
      for (int f = 0; f < feat && f < 8; ++f) o[f] = v[f];
    }
    __syncthreads();
    if (t == 0) base_s = base + total;
    __syncthreads();
  }
  if (t == 0) count[b] = base_s;
}

// PointSample (mmdet3d, recalled: np.random.choice(n, num_points, replace = n < num_points)).  The numpy stream cannot be replayed
// on the device; what is kept is the distribution: n >= num_points -> num_points DISTINCT rows, the first num_points values of a
// keyed pseudo-random PERMUTATION of [0, n) (4-round Feistel network on ceil(log2 n) bits + cycle walking: a bijection, no sort, no
// scratch); n < num_points -> independent uniform draws with replacement.  Row i of scene b lands in out[b * num_points + i].
__device__ __forceinline__ unsigned dp_mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned dp_perm(unsigned i, unsigned n, unsigned key) {
  unsigned bits = 2;
  while ((1u << bits) < n) ++bits;
  if (bits & 1u) ++bits;                               // balanced halves
  const unsigned half = bits >> 1, mask = (1u << half) - 1u;
  unsigned x = i;
  do {
    unsigned l = x >> half, r = x & mask;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const unsigned f = dp_mix(r ^ (key + 0x9E3779B9u * (unsigned)(rd + 1))) & mask;
      const unsigned nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (l << half) | r;
  } while (x >= n);                                    // cycle walking: the permutation of [0, 2^bits) restricted to [0, n)
  return x;
}
__global__ void k_point_sample(const float* __restrict__ pts, const int* __restrict__ scene_off, const int* __restrict__ count, int batch,
                               int feat, int num_points, const unsigned long long* __restrict__ seed, float* __restrict__ out,
                               int* __restrict__ idx_out) {
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= (long long)batch * num_points) return;
  const int b = (int)(gi / num_points), i = (int)(gi % num_points);
  const int r0 = scene_off[b];
  int n = count ? count[b] : scene_off[b + 1] - r0;
  const int seg = scene_off[b + 1] - r0;
  if (n > seg) n = seg;
  float* o = out + gi * feat;
  if (n <= 0) {
    for (int f = 0; f < feat; ++f) o[f] = 0.f;
    if (idx_out) idx_out[gi] = -1;
    return;
  }
  const unsigned long long sd = *seed;
  const unsigned key = dp_mix((unsigned)sd ^ dp_mix((unsigned)(sd >> 32) + 0x85ebca6bu * (unsigned)(b + 1)));
  const unsigned j = n >= num_points ? dp_perm((unsigned)i, (unsigned)n, key) : dp_mix((unsigned)i * 0x9E3779B1u ^ key) % (unsigned)n;
  const float* r = pts + (long long)(r0 + (int)j) * feat;
  for (int f = 0; f < feat; ++f) o[f] = r[f];
  if (idx_out) idx_out[gi] = (int)j;
}

// ObjectRangeFilter (mmdet3d, recalled: keep boxes whose BEV centre lies strictly inside (x0, y0, x1, y1) = in_range_bev, then
// limit_yaw(offset 0.5, period 2 pi): yaw -= floor(yaw / 2pi + 0.5) * 2pi).  One wave per scene compacts the survivors (order kept) to
// the front of the scene's own segment of boxes / labels; count[b] = survivors.  Segments longer than 64 boxes are walked in chunks.
__global__ __launch_bounds__(64) void k_boxes_range_filter(float* __restrict__ boxes, int* __restrict__ labels, const int* __restrict__ gt_off, int dim,
                                                           float x0, float y0, float x1, float y1, int* __restrict__ count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int r0 = gt_off[b], r1 = gt_off[b + 1];
  int base = 0;
  for (int c0 = r0; c0 < r1; c0 += 64) {
    const int i = c0 + lane;
    float v[9];
    int lab = 0;
    bool keep = false;
    if (i < r1) {
      for (int f = 0; f < dim; ++f) v[f] = boxes[(long long)i * dim + f];
      lab = labels ? labels[i] : 0;
      keep = v[0] > x0 && v[1] > y0 && v[0] < x1 && v[1] < y1;
      const float TWO_PI = 6.283185307179586f;
      v[6] = v[6] - floorf(v[6] / TWO_PI + 0.5f) * TWO_PI;
    }
    const unsigned long long m = __ballot(keep);          // every lane has read its row before any lane writes (one wave, in order)
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) {
      for (int f = 0; f < dim; ++f) boxes[(long long)(r0 + pos) * dim + f] = v[f];
      if (labels) labels[r0 + pos] = lab;
    }
    base += __popcll(m);
  }
  if (lane == 0) count[b] = base;
}

extern "C" int32_t u3d_points_augment(float* points, const int32_t* scene_off, int32_t batch, int32_t n_total, int32_t feat,
                                      const float* params, int32_t coord, int32_t height_dim, u3d_stream s) {
  U3D_REQUIRE(points && scene_off && params && batch > 0 && feat >= 3 && (coord == 0 || coord == 1), U3D_ERR_ARG);
  if (n_total <= 0) return U3D_OK;
  hipLaunchKernelGGL(k_points_augment, dim3(u3d_cdiv(n_total, 256)), dim3(256), 0, s, points, scene_off, batch, n_total, feat, params, coord, height_dim);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_boxes_augment(float* boxes, const int32_t* gt_off, int32_t batch, int32_t n, int32_t box_dim, const float* params,
                                     int32_t coord, u3d_stream s) {
  U3D_REQUIRE(boxes && gt_off && params && batch > 0 && (box_dim == 7 || box_dim == 9) && (coord == 0 || coord == 1), U3D_ERR_ARG);
  if (n <= 0) return U3D_OK;
  hipLaunchKernelGGL(k_boxes_augment, dim3(u3d_cdiv(n, 256)), dim3(256), 0, s, boxes, gt_off, batch, n, box_dim, params, coord);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_points_range_filter(const float* points, const int32_t* scene_off, int32_t batch, int32_t feat, const float* range6,
                                           float* out, int32_t* count, u3d_stream s) {
  U3D_REQUIRE(points && scene_off && range6 && out && count && batch > 0 && feat >= 3 && feat <= 8, U3D_ERR_ARG);
  hipLaunchKernelGGL(k_range_filter, dim3(batch), dim3(DP_RF_THREADS), 0, s, points, scene_off, feat, range6[0], range6[1], range6[2], range6[3],
                     range6[4], range6[5], out, count);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_point_sample(const float* points, const int32_t* scene_off, const int32_t* count, int32_t batch, int32_t feat,
                                    int32_t num_points, const uint64_t* seed, float* out, int32_t* idx_out, u3d_stream s) {
  U3D_REQUIRE(points && scene_off && seed && out && batch > 0 && feat >= 1 && num_points > 0, U3D_ERR_ARG);
  const long long total = (long long)batch * num_points;
  hipLaunchKernelGGL(k_point_sample, dim3(u3d_cdiv(total, 256)), dim3(256), 0, s, points, scene_off, count, batch, feat, num_points,
                     (const unsigned long long*)seed, out, idx_out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_boxes_range_filter(float* boxes, int32_t* labels, const int32_t* gt_off, int32_t batch, int32_t box_dim,
                                          const float* bev_range4, int32_t* count, u3d_stream s) {
  U3D_REQUIRE(boxes && gt_off && bev_range4 && count && batch > 0 && (box_dim == 7 || box_dim == 9), U3D_ERR_ARG);
  hipLaunchKernelGGL(k_boxes_range_filter, dim3(batch), dim3(64), 0, s, boxes, labels, gt_off, box_dim, bev_range4[0], bev_range4[1],
                     bev_range4[2], bev_range4[3], count);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
