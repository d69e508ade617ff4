// Fused decoder layer, forward (include/u3d_hip.h "Fused decoder layer"; ref: projects/mmdet3d_plugin/models/utils/
// uni3detr_transformer.py:145-212 (decoder loop), :33-65 (sine embedding), :271-360 (UniCrossAtten), mmcv BaseTransformerLayer /
// MultiheadAttention / FFN as the shipped configs build them, models/dense_heads/uni3detr_head.py:367-387 (branches)).
//
//   k_dec_pre   rows: sine embedding -> ref_point_head (3 GEMMs) [-> query_scale (3 GEMMs), product] -> q=k input -> in-projection
//   k_mha_fwd   (group, head, 64 queries): softmax(q k^T / sqrt(32)) v, online softmax over 128-key chunks held in LDS
//   k_dec_post  rows: out-proj + residual + LN1 -> gate + trilinear sample -> output_proj + position encoder + LN2 -> FFN + LN3
//               -> reg / iou / cls branches
// Each row kernel keeps its 32 rows in LDS from the first GEMM to the last; only what the backward needs goes to HBM.
#include "decoder_common.h"

struct DcPtrs {            // forward-save slot pointers (device), resolved on the host from the slot offsets
  u16 *sine, *rph1, *rph2, *raw, *qs1, *qs2, *qs, *pos, *qkin, *qk, *v;
  float* lse;
  u16* o;
  float *u1, *mr;
  u16 *qp, *samp, *gated, *peh0, *upe1;
  float* u2;
  u16 *x2c, *ffh;
  float* u3;
  u16 *r1, *r2, *i1, *i2, *uc1, *c1, *uc2, *c2;
};

// ---------------------------------------------------------------------------------------------------------------------------
// slot layout
// ---------------------------------------------------------------------------------------------------------------------------
static const int kSaveCols[U3D_DS_COUNT][2] = {   // (columns, bytes per element)
    {384, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // SINE RPH1 RPH2 RAW QS1 QS2 QS POS QKIN
    {512, 2}, {256, 2}, {8, 4}, {256, 2}, {256, 4}, {16, 4}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // QK V LSE O U1 MR QP SAMP GATED PEH0
    {256, 2}, {256, 4}, {256, 2}, {512, 2}, {256, 4}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // UPE1 U2 X2C FFH U3 R1 R2 I1 I2 UC1
    {256, 2}, {256, 2}, {256, 2}};                                                                        // C1 UC2 C2

extern "C" int32_t u3d_decoder_layer_blocks(int32_t m) { return u3d_cdiv(m > 0 ? m : 1, DC_BM); }

extern "C" int32_t u3d_decoder_layer_slots(int32_t m, int32_t ncls, int32_t code, int64_t* save_off, int64_t* grad_off) {
  U3D_REQUIRE(m > 0 && ncls > 0 && ncls <= 32 && code > 0 && code <= 32, U3D_ERR_ARG);
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  m = u3d_decoder_layer_blocks(m) * DC_BM;          // every slot holds whole 32-row blocks: the row kernels never branch on the row index
  if (save_off) {
    int64_t o = 0;
    for (int i = 0; i < U3D_DS_COUNT; ++i) {
      save_off[i] = o;
      o += al((int64_t)m * kSaveCols[i][0] * kSaveCols[i][1]);
    }
    save_off[U3D_DS_COUNT] = o;
  }
  if (grad_off) {
    const int nb = u3d_decoder_layer_blocks(m);
    int64_t o = 0;
    for (int i = 0; i < U3D_DG_COUNT; ++i) {
      grad_off[i] = o;
      int64_t bytes;
      switch (i) {
        case U3D_DG_CLSO: bytes = (int64_t)m * ncls * 2; break;
        case U3D_DG_REGO: bytes = (int64_t)m * code * 2; break;
        case U3D_DG_IOUO: case U3D_DG_WL: bytes = (int64_t)m * 2; break;
        case U3D_DG_FFH: case U3D_DG_DQK: bytes = (int64_t)m * 512 * 2; break;
        case U3D_DG_SINE: bytes = (int64_t)m * 384 * 2; break;
        case U3D_DG_LNP: bytes = (int64_t)U3D_DL_NLN * 2 * nb * DC_C * 4; break;      // [ln][dgamma|dbeta][block][256] f32
        case U3D_DG_DU1: bytes = (int64_t)m * DC_C * 4; break;
        default: bytes = (int64_t)m * DC_C * 2; break;
      }
      o += al(bytes);
    }
    grad_off[U3D_DG_COUNT] = o;
  }
  return U3D_OK;
}

static DcPtrs dc_resolve(void* save, int m) {
  int64_t off[U3D_DS_COUNT + 1];
  u3d_decoder_layer_slots(m, 1, 1, off, nullptr);
  char* b = (char*)save;
  DcPtrs p;
  p.sine = (u16*)(b + off[U3D_DS_SINE]); p.rph1 = (u16*)(b + off[U3D_DS_RPH1]); p.rph2 = (u16*)(b + off[U3D_DS_RPH2]);
  p.raw = (u16*)(b + off[U3D_DS_RAW]); p.qs1 = (u16*)(b + off[U3D_DS_QS1]); p.qs2 = (u16*)(b + off[U3D_DS_QS2]);
  p.qs = (u16*)(b + off[U3D_DS_QS]); p.pos = (u16*)(b + off[U3D_DS_POS]); p.qkin = (u16*)(b + off[U3D_DS_QKIN]);
  p.qk = (u16*)(b + off[U3D_DS_QK]); p.v = (u16*)(b + off[U3D_DS_V]); p.lse = (float*)(b + off[U3D_DS_LSE]);
  p.o = (u16*)(b + off[U3D_DS_O]); p.u1 = (float*)(b + off[U3D_DS_U1]); p.mr = (float*)(b + off[U3D_DS_MR]);
  p.qp = (u16*)(b + off[U3D_DS_QP]); p.samp = (u16*)(b + off[U3D_DS_SAMP]); p.gated = (u16*)(b + off[U3D_DS_GATED]);
  p.peh0 = (u16*)(b + off[U3D_DS_PEH0]); p.upe1 = (u16*)(b + off[U3D_DS_UPE1]); p.u2 = (float*)(b + off[U3D_DS_U2]);
  p.x2c = (u16*)(b + off[U3D_DS_X2C]); p.ffh = (u16*)(b + off[U3D_DS_FFH]); p.u3 = (float*)(b + off[U3D_DS_U3]);
  p.r1 = (u16*)(b + off[U3D_DS_R1]); p.r2 = (u16*)(b + off[U3D_DS_R2]); p.i1 = (u16*)(b + off[U3D_DS_I1]);
  p.i2 = (u16*)(b + off[U3D_DS_I2]); p.uc1 = (u16*)(b + off[U3D_DS_UC1]); p.c1 = (u16*)(b + off[U3D_DS_C1]);
  p.uc2 = (u16*)(b + off[U3D_DS_UC2]); p.c2 = (u16*)(b + off[U3D_DS_C2]);
  return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight packing: f32 master -> bf16 [n_pad][k] and its transpose [k][n_pad_t]
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wpack(const u3d_wpack_desc* __restrict__ descs) {
  const u3d_wpack_desc d = descs[blockIdx.y];
  const int total = d.n_pad * d.k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int n = i / d.k, k = i % d.k;
    ((u16*)d.dst)[i] = n < d.n ? dc_f2bf(d.src[(size_t)n * d.k + k]) : (u16)0;
  }
  if (d.dst_t) {
    const int tt = d.k * d.n_pad_t;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tt; i += gridDim.x * 256) {
      const int k = i / d.n_pad_t, n = i % d.n_pad_t;
      ((u16*)d.dst_t)[i] = n < d.n ? dc_f2bf(d.src[(size_t)n * d.k + k]) : (u16)0;
    }
  }
}
extern "C" int32_t u3d_wpack_bf16(const u3d_wpack_desc* descs_dev, int32_t count, int32_t max_elems, u3d_stream s) {
  U3D_REQUIRE(descs_dev && count >= 0 && max_elems > 0, U3D_ERR_ARG);
  if (count == 0) return U3D_OK;
  int gx = u3d_cdiv(max_elems, 256 * 4);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_wpack, dim3(gx, count), dim3(256), 0, s, descs_dev);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

__global__ void k_dropout_mask(const unsigned long long* rng, int layer, int site, long long n, unsigned thresh, unsigned char* keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keep[i] = thresh == 0u ? 1 : (dc_keep(dc_rng_load(rng), dc_site_key(layer, site), (unsigned)i, thresh) ? 1 : 0);
}
extern "C" int32_t u3d_dropout_mask(const uint64_t* rng, int32_t layer, int32_t site, int64_t n, float p, uint8_t* keep, u3d_stream s) {
  U3D_REQUIRE(rng && keep && n >= 0 && p >= 0.f && p < 1.f, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned long long*)rng, layer, site,
                     (long long)n, dc_thresh(p), keep);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_pre
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DC_THREADS) void k_dec_pre(u3d_declayer_params P, u3d_declayer_dims dm, const u16* __restrict__ xc,
                                                        const float* __restrict__ ref, DcPtrs S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u16* A0 = (u16*)(lds + DC_OFF_A0);
  u16* A1 = (u16*)(lds + DC_OFF_A1);
  u16* A2 = (u16*)(lds + DC_OFF_A2);
  float* misc = (float*)(lds + DC_OFF_MISC);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int row0 = blockIdx.x * DC_BM, M = dm.m;
  dc_poison_lds(lds, tid);
  (void)misc;
  // sine embedding of the reference points -> A0 (ldk 384), saved for the first GEMM's weight gradient
  DC_FOR_TID(c, DC_BM * 48) {
    const int row = c / 48, ch = c % 48, coord = ch >> 4, f0 = (ch & 15) * 8;
    const float pos = dc_sigmoid(ref[(size_t)min(row0 + row, M - 1) * 3 + coord]);
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sarg = pos * 6.283185307179586f / P.dim_t[f0 + e];
      v[e] = dc_f2bf((e & 1) ? cosf(sarg) : sinf(sarg));
    }
    *(u16x8*)(A0 + dc_aoff(row, ch * 8, 384)) = v;
    *(u16x8*)(S.sine + (size_t)(row0 + row) * 384 + ch * 8) = v;
  }
  __syncthreads();
  // linear + ReLU into an activation tile and its save slot
  auto relu_to = [&](u16* dst, u16* gsave, const float* bias) {
    return [=](int row, int col, f32x4 v) {
      const u16x4 o = dc_pack4(dc_relu4(v + dc_bias4(bias, col)));
      *(u16x4*)(dst + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gsave + (size_t)(row0 + row) * DC_C + col) = o;
    };
  };
  dc_linear<384, 4>(A0, (const u16*)P.w[U3D_DL_RPH0], wave * 64, lane, relu_to(A1, S.rph1, P.b[U3D_DL_RPH0]));
  __syncthreads();
  dc_linear<256, 4>(A1, (const u16*)P.w[U3D_DL_RPH1], wave * 64, lane, relu_to(A0, S.rph2, P.b[U3D_DL_RPH1]));
  __syncthreads();
  // raw = ref_point_head's output (bf16) -> A2; it is the position embedding itself in the first layer
  {
    const float* bias = P.b[U3D_DL_RPH2];
    u16* gsave = dm.has_qs ? S.raw : S.pos;
    dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_RPH2], wave * 64, lane, [=](int row, int col, f32x4 v) {
      const u16x4 o = dc_pack4(v + dc_bias4(bias, col));
      *(u16x4*)(A2 + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gsave + (size_t)(row0 + row) * DC_C + col) = o;
    });
  }
  __syncthreads();
  dc_load_a<256, true>(A0, xc, DC_C, row0, M, tid);           // x (bf16): input of query_scale and of the value projection
  __syncthreads();
  if (dm.has_qs) {
    dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_QS0], wave * 64, lane, relu_to(A1, S.qs1, P.b[U3D_DL_QS0]));
    __syncthreads();
    // A0 still holds x (the value projection needs it): the second hidden layer goes to the second 32x256 half of A1
    u16* A1b = A1 + DC_BM * DC_C;
    dc_linear<256, 4>(A1, (const u16*)P.w[U3D_DL_QS1], wave * 64, lane, relu_to(A1b, S.qs2, P.b[U3D_DL_QS1]));
    __syncthreads();
    const float* bias = P.b[U3D_DL_QS2];
    // pos = query_scale(x) * raw (both bf16 tensors in the layer-by-layer formulation) -> A2 in place (same element, same lane)
    dc_linear<256, 4>(A1b, (const u16*)P.w[U3D_DL_QS2], wave * 64, lane, [=](int row, int col, f32x4 v) {
      const u16x4 q = dc_pack4(v + dc_bias4(bias, col));
      u16* ap = A2 + dc_aoff(row, col, DC_C);
      const f32x4 raw = dc_unpack4(*(const u16x4*)ap);
      const u16x4 p = dc_pack4(dc_unpack4(q) * raw);
      *(u16x4*)ap = p;
      *(u16x4*)(S.qs + (size_t)(row0 + row) * DC_C + col) = q;
      *(u16x4*)(S.pos + (size_t)(row0 + row) * DC_C + col) = p;
    });
    __syncthreads();
  }
  // q = k input: x + pos (bf16 add) -> A1
  DC_FOR_TID(c, DC_BM * 32) {
    const int off = c * 8;                                  // A0, A1, A2 share one (row, chunk) permutation at ldk = 256
    const u16x8 a = *(const u16x8*)(A0 + off), p = *(const u16x8*)(A2 + off);
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = dc_f2bf(dc_bf2f(a[e]) + dc_bf2f(p[e]));
    *(u16x8*)(A1 + off) = r;
  }
  __syncthreads();
  dc_store_a<256>(A1, S.qkin, DC_C, row0, tid);
  // in-projection: (q | k) = A1 . Wqk^T + b, v = A0 . Wv^T + b -> HBM (the attention kernel regroups rows by (group, head))
  auto to_global = [&](u16* dst, int ld, const float* bias) {
    return [=](int row, int col, f32x4 v) {
      *(u16x4*)(dst + (size_t)(row0 + row) * ld + col) = dc_pack4(v + dc_bias4(bias, col));
    };
  };
  dc_linear<256, 4>(A1, (const u16*)P.w[U3D_DL_INQK], wave * 64, lane, to_global(S.qk, 512, P.b[U3D_DL_INQK]));
  dc_linear<256, 4>(A1, (const u16*)P.w[U3D_DL_INQK], 256 + wave * 64, lane, to_global(S.qk, 512, P.b[U3D_DL_INQK]));
  dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_INV], wave * 64, lane, to_global(S.v, 256, P.b[U3D_DL_INV]));
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_mha_fwd: grid (ceil(nq/64), groups*8), 4 waves x 16 queries.  Scores are computed transposed (S^T = K Q^T) so that a lane
// owns ONE query (column) and 4 keys per tile: the softmax statistics are per-lane scalars, and the exponentials of two
// consecutive tiles are exactly the 8 reduction elements of the P.V MFMA's b-operand (V^T from LDS as the a-operand with the
// same key permutation) - probabilities never leave registers.  lse is kept in log2 units.
// ---------------------------------------------------------------------------------------------------------------------------
#define MHA_KC 128                 /* keys per LDS chunk */
#define MHA_VT_LD (MHA_KC + 8)     /* row stride of V^T [d][key]: 272 B -> conflict-free ds_read_b64 */
__device__ __forceinline__ int mha_koff(int key, int part) { return key * 32 + (((part ^ ((-(key >> 2)) & 3)) & 3) << 3); }

// stage `n` rows (64 B each, head slice) of a row matrix into LDS: row-major swizzled copy and/or the transpose [32][ld_t]
__device__ __forceinline__ void mha_stage(const u16* __restrict__ src, int ld, long long base_row, int first, int nvalid, u16* rowmajor,
                                          u16* transposed, int tid, int nthreads) {
  (void)nthreads;
  DC_FOR_TID(c, MHA_KC * 4) {
    const int key = c >> 2, part = c & 3;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (first + key < nvalid) v = *(const u16x8*)(src + (base_row + first + key) * ld + part * 8);
    if (rowmajor) *(u16x8*)(rowmajor + mha_koff(key, part)) = v;
    if (transposed) {
#pragma unroll
      for (int e = 0; e < 8; ++e) transposed[(part * 8 + e) * MHA_VT_LD + key] = v[e];
    }
  }
}

__global__ __launch_bounds__(256) void k_mha_fwd(const u16* __restrict__ qk, const u16* __restrict__ vv, int nq, float scale_log2,
                                                 unsigned thresh, float inv_keep, int layer, const unsigned long long* __restrict__ rng,
                                                 u16* __restrict__ o, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) u16 Ks[MHA_KC * 32];
  __shared__ __attribute__((aligned(16))) u16 Vt[32 * MHA_VT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const int q = blockIdx.x * 64 + wave * 16 + r16;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  bf16x8 qf = {0, 0, 0, 0, 0, 0, 0, 0};
  if (q < nq) qf = *(const bf16x8*)(qk + (base + q) * 512 + h * DC_HD + kq * 8);
  f32x4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float m_run = -INFINITY, l_run = 0.f;
  for (int kc0 = 0; kc0 < nq; kc0 += MHA_KC) {
    __syncthreads();
    mha_stage(qk + 256 + h * DC_HD, 512, base, kc0, nq, Ks, nullptr, tid, 256);
    mha_stage(vv + h * DC_HD, 256, base, kc0, nq, nullptr, Vt, tid, 256);
    __syncthreads();
    const int nkeys = min(MHA_KC, nq - kc0);
    const int ntile = (nkeys + 15) >> 4;
    f32x4 s[MHA_KC / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MHA_KC / 16; ++t) {
      s[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (t < ntile) {
        const bf16x8 kf = *(const bf16x8*)(Ks + mha_koff(t * 16 + r16, kq));
        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + kq * 4 + r;
          a[r] = key < nkeys ? a[r] * scale_log2 : -INFINITY;
          mx = fmaxf(mx, a[r]);
        }
        s[t] = a;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);            // first chunk: exp2(-inf) = 0
    l_run *= corr;
    oacc[0] *= corr; oacc[1] *= corr;
    m_run = m_new;
#pragma unroll
    for (int tp = 0; tp < MHA_KC / 32; ++tp) {
      if (tp * 2 < ntile) {
        f32x4 p0, p1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[r] = __builtin_amdgcn_exp2f(s[2 * tp][r] - m_new);
          p1[r] = __builtin_amdgcn_exp2f(s[2 * tp + 1][r] - m_new);
          l_run += p0[r] + p1[r];
        }
        if (thresh) {
          const unsigned rowidx = (unsigned)(((long long)bh * nq + q) * nq + kc0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p0[r] = dc_keep(rg, key_site, rowidx + (2 * tp) * 16 + kq * 4 + r, thresh) ? p0[r] * inv_keep : 0.f;
            p1[r] = dc_keep(rg, key_site, rowidx + (2 * tp + 1) * 16 + kq * 4 + r, thresh) ? p1[r] * inv_keep : 0.f;
          }
        }
        const u16x4 b0 = dc_pack4(p0), b1 = dc_pack4(p1);
        const u16x8 pb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const u16* vp = Vt + (dt * 16 + r16) * MHA_VT_LD + tp * 32 + kq * 4;
          const u16x4 v0 = *(const u16x4*)vp, v1 = *(const u16x4*)(vp + 16);
          const u16x8 vb = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vb), pf, oacc[dt], 0, 0, 0);
        }
      }
    }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (q < nq) {
    const float inv = 1.f / l_run;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) *(u16x4*)(o + (base + q) * DC_C + h * DC_HD + dt * 16 + kq * 4) = dc_pack4(oacc[dt] * inv);
    if (kq == 0) lse[(base + q) * DC_NHEAD + h] = m_run + log2f(l_run);
  }
}

extern "C" int32_t u3d_mha_fwd(const void* qk, const void* v, int32_t m, int32_t nq, float p_attn, int32_t layer, const uint64_t* rng,
                               void* o, float* lse, u3d_stream s) {
  U3D_REQUIRE(qk && v && o && lse && m > 0 && nq > 0 && m % nq == 0 && p_attn >= 0.f && p_attn < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE(p_attn == 0.f || rng, U3D_ERR_ARG);
  U3D_REQUIRE((long long)(m / nq) * DC_NHEAD * nq * nq < (1ll << 32), U3D_ERR_UNSUPPORTED);
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)DC_HD);
  hipLaunchKernelGGL(k_mha_fwd, dim3(u3d_cdiv(nq, 64), (m / nq) * DC_NHEAD), dim3(256), 0, s, (const u16*)qk, (const u16*)v, nq, scale_log2,
                     dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (u16*)o, lse);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_post
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DC_THREADS) void k_dec_post(u3d_declayer_params P, u3d_declayer_dims dm, const float* __restrict__ x,
                                                         const float* __restrict__ ref, const u16* __restrict__ value,
                                                         const unsigned long long* __restrict__ rng, DcPtrs S, float* __restrict__ x_out,
                                                         u16* __restrict__ xc_out, float* __restrict__ reg_out,
                                                         float* __restrict__ cls_out, float* __restrict__ iou_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u16* A0 = (u16*)(lds + DC_OFF_A0);
  u16* A1 = (u16*)(lds + DC_OFF_A1);
  u16* A2 = (u16*)(lds + DC_OFF_A2);
  float* F = (float*)(lds + DC_OFF_F);
  float* G = (float*)(lds + DC_OFF_G);
  float* misc = (float*)(lds + DC_OFF_MISC);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int row0 = blockIdx.x * DC_BM, M = dm.m;
  dc_poison_lds(lds, tid);
  DcDrop drop = {dc_rng_load(rng), dc_thresh(dm.p_drop), dc_inv_keep(dm.p_drop), dm.layer};

  dc_load_a<256, true>(A0, S.o, DC_C, row0, M, tid);           // the attention kernel writes m rows only: padded rows repeat row m-1 (finite)
  dc_load_f<false>(F, x, row0, M, tid);
  dc_load_a<256, false>(A2, S.pos, DC_C, row0, M, tid);
  {                                           // reference-point logits of the block's rows: every lane of a row's wave reads them later
    const int row = tid >> 3, j = tid & 7;    // 32 rows x 8 slots, slots 3..7 unused
    misc[row * DC_MISC_LD + j] = ref[(size_t)min(row0 + row, M - 1) * 3 + min(j, 2)];
  }
  __syncthreads();
  // residual stream += dropout(linear(A)) (the linear's output is a bf16 tensor in the layer-by-layer formulation)
  auto add_to_F = [&](const float* bias, int site) {
    return [=](int row, int col, f32x4 v) {
      v = dc_round4(v + dc_bias4(bias, col));
      v = drop.apply(v, site, (unsigned)((row0 + row) * DC_C + col));
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + v;
    };
  };
  dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_OUTP], wave * 64, lane, add_to_F(P.b[U3D_DL_OUTP], 0));
  __syncthreads();
  {
    DcLnOut o = {F, A0, DC_C, nullptr, nullptr, S.mr, U3D_DLN_1, false, S.u1};
    dc_layernorm(F, P.ln_g[U3D_DLN_1], P.ln_b[U3D_DLN_1], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  // cross "attention": gate = sigmoid(attention_weights(x1 + pos)), one trilinear sample of the value volume per query
  {
    const f32x4 aw = *(const f32x4*)(P.attw_w + lane * 4);
    const float ab = P.attw_b[0];
#pragma unroll 2
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      const size_t gr = (size_t)(row0 + row);
      const int ao = dc_aoff(row, lane * 4, DC_C);
      const f32x4 x1 = dc_unpack4(*(const u16x4*)(A0 + ao)), pp = dc_unpack4(*(const u16x4*)(A2 + ao));
      const u16x4 qpb = dc_pack4(x1 + pp);
      const f32x4 qp = dc_unpack4(qpb);
      *(u16x4*)(S.qp + gr * DC_C + lane * 4) = qpb;
      const float wl = dc_round(u3d_wave_sum(qp[0] * aw[0] + qp[1] * aw[1] + qp[2] * aw[2] + qp[3] * aw[3]) + ab);
      const float gate = dc_round(dc_sigmoid(wl));
      S.mr[gr * 16 + 14] = wl;                     // all lanes, same value
      DcCorners tc;
      dc_corners(misc + row * DC_MISC_LD, min(row0 + row, M - 1) / dm.qps, dm.dz, dm.dy, dm.dx, tc);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (tc.row[c] >= 0) acc += tc.w[c] * dc_unpack4(*(const u16x4*)(value + (size_t)tc.row[c] * DC_C + lane * 4));
      const u16x4 sb = dc_pack4(acc);
      const u16x4 gb = dc_pack4(dc_unpack4(sb) * gate);
      *(u16x4*)(A1 + ao) = gb;
      *(u16x4*)(S.samp + gr * DC_C + lane * 4) = sb;
      *(u16x4*)(S.gated + gr * DC_C + lane * 4) = gb;
    }
  }
  // position encoder, first layer (3 -> 256): plain VALU into G
  {
    const int col = tid;
    const float w0 = dc_round(P.pe0_w[col * 3 + 0]), w1 = dc_round(P.pe0_w[col * 3 + 1]), w2 = dc_round(P.pe0_w[col * 3 + 2]);
    const float b = P.pe0_b[col];
    for (int row = 0; row < DC_BM; ++row) {
      const float* r3 = misc + row * DC_MISC_LD;
      G[row * DC_TS + col] = dc_round(dc_round(r3[0]) * w0 + dc_round(r3[1]) * w1 + dc_round(r3[2]) * w2 + b);
    }
  }
  __syncthreads();
  dc_linear<256, 4>(A1, (const u16*)P.w[U3D_DL_OPROJ], wave * 64, lane, add_to_F(P.b[U3D_DL_OPROJ], 1));
  {
    DcLnOut o = {nullptr, A0, DC_C, nullptr, S.peh0, S.mr, U3D_DLN_PE0, false, nullptr};
    dc_layernorm(G, P.ln_g[U3D_DLN_PE0], P.ln_b[U3D_DLN_PE0], dm.ln_eps, true, o, row0, wave, lane);   // overwrites A0 (x1 bf16: no longer needed)
  }
  __syncthreads();
  {
    const float* bias = P.b[U3D_DL_PE1];
    dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_PE1], wave * 64, lane, [=](int row, int col, f32x4 v) {
      const u16x4 ub = dc_pack4(v + dc_bias4(bias, col));
      *(f32x4*)(G + row * DC_TS + col) = dc_unpack4(ub);
      *(u16x4*)(S.upe1 + (size_t)(row0 + row) * DC_C + col) = ub;
    });
  }
  __syncthreads();
  {
    DcLnOut o = {G, nullptr, 0, nullptr, nullptr, S.mr, U3D_DLN_PE1, true, nullptr};
    dc_layernorm(G, P.ln_g[U3D_DLN_PE1], P.ln_b[U3D_DLN_PE1], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  DC_FOR_TID(c, DC_BM * 64) {                                   // x2 (pre-norm) = x1 + cross output + position feature
    const int o = (c >> 6) * DC_TS + (c & 63) * 4;
    *(f32x4*)(F + o) = *(const f32x4*)(F + o) + *(const f32x4*)(G + o);
  }
  __syncthreads();
  {
    DcLnOut o = {F, A0, DC_C, nullptr, S.x2c, S.mr, U3D_DLN_2, false, S.u2};
    dc_layernorm(F, P.ln_g[U3D_DLN_2], P.ln_b[U3D_DLN_2], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  // FFN
  {
    const float* bias = P.b[U3D_DL_FFN0];
    auto ffh = [=](int row, int col, f32x4 v) {
      v = dc_round4(dc_relu4(v + dc_bias4(bias, col)));
      const u16x4 hb = dc_pack4(drop.apply(v, 2, (unsigned)((row0 + row) * DC_FF + col)));
      *(u16x4*)(A1 + dc_aoff(row, col, DC_FF)) = hb;
      *(u16x4*)(S.ffh + (size_t)(row0 + row) * DC_FF + col) = hb;
    };
    dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_FFN0], wave * 64, lane, ffh);
    dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_FFN0], 256 + wave * 64, lane, ffh);
  }
  __syncthreads();
  dc_linear<512, 4>(A1, (const u16*)P.w[U3D_DL_FFN1], wave * 64, lane, add_to_F(P.b[U3D_DL_FFN1], 3));
  __syncthreads();
  {
    DcLnOut o = {nullptr, A2, DC_C, x_out, xc_out, S.mr, U3D_DLN_3, false, S.u3};
    dc_layernorm(F, P.ln_g[U3D_DLN_3], P.ln_b[U3D_DLN_3], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  // branches on the layer state x3 (A2)
  auto relu_to = [&](u16* dst, u16* gsave, const float* bias) {
    return [=](int row, int col, f32x4 v) {
      const u16x4 o = dc_pack4(dc_relu4(v + dc_bias4(bias, col)));
      *(u16x4*)(dst + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gsave + (size_t)(row0 + row) * DC_C + col) = o;
    };
  };
  auto narrow_out = [&](float* dst, int n, const float* bias) {     // final layer of a branch: n <= 32 real columns of the 64 computed
    return [=](int row, int col, f32x4 v) {       // n is uniform: the column test is the only lane-dependent branch (stores only)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < n) dst[(size_t)(row0 + row) * n + col + r] = dc_round(v[r] + bias[col + r]);
    };
  };
  u16* A1b = A1 + DC_BM * DC_C;
  dc_linear<256, 4>(A2, (const u16*)P.w[U3D_DL_REG0], wave * 64, lane, relu_to(A0, S.r1, P.b[U3D_DL_REG0]));
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_REG1], wave * 64, lane, relu_to(A1, S.r2, P.b[U3D_DL_REG1]));
  __syncthreads();
  dc_linear<256, 1>(A1, (const u16*)P.w[U3D_DL_REG2], wave * 16, lane, narrow_out(reg_out, dm.code, P.b[U3D_DL_REG2]));
  dc_linear<256, 4>(A2, (const u16*)P.w[U3D_DL_IOU0], wave * 64, lane, relu_to(A0, S.i1, P.b[U3D_DL_IOU0]));
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_IOU1], wave * 64, lane, relu_to(A1b, S.i2, P.b[U3D_DL_IOU1]));
  __syncthreads();
  dc_linear<256, 1>(A1b, (const u16*)P.w[U3D_DL_IOU2], wave * 16, lane, narrow_out(iou_out, 1, P.b[U3D_DL_IOU2]));
  // cls: Linear -> LN -> ReLU twice, then the class logits
  auto to_G = [&](u16* gsave, const float* bias) {
    return [=](int row, int col, f32x4 v) {
      const u16x4 ub = dc_pack4(v + dc_bias4(bias, col));
      *(f32x4*)(G + row * DC_TS + col) = dc_unpack4(ub);
      *(u16x4*)(gsave + (size_t)(row0 + row) * DC_C + col) = ub;
    };
  };
  dc_linear<256, 4>(A2, (const u16*)P.w[U3D_DL_CLS0], wave * 64, lane, to_G(S.uc1, P.b[U3D_DL_CLS0]));
  __syncthreads();
  {
    DcLnOut o = {nullptr, A0, DC_C, nullptr, S.c1, S.mr, U3D_DLN_C1, false, nullptr};
    dc_layernorm(G, P.ln_g[U3D_DLN_C1], P.ln_b[U3D_DLN_C1], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.w[U3D_DL_CLS1], wave * 64, lane, to_G(S.uc2, P.b[U3D_DL_CLS1]));
  __syncthreads();
  {
    DcLnOut o = {nullptr, A1, DC_C, nullptr, S.c2, S.mr, U3D_DLN_C2, false, nullptr};
    dc_layernorm(G, P.ln_g[U3D_DLN_C2], P.ln_b[U3D_DLN_C2], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  dc_linear<256, 1>(A1, (const u16*)P.w[U3D_DL_CLS2], wave * 16, lane, narrow_out(cls_out, dm.ncls, P.b[U3D_DL_CLS2]));
}

static int32_t dc_check(const u3d_declayer_params* p, const u3d_declayer_dims* d) {
  U3D_REQUIRE(p && d, U3D_ERR_ARG);
  U3D_REQUIRE(d->m > 0 && d->nq > 0 && d->qps > 0 && d->qps % d->nq == 0 && d->m % d->qps == 0 && d->batch == d->m / d->qps, U3D_ERR_ARG);
  U3D_REQUIRE(d->ncls > 0 && d->ncls <= 32 && d->code > 0 && d->code <= 32, U3D_ERR_UNSUPPORTED);
  U3D_REQUIRE(d->p_attn >= 0.f && d->p_attn < 1.f && d->p_drop >= 0.f && d->p_drop < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE((long long)d->m * DC_FF < (1ll << 32), U3D_ERR_UNSUPPORTED);
  for (int i = 0; i < U3D_DL_NLIN; ++i) {
    if (!d->has_qs && (i == U3D_DL_QS0 || i == U3D_DL_QS1 || i == U3D_DL_QS2)) continue;
    U3D_REQUIRE(p->w[i] && p->b[i], U3D_ERR_ARG);
  }
  for (int i = 0; i < U3D_DL_NLN; ++i) U3D_REQUIRE(p->ln_g[i] && p->ln_b[i], U3D_ERR_ARG);
  U3D_REQUIRE(p->attw_w && p->attw_b && p->pe0_w && p->pe0_b && p->dim_t, U3D_ERR_ARG);
  return U3D_OK;
}

extern "C" int32_t u3d_decoder_layer_fwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc,
                                         const float* ref, const void* value, const uint64_t* rng, float* x_out, void* xc_out,
                                         float* reg_out, float* cls_out, float* iou_out, void* save, int64_t save_bytes, u3d_stream s) {
  int32_t rc = dc_check(p, d);
  if (rc != U3D_OK) return rc;
  U3D_REQUIRE(x && xc && ref && value && x_out && xc_out && reg_out && cls_out && iou_out && save, U3D_ERR_ARG);
  U3D_REQUIRE((d->p_attn == 0.f && d->p_drop == 0.f) || rng, U3D_ERR_ARG);
  int64_t off[U3D_DS_COUNT + 1];
  u3d_decoder_layer_slots(d->m, d->ncls, d->code, off, nullptr);
  U3D_REQUIRE(save_bytes >= off[U3D_DS_COUNT], U3D_ERR_WORKSPACE);
  const DcPtrs S = dc_resolve(save, d->m);
  const int nb = u3d_decoder_layer_blocks(d->m);
  U3D_ALLOW_LDS(k_dec_pre, DC_LDS_BYTES);
  U3D_ALLOW_LDS(k_dec_post, DC_LDS_BYTES);
  hipLaunchKernelGGL(k_dec_pre, dim3(nb), dim3(DC_THREADS), DC_LDS_BYTES, s, *p, *d, (const u16*)xc, ref, S);
  rc = u3d_mha_fwd(S.qk, S.v, d->m, d->nq, d->p_attn, d->layer, rng, S.o, S.lse, s);
  if (rc != U3D_OK) return rc;
  hipLaunchKernelGGL(k_dec_post, dim3(nb), dim3(DC_THREADS), DC_LDS_BYTES, s, *p, *d, x, ref, (const u16*)value,
                     (const unsigned long long*)rng, S, x_out, (u16*)xc_out, reg_out, cls_out, iou_out);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
