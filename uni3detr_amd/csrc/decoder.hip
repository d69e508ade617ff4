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

#ifdef DC_PHASE_TIMING       /* tools/dec_bench.py: shader-clock stamps of ONE workgroup at the phase boundaries of k_dec_post */
__device__ unsigned long long dc_dbg_fwd[64];
#define DC_MARK(id) do { if (blockIdx.x == 7 && threadIdx.x == 0) dc_dbg_fwd[id] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t u3d_debug_fwd_times(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dc_dbg_fwd), 64 * 8) == hipSuccess ? 0 : -1; }
#define MP_MARK(id) do { if (blockIdx.x == 7 && threadIdx.x == 0) dc_dbg_fwd[32 + (id)] = __builtin_readcyclecounter(); } while (0)
#else
#define DC_MARK(id)
#define MP_MARK(id)
#endif

template <typename E>
struct DcPtrs {            // forward-save slot pointers (device), resolved on the host from the slot offsets
  typedef typename E::T T;
  T *sine, *rph1, *rph2, *raw, *qs1, *qs2, *qs, *pos, *qkin, *qk, *v;
  float* lse;
  T* o;
  float *u1, *mr;
  T *qp, *samp, *gated, *peh0, *upe1;
  float* u2;
  T *x2c, *ffh;
  float* u3;
  T *r1, *r2, *i1, *i2, *uc1, *c1, *uc2, *c2;
  unsigned* amask;
};

// ---------------------------------------------------------------------------------------------------------------------------
// slot layout
// ---------------------------------------------------------------------------------------------------------------------------
static const int kSaveCols[U3D_DS_COUNT][2] = {   // (columns, bytes per element: 2 = element type of the mode (bf16 / f32), 4 = always f32)
    {384, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // SINE RPH1 RPH2 RAW QS1 QS2 QS POS QKIN
    {512, 2}, {256, 2}, {8, 4}, {256, 2}, {256, 4}, {16, 4}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // QK V LSE O U1 MR QP SAMP GATED PEH0
    {256, 2}, {256, 4}, {256, 2}, {512, 2}, {256, 4}, {256, 2}, {256, 2}, {256, 2}, {256, 2}, {256, 2},   // UPE1 U2 X2C FFH U3 R1 R2 I1 I2 UC1
    {256, 2}, {256, 2}, {256, 2},                                                                         // C1 UC2 C2
    {80, 4}};                                                                                             // AMASK (8 heads x 10 words per query row)

extern "C" int32_t u3d_decoder_layer_blocks_dt(int32_t m, int32_t dtype) { return u3d_cdiv(m > 0 ? m : 1, dc_bm(dtype)); }
extern "C" int32_t u3d_decoder_layer_blocks(int32_t m) { return u3d_decoder_layer_blocks_dt(m, U3D_BF16); }

extern "C" int32_t u3d_decoder_layer_slots_dt(int32_t m, int32_t ncls, int32_t code, int32_t dtype, int64_t* save_off, int64_t* grad_off) {
  U3D_REQUIRE(m > 0 && ncls > 0 && ncls <= 32 && code > 0 && code <= 32, U3D_ERR_ARG);
  U3D_REQUIRE(dtype == U3D_BF16 || dtype == U3D_F32, U3D_ERR_ARG);
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  const int es = dc_esize(dtype);
  const int nb = u3d_decoder_layer_blocks_dt(m, dtype);
  m = nb * dc_bm(dtype);          // every slot holds whole row blocks: the row kernels never branch on the row index
  if (save_off) {
    int64_t o = 0;
    for (int i = 0; i < U3D_DS_COUNT; ++i) {
      save_off[i] = o;
      o += al((int64_t)m * kSaveCols[i][0] * (kSaveCols[i][1] == 2 ? es : 4));
    }
    save_off[U3D_DS_COUNT] = o;
  }
  if (grad_off) {
    int64_t o = 0;
    for (int i = 0; i < U3D_DG_COUNT; ++i) {
      grad_off[i] = o;
      int64_t bytes;
      switch (i) {
        case U3D_DG_CLSO: bytes = (int64_t)m * ncls * es; break;
        case U3D_DG_REGO: bytes = (int64_t)m * code * es; break;
        case U3D_DG_IOUO: case U3D_DG_WL: bytes = (int64_t)m * es; break;
        case U3D_DG_FFH: case U3D_DG_DQK: bytes = (int64_t)m * 512 * es; break;
        case U3D_DG_SINE: bytes = (int64_t)m * 384 * es; break;
        case U3D_DG_LNP: bytes = (int64_t)U3D_DL_NLN * 2 * nb * DC_C * 4; break;      // [ln][dgamma|dbeta][block][256] f32
        case U3D_DG_DU1: bytes = (int64_t)m * DC_C * 4; break;
        default: bytes = (int64_t)m * DC_C * es; break;
      }
      o += al(bytes);
    }
    grad_off[U3D_DG_COUNT] = o;
  }
  return U3D_OK;
}
extern "C" int32_t u3d_decoder_layer_slots(int32_t m, int32_t ncls, int32_t code, int64_t* save_off, int64_t* grad_off) {
  return u3d_decoder_layer_slots_dt(m, ncls, code, U3D_BF16, save_off, grad_off);
}

template <typename E>
static DcPtrs<E> dc_resolve(void* save, int m) {
  typedef typename E::T T;
  int64_t off[U3D_DS_COUNT + 1];
  u3d_decoder_layer_slots_dt(m, 1, 1, E::DT, off, nullptr);
  char* b = (char*)save;
  DcPtrs<E> p;
  p.sine = (T*)(b + off[U3D_DS_SINE]); p.rph1 = (T*)(b + off[U3D_DS_RPH1]); p.rph2 = (T*)(b + off[U3D_DS_RPH2]);
  p.raw = (T*)(b + off[U3D_DS_RAW]); p.qs1 = (T*)(b + off[U3D_DS_QS1]); p.qs2 = (T*)(b + off[U3D_DS_QS2]);
  p.qs = (T*)(b + off[U3D_DS_QS]); p.pos = (T*)(b + off[U3D_DS_POS]); p.qkin = (T*)(b + off[U3D_DS_QKIN]);
  p.qk = (T*)(b + off[U3D_DS_QK]); p.v = (T*)(b + off[U3D_DS_V]); p.lse = (float*)(b + off[U3D_DS_LSE]);
  p.o = (T*)(b + off[U3D_DS_O]); p.u1 = (float*)(b + off[U3D_DS_U1]); p.mr = (float*)(b + off[U3D_DS_MR]);
  p.qp = (T*)(b + off[U3D_DS_QP]); p.samp = (T*)(b + off[U3D_DS_SAMP]); p.gated = (T*)(b + off[U3D_DS_GATED]);
  p.peh0 = (T*)(b + off[U3D_DS_PEH0]); p.upe1 = (T*)(b + off[U3D_DS_UPE1]); p.u2 = (float*)(b + off[U3D_DS_U2]);
  p.x2c = (T*)(b + off[U3D_DS_X2C]); p.ffh = (T*)(b + off[U3D_DS_FFH]); p.u3 = (float*)(b + off[U3D_DS_U3]);
  p.r1 = (T*)(b + off[U3D_DS_R1]); p.r2 = (T*)(b + off[U3D_DS_R2]); p.i1 = (T*)(b + off[U3D_DS_I1]);
  p.i2 = (T*)(b + off[U3D_DS_I2]); p.uc1 = (T*)(b + off[U3D_DS_UC1]); p.c1 = (T*)(b + off[U3D_DS_C1]);
  p.uc2 = (T*)(b + off[U3D_DS_UC2]); p.c2 = (T*)(b + off[U3D_DS_C2]);
  p.amask = (unsigned*)(b + off[U3D_DS_AMASK]);
  return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight packing: f32 master -> T [n_pad][k] and its transpose [k][n_pad_t] (T = bf16: rounded copies; T = f32: padded copies), both in
// the MFMA FRAGMENT ORDER dc_gemm reads (decoder_common.h); t_plain keeps the transpose row-major
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void k_wpack(const u3d_wpack_desc* __restrict__ descs) {
  typedef typename E::T T;
  const u3d_wpack_desc d = descs[blockIdx.y];
  constexpr int WBLK = 64 * E::CH;
  // dst: W [n_pad][k] in fragment order - block (tile = n / 16, ks = k / KSTEP) holds lane l = kq * 16 + r16 -> elements
  // W[tile * 16 + r16][ks * KSTEP + kq * CH + e]
  const int total = d.n_pad * d.k, KS = d.k / E::KSTEP;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int e = i % E::CH, l = (i / E::CH) & 63, blk = i / WBLK, ks = blk % KS, tile = blk / KS;
    const int n = tile * 16 + (l & 15), k = ks * E::KSTEP + (l >> 4) * E::CH + e;
    ((T*)d.dst)[i] = n < d.n ? E::from_f(d.src[(size_t)n * d.k + k]) : E::from_f(0.f);
  }
  if (d.dst_t) {
    const int tt = d.k * d.n_pad_t;
    if (d.t_plain) {                                   // [k][n_pad_t] row-major (narrow final layers: read by plain VALU code)
      for (int i = blockIdx.x * 256 + threadIdx.x; i < tt; i += gridDim.x * 256) {
        const int k = i / d.n_pad_t, n = i % d.n_pad_t;
        ((T*)d.dst_t)[i] = n < d.n ? E::from_f(d.src[(size_t)n * d.k + k]) : E::from_f(0.f);
      }
    } else {                                           // W^T [k][n_pad_t] in fragment order (rows = k, reduction index = n)
      const int KST = d.n_pad_t / E::KSTEP;
      for (int i = blockIdx.x * 256 + threadIdx.x; i < tt; i += gridDim.x * 256) {
        const int e = i % E::CH, l = (i / E::CH) & 63, blk = i / WBLK, ks = blk % KST, tile = blk / KST;
        const int k = tile * 16 + (l & 15), n = ks * E::KSTEP + (l >> 4) * E::CH + e;
        ((T*)d.dst_t)[i] = n < d.n ? E::from_f(d.src[(size_t)n * d.k + k]) : E::from_f(0.f);
      }
    }
  }
}
extern "C" int32_t u3d_wpack(const u3d_wpack_desc* descs_dev, int32_t count, int32_t max_elems, int32_t dtype, u3d_stream s) {
  U3D_REQUIRE(descs_dev && count >= 0 && max_elems > 0 && (dtype == U3D_BF16 || dtype == U3D_F32), U3D_ERR_ARG);
  if (count == 0) return U3D_OK;
  int gx = u3d_cdiv(max_elems, 256 * 4);
  if (gx > 64) gx = 64;
  if (dtype == U3D_BF16) hipLaunchKernelGGL(k_wpack<EB>, dim3(gx, count), dim3(256), 0, s, descs_dev);
  else hipLaunchKernelGGL(k_wpack<EF>, dim3(gx, count), dim3(256), 0, s, descs_dev);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_wpack_bf16(const u3d_wpack_desc* descs_dev, int32_t count, int32_t max_elems, u3d_stream s) {
  return u3d_wpack(descs_dev, count, max_elems, U3D_BF16, s);
}

__global__ void k_dropout_mask(const unsigned long long* rng, int layer, int site, long long n, unsigned thresh, int cols, unsigned char* keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // cols > 0: element i is (row i / cols, column i % cols) of a matrix whose rows are padded to an even length in index space (the
  // attention weights, dc_att_idx)
  const unsigned idx = cols > 0 ? dc_att_idx((unsigned)(i / cols), (unsigned)(i % cols), (unsigned)(cols + 1) & ~1u) : (unsigned)i;
  keep[i] = thresh == 0u ? 1 : (dc_keep(dc_rng_load(rng), dc_site_key(layer, site), idx, thresh) ? 1 : 0);
}
extern "C" int32_t u3d_dropout_mask(const uint64_t* rng, int32_t layer, int32_t site, int64_t n, float p, int32_t cols, uint8_t* keep,
                                    u3d_stream s) {
  U3D_REQUIRE(rng && keep && n >= 0 && p >= 0.f && p < 1.f && cols >= 0, U3D_ERR_ARG);
  if (n == 0) return U3D_OK;
  hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned long long*)rng, layer, site,
                     (long long)n, dc_thresh(p), (int)cols, keep);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_pre
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(DC_THREADS) void k_dec_pre(u3d_declayer_params P, u3d_declayer_dims dm, const typename E::T* __restrict__ xc,
                                                        const float* __restrict__ ref, DcPtrs<E> S, int skip_inproj) {
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef typename E::VC VC;
  typedef DcLds<E> L;
  constexpr int BM = E::BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  T* A0 = (T*)(lds + L::A0);
  T* A1 = (T*)(lds + L::A1);
  T* A2 = (T*)(lds + L::A2);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int wv = (wave + (int)(blockIdx.x >> 3)) & 3;      // which 64 output columns this wave computes: rotated per workgroup (see dc_gemm)
  const int row0 = blockIdx.x * BM, M = dm.m;
  dc_poison_lds<E>(lds, tid);
  // sine embedding of the reference points -> A0 (ldk 384), saved for the first GEMM's weight gradient
  constexpr int SCH = 384 / E::CH;                 // chunks per row
  DC_FOR_TID(c, BM * SCH) {
    const int row = c / SCH, ch = c % SCH, col0 = ch * E::CH, coord = col0 >> 7, f0 = col0 & 127;
    const float pos = E::sigmoid(ref[(size_t)min(row0 + row, M - 1) * 3 + coord]);
    VC v;
#pragma unroll
    for (int e = 0; e < E::CH; ++e) {
      const float sarg = pos * 6.283185307179586f / P.dim_t[f0 + e];
      v[e] = E::from_f((e & 1) ? cosf(sarg) : sinf(sarg));
    }
    *(VC*)(A0 + dc_aoff<E>(row, col0, 384)) = v;
    *(VC*)(S.sine + (size_t)(row0 + row) * 384 + col0) = v;
  }
  __syncthreads();
  // Every linear's output lands in an LDS activation tile; what the backward (or the attention kernel) needs in HBM is stored FROM
  // THE TILE after the barrier, 16 bytes per lane over whole 512-byte rows (dc_store_a).  Stores straight from the MFMA fragments
  // (8 bytes per lane, 16 rows per instruction) were store-issue bound: ~3-4 k clocks of a 10 k-clock linear (tools/dec_bench.py).
  auto relu_to = [&](T* dst, const float* bias) {
    return [=](int row, int col, f32x4 v) {
      *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(dc_relu4(v + dc_bias4(bias, col)));
    };
  };
  auto lin_to = [&](T* dst, const float* bias) {
    return [=](int row, int col, f32x4 v) { *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(v + dc_bias4(bias, col)); };
  };
  T* A1b = A1 + BM * DC_C;
  dc_linear<E, 384, 4>(A0, (const T*)P.w[U3D_DL_RPH0], wv * 64, lane, relu_to(A1, P.b[U3D_DL_RPH0]));
  __syncthreads();
  dc_store_a<E, 256>(A1, S.rph1, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A1, (const T*)P.w[U3D_DL_RPH1], wv * 64, lane, relu_to(A0, P.b[U3D_DL_RPH1]));
  __syncthreads();
  dc_store_a<E, 256>(A0, S.rph2, DC_C, row0, tid);
  // raw = ref_point_head's output -> A2; it is the position embedding itself in the first layer
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_RPH2], wv * 64, lane, lin_to(A2, P.b[U3D_DL_RPH2]));
  __syncthreads();
  dc_store_a<E, 256>(A2, dm.has_qs ? S.raw : S.pos, DC_C, row0, tid);
  dc_load_a<E, 256, true>(A0, xc, DC_C, row0, M, tid);           // x: input of query_scale and of the value projection
  __syncthreads();
  if (dm.has_qs) {
    dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_QS0], wv * 64, lane, relu_to(A1, P.b[U3D_DL_QS0]));
    __syncthreads();
    dc_store_a<E, 256>(A1, S.qs1, DC_C, row0, tid);
    // A0 still holds x (the value projection needs it): the second hidden layer goes to the second BM x 256 half of A1
    dc_linear<E, 256, 4>(A1, (const T*)P.w[U3D_DL_QS1], wv * 64, lane, relu_to(A1b, P.b[U3D_DL_QS1]));
    __syncthreads();
    dc_store_a<E, 256>(A1b, S.qs2, DC_C, row0, tid);
    const float* bias = P.b[U3D_DL_QS2];
    // pos = query_scale(x) * raw (both T tensors in the layer-by-layer formulation) -> A2 in place (same element, same lane);
    // the scale itself -> first half of A1 (free: its reader, the linear before, finished at the barrier)
    dc_linear<E, 256, 4>(A1b, (const T*)P.w[U3D_DL_QS2], wv * 64, lane, [=](int row, int col, f32x4 v) {
      const V4 q = E::pack4(v + dc_bias4(bias, col));
      const int ao = dc_aoff<E>(row, col, DC_C);
      const f32x4 raw = E::unpack4(*(const V4*)(A2 + ao));
      *(V4*)(A2 + ao) = E::pack4(E::unpack4(q) * raw);
      *(V4*)(A1 + ao) = q;
    });
    __syncthreads();
    dc_store_a<E, 256>(A1, S.qs, DC_C, row0, tid);
    dc_store_a<E, 256>(A2, S.pos, DC_C, row0, tid);
  }
  // q = k input: x + pos -> A1b (free in both paths)
  DC_FOR_TID(c, BM * (DC_C / E::CH)) {
    const int off = c * E::CH;                              // the tiles share one (row, chunk) permutation at ldk = 256
    const VC a = *(const VC*)(A0 + off), p = *(const VC*)(A2 + off);
    VC r;
#pragma unroll
    for (int e = 0; e < E::CH; ++e) r[e] = E::from_f(E::to_f(a[e]) + E::to_f(p[e]));
    *(VC*)(A1b + off) = r;
  }
  __syncthreads();
  dc_store_a<E, 256>(A1b, S.qkin, DC_C, row0, tid);
  if (skip_inproj) return;                               // (uniform) the attention launch projects its own head: k_mha_proj_fwd
  // in-projection: q = A1b . Wq^T + b -> A1, k -> A2 (pos is saved), v = A0 . Wv^T + b -> A1b; each block leaves for HBM from its tile
  // (the attention kernel regroups rows by (group, head))
  dc_linear<E, 256, 4>(A1b, (const T*)P.w[U3D_DL_INQK], wv * 64, lane, lin_to(A1, P.b[U3D_DL_INQK]));
  {
    const float* bias = P.b[U3D_DL_INQK];
    dc_linear<E, 256, 4>(A1b, (const T*)P.w[U3D_DL_INQK], 256 + wv * 64, lane, [=](int row, int col, f32x4 v) {
      *(V4*)(A2 + dc_aoff<E>(row, col - 256, DC_C)) = E::pack4(v + dc_bias4(bias, col));
    });
  }
  __syncthreads();
  dc_store_a<E, 256>(A1, S.qk, 512, row0, tid);
  dc_store_a<E, 256>(A2, S.qk + 256, 512, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_INV], wv * 64, lane, lin_to(A1b, P.b[U3D_DL_INV]));
  __syncthreads();
  dc_store_a<E, 256>(A1b, S.v, DC_C, row0, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------
// One staged chunk of keys / values against the 16 queries of a wave (lane: query column r16, keys kq*4 .. +3 of every 16-key tile):
// scores, running-maximum rescale, exponentials, dropout, P.V.  Shared by k_mha_fwd and k_mha_proj_fwd.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__device__ __forceinline__ void mha_chunk(const typename E::T* Ks, const typename E::T* Vs, const typename E::T* Vt, int kc0, int nkeys,
                                          const typename Mha<E>::RowFrag& qf, unsigned row, unsigned nq_pad, DcRng rg, unsigned key_site,
                                          unsigned thresh, float inv_keep, float scale_log2, int r16, int kq, float& m_run, float& l_run,
                                          f32x4 (&oacc)[2], const unsigned* __restrict__ keep_bits = nullptr) {
  // keep_bits (nullable, uniform): the query's keep-mask words of this chunk, bit b of word tp = key kc0 + 32 tp + b - the SAME
  // decisions as the hash below (it made them), read instead of recomputed
  typedef Mha<E> H;
  constexpr int KC = H::KC;
  const int ntile = (nkeys + 15) >> 4;
  f32x4 s[KC / 16];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < KC / 16; ++t) {
    s[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (t < ntile) {
      f32x4 a = H::scores(Ks, t * 16, r16, kq, qf);
      if (t == ntile - 1) {                          // only the last tile can hold keys past the group (zero rows in the image)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = t * 16 + kq * 4 + r < nkeys ? a[r] : -INFINITY;
      }
      mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
      s[t] = a;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx);
  const float corr = E::exp2((m_run - m_new) * scale_log2);            // first chunk: exp2(-inf) = 0
  const float neg_m = -m_new * scale_log2;
  l_run *= corr;
  oacc[0] *= corr; oacc[1] *= corr;
  m_run = m_new;
#pragma unroll
  for (int tp = 0; tp < KC / 32; ++tp) {
    if (tp * 2 < ntile) {
      f32x4 p0, p1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] = E::exp2(fmaf(s[2 * tp][r], scale_log2, neg_m));
        p1[r] = E::exp2(fmaf(s[2 * tp + 1][r], scale_log2, neg_m));
        l_run += p0[r] + p1[r];
      }
      if (thresh && keep_bits) {
        // bit -> all-ones / zero word in one v_bfe_i32, applied to the (non-negative) probability's bits; 1 / (1 - p) multiplies
        const int w = (int)keep_bits[tp];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[r] = __int_as_float(__float_as_int(p0[r] * inv_keep) & __builtin_amdgcn_sbfe(w, kq * 4 + r, 1));
          p1[r] = __int_as_float(__float_as_int(p1[r] * inv_keep) & __builtin_amdgcn_sbfe(w, 16 + kq * 4 + r, 1));
        }
      } else if (thresh) {
        bool k0[4], k1[4];
        dc_keep4(rg, key_site, dc_att_idx(row, (unsigned)(kc0 + (2 * tp) * 16 + kq * 4), nq_pad), thresh, k0);
        dc_keep4(rg, key_site, dc_att_idx(row, (unsigned)(kc0 + (2 * tp + 1) * 16 + kq * 4), nq_pad), thresh, k1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[r] *= k0[r] ? inv_keep : 0.f;
          p1[r] *= k1[r] ? inv_keep : 0.f;
        }
      }
      H::pv(Vs, Vt, tp, r16, kq, p0, p1, oacc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_mha_fwd: grid (ceil(nq / 128), groups*8), 4 waves.  A workgroup stages a chunk of keys / values of its (group, head) ONCE and walks
// NQI = 2 tiles of 64 queries over it (wave w: 16 queries of each).  Scores are computed transposed (S^T = K Q^T) so that a lane
// owns ONE query (column) and 4 keys per tile: the softmax statistics are per-lane scalars, and the exponentials of two
// consecutive tiles are exactly the reduction elements of the P.V MFMA's b-operand; its a-operand V^T comes out of the row-major V
// image through transpose reads - probabilities never leave registers, nothing is transposed in LDS.  lse is kept in log2 units.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256, 2) void k_mha_fwd(const typename E::T* __restrict__ qk, const typename E::T* __restrict__ vv, int nq, int qt_per_wg,
                                                    float scale_log2, unsigned thresh, float inv_keep, int layer,
                                                    const unsigned long long* __restrict__ rng, typename E::T* __restrict__ o,
                                                    float* __restrict__ lse) {
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef Mha<E> H;
  constexpr int KC = H::KC;
  __shared__ __attribute__((aligned(16))) T Ks[H::RM_ELEMS];
  __shared__ __attribute__((aligned(16))) T Vs[H::TR ? H::RM_ELEMS : 8];
  __shared__ __attribute__((aligned(16))) T Vt[H::TP_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const unsigned nq_pad = (unsigned)(nq + 1) & ~1u;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  const bool one_chunk = nq <= KC;                       // the whole group is resident: staged once, every query tile walks it
  for (int qt = 0; qt < qt_per_wg; ++qt) {
    const int q0 = (blockIdx.x * qt_per_wg + qt) * 64;
    if (q0 >= nq) break;                                 // uniform
    const int q = q0 + wave * 16 + r16;
    typename H::RowFrag qf = H::zero_frag();
    if (q < nq) qf = H::load_frag(qk + (base + q) * 512 + h * DC_HD, kq);
    f32x4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float m_run = -INFINITY, l_run = 0.f;                // running maximum of the RAW scores (the scale is positive)
    const unsigned row = (unsigned)(bh * nq + q);
    for (int kc0 = 0; kc0 < nq; kc0 += KC) {
      if (qt == 0 || !one_chunk) {
        __syncthreads();
        H::stage(qk + 256 + h * DC_HD, 512, base, kc0, nq, Ks, nullptr, tid);
        H::stage(vv + h * DC_HD, 256, base, kc0, nq, H::TR ? Vs : nullptr, Vt, tid);
        __syncthreads();
      }
      mha_chunk<E>(Ks, Vs, Vt, kc0, min(KC, nq - kc0), qf, row, nq_pad, rg, key_site, thresh, inv_keep, scale_log2, r16, kq, m_run, l_run, oacc);
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (q < nq) {
      const float inv = 1.f / l_run;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) *(V4*)(o + (base + q) * DC_C + h * DC_HD + dt * 16 + kq * 4) = E::pack4(oacc[dt] * inv);
      if (kq == 0) lse[(base + q) * DC_NHEAD + h] = m_run * scale_log2 + log2f(l_run);
    }
  }
}

extern "C" int32_t u3d_mha_fwd_dt(const void* qk, const void* v, int32_t m, int32_t nq, float p_attn, int32_t layer, const uint64_t* rng,
                                  void* o, float* lse, int32_t dtype, u3d_stream s) {
  U3D_REQUIRE(qk && v && o && lse && m > 0 && nq > 0 && m % nq == 0 && p_attn >= 0.f && p_attn < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE(p_attn == 0.f || rng, U3D_ERR_ARG);
  U3D_REQUIRE(dtype == U3D_BF16 || dtype == U3D_F32, U3D_ERR_ARG);
  U3D_REQUIRE((long long)(m / nq) * DC_NHEAD * nq * (nq + 1) < (1ll << 32), U3D_ERR_UNSUPPORTED);
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)DC_HD);
  if (dtype == U3D_BF16) {
    const int qt = mha_tiles_per_wg<EB>(nq);
    hipLaunchKernelGGL(k_mha_fwd<EB>, dim3(u3d_cdiv(nq, 64 * qt), (m / nq) * DC_NHEAD), dim3(256), 0, s, (const u16*)qk, (const u16*)v, nq, qt,
                       scale_log2, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (u16*)o, lse);
  } else {
    const int qt = mha_tiles_per_wg<EF>(nq);
    hipLaunchKernelGGL(k_mha_fwd<EF>, dim3(u3d_cdiv(nq, 64 * qt), (m / nq) * DC_NHEAD), dim3(256), 0, s, (const float*)qk, (const float*)v, nq,
                       qt, scale_log2, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (float*)o, lse);
  }
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_mha_fwd(const void* qk, const void* v, int32_t m, int32_t nq, float p_attn, int32_t layer, const uint64_t* rng,
                               void* o, float* lse, u3d_stream s) {
  return u3d_mha_fwd_dt(qk, v, m, nq, p_attn, layer, rng, o, lse, U3D_BF16, s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_mha_proj_fwd (bf16, groups of at most KC = 320 queries): the in-projection AND the attention of ONE (group, head) per workgroup
// (ref: mmcv MultiheadAttention = nn.MultiheadAttention(q = k = x + pos, v = x), uni3detr_sunrgbd.py:79-83).  The three launches of a
// layer were k_dec_pre (... -> in-projection -> qk / v slots) | k_mha_fwd | k_dec_post; a (group, head) workgroup of k_mha_fwd then
// spent a launch, a staging round and a softmax on 11.5 MFLOP of MFMA work.  Here the workgroup also computes its head's
// q / k / v = [x + pos | x] . W_h^T + b (14.7 MFLOP; 96 of the 768 in-projection rows: 48 KiB of fragment-packed weights, staged in
// LDS once and read as a-operands, activation rows straight from L2 as b-operands, a row block ahead), writes them into the LDS
// images the attention reads (and to the qk / v slots the backward reads), and walks ALL query tiles of the group over them:
// 8 waves, wave w takes the 16-query blocks w, w + 8, w + 16.  k_dec_pre stops after the x + pos slot.
// ---------------------------------------------------------------------------------------------------------------------------
#define MHA_PROJ_THREADS 512
struct MhaProjLds {
  static constexpr int IMG = Mha<EB>::KC * Mha<EB>::RS;          // elements of one row-major image (q, k, v)
  static constexpr int WT = 6 * (DC_C / 32) * 512;               // six 16-row weight tiles x 8 k-steps x (64 lanes x 8 elements)
  static constexpr int MW = Mha<EB>::KC / 32;                    // dropout keep-mask words per query row (one bit per key)
  static constexpr int MASK = Mha<EB>::KC * MW;                  // 32-bit words
  static constexpr int BYTES = (3 * IMG + WT) * 2 + MASK * 4;
};
static_assert(MhaProjLds::BYTES <= 160 * 1024, "q / k / v images + the head's weight tiles must fit the CU's LDS");
static_assert(Mha<EB>::KC / 16 <= 3 * (MHA_PROJ_THREADS / 64), "three row blocks per wave cover a chunk");
__global__ __launch_bounds__(MHA_PROJ_THREADS) void k_mha_proj_fwd(const u16* __restrict__ qkin, const u16* __restrict__ xc,
                                                                   const u16* __restrict__ wqk, const float* __restrict__ bqk,
                                                                   const u16* __restrict__ wv, const float* __restrict__ bv, int nq,
                                                                   float scale_log2, unsigned thresh, float inv_keep, int layer,
                                                                   const unsigned long long* __restrict__ rng, u16* __restrict__ qk_out,
                                                                   u16* __restrict__ v_out, u16* __restrict__ o, float* __restrict__ lse,
                                                                   unsigned* __restrict__ amask) {
  typedef Mha<EB> H;
  typedef u16x8 VC;
  typedef u16x4 V4;
  constexpr int KC = H::KC, RS = H::RS, KS = DC_C / 32, WBLK = 512, NW = MHA_PROJ_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) u16 sm_[];
  u16* Qs = sm_;
  u16* Ks = Qs + MhaProjLds::IMG;
  u16* Vs = Ks + MhaProjLds::IMG;
  u16* Ws = Vs + MhaProjLds::IMG;
  unsigned* Ms = (unsigned*)(Ws + MhaProjLds::WT);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
  // workgroups are dealt to the 8 XCDs round-robin by their linear id: the eight HEADS of a group read the same 2 x nq activation
  // rows, so they are given ids that are congruent mod 8 (one XCD, one L2 fill per row) - head-major inside blocks of 64 ids
  int bh = blockIdx.x;
  {
    const int ngrp = gridDim.x >> 3, blk = bh >> 6, in = bh & 63, g_ = blk * 8 + (in & 7);
    if ((blk + 1) * 8 <= ngrp) bh = g_ * 8 + (in >> 3);     // whole blocks of 8 groups; a ragged tail keeps the plain order
  }
  const int g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const int nrb = (nq + 15) >> 4;                          // 16-row blocks of the group (<= KC / 16 = 20)

  // ---- phase A: q | k | v of this head for all rows of the group -------------------------------------------------------------
  // every request of the phase goes out first, in the order it is needed (vmcnt retires in order): the head's six weight tiles
  // (fragment-packed: copied as they are), the six bias quads, then the b-operands (x + pos | x) of ALL of this wave's row blocks
  constexpr int RBW = 3;                                   // row blocks per wave: ceil(20 / 8)
  VC wt[6];
  {
    const u16* src[6] = {wqk + (size_t)(2 * h) * KS * WBLK,      wqk + (size_t)(2 * h + 1) * KS * WBLK,
                         wqk + (size_t)(16 + 2 * h) * KS * WBLK, wqk + (size_t)(17 + 2 * h) * KS * WBLK,
                         wv + (size_t)(2 * h) * KS * WBLK,       wv + (size_t)(2 * h + 1) * KS * WBLK};
#pragma unroll
    for (int t = 0; t < 6; ++t) wt[t] = *(const VC*)(src[t] + tid * 8);
  }
  const int bcol = h * DC_HD + kq * 4;
  f32x4 bias[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) bias[t] = *(const f32x4*)((t < 2 ? bqk : (t < 4 ? bqk + 256 : bv)) + bcol + (t & 1) * 16);
  VC a1[2][KS], a2[2][KS];                                 // two row blocks in flight; the third takes the first one's registers
  auto load_rows = [&](int rb, VC (&p1)[KS], VC (&p2)[KS]) {
    const int row = min(rb * 16 + r16, nq - 1);
    const u16* s1 = qkin + (base + row) * DC_C + kq * 8;
    const u16* s2 = xc + (base + row) * DC_C + kq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      p1[ks] = *(const VC*)(s1 + ks * 32);
      p2[ks] = *(const VC*)(s2 + ks * 32);
    }
  };
  MP_MARK(0);
  if (wave < nrb) load_rows(wave, a1[0], a2[0]);           // wave-uniform
  if (wave + NW < nrb) load_rows(wave + NW, a1[1], a2[1]);
  const unsigned nq_pad = (unsigned)(nq + 1) & ~1u;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  __builtin_amdgcn_sched_barrier(0);                       // every request above is issued before the first hash instruction
  if (thresh) {
    // the (group, head)'s dropout keep-mask, one bit per (query, key), hashed HERE - underneath the requests above, whose data takes
    // ~20 k clocks to arrive while the vector ALU idles - instead of on the score path, where the hash was a third of the softmax
    constexpr int MW = MhaProjLds::MW;
    const int words = nrb * 16 * MW, iters = (words + MHA_PROJ_THREADS - 1) / MHA_PROJ_THREADS;
    for (int it = 0; it < iters; ++it) {
      const int wd = it * MHA_PROJ_THREADS + tid;
      if (wd < words) {
        const int q = wd / MW, tp = wd - q * MW;
        const unsigned idx0 = dc_att_idx((unsigned)(bh * nq + q), (unsigned)(tp * 32), nq_pad);      // even: whole hash pairs
        unsigned bits = 0u;
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
          const unsigned hw = dc_hash_pair(rg, key_site, (idx0 >> 1) + pr);
          bits |= (dc_keep_half(hw, 0u, thresh) ? 1u : 0u) << (2 * pr);
          bits |= (dc_keep_half(hw, 1u, thresh) ? 1u : 0u) << (2 * pr + 1);
        }
        Ms[wd] = bits;
      }
    }
  }
  {
#pragma unroll
    for (int t = 0; t < 6; ++t) *(VC*)(Ws + t * KS * WBLK + tid * 8) = wt[t];
    // value rows past the last row block that the paired P.V tiles still touch: zero (an odd number of row blocks)
    const int z0 = nrb * 16 * RS, zn = (KC * RS - z0) / 8;
    for (int c = tid; c < zn; c += MHA_PROJ_THREADS) *(VC*)(Vs + z0 + c * 8) = (VC){0, 0, 0, 0, 0, 0, 0, 0};
  }
  MP_MARK(1);
  __syncthreads();
  MP_MARK(2);
#pragma unroll
  for (int i = 0; i < RBW; ++i) {
    const int rb = wave + NW * i;                          // wave-uniform
    if (rb < nrb) {
      f32x4 acc[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 6; ++t) EB::mma(*(const VC*)(Ws + (t * KS + ks) * WBLK + lane * 8), t < 4 ? a1[i & 1][ks] : a2[i & 1][ks], acc[t]);
      if (i == 0 && rb + 2 * NW < nrb) load_rows(rb + 2 * NW, a1[0], a2[0]);
      const int row = rb * 16 + r16;
      const bool valid = row < nq;                         // rows past the group: zero rows in all three images
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        u16* img = t < 2 ? Qs : (t < 4 ? Ks : Vs);
        const V4 pk = EB::pack4(acc[t] + bias[t]);
        *(V4*)(img + row * RS + (t & 1) * 16 + kq * 4) = valid ? pk : (V4){0, 0, 0, 0};
      }
    }
  }
  MP_MARK(3);
  __syncthreads();
  MP_MARK(4);
  // the slots the backward reads (and the unfused attention entry points): 16 bytes per lane out of the images
  {
    const int per = nq * 4, total = 3 * per, iters = (total + MHA_PROJ_THREADS - 1) / MHA_PROJ_THREADS;
    for (int it = 0; it < iters; ++it) {
      const int c = it * MHA_PROJ_THREADS + tid;
      if (c < total) {
        const int which = c / per, rem = c - which * per, row = rem >> 2, part = rem & 3;
        const VC v = *(const VC*)((which == 0 ? Qs : (which == 1 ? Ks : Vs)) + row * RS + part * 8);
        u16* dst = which == 2 ? v_out + (base + row) * DC_C + h * DC_HD + part * 8
                              : qk_out + (base + row) * 512 + which * 256 + h * DC_HD + part * 8;
        *(VC*)dst = v;
      }
    }
    if (thresh) {                                          // the keep bits, for the backward kernels (same decisions, no second hash)
      const int words = nq * MhaProjLds::MW, wit = (words + MHA_PROJ_THREADS - 1) / MHA_PROJ_THREADS;
      for (int it = 0; it < wit; ++it) {
        const int wd = it * MHA_PROJ_THREADS + tid;
        if (wd < words) amask[(size_t)bh * words + wd] = Ms[wd];
      }
    }
  }
  MP_MARK(5);
  // ---- phase B: softmax(q k^T / sqrt(32)) v for every query block of the group (one chunk: no running-maximum rescale) ----
  for (int qb = wave; qb < nrb; qb += NW) {
    const int q = qb * 16 + r16;
    H::RowFrag qf;
    qf.c[0] = *(const VC*)(Qs + q * RS + kq * 8);
    f32x4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float m_run = -INFINITY, l_run = 0.f;
    const unsigned row = (unsigned)(bh * nq + q);
    unsigned kb[MhaProjLds::MW];
#pragma unroll
    for (int j = 0; j < MhaProjLds::MW; ++j) kb[j] = Ms[q * MhaProjLds::MW + j];
    mha_chunk<EB>(Ks, Vs, nullptr, 0, nq, qf, row, nq_pad, rg, key_site, thresh, inv_keep, scale_log2, r16, kq, m_run, l_run, oacc, kb);
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (q < nq) {
      const float inv = 1.f / l_run;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) *(V4*)(o + (base + q) * DC_C + h * DC_HD + dt * 16 + kq * 4) = EB::pack4(oacc[dt] * inv);
      if (kq == 0) lse[(base + q) * DC_NHEAD + h] = m_run * scale_log2 + log2f(l_run);
    }
    MP_MARK(6 + (qb >> 3));
  }
  MP_MARK(9);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_post
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(DC_THREADS) void k_dec_post(u3d_declayer_params P, u3d_declayer_dims dm, const float* __restrict__ x,
                                                         const float* __restrict__ ref, const typename E::T* __restrict__ value,
                                                         const unsigned long long* __restrict__ rng, DcPtrs<E> S, float* __restrict__ x_out,
                                                         typename E::T* __restrict__ xc_out, float* __restrict__ reg_out,
                                                         float* __restrict__ cls_out, float* __restrict__ iou_out) {
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef DcLds<E> L;
  constexpr int BM = E::BM, RPW = E::BM / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  T* A0 = (T*)(lds + L::A0);
  T* A1 = (T*)(lds + L::A1);
  T* A2 = (T*)(lds + L::A2);
  float* F = (float*)(lds + L::F);
  float* G = (float*)(lds + L::G);
  float* misc = (float*)(lds + L::MISC);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int wv = (wave + (int)(blockIdx.x >> 3)) & 3;      // which 64 output columns this wave computes: rotated per workgroup (see dc_gemm)
  const int row0 = blockIdx.x * BM, M = dm.m;
  dc_poison_lds<E>(lds, tid);
  DcDrop drop = {dc_rng_load(rng), dc_thresh(dm.p_drop), dc_inv_keep(dm.p_drop), dm.layer};

  DC_MARK(0);
  dc_load_a<E, 256, true>(A0, S.o, DC_C, row0, M, tid);           // the attention kernel writes m rows only: padded rows repeat row m-1 (finite)
  dc_load_f<E, false>(F, x, row0, M, tid);
  dc_load_a<E, 256, false>(A2, S.pos, DC_C, row0, M, tid);
  {                                           // reference-point logits of the block's rows: every lane of a row's wave reads them later
    const int row = (tid >> 3) & (BM - 1), j = tid & 7;    // BM rows x 8 slots, slots 3..7 unused (BM = 16: two threads store the same value)
    misc[row * DC_MISC_LD + j] = ref[(size_t)min(row0 + row, M - 1) * 3 + min(j, 2)];
  }
  __syncthreads();
  // residual stream += dropout(linear(A)) (the linear's output is a T tensor in the layer-by-layer formulation)
  auto add_to_F = [&](const float* bias, int site) {
    return [=](int row, int col, f32x4 v) {
      v = E::round4(v + dc_bias4(bias, col));
      v = drop.apply(v, site, (unsigned)((row0 + row) * DC_C + col));
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + v;
    };
  };
  DC_MARK(1);
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_OUTP], wv * 64, lane, add_to_F(P.b[U3D_DL_OUTP], 0));
  __syncthreads();
  DC_MARK(2);
  {
    DcLnOut<E> o = {F, A0, DC_C, nullptr, nullptr, S.mr, U3D_DLN_1, false, S.u1, nullptr};
    dc_layernorm<E>(F, P.ln_g[U3D_DLN_1], P.ln_b[U3D_DLN_1], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  DC_MARK(3);
  // cross "attention": gate = sigmoid(attention_weights(x1 + pos)), one trilinear sample of the value volume per query.
  // The 8 corner rows of GRP queries are requested together (8 * GRP row loads of 512 B in flight per wave) before any of them is
  // used: one query at a time, every query waited out a full miss latency (phase stamps: 82 k clocks = 21 % of the kernel).
  // Out-of-volume corners load row 0 with weight 0: no branch between the loads.
  {
    const f32x4 aw = *(const f32x4*)(P.attw_w + lane * 4);
    const float ab = P.attw_b[0];
    constexpr int GRP = 4;
    static_assert(RPW % GRP == 0, "rows per wave must split into gather groups");
#pragma unroll 1
    for (int r0g = 0; r0g < RPW; r0g += GRP) {
      V4 cv[GRP][8];
      float cw[GRP][8];
#ifndef DC_ABL_NOGATHER
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
        const int row = wave * RPW + r0g + g;
        DcCorners tc;
        dc_corners<E>(misc + row * DC_MISC_LD, min(row0 + row, M - 1) / dm.qps, dm.dz, dm.dy, dm.dx, tc);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int vr = max(tc.row[c], 0);                  // scalar
          cw[g][c] = tc.row[c] >= 0 ? tc.w[c] : 0.f;
          cv[g][c] = *(const V4*)(value + (size_t)vr * DC_C + lane * 4);
        }
      }
#endif
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
        const int row = wave * RPW + r0g + g;
        const size_t gr = (size_t)(row0 + row);
        const int ao = dc_aoff<E>(row, lane * 4, DC_C);
        const f32x4 x1 = E::unpack4(*(const V4*)(A0 + ao)), pp = E::unpack4(*(const V4*)(A2 + ao));
        const V4 qpb = E::pack4(x1 + pp);
        const f32x4 qp = E::unpack4(qpb);
        *(V4*)(S.qp + gr * DC_C + lane * 4) = qpb;
        const float wl = E::round(dc_wave_sum(qp[0] * aw[0] + qp[1] * aw[1] + qp[2] * aw[2] + qp[3] * aw[3]) + ab);
        const float gate = E::round(E::sigmoid(wl));
        S.mr[gr * 16 + 14] = wl;                     // all lanes, same value
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifndef DC_ABL_NOGATHER
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += cw[g][c] * E::unpack4(cv[g][c]);
#endif
        const V4 sb = E::pack4(acc);
        const V4 gb = E::pack4(E::unpack4(sb) * gate);
        *(V4*)(A1 + ao) = gb;
        *(V4*)(S.samp + gr * DC_C + lane * 4) = sb;
        *(V4*)(S.gated + gr * DC_C + lane * 4) = gb;
      }
    }
  }
  DC_MARK(4);
  // position encoder, first layer (3 -> 256): plain VALU into G
  {
    const int col = tid;
    const float w0 = E::round(P.pe0_w[col * 3 + 0]), w1 = E::round(P.pe0_w[col * 3 + 1]), w2 = E::round(P.pe0_w[col * 3 + 2]);
    const float b = P.pe0_b[col];
    for (int row = 0; row < BM; ++row) {
      const float* r3 = misc + row * DC_MISC_LD;
      G[row * DC_TS + col] = E::round(E::round(r3[0]) * w0 + E::round(r3[1]) * w1 + E::round(r3[2]) * w2 + b);
    }
  }
  __syncthreads();
  DC_MARK(5);
  dc_linear<E, 256, 4>(A1, (const T*)P.w[U3D_DL_OPROJ], wv * 64, lane, add_to_F(P.b[U3D_DL_OPROJ], 1));
  {
    DcLnOut<E> o = {nullptr, A0, DC_C, nullptr, S.peh0, S.mr, U3D_DLN_PE0, false, nullptr, nullptr};
    dc_layernorm<E>(G, P.ln_g[U3D_DLN_PE0], P.ln_b[U3D_DLN_PE0], dm.ln_eps, true, o, row0, wave, lane);   // overwrites A0 (x1: no longer needed)
  }
  __syncthreads();
  DC_MARK(6);
  {
    const float* bias = P.b[U3D_DL_PE1];
    dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_PE1], wv * 64, lane, [=](int row, int col, f32x4 v) {
      *(f32x4*)(G + row * DC_TS + col) = E::round4(v + dc_bias4(bias, col));       // saved by the LayerNorm below (its input rows)
    });
  }
  __syncthreads();
  DC_MARK(7);
  {
    DcLnOut<E> o = {G, nullptr, 0, nullptr, nullptr, S.mr, U3D_DLN_PE1, true, nullptr, S.upe1};
    dc_layernorm<E>(G, P.ln_g[U3D_DLN_PE1], P.ln_b[U3D_DLN_PE1], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  DC_FOR_TID(c, BM * 64) {                                   // x2 (pre-norm) = x1 + cross output + position feature
    const int o = (c >> 6) * DC_TS + (c & 63) * 4;
    *(f32x4*)(F + o) = *(const f32x4*)(F + o) + *(const f32x4*)(G + o);
  }
  __syncthreads();
  DC_MARK(8);
  {
    DcLnOut<E> o = {F, A0, DC_C, nullptr, S.x2c, S.mr, U3D_DLN_2, false, S.u2, nullptr};
    dc_layernorm<E>(F, P.ln_g[U3D_DLN_2], P.ln_b[U3D_DLN_2], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  DC_MARK(9);
  // FFN
  {
    const float* bias = P.b[U3D_DL_FFN0];
    auto ffh = [=](int row, int col, f32x4 v) {
      v = E::round4(dc_relu4(v + dc_bias4(bias, col)));
      const V4 hb = E::pack4(drop.apply(v, 2, (unsigned)((row0 + row) * DC_FF + col)));
      *(V4*)(A1 + dc_aoff<E>(row, col, DC_FF)) = hb;
    };
    dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_FFN0], wv * 64, lane, ffh);
    dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_FFN0], 256 + wv * 64, lane, ffh);
  }
  __syncthreads();
  DC_MARK(10);
  dc_store_a<E, 512>(A1, S.ffh, DC_FF, row0, tid);
  dc_linear<E, 512, 4>(A1, (const T*)P.w[U3D_DL_FFN1], wv * 64, lane, add_to_F(P.b[U3D_DL_FFN1], 3));
  __syncthreads();
  DC_MARK(11);
  {
    DcLnOut<E> o = {nullptr, A2, DC_C, x_out, xc_out, S.mr, U3D_DLN_3, false, S.u3, nullptr};
    dc_layernorm<E>(F, P.ln_g[U3D_DLN_3], P.ln_b[U3D_DLN_3], dm.ln_eps, false, o, row0, wave, lane);
  }
  __syncthreads();
  DC_MARK(12);
  // branches on the layer state x3 (A2)
  auto relu_to = [&](T* dst, const float* bias) {         // the slot copy leaves from the tile after the barrier (see k_dec_pre)
    return [=](int row, int col, f32x4 v) {
      *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(dc_relu4(v + dc_bias4(bias, col)));
    };
  };
  auto narrow_out = [&](float* dst, int n, const float* bias) {     // final layer of a branch: n <= 32 real columns of the 64 computed
    return [=](int row, int col, f32x4 v) {       // n is uniform: the column test is the only lane-dependent branch (stores only)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < n) dst[(size_t)(row0 + row) * n + col + r] = E::round(v[r] + bias[col + r]);
    };
  };
  T* A1b = A1 + BM * DC_C;
#ifdef DC_PHASE_TIMING
  {     // one linear taken apart: burst + MFMAs | epilogue | barrier
    DC_MARK(20);
    f32x4 acc[E::MT][4];
#pragma unroll
    for (int mt = 0; mt < E::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dc_gemm<E, 256, 4>(A2, (const T*)P.w[U3D_DL_REG0] + (size_t)(wv * 64) * 256, acc, lane);
    __builtin_amdgcn_sched_barrier(0);
    DC_MARK(21);
    auto epi = relu_to(A0, P.b[U3D_DL_REG0]);
#pragma unroll
    for (int mt = 0; mt < E::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) epi(mt * 16 + (lane & 15), wv * 64 + nt * 16 + (lane >> 4) * 4, acc[mt][nt]);
    __builtin_amdgcn_sched_barrier(0);
    DC_MARK(22);
    __syncthreads();
    DC_MARK(23);
  }
#else
  dc_linear<E, 256, 4>(A2, (const T*)P.w[U3D_DL_REG0], wv * 64, lane, relu_to(A0, P.b[U3D_DL_REG0]));
  __syncthreads();
#endif
  dc_store_a<E, 256>(A0, S.r1, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_REG1], wv * 64, lane, relu_to(A1, P.b[U3D_DL_REG1]));
  __syncthreads();
  dc_store_a<E, 256>(A1, S.r2, DC_C, row0, tid);
  dc_linear<E, 256, 1>(A1, (const T*)P.w[U3D_DL_REG2], wv * 16, lane, narrow_out(reg_out, dm.code, P.b[U3D_DL_REG2]));
  dc_linear<E, 256, 4>(A2, (const T*)P.w[U3D_DL_IOU0], wv * 64, lane, relu_to(A0, P.b[U3D_DL_IOU0]));
  __syncthreads();
  dc_store_a<E, 256>(A0, S.i1, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_IOU1], wv * 64, lane, relu_to(A1b, P.b[U3D_DL_IOU1]));
  __syncthreads();
  dc_store_a<E, 256>(A1b, S.i2, DC_C, row0, tid);
  dc_linear<E, 256, 1>(A1b, (const T*)P.w[U3D_DL_IOU2], wv * 16, lane, narrow_out(iou_out, 1, P.b[U3D_DL_IOU2]));
  // cls: Linear -> LN -> ReLU twice, then the class logits (each linear's output is saved by the LayerNorm that reads it)
  auto to_G = [&](const float* bias) {
    return [=](int row, int col, f32x4 v) { *(f32x4*)(G + row * DC_TS + col) = E::round4(v + dc_bias4(bias, col)); };
  };
  dc_linear<E, 256, 4>(A2, (const T*)P.w[U3D_DL_CLS0], wv * 64, lane, to_G(P.b[U3D_DL_CLS0]));
  __syncthreads();
  {
    DcLnOut<E> o = {nullptr, A0, DC_C, nullptr, S.c1, S.mr, U3D_DLN_C1, false, nullptr, S.uc1};
    dc_layernorm<E>(G, P.ln_g[U3D_DLN_C1], P.ln_b[U3D_DLN_C1], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  dc_linear<E, 256, 4>(A0, (const T*)P.w[U3D_DL_CLS1], wv * 64, lane, to_G(P.b[U3D_DL_CLS1]));
  __syncthreads();
  {
    DcLnOut<E> o = {nullptr, A1, DC_C, nullptr, S.c2, S.mr, U3D_DLN_C2, false, nullptr, S.uc2};
    dc_layernorm<E>(G, P.ln_g[U3D_DLN_C2], P.ln_b[U3D_DLN_C2], dm.ln_eps, true, o, row0, wave, lane);
  }
  __syncthreads();
  dc_linear<E, 256, 1>(A1, (const T*)P.w[U3D_DL_CLS2], wv * 16, lane, narrow_out(cls_out, dm.ncls, P.b[U3D_DL_CLS2]));
  DC_MARK(14);
}

static int32_t dc_check(const u3d_declayer_params* p, const u3d_declayer_dims* d) {
  U3D_REQUIRE(p && d, U3D_ERR_ARG);
  U3D_REQUIRE(d->m > 0 && d->nq > 0 && d->qps > 0 && d->qps % d->nq == 0 && d->m % d->qps == 0 && d->batch == d->m / d->qps, U3D_ERR_ARG);
  U3D_REQUIRE(d->ncls > 0 && d->ncls <= 32 && d->code > 0 && d->code <= 32, U3D_ERR_UNSUPPORTED);
  U3D_REQUIRE(d->p_attn >= 0.f && d->p_attn < 1.f && d->p_drop >= 0.f && d->p_drop < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE(d->dtype == U3D_BF16 || d->dtype == U3D_F32, U3D_ERR_ARG);
  U3D_REQUIRE((long long)d->m * DC_FF < (1ll << 32), U3D_ERR_UNSUPPORTED);
  for (int i = 0; i < U3D_DL_NLIN; ++i) {
    if (!d->has_qs && (i == U3D_DL_QS0 || i == U3D_DL_QS1 || i == U3D_DL_QS2)) continue;
    U3D_REQUIRE(p->w[i] && p->b[i], U3D_ERR_ARG);
  }
  for (int i = 0; i < U3D_DL_NLN; ++i) U3D_REQUIRE(p->ln_g[i] && p->ln_b[i], U3D_ERR_ARG);
  U3D_REQUIRE(p->attw_w && p->attw_b && p->pe0_w && p->pe0_b && p->dim_t, U3D_ERR_ARG);
  return U3D_OK;
}

template <typename E>
static int32_t dc_layer_fwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc, const float* ref,
                            const void* value, const uint64_t* rng, float* x_out, void* xc_out, float* reg_out, float* cls_out,
                            float* iou_out, void* save, u3d_stream s) {
  typedef typename E::T T;
  const DcPtrs<E> S = dc_resolve<E>(save, d->m);
  const int nb = u3d_decoder_layer_blocks_dt(d->m, E::DT);
  U3D_ALLOW_LDS(k_dec_pre<E>, DcLds<E>::BYTES);
  U3D_ALLOW_LDS(k_dec_post<E>, DcLds<E>::BYTES);
  // bf16, groups of at most one LDS chunk of queries: in-projection + attention in ONE launch per layer (k_mha_proj_fwd)
  const bool fused = mha_fused_inproj(E::DT, d->nq);
  hipLaunchKernelGGL(k_dec_pre<E>, dim3(nb), dim3(DC_THREADS), DcLds<E>::BYTES, s, *p, *d, (const T*)xc, ref, S, fused ? 1 : 0);
  if (fused) {
    U3D_ALLOW_LDS(k_mha_proj_fwd, MhaProjLds::BYTES);
    const float scale_log2 = 1.4426950408889634f / sqrtf((float)DC_HD);
    hipLaunchKernelGGL(k_mha_proj_fwd, dim3((d->m / d->nq) * DC_NHEAD), dim3(MHA_PROJ_THREADS), MhaProjLds::BYTES, s, (const u16*)S.qkin,
                       (const u16*)xc, (const u16*)p->w[U3D_DL_INQK], p->b[U3D_DL_INQK], (const u16*)p->w[U3D_DL_INV], p->b[U3D_DL_INV],
                       d->nq, scale_log2, dc_thresh(d->p_attn), dc_inv_keep(d->p_attn), d->layer, (const unsigned long long*)rng,
                       (u16*)S.qk, (u16*)S.v, (u16*)S.o, S.lse, S.amask);
  } else {
    int32_t rc = u3d_mha_fwd_dt(S.qk, S.v, d->m, d->nq, d->p_attn, d->layer, rng, S.o, S.lse, E::DT, s);
    if (rc != U3D_OK) return rc;
  }
  hipLaunchKernelGGL(k_dec_post<E>, dim3(nb), dim3(DC_THREADS), DcLds<E>::BYTES, s, *p, *d, x, ref, (const T*)value,
                     (const unsigned long long*)rng, S, x_out, (T*)xc_out, reg_out, cls_out, iou_out);
  U3D_CHECK_LAUNCH();
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
  u3d_decoder_layer_slots_dt(d->m, d->ncls, d->code, d->dtype, off, nullptr);
  U3D_REQUIRE(save_bytes >= off[U3D_DS_COUNT], U3D_ERR_WORKSPACE);
  if (d->dtype == U3D_BF16) return dc_layer_fwd<EB>(p, d, x, xc, ref, value, rng, x_out, xc_out, reg_out, cls_out, iou_out, save, s);
  return dc_layer_fwd<EF>(p, d, x, xc, ref, value, rng, x_out, xc_out, reg_out, cls_out, iou_out, save, s);
}
