// Fused decoder layer, backward (see decoder.hip / include/u3d_hip.h).  Four launches:
//   k_dec_post_bwd   rows: branches -> LN3 -> FFN -> LN2 -> position encoder / output_proj / gate / trilinear scatter -> LN1 -> out-proj
//   k_mha_bwd_dq     (group, head, 64 queries): dQ
//   k_mha_bwd_dkv    (group, head, 64 keys): dK, dV
//   k_dec_pre_bwd    rows: in-projection -> query_scale / ref_point_head chains -> gradient w.r.t. the layer input
// Input gradients of every linear are GEMMs against the transposed bf16 weight copies (wt[.] = [K][N]); the dY of every linear is
// left in the gradient workspace (bf16) for the caller's batched weight-gradient pass, LayerNorm (dgamma, dbeta) as per-workgroup
// partial sums.
#define DC_PF 4      /* weight-prefetch burst (k-steps): the backward row kernels carry more live state per lane than the forward ones */
#include "decoder_common.h"

#ifdef DC_PHASE_TIMING       /* tools/dec_bench.py: shader-clock stamps of ONE workgroup at the phase boundaries of k_dec_post_bwd */
__device__ unsigned long long dc_dbg_bwd[64];
#define DC_MARK(id) do { if (blockIdx.x == 7 && threadIdx.x == 0) dc_dbg_bwd[id] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t u3d_debug_bwd_times(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dc_dbg_bwd), 64 * 8) == hipSuccess ? 0 : -1; }
#else
#define DC_MARK(id)
#endif

template <typename E>
struct DcSave {            // forward-save slots (read-only here)
  typedef typename E::T T;
  const T *sine, *rph1, *rph2, *raw, *qs1, *qs2, *qs, *pos, *qkin, *qk, *v;
  const float* lse;
  const T* o;
  const float *u1, *mr;
  const T *qp, *samp, *gated, *peh0, *upe1;
  const float* u2;
  const T *x2c, *ffh;
  const float* u3;
  const T *r1, *r2, *i1, *i2, *uc1, *c1, *uc2, *c2;
  const unsigned* amask;
};
template <typename E>
struct DcGrad {            // gradient-workspace slots
  typedef typename E::T T;
  T *clso, *c2u, *c1u, *iouo, *i2, *i1, *rego, *r2, *r1, *f, *ffh, *out, *upe1, *p0, *wl, *o2, *d_o, *dqk, *dv, *qs, *qs2, *qs1, *raw,
      *rph2, *rph1;
  float *lnp, *du1;
  T *dposa, *sine;
  int nb;                  // workgroups = rows of each LayerNorm partial matrix
};

extern "C" int32_t u3d_decoder_layer_slots_dt(int32_t m, int32_t ncls, int32_t code, int32_t dtype, int64_t* save_off, int64_t* grad_off);
extern "C" int32_t u3d_decoder_layer_blocks_dt(int32_t m, int32_t dtype);

template <typename E>
static DcSave<E> dcb_resolve_save(const void* save, int m) {
  typedef typename E::T T;
  int64_t off[U3D_DS_COUNT + 1];
  u3d_decoder_layer_slots_dt(m, 1, 1, E::DT, off, nullptr);
  const char* b = (const char*)save;
  DcSave<E> p;
  p.sine = (const T*)(b + off[U3D_DS_SINE]); p.rph1 = (const T*)(b + off[U3D_DS_RPH1]); p.rph2 = (const T*)(b + off[U3D_DS_RPH2]);
  p.raw = (const T*)(b + off[U3D_DS_RAW]); p.qs1 = (const T*)(b + off[U3D_DS_QS1]); p.qs2 = (const T*)(b + off[U3D_DS_QS2]);
  p.qs = (const T*)(b + off[U3D_DS_QS]); p.pos = (const T*)(b + off[U3D_DS_POS]); p.qkin = (const T*)(b + off[U3D_DS_QKIN]);
  p.qk = (const T*)(b + off[U3D_DS_QK]); p.v = (const T*)(b + off[U3D_DS_V]); p.lse = (const float*)(b + off[U3D_DS_LSE]);
  p.o = (const T*)(b + off[U3D_DS_O]); p.u1 = (const float*)(b + off[U3D_DS_U1]); p.mr = (const float*)(b + off[U3D_DS_MR]);
  p.qp = (const T*)(b + off[U3D_DS_QP]); p.samp = (const T*)(b + off[U3D_DS_SAMP]); p.gated = (const T*)(b + off[U3D_DS_GATED]);
  p.peh0 = (const T*)(b + off[U3D_DS_PEH0]); p.upe1 = (const T*)(b + off[U3D_DS_UPE1]); p.u2 = (const float*)(b + off[U3D_DS_U2]);
  p.x2c = (const T*)(b + off[U3D_DS_X2C]); p.ffh = (const T*)(b + off[U3D_DS_FFH]); p.u3 = (const float*)(b + off[U3D_DS_U3]);
  p.r1 = (const T*)(b + off[U3D_DS_R1]); p.r2 = (const T*)(b + off[U3D_DS_R2]); p.i1 = (const T*)(b + off[U3D_DS_I1]);
  p.i2 = (const T*)(b + off[U3D_DS_I2]); p.uc1 = (const T*)(b + off[U3D_DS_UC1]); p.c1 = (const T*)(b + off[U3D_DS_C1]);
  p.uc2 = (const T*)(b + off[U3D_DS_UC2]); p.c2 = (const T*)(b + off[U3D_DS_C2]);
  p.amask = (const unsigned*)(b + off[U3D_DS_AMASK]);
  return p;
}
template <typename E>
static DcGrad<E> dcb_resolve_grad(void* grad, int m, int ncls, int code, int64_t* total) {
  typedef typename E::T T;
  int64_t off[U3D_DG_COUNT + 1];
  u3d_decoder_layer_slots_dt(m, ncls, code, E::DT, nullptr, off);
  char* b = (char*)grad;
  DcGrad<E> g;
  g.clso = (T*)(b + off[U3D_DG_CLSO]); g.c2u = (T*)(b + off[U3D_DG_C2U]); g.c1u = (T*)(b + off[U3D_DG_C1U]);
  g.iouo = (T*)(b + off[U3D_DG_IOUO]); g.i2 = (T*)(b + off[U3D_DG_I2]); g.i1 = (T*)(b + off[U3D_DG_I1]);
  g.rego = (T*)(b + off[U3D_DG_REGO]); g.r2 = (T*)(b + off[U3D_DG_R2]); g.r1 = (T*)(b + off[U3D_DG_R1]);
  g.f = (T*)(b + off[U3D_DG_F]); g.ffh = (T*)(b + off[U3D_DG_FFH]); g.out = (T*)(b + off[U3D_DG_OUT]);
  g.upe1 = (T*)(b + off[U3D_DG_UPE1]); g.p0 = (T*)(b + off[U3D_DG_P0]); g.wl = (T*)(b + off[U3D_DG_WL]);
  g.o2 = (T*)(b + off[U3D_DG_O2]); g.d_o = (T*)(b + off[U3D_DG_DO]); g.dqk = (T*)(b + off[U3D_DG_DQK]);
  g.dv = (T*)(b + off[U3D_DG_DV]); g.qs = (T*)(b + off[U3D_DG_QS]); g.qs2 = (T*)(b + off[U3D_DG_QS2]);
  g.qs1 = (T*)(b + off[U3D_DG_QS1]); g.raw = (T*)(b + off[U3D_DG_RAW]); g.rph2 = (T*)(b + off[U3D_DG_RPH2]);
  g.rph1 = (T*)(b + off[U3D_DG_RPH1]); g.lnp = (float*)(b + off[U3D_DG_LNP]); g.du1 = (float*)(b + off[U3D_DG_DU1]);
  g.dposa = (T*)(b + off[U3D_DG_DPOSA]); g.sine = (T*)(b + off[U3D_DG_SINE]);
  g.nb = u3d_decoder_layer_blocks_dt(m, E::DT);
  if (total) *total = off[U3D_DG_COUNT];
  return g;
}

// ---- LayerNorm backward over an f32 tile of incoming gradients --------------------------------------------------------------
// dy: tile T (rows RPW*w .. per wave); u(row) -> the forward input row (4 columns per lane); writes du to `tile_out` (f32) and/or the
// activation tile `a_out` (T) + global T `g16`; accumulates (dgamma, dbeta) of this workgroup into lnp[ln][0|1][block][256].
template <typename E>
struct DcLnBwdOut { float* tile; typename E::T* a; typename E::T* g16; };
template <typename E, typename LoadU>
__device__ __forceinline__ void dc_layernorm_bwd(const float* T, LoadU load_u, const float* __restrict__ mr, int mr_idx,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, bool relu,
                                                 const DcLnBwdOut<E>& o, float* __restrict__ lnp, int ln, int nb, float* red /* LDS [4][2][256] */,
                                                 int row0, int wave, int lane, int tid) {
  typedef typename E::V4 V4;
  constexpr int RPW = E::BM / 4;
  const f32x4 ga = *(const f32x4*)(gamma + lane * 4), be = *(const f32x4*)(beta + lane * 4);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr;
    const int gr = row0 + row;                 // padded slots: rows past m carry zero gradients and finite saved values
    const float mu = mr[(size_t)gr * 16 + 2 * mr_idx], rs = mr[(size_t)gr * 16 + 2 * mr_idx + 1];
    const f32x4 u = load_u(row, gr);
    f32x4 dy = *(const f32x4*)(T + row * DC_TS + lane * 4);
    f32x4 xh, dxh;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = (u[j] - mu) * rs;
      if (relu && !(xh[j] * ga[j] + be[j] > 0.f)) dy[j] = 0.f;
      dxh[j] = dy[j] * ga[j];
      s1 += dxh[j];
      s2 += dxh[j] * xh[j];
      dg[j] += dy[j] * xh[j];
      db[j] += dy[j];
    }
    s1 = dc_wave_sum(s1) * (1.f / DC_C);
    s2 = dc_wave_sum(s2) * (1.f / DC_C);
    f32x4 du;
#pragma unroll
    for (int j = 0; j < 4; ++j) du[j] = rs * (dxh[j] - s1 - xh[j] * s2);
    if (o.tile) *(f32x4*)(o.tile + row * DC_TS + lane * 4) = du;
    const V4 dub = E::pack4(du);
    if (o.a) *(V4*)(o.a + dc_aoff<E>(row, lane * 4, DC_C)) = dub;
    if (o.g16) *(V4*)(o.g16 + (size_t)gr * DC_C + lane * 4) = dub;
  }
  *(f32x4*)(red + (wave * 2 + 0) * DC_C + lane * 4) = dg;
  *(f32x4*)(red + (wave * 2 + 1) * DC_C + lane * 4) = db;
  __syncthreads();
  DC_FOR_TID(i, 2 * DC_C) {
    const int which = i >> 8, col = i & 255;
    const float v = red[(0 * 2 + which) * DC_C + col] + red[(1 * 2 + which) * DC_C + col] + red[(2 * 2 + which) * DC_C + col] +
                    red[(3 * 2 + which) * DC_C + col];
    lnp[((size_t)(ln * 2 + which) * nb + blockIdx.x) * DC_C + col] = v;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_post_bwd
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(DC_THREADS) void k_dec_post_bwd(u3d_declayer_params P, u3d_declayer_dims dm, const float* __restrict__ ref,
                                                             const typename E::T* __restrict__ value,
                                                             const unsigned long long* __restrict__ rng, DcSave<E> S, DcGrad<E> Gd,
                                                             const float* __restrict__ dx_out, const float* __restrict__ dreg,
                                                             const float* __restrict__ dcls, const float* __restrict__ diou,
                                                             void* __restrict__ dvalue_v, float* __restrict__ dref) {
  float* const dvalue = (float*)dvalue_v;
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
  float* red = (float*)(A1 + BM * DC_C);               // LayerNorm partial reduce scratch: second half of A1 (free whenever it is used)
  static_assert(BM * DC_C * sizeof(T) >= 4 * 2 * DC_C * 4, "LayerNorm reduce scratch must fit the second half of A1");
  // narrow output gradients [BM][32] (zero-padded) as an MFMA operand: rows of NLD elements in the A2 region (unused otherwise here);
  // the 80-byte (bf16) / 160-byte (f32) row stride spreads the 16 rows of a fragment read over all banks
  constexpr int NLD = 40;
  T* Dn = A2;
  static_assert(BM * NLD <= BM * DC_C, "narrow-gradient tile must fit the A2 region");
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int wv = (wave + (int)(blockIdx.x >> 3)) & 3;      // which 64 output columns this wave computes: rotated per workgroup (see dc_gemm)
  const int row0 = blockIdx.x * BM, M = dm.m, nb = Gd.nb;
  dc_poison_lds<E>(lds, tid);
  DcDrop drop = {dc_rng_load(rng), dc_thresh(dm.p_drop), dc_inv_keep(dm.p_drop), dm.layer};

  DC_MARK(0);
  dc_load_f<E, true>(F, dx_out, row0, M, tid);         // F = gradient w.r.t. the layer output x3 (zero when dx_out is null, zero past row m)
  {
    const int row = (tid >> 3) & (BM - 1), j = tid & 7;      // reference-point logits, slots 3..7 unused
    misc[row * DC_MISC_LD + j] = ref[(size_t)min(row0 + row, M - 1) * 3 + min(j, 2)];
  }
  // narrow gradient [BM][n] (f32, m rows) -> the Dn tile (T, zero-padded to 32 columns), plus its T copy (padded rows) for the
  // caller's weight gradient
  auto load_narrow = [&](const float* src, int n, T* gcopy) {
    DC_FOR_TID(i, BM * 32) {
      const int row = i >> 5, j = i & 31;
      const float raw = src[(size_t)min(row0 + row, M - 1) * n + min(j, n - 1)];      // always in range: selects below, no branch
      const float vr = row0 + row < M ? raw : 0.f;                                     // value of column min(j, n-1)
      gcopy[(size_t)(row0 + row) * n + min(j, n - 1)] = E::from_f(vr);                 // lanes j >= n repeat column n-1's store
      Dn[row * NLD + j] = E::from_f(j < n ? vr : 0.f);
    }
  };
  // dX [BM][256] = dY [BM][32] . W [32][256] for a final branch layer on the matrix pipe (one 32-deep reduction: wt = W^T in fragment
  // order, 16 tiles of one / two k-steps), masked by the saved ReLU output `mask_src` when given; result (T) -> activation tile `dst`
  // (its slot copy leaves from the tile after the barrier), or -> tile G (f32) when dst is null.
  // (As plain VALU code - 32 x 32 multiply-adds per thread and row-at-a-time 2-byte stores - this step cost 25-30 k clocks, a
  // quarter of each branch's backward: tools/dec_bench.py phase stamps.)
  auto narrow_dgrad = [&](const T* wt, const T* mask_src, T* dst) {
    typedef typename E::VC VC;
    constexpr int KSN = 32 / E::KSTEP, WBLK = 64 * E::CH;
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 acc[E::MT][4];
#pragma unroll
    for (int mt = 0; mt < E::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      VC a[E::MT];
#pragma unroll
      for (int mt = 0; mt < E::MT; ++mt) a[mt] = *(const VC*)(Dn + (mt * 16 + r16) * NLD + (ks * 4 + kq) * E::CH);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const VC wf = *(const VC*)(wt + (size_t)((wv * 4 + nt) * KSN + ks) * WBLK + lane * E::CH);
#pragma unroll
        for (int mt = 0; mt < E::MT; ++mt) E::mma(wf, a[mt], acc[mt][nt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < E::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = mt * 16 + r16, col = wv * 64 + nt * 16 + kq * 4;
        f32x4 v = acc[mt][nt];
        if (mask_src) {
          const V4 y = *(const V4*)(mask_src + (size_t)(row0 + row) * DC_C + col);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = E::to_f(y[r]) > 0.f ? v[r] : 0.f;
        }
        if (dst) *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(v);
        else *(f32x4*)(G + row * DC_TS + col) = E::round4(v);
      }
  };
  // dgrad GEMM epilogues
  auto masked_to = [&](T* dst, const T* mask_src) {          // (dY W) * [saved ReLU output > 0] -> tile (slot copy: dc_store_a)
    return [=](int row, int col, f32x4 v) {
      const V4 y = *(const V4*)(mask_src + (size_t)(row0 + row) * DC_C + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = E::to_f(y[r]) > 0.f ? v[r] : 0.f;
      *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(v);
    };
  };
  auto acc_to_F = [&]() {
    return [=](int row, int col, f32x4 v) {
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + E::round4(v);
    };
  };
  auto to_G = [&]() {
    return [=](int row, int col, f32x4 v) { *(f32x4*)(G + row * DC_TS + col) = E::round4(v); };
  };
  auto u_elem = [&](const T* src) {
    return [=](int row, int gr) { return E::unpack4(*(const V4*)(src + (size_t)gr * DC_C + lane * 4)); };
  };
  auto u_f32 = [&](const float* src) {
    return [=](int row, int gr) { return *(const f32x4*)(src + (size_t)gr * DC_C + lane * 4); };
  };

  DC_MARK(1);
  // ---- cls branch --------------------------------------------------------------------------------------------------------
  load_narrow(dcls, dm.ncls, Gd.clso);
  __syncthreads();
  narrow_dgrad((const T*)P.wt[U3D_DL_CLS2], nullptr, nullptr);          // -> G
  __syncthreads();
  {
    DcLnBwdOut<E> o = {nullptr, A0, Gd.c2u};
    dc_layernorm_bwd<E>(G, u_elem(S.uc2), S.mr, U3D_DLN_C2, P.ln_g[U3D_DLN_C2], P.ln_b[U3D_DLN_C2], true, o, Gd.lnp, U3D_DLN_C2, nb, red, row0,
                        wave, lane, tid);
  }
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_CLS1], wv * 64, lane, to_G());
  __syncthreads();
  {
    DcLnBwdOut<E> o = {nullptr, A0, Gd.c1u};
    dc_layernorm_bwd<E>(G, u_elem(S.uc1), S.mr, U3D_DLN_C1, P.ln_g[U3D_DLN_C1], P.ln_b[U3D_DLN_C1], true, o, Gd.lnp, U3D_DLN_C1, nb, red, row0,
                        wave, lane, tid);
  }
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_CLS0], wv * 64, lane, acc_to_F());
  __syncthreads();
  DC_MARK(2);
  // ---- iou branch ----------------------------------------------------------------------------------------------------------
  load_narrow(diou, 1, Gd.iouo);
  __syncthreads();
  narrow_dgrad((const T*)P.wt[U3D_DL_IOU2], S.i2, A0);
  __syncthreads();
  dc_store_a<E, 256>(A0, Gd.i2, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_IOU1], wv * 64, lane, masked_to(A1, S.i1));
  __syncthreads();
  dc_store_a<E, 256>(A1, Gd.i1, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A1, (const T*)P.wt[U3D_DL_IOU0], wv * 64, lane, acc_to_F());
  __syncthreads();
  DC_MARK(3);
  // ---- reg branch ----------------------------------------------------------------------------------------------------------
  load_narrow(dreg, dm.code, Gd.rego);
  __syncthreads();
  narrow_dgrad((const T*)P.wt[U3D_DL_REG2], S.r2, A0);
  __syncthreads();
  dc_store_a<E, 256>(A0, Gd.r2, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_REG1], wv * 64, lane, masked_to(A1, S.r1));
  __syncthreads();
  dc_store_a<E, 256>(A1, Gd.r1, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A1, (const T*)P.wt[U3D_DL_REG0], wv * 64, lane, acc_to_F());
  __syncthreads();
  DC_MARK(4);
  // ---- LN3 -> du3 (F) -------------------------------------------------------------------------------------------------------
  {
    DcLnBwdOut<E> o = {F, nullptr, nullptr};
    dc_layernorm_bwd<E>(F, u_f32(S.u3), S.mr, U3D_DLN_3, P.ln_g[U3D_DLN_3], P.ln_b[U3D_DLN_3], false, o, Gd.lnp, U3D_DLN_3, nb, red, row0,
                        wave, lane, tid);
  }
  DC_MARK(5);
  // ---- FFN -----------------------------------------------------------------------------------------------------------------
  // dY of a residual branch: dropout mask of the forward applied to the residual-stream gradient, T -> tile + slot
  auto branch_grad = [&](int site, T* dst, T* gslot) {
    DC_FOR_TID(c, BM * 64) {
      const int row = c >> 6, col = (c & 63) * 4;
      f32x4 v = *(const f32x4*)(F + row * DC_TS + col);
      v = drop.apply(v, site, (unsigned)((row0 + row) * DC_C + col));
      const V4 o = E::pack4(v);
      *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = o;
      *(V4*)(gslot + (size_t)(row0 + row) * DC_C + col) = o;           // whole rows, 8 / 16 bytes per lane: coalesced as it is
    }
  };
  branch_grad(3, A0, Gd.f);
  __syncthreads();
  {
    const float ik = drop.inv_keep;
    auto dh = [=](int row, int col, f32x4 v) {
      const V4 y = *(const V4*)(S.ffh + (size_t)(row0 + row) * DC_FF + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = E::to_f(y[r]) > 0.f ? E::round(v[r]) * ik : 0.f;     // kept & positive <=> saved value > 0
      const V4 o = E::pack4(v);
      *(V4*)(A1 + dc_aoff<E>(row, col, DC_FF)) = o;
    };
    dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_FFN1], wv * 64, lane, dh);
    dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_FFN1], 256 + wv * 64, lane, dh);
  }
  __syncthreads();
  dc_store_a<E, 512>(A1, Gd.ffh, DC_FF, row0, tid);
  dc_linear<E, 512, 4>(A1, (const T*)P.wt[U3D_DL_FFN0], wv * 64, lane, acc_to_F());
  __syncthreads();
  DC_MARK(6);
  // ---- LN2 -> du2 (F) -------------------------------------------------------------------------------------------------------
  {
    DcLnBwdOut<E> o = {F, nullptr, nullptr};
    dc_layernorm_bwd<E>(F, u_f32(S.u2), S.mr, U3D_DLN_2, P.ln_g[U3D_DLN_2], P.ln_b[U3D_DLN_2], false, o, Gd.lnp, U3D_DLN_2, nb, red, row0,
                        wave, lane, tid);
  }
  DC_MARK(7);
  // ---- position encoder ---------------------------------------------------------------------------------------------------
  DC_FOR_TID(c, BM * 64) {                                       // its output was a T tensor: the gradient arrives rounded
    const int o = (c >> 6) * DC_TS + (c & 63) * 4;
    *(f32x4*)(G + o) = E::round4(*(const f32x4*)(F + o));
  }
  __syncthreads();
  {
    DcLnBwdOut<E> o = {nullptr, A0, Gd.upe1};
    dc_layernorm_bwd<E>(G, u_elem(S.upe1), S.mr, U3D_DLN_PE1, P.ln_g[U3D_DLN_PE1], P.ln_b[U3D_DLN_PE1], true, o, Gd.lnp, U3D_DLN_PE1, nb, red,
                        row0, wave, lane, tid);
  }
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_PE1], wv * 64, lane, to_G());
  __syncthreads();
  {
    // the first position-encoder layer's output is recomputed from the reference point (3 -> 256)
    const f32x4 b4 = *(const f32x4*)(P.pe0_b + lane * 4);
    float w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) w[j][k] = E::round(P.pe0_w[(lane * 4 + j) * 3 + k]);
    auto u_pe0 = [=](int row, int gr) {
      const float* r3 = misc + row * DC_MISC_LD;
      const float a = E::round(r3[0]), b = E::round(r3[1]), c = E::round(r3[2]);
      f32x4 u;
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = E::round(a * w[j][0] + b * w[j][1] + c * w[j][2] + b4[j]);
      return u;
    };
    DcLnBwdOut<E> o = {G, nullptr, Gd.p0};
    dc_layernorm_bwd<E>(G, u_pe0, S.mr, U3D_DLN_PE0, P.ln_g[U3D_DLN_PE0], P.ln_b[U3D_DLN_PE0], true, o, Gd.lnp, U3D_DLN_PE0, nb, red, row0,
                        wave, lane, tid);
    if (dm.need_dref) {          // gradient w.r.t. the reference-point logits through Linear(3, 256)
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        const f32x4 d = E::round4(*(const f32x4*)(G + row * DC_TS + lane * 4));
        float g3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) g3[k] = dc_wave_sum(d[0] * w[0][k] + d[1] * w[1][k] + d[2] * w[2][k] + d[3] * w[3][k]);
        misc[row * DC_MISC_LD + 8 + (lane & 3)] = (lane & 3) == 0 ? g3[0] : ((lane & 3) == 1 ? g3[1] : g3[2]);   // misc[8..10]: running dref of the row (every lane stores; slot 11 is a dummy)
      }
    }
  }
  __syncthreads();
  DC_MARK(8);
  // ---- output_proj, gate, trilinear scatter ------------------------------------------------------------------------------------
  branch_grad(1, A0, Gd.out);
  __syncthreads();
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_OPROJ], wv * 64, lane, to_G());      // d(gated) (T tensor)
  __syncthreads();
  DC_MARK(9);
  {
    const f32x4 aw = *(const f32x4*)(P.attw_w + lane * 4);
#pragma unroll 2
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr, gr = row0 + row;
      const bool ok = gr < M;                                        // wave-uniform (the wave id is a scalar)
      const f32x4 dgt = *(const f32x4*)(G + row * DC_TS + lane * 4);
      const f32x4 samp = E::unpack4(*(const V4*)(S.samp + (size_t)gr * DC_C + lane * 4));
      const float wl = S.mr[(size_t)gr * 16 + 14];
      const float gate = E::round(E::sigmoid(wl));
      float dw = dc_wave_sum(dgt[0] * samp[0] + dgt[1] * samp[1] + dgt[2] * samp[2] + dgt[3] * samp[3]);
      const float sg = E::sigmoid(wl);
      const float dwl = E::round(E::round(dw) * sg * (1.f - sg));
      Gd.wl[gr] = E::from_f(dwl);                                    // all lanes, same value
      // gradient w.r.t. (x1 + pos) through attention_weights: into the residual stream and into the pos gradient
      f32x4 dqp;
#pragma unroll
      for (int j = 0; j < 4; ++j) dqp[j] = E::round(dwl * E::round(aw[j]));
      float* fp = F + row * DC_TS + lane * 4;
      *(f32x4*)fp = *(const f32x4*)fp + dqp;
      *(V4*)(Gd.dposa + (size_t)gr * DC_C + lane * 4) = E::pack4(dqp);
      // trilinear scatter of d(sample) = d(gated) * gate
      f32x4 ds;
#pragma unroll
      for (int j = 0; j < 4; ++j) ds[j] = E::round(dgt[j] * gate);
      float dsc[4];                                                  // the same d(sample) values, channel j*64 + lane
#pragma unroll
      for (int j = 0; j < 4; ++j) dsc[j] = E::round(G[row * DC_TS + j * 64 + lane] * gate);
      float dsp[2][2];                                               // ... and as channel pairs (j*128 + 2*lane, + 1) for the packed bf16 atomics
#pragma unroll
      for (int j = 0; j < 2; ++j) { dsp[j][0] = E::round(G[row * DC_TS + j * 128 + 2 * lane] * gate); dsp[j][1] = E::round(G[row * DC_TS + j * 128 + 2 * lane + 1] * gate); }
      DcCorners tc;
      dc_corners<E>(misc + row * DC_MISC_LD, min(gr, M - 1) / dm.qps, dm.dz, dm.dy, dm.dx, tc);
      float gx = 0.f, gy = 0.f, gz = 0.f;
#ifndef DC_ABL_NOGATHER
      if (ok) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (tc.row[c] >= 0) {
            // lane l adds channels l, l + 64, l + 128, l + 192: one atomic instruction then covers 64 CONSECUTIVE floats (two cache
            // lines) - with the MFMA-side mapping (channels 4l .. 4l+3) each of the four instructions touched all eight lines of
            // the row, and the L2 atomic unit works a line at a time (125 of this kernel's 316 us were these atomics)
            if (E::DT == U3D_BF16 && dm.dvalue_bf16) {
              // bf16 accumulator: lane l adds the channel PAIRS (2l, 2l+1) and (128 + 2l, 128 + 2l + 1) - global_atomic_pk_add_bf16,
              // 64 lanes x 4 B = 256 consecutive bytes per instruction, two instructions per corner instead of four, and the volume
              // gradient needs neither a 196 MB f32 zero-fill nor a cast pass (a cell collects a handful of contributions at most)
              typedef short s16x2 __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              u16* dvp = (u16*)dvalue_v + (size_t)tc.row[c] * DC_C + 2 * lane;
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const bf16x2_t pv = {(__bf16)(tc.w[c] * dsp[j][0]), (__bf16)(tc.w[c] * dsp[j][1])};
                __builtin_amdgcn_global_atomic_fadd_v2bf16((s16x2 __attribute__((address_space(1)))*)(dvp + j * 128), __builtin_bit_cast(s16x2, pv));
              }
            } else {
              float* dvr = dvalue + (size_t)tc.row[c] * DC_C + lane;
#pragma unroll
              for (int j = 0; j < 4; ++j) atomicAdd(dvr + j * 64, tc.w[c] * dsc[j]);
            }
            if (dm.need_dref) {
              const f32x4 val = E::unpack4(*(const V4*)(value + (size_t)tc.row[c] * DC_C + lane * 4));
              const float dot = val[0] * ds[0] + val[1] * ds[1] + val[2] * ds[2] + val[3] * ds[3];
              gx += tc.dwx[c] * dot; gy += tc.dwy[c] * dot; gz += tc.dwz[c] * dot;
            }
          }
      }
#endif
      if (dm.need_dref) {
        gx = dc_wave_sum(gx); gy = dc_wave_sum(gy); gz = dc_wave_sum(gz);
        {                                        // every lane stores component lane % 3 (same address -> same value): no lane branch
          const float* r3 = misc + row * DC_MISC_LD;
          const int k = lane % 3;
          const float gk = k == 0 ? gx * dm.dx * 0.5f : (k == 1 ? gy * dm.dy * 0.5f : gz * dm.dz * 0.5f);   // d/d(grid) -> d/d(logit): grid = 2 sigmoid(l) - 1
          const float sk = E::sigmoid(r3[k]);
          dref[(size_t)gr * 3 + k] = r3[8 + k] + gk * 2.f * sk * (1.f - sk);
        }
      }
    }
  }
  __syncthreads();
  DC_MARK(10);
  // ---- LN1 -> du1 (F): gradient of the residual stream at the layer input, and of the attention output -------------------------
  {
    DcLnBwdOut<E> o = {F, nullptr, nullptr};
    dc_layernorm_bwd<E>(F, u_f32(S.u1), S.mr, U3D_DLN_1, P.ln_g[U3D_DLN_1], P.ln_b[U3D_DLN_1], false, o, Gd.lnp, U3D_DLN_1, nb, red, row0,
                        wave, lane, tid);
  }
  dc_store_f<E>(F, Gd.du1, row0, tid);
  branch_grad(0, A0, Gd.o2);
  __syncthreads();
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_OUTP], wv * 64, lane, [=](int row, int col, f32x4 v) {
    *(V4*)(A1 + dc_aoff<E>(row, col, DC_C)) = E::pack4(v);              // A1 is free by now: d(attention output) leaves from the tile
  });
  __syncthreads();
  dc_store_a<E, 256>(A1, Gd.d_o, DC_C, row0, tid);
  DC_MARK(11);
}

// ---------------------------------------------------------------------------------------------------------------------------
// attention backward
// ---------------------------------------------------------------------------------------------------------------------------
// dQ: a lane owns one query (column of the transposed score tiles); the workgroup walks its query tiles over the staged key chunk
template <typename E>
__global__ __launch_bounds__(256, 2) void k_mha_bwd_dq(const typename E::T* __restrict__ qk, const typename E::T* __restrict__ vv,
                                                       const typename E::T* __restrict__ o, const typename E::T* __restrict__ d_o,
                                                       const float* __restrict__ lse, int nq, int qt_per_wg, float scale_log2, float scale,
                                                       unsigned thresh, float inv_keep, int layer, const unsigned long long* __restrict__ rng,
                                                       typename E::T* __restrict__ dqk, const unsigned* __restrict__ amask) {
  // amask (nullable, uniform): the keep bits the forward launch published (U3D_DS_AMASK; one chunk: nq <= KC) - read instead of hashed
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef Mha<E> H;
  constexpr int KC = H::KC, MW = KC / 32;
  __shared__ __attribute__((aligned(16))) T Ks[H::RM_ELEMS];
  __shared__ __attribute__((aligned(16))) T Vs[H::RM_ELEMS];
  __shared__ __attribute__((aligned(16))) T Kt[H::TP_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const unsigned nq_pad = (unsigned)(nq + 1) & ~1u;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  const bool one_chunk = nq <= KC;
  for (int qt = 0; qt < qt_per_wg; ++qt) {
    const int q0 = (blockIdx.x * qt_per_wg + qt) * 64;
    if (q0 >= nq) break;                                 // uniform
    const int q = q0 + wave * 16 + r16;
    const bool qok = q < nq;
    typename H::RowFrag qf = H::zero_frag(), dof = H::zero_frag();
    float Dq = 0.f, lq = INFINITY;                       // a query past the group: exp2(-inf) = 0 everywhere (its lanes are never stored)
    if (qok) {
      qf = H::load_frag(qk + (base + q) * 512 + h * DC_HD, kq);
      dof = H::load_frag(d_o + (base + q) * DC_C + h * DC_HD, kq);
      Dq = H::dot(dof, H::load_frag(o + (base + q) * DC_C + h * DC_HD, kq));
      lq = lse[(base + q) * DC_NHEAD + h];
    }
    Dq += __shfl_xor(Dq, 16, 64);
    Dq += __shfl_xor(Dq, 32, 64);
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned row = (unsigned)(bh * nq + q);
    unsigned kb[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) kb[j] = (amask && qok) ? amask[(size_t)row * MW + j] : 0u;
    for (int kc0 = 0; kc0 < nq; kc0 += KC) {
      if (qt == 0 || !one_chunk) {
        __syncthreads();
        H::stage(qk + 256 + h * DC_HD, 512, base, kc0, nq, Ks, Kt, tid);
        H::stage(vv + h * DC_HD, 256, base, kc0, nq, Vs, nullptr, tid);
        __syncthreads();
      }
      const int nkeys = min(KC, nq - kc0);
      const int ntile = (nkeys + 15) >> 4;
#pragma unroll
      for (int tp = 0; tp < KC / 32; ++tp) {
        if (tp * 2 < ntile) {
          f32x4 dsv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = tp * 2 + u;
            const f32x4 s = H::scores(Ks, t * 16, r16, kq, qf);
            f32x4 dp = H::scores(Vs, t * 16, r16, kq, dof);
            f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = E::exp2(fmaf(s[r], scale_log2, -lq));
            if (t >= ntile - 1) {                         // keys past the group (zero rows of the image): only the last tile (pair)
#pragma unroll
              for (int r = 0; r < 4; ++r) p[r] = t * 16 + kq * 4 + r < nkeys ? p[r] : 0.f;
            }
            if (thresh && amask) {
              const int w = (int)kb[tp];
#pragma unroll
              for (int r = 0; r < 4; ++r) dp[r] = __int_as_float(__float_as_int(dp[r] * inv_keep) & __builtin_amdgcn_sbfe(w, u * 16 + kq * 4 + r, 1));
            } else if (thresh) {
              bool keep[4];
              dc_keep4(rg, key_site, dc_att_idx(row, (unsigned)(kc0 + t * 16 + kq * 4), nq_pad), thresh, keep);
#pragma unroll
              for (int r = 0; r < 4; ++r) dp[r] *= keep[r] ? inv_keep : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dsv[u][r] = p[r] * (dp[r] - Dq) * scale;
          }
          H::pv(Ks, Kt, tp, r16, kq, dsv[0], dsv[1], dq);
        }
      }
    }
    if (qok) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) *(V4*)(dqk + (base + q) * 512 + h * DC_HD + dt * 16 + kq * 4) = E::pack4(dq[dt]);
    }
  }
}

// dK, dV: a lane owns one key; the workgroup walks its key tiles over the staged query chunk.  No masks: a key lane past the group is
// never stored, and a query row past the group carries lse = +inf (probability 0).
template <typename E>
__global__ __launch_bounds__(256, 2) void k_mha_bwd_dkv(const typename E::T* __restrict__ qk, const typename E::T* __restrict__ vv,
                                                        const typename E::T* __restrict__ o, const typename E::T* __restrict__ d_o,
                                                        const float* __restrict__ lse, int nq, int kt_per_wg, float scale_log2, float scale,
                                                        unsigned thresh, float inv_keep, int layer, const unsigned long long* __restrict__ rng,
                                                        typename E::T* __restrict__ dqk, typename E::T* __restrict__ dv,
                                                        const unsigned* __restrict__ amask) {
  // amask (nullable, uniform; EB only): the forward's keep bits - the (group, head)'s words are staged in LDS with the query chunk,
  // a lane reads the word holding ITS key of four consecutive query rows per tile (this kernel hashed every element on its own)
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef typename E::VC VC;
  typedef Mha<E> H;
  constexpr int KC = H::KC, NP = H::NP, MW = KC / 32;
  __shared__ unsigned Ms[H::TR ? KC * MW : 4];
  __shared__ __attribute__((aligned(16))) T Qs[H::RM_ELEMS];
  __shared__ __attribute__((aligned(16))) T Os[H::RM_ELEMS];        // dO rows
  __shared__ __attribute__((aligned(16))) T Qt[H::TP_ELEMS];
  __shared__ __attribute__((aligned(16))) T Ot[H::TP_ELEMS];        // dO^T (EF)
  __shared__ float lse_s[KC], D_s[KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const unsigned nq_pad = (unsigned)(nq + 1) & ~1u;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  const bool one_chunk = nq <= KC;
  for (int kt = 0; kt < kt_per_wg; ++kt) {
    const int k0 = (blockIdx.x * kt_per_wg + kt) * 64;
    if (k0 >= nq) break;                                 // uniform
    const int key = k0 + wave * 16 + r16;
    const bool kok = key < nq;
    typename H::RowFrag kf = H::zero_frag(), vf = H::zero_frag();
    if (kok) {
      kf = H::load_frag(qk + (base + key) * 512 + 256 + h * DC_HD, kq);
      vf = H::load_frag(vv + (base + key) * DC_C + h * DC_HD, kq);
    }
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dvv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int qc0 = 0; qc0 < nq; qc0 += KC) {
      if (kt == 0 || !one_chunk) {
        __syncthreads();
        H::stage(qk + h * DC_HD, 512, base, qc0, nq, Qs, Qt, tid);
        H::stage(d_o + h * DC_HD, 256, base, qc0, nq, Os, Ot, tid);
        DC_FOR_TID(c, KC * NP) {                                // D[q] = dO[q] . O[q] over this head's 32 columns; lse[q]
          const int qq = c / NP, part = c % NP;
          float d = 0.f;
          if (qc0 + qq < nq) {
            const VC a = *(const VC*)(d_o + (base + qc0 + qq) * DC_C + h * DC_HD + part * E::CH);
            const VC b = *(const VC*)(o + (base + qc0 + qq) * DC_C + h * DC_HD + part * E::CH);
#pragma unroll
            for (int e = 0; e < E::CH; ++e) d += E::chunk_elem(a, e) * E::chunk_elem(b, e);
          }
#pragma unroll
          for (int sh = 1; sh < NP; sh <<= 1) d += __shfl_xor(d, sh, 64);
          if (part == 0) {
            D_s[qq] = d;
            lse_s[qq] = qc0 + qq < nq ? lse[(base + qc0 + qq) * DC_NHEAD + h] : INFINITY;
          }
        }
        if constexpr (H::TR) {
          if (amask) {
            DC_FOR_TID(c, KC * MW) Ms[c] = c < nq * MW ? amask[(size_t)bh * nq * MW + c] : 0u;
          }
        }
        __syncthreads();
      }
      const int nqs = min(KC, nq - qc0);
      const int ntile = (nqs + 15) >> 4;
#pragma unroll 2
      for (int tp = 0; tp < KC / 32; ++tp) {
        if (tp * 2 < ntile) {
          f32x4 pd[2], dsv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = tp * 2 + u;
            const f32x4 s = H::scores(Qs, t * 16, r16, kq, kf);
            const f32x4 dp = H::scores(Os, t * 16, r16, kq, vf);
            const f32x4 l4 = *(const f32x4*)(lse_s + t * 16 + kq * 4), D4 = *(const f32x4*)(D_s + t * 16 + kq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = E::exp2(fmaf(s[r], scale_log2, -l4[r]));
              float keepf = 1.f;
              if (H::TR && thresh && amask)
                keepf = __int_as_float(__float_as_int(inv_keep) & __builtin_amdgcn_sbfe((int)Ms[(t * 16 + kq * 4 + r) * MW + (key >> 5)], key & 31, 1));
              else if (thresh) keepf = dc_keep(rg, key_site, dc_att_idx((unsigned)(bh * nq + qc0 + t * 16 + kq * 4 + r), (unsigned)key, nq_pad), thresh) ? inv_keep : 0.f;
              pd[u][r] = p * keepf;
              dsv[u][r] = p * (dp[r] * keepf - D4[r]) * scale;
            }
          }
          H::pv(Os, Ot, tp, r16, kq, pd[0], pd[1], dvv);
          H::pv(Qs, Qt, tp, r16, kq, dsv[0], dsv[1], dk);
        }
      }
    }
    if (kok) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        *(V4*)(dqk + (base + key) * 512 + 256 + h * DC_HD + dt * 16 + kq * 4) = E::pack4(dk[dt]);
        *(V4*)(dv + (base + key) * DC_C + h * DC_HD + dt * 16 + kq * 4) = E::pack4(dvv[dt]);
      }
    }
  }
}

template <typename E>
static void mha_bwd_launch(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                           float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, u3d_stream s, const unsigned* amask = nullptr) {
  typedef typename E::T T;
  if (nq > Mha<E>::KC || E::DT != U3D_BF16) amask = nullptr;      // the published bits cover one-chunk bf16 groups only
  const float scale = 1.f / sqrtf((float)DC_HD), scale_log2 = 1.4426950408889634f * scale;
  const int qt = mha_tiles_per_wg<E>(nq);
  const dim3 grid(u3d_cdiv(nq, 64 * qt), (m / nq) * DC_NHEAD);
  hipLaunchKernelGGL(k_mha_bwd_dq<E>, grid, dim3(256), 0, s, (const T*)qk, (const T*)v, (const T*)o, (const T*)d_o, lse, nq, qt, scale_log2,
                     scale, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (T*)dqk, amask);
  hipLaunchKernelGGL(k_mha_bwd_dkv<E>, grid, dim3(256), 0, s, (const T*)qk, (const T*)v, (const T*)o, (const T*)d_o, lse, nq, qt, scale_log2,
                     scale, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (T*)dqk, (T*)dv, amask);
}
extern "C" int32_t u3d_mha_bwd_dt(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                                  float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, int32_t dtype, u3d_stream s) {
  U3D_REQUIRE(qk && v && o && d_o && lse && dqk && dv && m > 0 && nq > 0 && m % nq == 0 && p_attn >= 0.f && p_attn < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE(p_attn == 0.f || rng, U3D_ERR_ARG);
  U3D_REQUIRE(dtype == U3D_BF16 || dtype == U3D_F32, U3D_ERR_ARG);
  U3D_REQUIRE((long long)(m / nq) * DC_NHEAD * nq * (nq + 1) < (1ll << 32), U3D_ERR_UNSUPPORTED);
  if (dtype == U3D_BF16) mha_bwd_launch<EB>(qk, v, o, d_o, lse, m, nq, p_attn, layer, rng, dqk, dv, s);
  else mha_bwd_launch<EF>(qk, v, o, d_o, lse, m, nq, p_attn, layer, rng, dqk, dv, s);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
extern "C" int32_t u3d_mha_bwd(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                               float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, u3d_stream s) {
  return u3d_mha_bwd_dt(qk, v, o, d_o, lse, m, nq, p_attn, layer, rng, dqk, dv, U3D_BF16, s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_pre_bwd
// ---------------------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(DC_THREADS) void k_dec_pre_bwd(u3d_declayer_params P, u3d_declayer_dims dm, DcSave<E> S, DcGrad<E> Gd,
                                                            float* __restrict__ dx) {
  typedef typename E::T T;
  typedef typename E::V4 V4;
  typedef DcLds<E> L;
  constexpr int BM = E::BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  T* A0 = (T*)(lds + L::A0);
  T* A1 = (T*)(lds + L::A1);
  T* A2 = (T*)(lds + L::A2);
  float* F = (float*)(lds + L::F);
  float* G = (float*)(lds + L::G);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int wv = (wave + (int)(blockIdx.x >> 3)) & 3;      // which 64 output columns this wave computes: rotated per workgroup (see dc_gemm)
  const int row0 = blockIdx.x * BM, M = dm.m;
  dc_poison_lds<E>(lds, tid);

  // dqk / dv rows past m were never written by the attention kernels: re-read row m-1 (finite); the gradients those rows produce
  // land in padded slot rows nobody reads
  dc_load_a<E, 512, true>(A1, Gd.dqk, 512, row0, M, tid);
  dc_load_a<E, 256, true>(A0, Gd.dv, DC_C, row0, M, tid);
  dc_load_f_rows<E>(F, Gd.du1, row0, tid);
  __syncthreads();
  auto acc_to_F = [&]() {
    return [=](int row, int col, f32x4 v) {
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + E::round4(v);
    };
  };
  auto masked_to = [&](T* dst, const T* mask_src) {          // (dY W) * [saved ReLU output > 0] -> tile (slot copy: dc_store_a)
    return [=](int row, int col, f32x4 v) {
      const V4 y = *(const V4*)(mask_src + (size_t)(row0 + row) * DC_C + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = E::to_f(y[r]) > 0.f ? v[r] : 0.f;
      *(V4*)(dst + dc_aoff<E>(row, col, DC_C)) = E::pack4(v);
    };
  };
  // gradient w.r.t. the q = k input (x + pos) -> G; it reaches x (F) and pos
  dc_linear<E, 512, 4>(A1, (const T*)P.wt[U3D_DL_INQK], wv * 64, lane,
                       [=](int row, int col, f32x4 v) { *(f32x4*)(G + row * DC_TS + col) = E::round4(v); });
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_INV], wv * 64, lane, acc_to_F());
  __syncthreads();
  // dpos = d(q=k input) + gate path (post kernel); x gets d(q=k input) too.  d(raw), d(query_scale output) by the product rule.
  DC_FOR_TID(c, BM * 64) {
    const int row = c >> 6, col = (c & 63) * 4;
    const size_t go = (size_t)(row0 + row) * DC_C + col;
    const f32x4 dqk = *(const f32x4*)(G + row * DC_TS + col);
    float* fp = F + row * DC_TS + col;
    *(f32x4*)fp = *(const f32x4*)fp + dqk;
    const f32x4 dpos = E::round4(dqk + E::unpack4(*(const V4*)(Gd.dposa + go)));
    const int ao = dc_aoff<E>(row, col, DC_C);
    if (dm.has_qs) {
      const f32x4 raw = E::unpack4(*(const V4*)(S.raw + go)), qs = E::unpack4(*(const V4*)(S.qs + go));
      const V4 dqs = E::pack4(dpos * raw), draw = E::pack4(dpos * qs);
      *(V4*)(A0 + ao) = dqs;
      *(V4*)(A2 + ao) = draw;
      *(V4*)(Gd.qs + go) = dqs;
      *(V4*)(Gd.raw + go) = draw;
    } else {
      const V4 draw = E::pack4(dpos);
      *(V4*)(A2 + ao) = draw;
      *(V4*)(Gd.raw + go) = draw;
    }
  }
  __syncthreads();
  T* A1b = A1 + BM * DC_C;
  if (dm.has_qs) {
    dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_QS2], wv * 64, lane, masked_to(A1, S.qs2));
    __syncthreads();
    dc_store_a<E, 256>(A1, Gd.qs2, DC_C, row0, tid);
    dc_linear<E, 256, 4>(A1, (const T*)P.wt[U3D_DL_QS1], wv * 64, lane, masked_to(A1b, S.qs1));
    __syncthreads();
    dc_store_a<E, 256>(A1b, Gd.qs1, DC_C, row0, tid);
    dc_linear<E, 256, 4>(A1b, (const T*)P.wt[U3D_DL_QS0], wv * 64, lane, acc_to_F());
    __syncthreads();
  }
  dc_linear<E, 256, 4>(A2, (const T*)P.wt[U3D_DL_RPH2], wv * 64, lane, masked_to(A0, S.rph2));
  __syncthreads();
  dc_store_a<E, 256>(A0, Gd.rph2, DC_C, row0, tid);
  dc_linear<E, 256, 4>(A0, (const T*)P.wt[U3D_DL_RPH1], wv * 64, lane, masked_to(A1, S.rph1));
  __syncthreads();
  dc_store_a<E, 256>(A1, Gd.rph1, DC_C, row0, tid);
  if (dm.need_dref) {      // gradient w.r.t. the sine embedding [BM][384] -> slot (u3d_sine_embed_bwd turns it into d(logits))
    auto to_sine = [=](int row, int col, f32x4 v) { *(V4*)(Gd.sine + (size_t)(row0 + row) * 384 + col) = E::pack4(v); };
    dc_linear<E, 256, 4>(A1, (const T*)P.wt[U3D_DL_RPH0], wv * 64, lane, to_sine);
    dc_linear<E, 256, 2>(A1, (const T*)P.wt[U3D_DL_RPH0], 256 + wv * 32, lane, to_sine);
  }
  dc_store_f<E>(F, dx, row0, tid);
}

static int32_t dcb_check(const u3d_declayer_params* p, const u3d_declayer_dims* d) {
  U3D_REQUIRE(p && d, U3D_ERR_ARG);
  U3D_REQUIRE(d->m > 0 && d->nq > 0 && d->qps > 0 && d->qps % d->nq == 0 && d->m % d->qps == 0 && d->batch == d->m / d->qps, U3D_ERR_ARG);
  U3D_REQUIRE(d->ncls > 0 && d->ncls <= 32 && d->code > 0 && d->code <= 32, U3D_ERR_UNSUPPORTED);
  U3D_REQUIRE(d->dtype == U3D_BF16 || d->dtype == U3D_F32, U3D_ERR_ARG);
  for (int i = 0; i < U3D_DL_NLIN; ++i) {
    if (!d->has_qs && (i == U3D_DL_QS0 || i == U3D_DL_QS1 || i == U3D_DL_QS2)) continue;
    U3D_REQUIRE(p->w[i], U3D_ERR_ARG);
    U3D_REQUIRE(p->wt[i], U3D_ERR_ARG);
  }
  for (int i = 0; i < U3D_DL_NLN; ++i) U3D_REQUIRE(p->ln_g[i] && p->ln_b[i], U3D_ERR_ARG);
  U3D_REQUIRE(p->attw_w && p->pe0_w && p->pe0_b && p->dim_t, U3D_ERR_ARG);
  return U3D_OK;
}

template <typename E>
static int32_t dcb_layer_bwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* ref, const void* value,
                             const uint64_t* rng, const void* save, const float* dx_out, const float* dreg, const float* dcls,
                             const float* diou, float* dx, void* dvalue, float* dref, void* grad, int64_t grad_bytes, u3d_stream s) {
  typedef typename E::T T;
  int64_t total = 0;
  const DcGrad<E> G = dcb_resolve_grad<E>(grad, d->m, d->ncls, d->code, &total);
  U3D_REQUIRE(grad_bytes >= total, U3D_ERR_WORKSPACE);
  const DcSave<E> S = dcb_resolve_save<E>(save, d->m);
  U3D_ALLOW_LDS(k_dec_post_bwd<E>, DcLds<E>::BYTES);
  U3D_ALLOW_LDS(k_dec_pre_bwd<E>, DcLds<E>::BYTES);
  hipLaunchKernelGGL(k_dec_post_bwd<E>, dim3(G.nb), dim3(DC_THREADS), DcLds<E>::BYTES, s, *p, *d, ref, (const T*)value,
                     (const unsigned long long*)rng, S, G, dx_out, dreg, dcls, diou, dvalue, dref);
  if (mha_fused_inproj(E::DT, d->nq) && d->p_attn > 0.f) {
    // the forward launch of this layer published its keep bits: the same decisions, read instead of hashed
    U3D_REQUIRE((long long)(d->m / d->nq) * DC_NHEAD * d->nq * (d->nq + 1) < (1ll << 32), U3D_ERR_UNSUPPORTED);
    mha_bwd_launch<E>(S.qk, S.v, S.o, G.d_o, S.lse, d->m, d->nq, d->p_attn, d->layer, rng, G.dqk, G.dv, s, S.amask);
  } else {
    int32_t rc = u3d_mha_bwd_dt(S.qk, S.v, S.o, G.d_o, S.lse, d->m, d->nq, d->p_attn, d->layer, rng, G.dqk, G.dv, E::DT, s);
    if (rc != U3D_OK) return rc;
  }
  hipLaunchKernelGGL(k_dec_pre_bwd<E>, dim3(G.nb), dim3(DC_THREADS), DcLds<E>::BYTES, s, *p, *d, S, G, dx);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

extern "C" int32_t u3d_decoder_layer_bwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc,
                                         const float* ref, const void* value, const uint64_t* rng, const void* xc_out, const void* save,
                                         const float* dx_out, const float* dreg, const float* dcls, const float* diou, float* dx,
                                         void* dvalue, float* dref, void* grad, int64_t grad_bytes, u3d_stream s) {
  (void)x; (void)xc; (void)xc_out;
  int32_t rc = dcb_check(p, d);
  if (rc != U3D_OK) return rc;
  U3D_REQUIRE(ref && value && save && dreg && dcls && diou && dx && dvalue && grad, U3D_ERR_ARG);
  U3D_REQUIRE(!d->dvalue_bf16 || d->dtype == U3D_BF16, U3D_ERR_ARG);
  U3D_REQUIRE(!d->need_dref || dref, U3D_ERR_ARG);
  U3D_REQUIRE((d->p_attn == 0.f && d->p_drop == 0.f) || rng, U3D_ERR_ARG);
  if (d->dtype == U3D_BF16)
    return dcb_layer_bwd<EB>(p, d, ref, value, rng, save, dx_out, dreg, dcls, diou, dx, dvalue, dref, grad, grad_bytes, s);
  return dcb_layer_bwd<EF>(p, d, ref, value, rng, save, dx_out, dreg, dcls, diou, dx, dvalue, dref, grad, grad_bytes, s);
}
