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

struct DcSave {            // forward-save slots (read-only here)
  const u16 *sine, *rph1, *rph2, *raw, *qs1, *qs2, *qs, *pos, *qkin, *qk, *v;
  const float* lse;
  const u16* o;
  const float *u1, *mr;
  const u16 *qp, *samp, *gated, *peh0, *upe1;
  const float* u2;
  const u16 *x2c, *ffh;
  const float* u3;
  const u16 *r1, *r2, *i1, *i2, *uc1, *c1, *uc2, *c2;
};
struct DcGrad {            // gradient-workspace slots
  u16 *clso, *c2u, *c1u, *iouo, *i2, *i1, *rego, *r2, *r1, *f, *ffh, *out, *upe1, *p0, *wl, *o2, *d_o, *dqk, *dv, *qs, *qs2, *qs1, *raw,
      *rph2, *rph1;
  float *lnp, *du1;
  u16 *dposa, *sine;
  int nb;                  // workgroups = rows of each LayerNorm partial matrix
};

static DcSave dcb_resolve_save(const void* save, int m) {
  int64_t off[U3D_DS_COUNT + 1];
  u3d_decoder_layer_slots(m, 1, 1, off, nullptr);
  const char* b = (const char*)save;
  DcSave p;
  p.sine = (const u16*)(b + off[U3D_DS_SINE]); p.rph1 = (const u16*)(b + off[U3D_DS_RPH1]); p.rph2 = (const u16*)(b + off[U3D_DS_RPH2]);
  p.raw = (const u16*)(b + off[U3D_DS_RAW]); p.qs1 = (const u16*)(b + off[U3D_DS_QS1]); p.qs2 = (const u16*)(b + off[U3D_DS_QS2]);
  p.qs = (const u16*)(b + off[U3D_DS_QS]); p.pos = (const u16*)(b + off[U3D_DS_POS]); p.qkin = (const u16*)(b + off[U3D_DS_QKIN]);
  p.qk = (const u16*)(b + off[U3D_DS_QK]); p.v = (const u16*)(b + off[U3D_DS_V]); p.lse = (const float*)(b + off[U3D_DS_LSE]);
  p.o = (const u16*)(b + off[U3D_DS_O]); p.u1 = (const float*)(b + off[U3D_DS_U1]); p.mr = (const float*)(b + off[U3D_DS_MR]);
  p.qp = (const u16*)(b + off[U3D_DS_QP]); p.samp = (const u16*)(b + off[U3D_DS_SAMP]); p.gated = (const u16*)(b + off[U3D_DS_GATED]);
  p.peh0 = (const u16*)(b + off[U3D_DS_PEH0]); p.upe1 = (const u16*)(b + off[U3D_DS_UPE1]); p.u2 = (const float*)(b + off[U3D_DS_U2]);
  p.x2c = (const u16*)(b + off[U3D_DS_X2C]); p.ffh = (const u16*)(b + off[U3D_DS_FFH]); p.u3 = (const float*)(b + off[U3D_DS_U3]);
  p.r1 = (const u16*)(b + off[U3D_DS_R1]); p.r2 = (const u16*)(b + off[U3D_DS_R2]); p.i1 = (const u16*)(b + off[U3D_DS_I1]);
  p.i2 = (const u16*)(b + off[U3D_DS_I2]); p.uc1 = (const u16*)(b + off[U3D_DS_UC1]); p.c1 = (const u16*)(b + off[U3D_DS_C1]);
  p.uc2 = (const u16*)(b + off[U3D_DS_UC2]); p.c2 = (const u16*)(b + off[U3D_DS_C2]);
  return p;
}
static DcGrad dcb_resolve_grad(void* grad, int m, int ncls, int code, int64_t* total) {
  int64_t off[U3D_DG_COUNT + 1];
  u3d_decoder_layer_slots(m, ncls, code, nullptr, off);
  char* b = (char*)grad;
  DcGrad g;
  g.clso = (u16*)(b + off[U3D_DG_CLSO]); g.c2u = (u16*)(b + off[U3D_DG_C2U]); g.c1u = (u16*)(b + off[U3D_DG_C1U]);
  g.iouo = (u16*)(b + off[U3D_DG_IOUO]); g.i2 = (u16*)(b + off[U3D_DG_I2]); g.i1 = (u16*)(b + off[U3D_DG_I1]);
  g.rego = (u16*)(b + off[U3D_DG_REGO]); g.r2 = (u16*)(b + off[U3D_DG_R2]); g.r1 = (u16*)(b + off[U3D_DG_R1]);
  g.f = (u16*)(b + off[U3D_DG_F]); g.ffh = (u16*)(b + off[U3D_DG_FFH]); g.out = (u16*)(b + off[U3D_DG_OUT]);
  g.upe1 = (u16*)(b + off[U3D_DG_UPE1]); g.p0 = (u16*)(b + off[U3D_DG_P0]); g.wl = (u16*)(b + off[U3D_DG_WL]);
  g.o2 = (u16*)(b + off[U3D_DG_O2]); g.d_o = (u16*)(b + off[U3D_DG_DO]); g.dqk = (u16*)(b + off[U3D_DG_DQK]);
  g.dv = (u16*)(b + off[U3D_DG_DV]); g.qs = (u16*)(b + off[U3D_DG_QS]); g.qs2 = (u16*)(b + off[U3D_DG_QS2]);
  g.qs1 = (u16*)(b + off[U3D_DG_QS1]); g.raw = (u16*)(b + off[U3D_DG_RAW]); g.rph2 = (u16*)(b + off[U3D_DG_RPH2]);
  g.rph1 = (u16*)(b + off[U3D_DG_RPH1]); g.lnp = (float*)(b + off[U3D_DG_LNP]); g.du1 = (float*)(b + off[U3D_DG_DU1]);
  g.dposa = (u16*)(b + off[U3D_DG_DPOSA]); g.sine = (u16*)(b + off[U3D_DG_SINE]);
  g.nb = u3d_decoder_layer_blocks(m);
  if (total) *total = off[U3D_DG_COUNT];
  return g;
}

// ---- LayerNorm backward over an f32 tile of incoming gradients --------------------------------------------------------------
// dy: tile T (rows 8w..8w+7 per wave); u(row) -> the forward input row (4 columns per lane); writes du to `tile_out` (f32) and/or the
// activation tile `a_out` (bf16) + global bf16 `g16`; accumulates (dgamma, dbeta) of this workgroup into lnp[ln][0|1][block][256].
struct DcLnBwdOut { float* tile; u16* a; u16* g16; };
template <typename LoadU>
__device__ __forceinline__ void dc_layernorm_bwd(const float* T, LoadU load_u, const float* __restrict__ mr, int mr_idx,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, bool relu,
                                                 const DcLnBwdOut& o, float* __restrict__ lnp, int ln, int nb, float* red /* LDS [4][2][256] */,
                                                 int row0, int wave, int lane, int tid) {
  const f32x4 ga = *(const f32x4*)(gamma + lane * 4), be = *(const f32x4*)(beta + lane * 4);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
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
    s1 = u3d_wave_sum(s1) * (1.f / DC_C);
    s2 = u3d_wave_sum(s2) * (1.f / DC_C);
    f32x4 du;
#pragma unroll
    for (int j = 0; j < 4; ++j) du[j] = rs * (dxh[j] - s1 - xh[j] * s2);
    if (o.tile) *(f32x4*)(o.tile + row * DC_TS + lane * 4) = du;
    const u16x4 dub = dc_pack4(du);
    if (o.a) *(u16x4*)(o.a + dc_aoff(row, lane * 4, DC_C)) = dub;
    if (o.g16) *(u16x4*)(o.g16 + (size_t)gr * DC_C + lane * 4) = dub;
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
__global__ __launch_bounds__(DC_THREADS) void k_dec_post_bwd(u3d_declayer_params P, u3d_declayer_dims dm, const float* __restrict__ ref,
                                                             const u16* __restrict__ value, const unsigned long long* __restrict__ rng,
                                                             DcSave S, DcGrad Gd, const float* __restrict__ dx_out,
                                                             const float* __restrict__ dreg, const float* __restrict__ dcls,
                                                             const float* __restrict__ diou, float* __restrict__ dvalue,
                                                             float* __restrict__ dref) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u16* A0 = (u16*)(lds + DC_OFF_A0);
  u16* A1 = (u16*)(lds + DC_OFF_A1);
  u16* A2 = (u16*)(lds + DC_OFF_A2);
  float* F = (float*)(lds + DC_OFF_F);
  float* G = (float*)(lds + DC_OFF_G);
  float* misc = (float*)(lds + DC_OFF_MISC);
  float* red = (float*)(A1 + DC_BM * DC_C);            // LayerNorm partial reduce scratch: second half of A1 (free whenever it is used)
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int row0 = blockIdx.x * DC_BM, M = dm.m, nb = Gd.nb;
  dc_poison_lds(lds, tid);
  DcDrop drop = {dc_rng_load(rng), dc_thresh(dm.p_drop), dc_inv_keep(dm.p_drop), dm.layer};

  dc_load_f<true>(F, dx_out, row0, M, tid);            // F = gradient w.r.t. the layer output x3 (zero when dx_out is null, zero past row m)
  {
    const int row = tid >> 3, j = tid & 7;             // reference-point logits, slots 3..7 unused
    misc[row * DC_MISC_LD + j] = ref[(size_t)min(row0 + row, M - 1) * 3 + min(j, 2)];
  }
  // narrow gradient [32][n] (f32, m rows) -> misc columns 16.., plus its bf16 copy (padded rows) for the caller's weight gradient
  auto load_narrow = [&](const float* src, int n, u16* gcopy) {
    DC_FOR_TID(i, DC_BM * 32) {
      const int row = i >> 5, j = i & 31;
      const float raw = src[(size_t)min(row0 + row, M - 1) * n + min(j, n - 1)];      // always in range: selects below, no branch
      const float vr = row0 + row < M ? raw : 0.f;                                     // value of column min(j, n-1)
      gcopy[(size_t)(row0 + row) * n + min(j, n - 1)] = dc_f2bf(vr);                   // lanes j >= n repeat column n-1's store
      misc[row * DC_MISC_LD + 16 + j] = dc_round(j < n ? vr : 0.f);
    }
  };
  // dX[row][col] = sum_j dY[row][j] * W[j][col] for a final branch layer (n <= 32 rows of the padded bf16 weight), masked by the
  // saved ReLU output `mask_src` when given; result (bf16) -> activation tile `dst` + global slot, or -> tile G (f32) when dst is null
  auto narrow_dgrad = [&](const u16* w, int n, const u16* mask_src, u16* dst, u16* gslot) {
    const int col = tid;
    float wc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) wc[j] = j < n ? dc_bf2f(w[j * DC_C + col]) : 0.f;
    for (int row = 0; row < DC_BM; ++row) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) acc += misc[row * DC_MISC_LD + 16 + j] * wc[j];
      if (mask_src) {
        const float y = dc_bf2f(mask_src[(size_t)(row0 + row) * DC_C + col]);
        acc = y > 0.f ? acc : 0.f;
      }
      const u16 ab = dc_f2bf(acc);
      if (dst) {
        dst[dc_aoff(row, col, DC_C)] = ab;
        gslot[(size_t)(row0 + row) * DC_C + col] = ab;
      } else {
        G[row * DC_TS + col] = dc_bf2f(ab);
      }
    }
  };
  // dgrad GEMM epilogues
  auto masked_to = [&](u16* dst, u16* gslot, const u16* mask_src) {          // (dY W) * [saved ReLU output > 0] -> tile + slot
    return [=](int row, int col, f32x4 v) {
      const u16x4 y = *(const u16x4*)(mask_src + (size_t)(row0 + row) * DC_C + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dc_bf2f(y[r]) > 0.f ? v[r] : 0.f;
      const u16x4 o = dc_pack4(v);
      *(u16x4*)(dst + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gslot + (size_t)(row0 + row) * DC_C + col) = o;
    };
  };
  auto acc_to_F = [&]() {
    return [=](int row, int col, f32x4 v) {
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + dc_round4(v);
    };
  };
  auto to_G = [&]() {
    return [=](int row, int col, f32x4 v) { *(f32x4*)(G + row * DC_TS + col) = dc_round4(v); };
  };
  auto u_bf16 = [&](const u16* src) {
    return [=](int row, int gr) { return dc_unpack4(*(const u16x4*)(src + (size_t)gr * DC_C + lane * 4)); };
  };
  auto u_f32 = [&](const float* src) {
    return [=](int row, int gr) { return *(const f32x4*)(src + (size_t)gr * DC_C + lane * 4); };
  };

  // ---- cls branch --------------------------------------------------------------------------------------------------------
  load_narrow(dcls, dm.ncls, Gd.clso);
  __syncthreads();
  narrow_dgrad((const u16*)P.w[U3D_DL_CLS2], dm.ncls, nullptr, nullptr, nullptr);          // -> G
  __syncthreads();
  {
    DcLnBwdOut o = {nullptr, A0, Gd.c2u};
    dc_layernorm_bwd(G, u_bf16(S.uc2), S.mr, U3D_DLN_C2, P.ln_g[U3D_DLN_C2], P.ln_b[U3D_DLN_C2], true, o, Gd.lnp, U3D_DLN_C2, nb, red, row0,
                     wave, lane, tid);
  }
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_CLS1], wave * 64, lane, to_G());
  __syncthreads();
  {
    DcLnBwdOut o = {nullptr, A0, Gd.c1u};
    dc_layernorm_bwd(G, u_bf16(S.uc1), S.mr, U3D_DLN_C1, P.ln_g[U3D_DLN_C1], P.ln_b[U3D_DLN_C1], true, o, Gd.lnp, U3D_DLN_C1, nb, red, row0,
                     wave, lane, tid);
  }
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_CLS0], wave * 64, lane, acc_to_F());
  __syncthreads();
  // ---- iou branch ----------------------------------------------------------------------------------------------------------
  load_narrow(diou, 1, Gd.iouo);
  __syncthreads();
  narrow_dgrad((const u16*)P.w[U3D_DL_IOU2], 1, S.i2, A0, Gd.i2);
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_IOU1], wave * 64, lane, masked_to(A1, Gd.i1, S.i1));
  __syncthreads();
  dc_linear<256, 4>(A1, (const u16*)P.wt[U3D_DL_IOU0], wave * 64, lane, acc_to_F());
  __syncthreads();
  // ---- reg branch ----------------------------------------------------------------------------------------------------------
  load_narrow(dreg, dm.code, Gd.rego);
  __syncthreads();
  narrow_dgrad((const u16*)P.w[U3D_DL_REG2], dm.code, S.r2, A0, Gd.r2);
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_REG1], wave * 64, lane, masked_to(A1, Gd.r1, S.r1));
  __syncthreads();
  dc_linear<256, 4>(A1, (const u16*)P.wt[U3D_DL_REG0], wave * 64, lane, acc_to_F());
  __syncthreads();
  // ---- LN3 -> du3 (F) -------------------------------------------------------------------------------------------------------
  {
    DcLnBwdOut o = {F, nullptr, nullptr};
    dc_layernorm_bwd(F, u_f32(S.u3), S.mr, U3D_DLN_3, P.ln_g[U3D_DLN_3], P.ln_b[U3D_DLN_3], false, o, Gd.lnp, U3D_DLN_3, nb, red, row0,
                     wave, lane, tid);
  }
  // ---- FFN -----------------------------------------------------------------------------------------------------------------
  // dY of a residual branch: dropout mask of the forward applied to the residual-stream gradient, bf16 -> tile + slot
  auto branch_grad = [&](int site, u16* dst, u16* gslot) {
    DC_FOR_TID(c, DC_BM * 64) {
      const int row = c >> 6, col = (c & 63) * 4;
      f32x4 v = *(const f32x4*)(F + row * DC_TS + col);
      v = drop.apply(v, site, (unsigned)((row0 + row) * DC_C + col));
      const u16x4 o = dc_pack4(v);
      *(u16x4*)(dst + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gslot + (size_t)(row0 + row) * DC_C + col) = o;
    }
  };
  branch_grad(3, A0, Gd.f);
  __syncthreads();
  {
    const float ik = drop.inv_keep;
    auto dh = [=](int row, int col, f32x4 v) {
      const u16x4 y = *(const u16x4*)(S.ffh + (size_t)(row0 + row) * DC_FF + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dc_bf2f(y[r]) > 0.f ? dc_round(v[r]) * ik : 0.f;     // kept & positive <=> saved value > 0
      const u16x4 o = dc_pack4(v);
      *(u16x4*)(A1 + dc_aoff(row, col, DC_FF)) = o;
      *(u16x4*)(Gd.ffh + (size_t)(row0 + row) * DC_FF + col) = o;
    };
    dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_FFN1], wave * 64, lane, dh);
    dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_FFN1], 256 + wave * 64, lane, dh);
  }
  __syncthreads();
  dc_linear<512, 4>(A1, (const u16*)P.wt[U3D_DL_FFN0], wave * 64, lane, acc_to_F());
  __syncthreads();
  // ---- LN2 -> du2 (F) -------------------------------------------------------------------------------------------------------
  {
    DcLnBwdOut o = {F, nullptr, nullptr};
    dc_layernorm_bwd(F, u_f32(S.u2), S.mr, U3D_DLN_2, P.ln_g[U3D_DLN_2], P.ln_b[U3D_DLN_2], false, o, Gd.lnp, U3D_DLN_2, nb, red, row0,
                     wave, lane, tid);
  }
  // ---- position encoder ---------------------------------------------------------------------------------------------------
  DC_FOR_TID(c, DC_BM * 64) {                                       // its output was a bf16 tensor: the gradient arrives rounded
    const int o = (c >> 6) * DC_TS + (c & 63) * 4;
    *(f32x4*)(G + o) = dc_round4(*(const f32x4*)(F + o));
  }
  __syncthreads();
  {
    DcLnBwdOut o = {nullptr, A0, Gd.upe1};
    dc_layernorm_bwd(G, u_bf16(S.upe1), S.mr, U3D_DLN_PE1, P.ln_g[U3D_DLN_PE1], P.ln_b[U3D_DLN_PE1], true, o, Gd.lnp, U3D_DLN_PE1, nb, red,
                     row0, wave, lane, tid);
  }
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_PE1], wave * 64, lane, to_G());
  __syncthreads();
  {
    // the first position-encoder layer's output is recomputed from the reference point (3 -> 256)
    const f32x4 b4 = *(const f32x4*)(P.pe0_b + lane * 4);
    float w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) w[j][k] = dc_round(P.pe0_w[(lane * 4 + j) * 3 + k]);
    auto u_pe0 = [=](int row, int gr) {
      const float* r3 = misc + row * DC_MISC_LD;
      const float a = dc_round(r3[0]), b = dc_round(r3[1]), c = dc_round(r3[2]);
      f32x4 u;
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = dc_round(a * w[j][0] + b * w[j][1] + c * w[j][2] + b4[j]);
      return u;
    };
    DcLnBwdOut o = {G, nullptr, Gd.p0};
    dc_layernorm_bwd(G, u_pe0, S.mr, U3D_DLN_PE0, P.ln_g[U3D_DLN_PE0], P.ln_b[U3D_DLN_PE0], true, o, Gd.lnp, U3D_DLN_PE0, nb, red, row0,
                     wave, lane, tid);
    if (dm.need_dref) {          // gradient w.r.t. the reference-point logits through Linear(3, 256)
      for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr;
        const f32x4 d = dc_round4(*(const f32x4*)(G + row * DC_TS + lane * 4));
        float g3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) g3[k] = u3d_wave_sum(d[0] * w[0][k] + d[1] * w[1][k] + d[2] * w[2][k] + d[3] * w[3][k]);
        misc[row * DC_MISC_LD + 8 + (lane & 3)] = (lane & 3) == 0 ? g3[0] : ((lane & 3) == 1 ? g3[1] : g3[2]);   // misc[8..10]: running dref of the row (every lane stores; slot 11 is a dummy)
      }
    }
  }
  __syncthreads();
  // ---- output_proj, gate, trilinear scatter ------------------------------------------------------------------------------------
  branch_grad(1, A0, Gd.out);
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_OPROJ], wave * 64, lane, to_G());      // d(gated) (bf16 tensor)
  __syncthreads();
  {
    const f32x4 aw = *(const f32x4*)(P.attw_w + lane * 4);
#pragma unroll 2
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr, gr = row0 + row;
      const bool ok = gr < M;                                        // wave-uniform (the wave id is a scalar)
      const f32x4 dgt = *(const f32x4*)(G + row * DC_TS + lane * 4);
      const f32x4 samp = dc_unpack4(*(const u16x4*)(S.samp + (size_t)gr * DC_C + lane * 4));
      const float wl = S.mr[(size_t)gr * 16 + 14];
      const float gate = dc_round(dc_sigmoid(wl));
      float dw = u3d_wave_sum(dgt[0] * samp[0] + dgt[1] * samp[1] + dgt[2] * samp[2] + dgt[3] * samp[3]);
      const float sg = dc_sigmoid(wl);
      const float dwl = dc_round(dc_round(dw) * sg * (1.f - sg));
      Gd.wl[gr] = dc_f2bf(dwl);                                    // all lanes, same value
      // gradient w.r.t. (x1 + pos) through attention_weights: into the residual stream and into the pos gradient
      f32x4 dqp;
#pragma unroll
      for (int j = 0; j < 4; ++j) dqp[j] = dc_round(dwl * dc_round(aw[j]));
      float* fp = F + row * DC_TS + lane * 4;
      *(f32x4*)fp = *(const f32x4*)fp + dqp;
      *(u16x4*)(Gd.dposa + (size_t)gr * DC_C + lane * 4) = dc_pack4(dqp);
      // trilinear scatter of d(sample) = d(gated) * gate
      f32x4 ds;
#pragma unroll
      for (int j = 0; j < 4; ++j) ds[j] = dc_round(dgt[j] * gate);
      float dsc[4];                                                  // the same d(sample) values, channel j*64 + lane
#pragma unroll
      for (int j = 0; j < 4; ++j) dsc[j] = dc_round(G[row * DC_TS + j * 64 + lane] * gate);
      DcCorners tc;
      dc_corners(misc + row * DC_MISC_LD, min(gr, M - 1) / dm.qps, dm.dz, dm.dy, dm.dx, tc);
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (ok) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (tc.row[c] >= 0) {
            // lane l adds channels l, l + 64, l + 128, l + 192: one atomic instruction then covers 64 CONSECUTIVE floats (two cache
            // lines) - with the MFMA-side mapping (channels 4l .. 4l+3) each of the four instructions touched all eight lines of
            // the row, and the L2 atomic unit works a line at a time (125 of this kernel's 316 us were these atomics)
            float* dvr = dvalue + (size_t)tc.row[c] * DC_C + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dvr + j * 64, tc.w[c] * dsc[j]);
            if (dm.need_dref) {
              const f32x4 val = dc_unpack4(*(const u16x4*)(value + (size_t)tc.row[c] * DC_C + lane * 4));
              const float dot = val[0] * ds[0] + val[1] * ds[1] + val[2] * ds[2] + val[3] * ds[3];
              gx += tc.dwx[c] * dot; gy += tc.dwy[c] * dot; gz += tc.dwz[c] * dot;
            }
          }
      }
      if (dm.need_dref) {
        gx = u3d_wave_sum(gx); gy = u3d_wave_sum(gy); gz = u3d_wave_sum(gz);
        {                                        // every lane stores component lane % 3 (same address -> same value): no lane branch
          const float* r3 = misc + row * DC_MISC_LD;
          const int k = lane % 3;
          const float gk = k == 0 ? gx * dm.dx * 0.5f : (k == 1 ? gy * dm.dy * 0.5f : gz * dm.dz * 0.5f);   // d/d(grid) -> d/d(logit): grid = 2 sigmoid(l) - 1
          const float sk = dc_sigmoid(r3[k]);
          dref[(size_t)gr * 3 + k] = r3[8 + k] + gk * 2.f * sk * (1.f - sk);
        }
      }
    }
  }
  __syncthreads();
  // ---- LN1 -> du1 (F): gradient of the residual stream at the layer input, and of the attention output -------------------------
  {
    DcLnBwdOut o = {F, nullptr, nullptr};
    dc_layernorm_bwd(F, u_f32(S.u1), S.mr, U3D_DLN_1, P.ln_g[U3D_DLN_1], P.ln_b[U3D_DLN_1], false, o, Gd.lnp, U3D_DLN_1, nb, red, row0,
                     wave, lane, tid);
  }
  dc_store_f(F, Gd.du1, row0, tid);
  branch_grad(0, A0, Gd.o2);
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_OUTP], wave * 64, lane, [=](int row, int col, f32x4 v) {
    *(u16x4*)(Gd.d_o + (size_t)(row0 + row) * DC_C + col) = dc_pack4(v);
  });
}

// ---------------------------------------------------------------------------------------------------------------------------
// attention backward
// ---------------------------------------------------------------------------------------------------------------------------
#define MHA_KC 128
#define MHA_VT_LD (MHA_KC + 8)
__device__ __forceinline__ int mhab_koff(int key, int part) { return key * 32 + (((part ^ ((-(key >> 2)) & 3)) & 3) << 3); }
__device__ __forceinline__ void mhab_stage(const u16* __restrict__ src, int ld, long long base_row, int first, int nvalid, u16* rowmajor,
                                           u16* transposed, int tid) {
  DC_FOR_TID(c, MHA_KC * 4) {
    const int key = c >> 2, part = c & 3;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (first + key < nvalid) v = *(const u16x8*)(src + (base_row + first + key) * ld + part * 8);
    if (rowmajor) *(u16x8*)(rowmajor + mhab_koff(key, part)) = v;
    if (transposed) {
#pragma unroll
      for (int e = 0; e < 8; ++e) transposed[(part * 8 + e) * MHA_VT_LD + key] = v[e];
    }
  }
}
__device__ __forceinline__ bf16x8 mhab_tfrag(const u16* T, int dt, int tp, int r16, int kq) {       // [d][key] operand of an output product
  const u16* vp = T + (dt * 16 + r16) * MHA_VT_LD + tp * 32 + kq * 4;
  const u16x4 v0 = *(const u16x4*)vp, v1 = *(const u16x4*)(vp + 16);
  const u16x8 vb = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(bf16x8, vb);
}
__device__ __forceinline__ bf16x8 mhab_pack8(f32x4 a, f32x4 b) {
  const u16x4 a4 = dc_pack4(a), b4 = dc_pack4(b);
  const u16x8 v = {a4[0], a4[1], a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// dQ: a lane owns one query (column of the transposed score tiles), loops over all keys of its group
__global__ __launch_bounds__(256) void k_mha_bwd_dq(const u16* __restrict__ qk, const u16* __restrict__ vv, const u16* __restrict__ o,
                                                    const u16* __restrict__ d_o, const float* __restrict__ lse, int nq, float scale_log2,
                                                    float scale, unsigned thresh, float inv_keep, int layer,
                                                    const unsigned long long* __restrict__ rng, u16* __restrict__ dqk) {
  __shared__ __attribute__((aligned(16))) u16 Ks[MHA_KC * 32];
  __shared__ __attribute__((aligned(16))) u16 Vs[MHA_KC * 32];
  __shared__ __attribute__((aligned(16))) u16 Kt[32 * MHA_VT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const int q = blockIdx.x * 64 + wave * 16 + r16;
  const bool qok = q < nq;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  bf16x8 qf = {0, 0, 0, 0, 0, 0, 0, 0}, dof = {0, 0, 0, 0, 0, 0, 0, 0};
  float Dq = 0.f, lq = 0.f;
  if (qok) {
    qf = *(const bf16x8*)(qk + (base + q) * 512 + h * DC_HD + kq * 8);
    const u16x8 dob = *(const u16x8*)(d_o + (base + q) * DC_C + h * DC_HD + kq * 8);
    const u16x8 ob = *(const u16x8*)(o + (base + q) * DC_C + h * DC_HD + kq * 8);
    dof = __builtin_bit_cast(bf16x8, dob);
#pragma unroll
    for (int e = 0; e < 8; ++e) Dq += dc_bf2f(dob[e]) * dc_bf2f(ob[e]);
    lq = lse[(base + q) * DC_NHEAD + h];
  }
  Dq += __shfl_xor(Dq, 16, 64);
  Dq += __shfl_xor(Dq, 32, 64);
  f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int kc0 = 0; kc0 < nq; kc0 += MHA_KC) {
    __syncthreads();
    mhab_stage(qk + 256 + h * DC_HD, 512, base, kc0, nq, Ks, Kt, tid);
    mhab_stage(vv + h * DC_HD, 256, base, kc0, nq, Vs, nullptr, tid);
    __syncthreads();
    const int nkeys = min(MHA_KC, nq - kc0);
    const int ntile = (nkeys + 15) >> 4;
#pragma unroll
    for (int tp = 0; tp < MHA_KC / 32; ++tp) {
      if (tp * 2 < ntile) {
        f32x4 dsv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = tp * 2 + u;
          const bf16x8 kf = *(const bf16x8*)(Ks + mhab_koff(t * 16 + r16, kq));
          const bf16x8 vf = *(const bf16x8*)(Vs + mhab_koff(t * 16 + r16, kq));
          const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + kq * 4 + r;
            const float p = (key < nkeys && qok) ? __builtin_amdgcn_exp2f(s[r] * scale_log2 - lq) : 0.f;
            if (thresh) {
              const unsigned idx = (unsigned)(((long long)bh * nq + q) * nq + kc0 + key);
              dp[r] = dc_keep(rg, key_site, idx, thresh) ? dp[r] * inv_keep : 0.f;
            }
            dsv[u][r] = p * (dp[r] - Dq) * scale;
          }
        }
        const bf16x8 dsf = mhab_pack8(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mhab_tfrag(Kt, dt, tp, r16, kq), dsf, dq[dt], 0, 0, 0);
      }
    }
  }
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) *(u16x4*)(dqk + (base + q) * 512 + h * DC_HD + dt * 16 + kq * 4) = dc_pack4(dq[dt]);
  }
}

// dK, dV: a lane owns one key, loops over all queries of its group
__global__ __launch_bounds__(256) void k_mha_bwd_dkv(const u16* __restrict__ qk, const u16* __restrict__ vv, const u16* __restrict__ o,
                                                     const u16* __restrict__ d_o, const float* __restrict__ lse, int nq, float scale_log2,
                                                     float scale, unsigned thresh, float inv_keep, int layer,
                                                     const unsigned long long* __restrict__ rng, u16* __restrict__ dqk, u16* __restrict__ dv) {
  __shared__ __attribute__((aligned(16))) u16 Qs[MHA_KC * 32];
  __shared__ __attribute__((aligned(16))) u16 Os[MHA_KC * 32];        // dO rows
  __shared__ __attribute__((aligned(16))) u16 Qt[32 * MHA_VT_LD];
  __shared__ __attribute__((aligned(16))) u16 Ot[32 * MHA_VT_LD];     // dO^T
  __shared__ float lse_s[MHA_KC], D_s[MHA_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int bh = blockIdx.y, g = bh >> 3, h = bh & 7;
  const long long base = (long long)g * nq;
  const int key = blockIdx.x * 64 + wave * 16 + r16;
  const bool kok = key < nq;
  const DcRng rg = dc_rng_load(rng);
  const unsigned key_site = dc_site_key(layer, 4);
  bf16x8 kf = {0, 0, 0, 0, 0, 0, 0, 0}, vf = {0, 0, 0, 0, 0, 0, 0, 0};
  if (kok) {
    kf = *(const bf16x8*)(qk + (base + key) * 512 + 256 + h * DC_HD + kq * 8);
    vf = *(const bf16x8*)(vv + (base + key) * DC_C + h * DC_HD + kq * 8);
  }
  f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dvv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int qc0 = 0; qc0 < nq; qc0 += MHA_KC) {
    __syncthreads();
    mhab_stage(qk + h * DC_HD, 512, base, qc0, nq, Qs, Qt, tid);
    mhab_stage(d_o + h * DC_HD, 256, base, qc0, nq, Os, Ot, tid);
    DC_FOR_TID(c, MHA_KC * 4) {                             // D[q] = dO[q] . O[q] over this head's 32 columns; lse[q]
      const int qq = c >> 2, part = c & 3;
      float d = 0.f;
      if (qc0 + qq < nq) {
        const u16x8 a = *(const u16x8*)(d_o + (base + qc0 + qq) * DC_C + h * DC_HD + part * 8);
        const u16x8 b = *(const u16x8*)(o + (base + qc0 + qq) * DC_C + h * DC_HD + part * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += dc_bf2f(a[e]) * dc_bf2f(b[e]);
      }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      if (part == 0) {
        D_s[qq] = d;
        lse_s[qq] = qc0 + qq < nq ? lse[(base + qc0 + qq) * DC_NHEAD + h] : 0.f;
      }
    }
    __syncthreads();
    const int nqs = min(MHA_KC, nq - qc0);
    const int ntile = (nqs + 15) >> 4;
#pragma unroll
    for (int tp = 0; tp < MHA_KC / 32; ++tp) {
      if (tp * 2 < ntile) {
        f32x4 pd[2], dsv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = tp * 2 + u;
          const bf16x8 qf = *(const bf16x8*)(Qs + mhab_koff(t * 16 + r16, kq));
          const bf16x8 df = *(const bf16x8*)(Os + mhab_koff(t * 16 + r16, kq));
          const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = t * 16 + kq * 4 + r;
            const float p = (qq < nqs && kok) ? __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse_s[qq]) : 0.f;
            float keepf = 1.f;
            if (thresh) {
              const unsigned idx = (unsigned)(((long long)bh * nq + qc0 + qq) * nq + key);
              keepf = dc_keep(rg, key_site, idx, thresh) ? inv_keep : 0.f;
            }
            pd[u][r] = p * keepf;
            dsv[u][r] = p * (dp[r] * keepf - D_s[qq]) * scale;
          }
        }
        const bf16x8 pf = mhab_pack8(pd[0], pd[1]), dsf = mhab_pack8(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dvv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mhab_tfrag(Ot, dt, tp, r16, kq), pf, dvv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mhab_tfrag(Qt, dt, tp, r16, kq), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (kok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      *(u16x4*)(dqk + (base + key) * 512 + 256 + h * DC_HD + dt * 16 + kq * 4) = dc_pack4(dk[dt]);
      *(u16x4*)(dv + (base + key) * DC_C + h * DC_HD + dt * 16 + kq * 4) = dc_pack4(dvv[dt]);
    }
  }
}

extern "C" int32_t u3d_mha_bwd(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                               float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, u3d_stream s) {
  U3D_REQUIRE(qk && v && o && d_o && lse && dqk && dv && m > 0 && nq > 0 && m % nq == 0 && p_attn >= 0.f && p_attn < 1.f, U3D_ERR_ARG);
  U3D_REQUIRE(p_attn == 0.f || rng, U3D_ERR_ARG);
  U3D_REQUIRE((long long)(m / nq) * DC_NHEAD * nq * nq < (1ll << 32), U3D_ERR_UNSUPPORTED);
  const float scale = 1.f / sqrtf((float)DC_HD), scale_log2 = 1.4426950408889634f * scale;
  const dim3 grid(u3d_cdiv(nq, 64), (m / nq) * DC_NHEAD);
  hipLaunchKernelGGL(k_mha_bwd_dq, grid, dim3(256), 0, s, (const u16*)qk, (const u16*)v, (const u16*)o, (const u16*)d_o, lse, nq, scale_log2,
                     scale, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (u16*)dqk);
  hipLaunchKernelGGL(k_mha_bwd_dkv, grid, dim3(256), 0, s, (const u16*)qk, (const u16*)v, (const u16*)o, (const u16*)d_o, lse, nq, scale_log2,
                     scale, dc_thresh(p_attn), dc_inv_keep(p_attn), layer, (const unsigned long long*)rng, (u16*)dqk, (u16*)dv);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_dec_pre_bwd
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DC_THREADS) void k_dec_pre_bwd(u3d_declayer_params P, u3d_declayer_dims dm, DcSave S, DcGrad Gd,
                                                            float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u16* A0 = (u16*)(lds + DC_OFF_A0);
  u16* A1 = (u16*)(lds + DC_OFF_A1);
  u16* A2 = (u16*)(lds + DC_OFF_A2);
  float* F = (float*)(lds + DC_OFF_F);
  float* G = (float*)(lds + DC_OFF_G);
  const int tid = threadIdx.x, lane = tid & 63, wave = dc_wave_id();
  const int row0 = blockIdx.x * DC_BM, M = dm.m;
  dc_poison_lds(lds, tid);

  // dqk / dv rows past m were never written by the attention kernels: re-read row m-1 (finite); the gradients those rows produce
  // land in padded slot rows nobody reads
  dc_load_a<512, true>(A1, Gd.dqk, 512, row0, M, tid);
  dc_load_a<256, true>(A0, Gd.dv, DC_C, row0, M, tid);
  dc_load_f_rows(F, Gd.du1, row0, tid);
  __syncthreads();
  auto acc_to_F = [&]() {
    return [=](int row, int col, f32x4 v) {
      float* fp = F + row * DC_TS + col;
      *(f32x4*)fp = *(const f32x4*)fp + dc_round4(v);
    };
  };
  auto masked_to = [&](u16* dst, u16* gslot, const u16* mask_src) {
    return [=](int row, int col, f32x4 v) {
      const u16x4 y = *(const u16x4*)(mask_src + (size_t)(row0 + row) * DC_C + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dc_bf2f(y[r]) > 0.f ? v[r] : 0.f;
      const u16x4 o = dc_pack4(v);
      *(u16x4*)(dst + dc_aoff(row, col, DC_C)) = o;
      *(u16x4*)(gslot + (size_t)(row0 + row) * DC_C + col) = o;
    };
  };
  // gradient w.r.t. the q = k input (x + pos) -> G; it reaches x (F) and pos
  dc_linear<512, 4>(A1, (const u16*)P.wt[U3D_DL_INQK], wave * 64, lane,
                    [=](int row, int col, f32x4 v) { *(f32x4*)(G + row * DC_TS + col) = dc_round4(v); });
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_INV], wave * 64, lane, acc_to_F());
  __syncthreads();
  // dpos = d(q=k input) + gate path (post kernel); x gets d(q=k input) too.  d(raw), d(query_scale output) by the product rule.
  DC_FOR_TID(c, DC_BM * 64) {
    const int row = c >> 6, col = (c & 63) * 4;
    const size_t go = (size_t)(row0 + row) * DC_C + col;
    const f32x4 dqk = *(const f32x4*)(G + row * DC_TS + col);
    float* fp = F + row * DC_TS + col;
    *(f32x4*)fp = *(const f32x4*)fp + dqk;
    const f32x4 dpos = dc_round4(dqk + dc_unpack4(*(const u16x4*)(Gd.dposa + go)));
    const int ao = dc_aoff(row, col, DC_C);
    if (dm.has_qs) {
      const f32x4 raw = dc_unpack4(*(const u16x4*)(S.raw + go)), qs = dc_unpack4(*(const u16x4*)(S.qs + go));
      const u16x4 dqs = dc_pack4(dpos * raw), draw = dc_pack4(dpos * qs);
      *(u16x4*)(A0 + ao) = dqs;
      *(u16x4*)(A2 + ao) = draw;
      *(u16x4*)(Gd.qs + go) = dqs;
      *(u16x4*)(Gd.raw + go) = draw;
    } else {
      const u16x4 draw = dc_pack4(dpos);
      *(u16x4*)(A2 + ao) = draw;
      *(u16x4*)(Gd.raw + go) = draw;
    }
  }
  __syncthreads();
  u16* A1b = A1 + DC_BM * DC_C;
  if (dm.has_qs) {
    dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_QS2], wave * 64, lane, masked_to(A1, Gd.qs2, S.qs2));
    __syncthreads();
    dc_linear<256, 4>(A1, (const u16*)P.wt[U3D_DL_QS1], wave * 64, lane, masked_to(A1b, Gd.qs1, S.qs1));
    __syncthreads();
    dc_linear<256, 4>(A1b, (const u16*)P.wt[U3D_DL_QS0], wave * 64, lane, acc_to_F());
    __syncthreads();
  }
  dc_linear<256, 4>(A2, (const u16*)P.wt[U3D_DL_RPH2], wave * 64, lane, masked_to(A0, Gd.rph2, S.rph2));
  __syncthreads();
  dc_linear<256, 4>(A0, (const u16*)P.wt[U3D_DL_RPH1], wave * 64, lane, masked_to(A1, Gd.rph1, S.rph1));
  __syncthreads();
  if (dm.need_dref) {      // gradient w.r.t. the sine embedding [32][384] -> slot (u3d_sine_embed_bwd turns it into d(logits))
    auto to_sine = [=](int row, int col, f32x4 v) { *(u16x4*)(Gd.sine + (size_t)(row0 + row) * 384 + col) = dc_pack4(v); };
    dc_linear<256, 4>(A1, (const u16*)P.wt[U3D_DL_RPH0], wave * 64, lane, to_sine);
    dc_linear<256, 2>(A1, (const u16*)P.wt[U3D_DL_RPH0], 256 + wave * 32, lane, to_sine);
  }
  dc_store_f(F, dx, row0, tid);
}

extern "C" int32_t u3d_sine_embed_bwd(const float* logits, const float* dim_t, const void* dout, int32_t dout_dtype, int32_t n, int32_t nc,
                                      int32_t nfeat, float* dlogits, u3d_stream s);

static int32_t dcb_check(const u3d_declayer_params* p, const u3d_declayer_dims* d) {
  U3D_REQUIRE(p && d, U3D_ERR_ARG);
  U3D_REQUIRE(d->m > 0 && d->nq > 0 && d->qps > 0 && d->qps % d->nq == 0 && d->m % d->qps == 0 && d->batch == d->m / d->qps, U3D_ERR_ARG);
  U3D_REQUIRE(d->ncls > 0 && d->ncls <= 32 && d->code > 0 && d->code <= 32, U3D_ERR_UNSUPPORTED);
  for (int i = 0; i < U3D_DL_NLIN; ++i) {
    if (!d->has_qs && (i == U3D_DL_QS0 || i == U3D_DL_QS1 || i == U3D_DL_QS2)) continue;
    U3D_REQUIRE(p->w[i], U3D_ERR_ARG);
    if (i != U3D_DL_REG2 && i != U3D_DL_CLS2 && i != U3D_DL_IOU2) U3D_REQUIRE(p->wt[i], U3D_ERR_ARG);
  }
  for (int i = 0; i < U3D_DL_NLN; ++i) U3D_REQUIRE(p->ln_g[i] && p->ln_b[i], U3D_ERR_ARG);
  U3D_REQUIRE(p->attw_w && p->pe0_w && p->pe0_b && p->dim_t, U3D_ERR_ARG);
  return U3D_OK;
}

extern "C" int32_t u3d_decoder_layer_bwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc,
                                         const float* ref, const void* value, const uint64_t* rng, const void* xc_out, const void* save,
                                         const float* dx_out, const float* dreg, const float* dcls, const float* diou, float* dx,
                                         float* dvalue, float* dref, void* grad, int64_t grad_bytes, u3d_stream s) {
  (void)x; (void)xc; (void)xc_out;
  int32_t rc = dcb_check(p, d);
  if (rc != U3D_OK) return rc;
  U3D_REQUIRE(ref && value && save && dreg && dcls && diou && dx && dvalue && grad, U3D_ERR_ARG);
  U3D_REQUIRE(!d->need_dref || dref, U3D_ERR_ARG);
  U3D_REQUIRE((d->p_attn == 0.f && d->p_drop == 0.f) || rng, U3D_ERR_ARG);
  int64_t total = 0;
  const DcGrad G = dcb_resolve_grad(grad, d->m, d->ncls, d->code, &total);
  U3D_REQUIRE(grad_bytes >= total, U3D_ERR_WORKSPACE);
  const DcSave S = dcb_resolve_save(save, d->m);
  U3D_ALLOW_LDS(k_dec_post_bwd, DC_LDS_BYTES);
  U3D_ALLOW_LDS(k_dec_pre_bwd, DC_LDS_BYTES);
  hipLaunchKernelGGL(k_dec_post_bwd, dim3(G.nb), dim3(DC_THREADS), DC_LDS_BYTES, s, *p, *d, ref, (const u16*)value,
                     (const unsigned long long*)rng, S, G, dx_out, dreg, dcls, diou, dvalue, dref);
  rc = u3d_mha_bwd(S.qk, S.v, S.o, G.d_o, S.lse, d->m, d->nq, d->p_attn, d->layer, rng, G.dqk, G.dv, s);
  if (rc != U3D_OK) return rc;
  hipLaunchKernelGGL(k_dec_pre_bwd, dim3(G.nb), dim3(DC_THREADS), DC_LDS_BYTES, s, *p, *d, S, G, dx);
  U3D_CHECK_LAUNCH();
  return U3D_OK;
}
