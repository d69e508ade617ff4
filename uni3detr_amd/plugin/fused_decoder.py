"""Fused decoder path: every decoder layer (+ the reference-point / query-scale MLPs in front of it and the reg / cls / iou
branches behind it) is ONE C-ABI call forward and one backward (u3d_decoder_layer_fwd / _bwd: csrc/decoder.hip, decoder_bwd.hip) -
own MFMA row-chain kernels and own attention kernels; no torch GEMM / SDPA / element-wise launch in between.

Two instantiations of the SAME kernels (csrc/decoder_common.h, element traits EB / EF): bf16 storage + v_mfma_f32_16x16x32_bf16 in
throughput mode (bf16 autocast), f32 storage + exact v_mfma_f32_16x16x4_f32 in parity mode (`set_precision('fp32')`): the 1e-3
reference-golden tests run the kernels the benchmark runs.

ref: projects/mmdet3d_plugin/models/utils/uni3detr_transformer.py:145-212 (decoder loop), :271-360 (UniCrossAtten),
models/dense_heads/uni3detr_head.py:367-387, 470-490 (branches; the box decode stays in Uni3DETRHead.forward).

The module layout is the registry's (plugin/transformer.py, plugin/head.py): this file only gathers the parameters, keeps bf16
copies of the weights (one refresh launch per step: u3d_wpack_bf16) and turns the kernels' gradient slots into parameter
gradients through the same deferred / batched mechanism the layer-by-layer path uses (plugin/transformer.py `_Deferred`).
"""
import os

import torch
from torch import nn

from .. import native as nv

ENABLED = True         # test-only module attribute: the GPU tests set it False to run the layer-by-layer formulation next to the fused one
# trilinear scatter of the value-volume gradient with global_atomic_pk_add_bf16 into a bf16 accumulator (two channels per atomic, no f32
# volume to zero and cast: -0.07 ms per step).  OPT-IN: every add rounds to 8 mantissa bits, and all 3 layers x 3 query groups land in one
# buffer - with queries clustered on objects a cell collects hundreds of contributions and the result drifts 9 % from the f32
# accumulator (tests/test_decoder_gpu.py::test_packed_bf16_volume_gradient_scatter_vs_f32_accumulator; 0.7 % with spread queries).
SHARED_DEFER = True     # test-only module attribute: shared linears' gradients summed by the deferred flush (False: autograd adds them)
PK_SCATTER = os.environ.get("U3D_PK_SCATTER", "0") == "1"
POISON = os.environ.get("U3D_DEC_POISON", "0") == "1"      # debugging: NaN-fill the workspaces (read-before-write shows up as NaN)
DEBUG_KEEP = None          # tests: a list that receives every backward call's gradient workspace

# (enum index, rows of the weight that belong to this linear) for the wide linears; narrow finals are padded
_WIDE = [nv.DL_RPH0, nv.DL_RPH1, nv.DL_RPH2, nv.DL_QS0, nv.DL_QS1, nv.DL_QS2, nv.DL_INQK, nv.DL_INV, nv.DL_OUTP, nv.DL_OPROJ, nv.DL_PE1,
         nv.DL_FFN0, nv.DL_FFN1, nv.DL_REG0, nv.DL_REG1, nv.DL_CLS0, nv.DL_CLS1, nv.DL_IOU0, nv.DL_IOU1]
_NARROW = [nv.DL_REG2, nv.DL_CLS2, nv.DL_IOU2]


def _is_seq(seq, kinds):
    return isinstance(seq, nn.Sequential) and len(seq) == len(kinds) and all(isinstance(m, k) for m, k in zip(seq, kinds))


class LayerSpec:
    """Parameters of one fused layer, in the order of the C enums: lin[i] = (weight tensor, first row, rows, bias)."""

    def __init__(self, decoder, lid, reg_branch, cls_branch, iou_branch):
        from .transformer import FFN, MultiheadAttention, UniCrossAtten
        layer = decoder.layers[lid]
        if layer.operation_order != ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"):
            raise ValueError("operation order")
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        if not (isinstance(sa, MultiheadAttention) and isinstance(ca, UniCrossAtten) and isinstance(ffn, FFN)):
            raise ValueError("layer members")
        if sa.embed_dims != 256 or sa.num_heads != 8 or ca.num_points != 1 or not ffn.add_identity:
            raise ValueError("dims")
        if not (_is_seq(ffn.layers, (nn.Sequential, nn.Linear, nn.Dropout)) and _is_seq(ffn.layers[0], (nn.Linear, nn.ReLU, nn.Dropout))):
            raise ValueError("ffn layout")
        f0, f1 = ffn.layers[0][0], ffn.layers[1]
        if f0.out_features != 512 or f1.in_features != 512:
            raise ValueError("ffn width")
        pe = ca.position_encoder
        if not _is_seq(pe, (nn.Linear, nn.LayerNorm, nn.ReLU, nn.Linear, nn.LayerNorm, nn.ReLU)):
            raise ValueError("position encoder")
        if not (_is_seq(reg_branch, (nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear))
                and _is_seq(iou_branch, (nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear))
                and _is_seq(cls_branch, (nn.Linear, nn.LayerNorm, nn.ReLU, nn.Linear, nn.LayerNorm, nn.ReLU, nn.Linear))):
            raise ValueError("branch layout")
        rph, qs = decoder.ref_point_head.layers, decoder.query_scale.layers
        if len(rph) != 3 or len(qs) != 3 or rph[0].in_features != 384:
            raise ValueError("mlp layout")
        ipw, ipb = sa.attn.in_proj_weight, sa.attn.in_proj_bias
        if ipw is None or ipb is None or sa.attn.out_proj.bias is None:
            raise ValueError("in_proj")
        L = lambda m: (m.weight, 0, m.weight.shape[0], m.bias)
        self.lin = [None] * nv.DL_NLIN
        for i, m in zip((nv.DL_RPH0, nv.DL_RPH1, nv.DL_RPH2), rph):
            self.lin[i] = L(m)
        for i, m in zip((nv.DL_QS0, nv.DL_QS1, nv.DL_QS2), qs):
            self.lin[i] = L(m)
        self.lin[nv.DL_INQK] = (ipw, 0, 512, ipb)
        self.lin[nv.DL_INV] = (ipw, 512, 256, ipb)
        self.lin[nv.DL_OUTP] = L(sa.attn.out_proj)
        self.lin[nv.DL_OPROJ] = L(ca.output_proj)
        self.lin[nv.DL_PE1] = L(pe[3])
        self.lin[nv.DL_FFN0], self.lin[nv.DL_FFN1] = L(f0), L(f1)
        self.lin[nv.DL_REG0], self.lin[nv.DL_REG1], self.lin[nv.DL_REG2] = L(reg_branch[0]), L(reg_branch[2]), L(reg_branch[4])
        self.lin[nv.DL_CLS0], self.lin[nv.DL_CLS1], self.lin[nv.DL_CLS2] = L(cls_branch[0]), L(cls_branch[3]), L(cls_branch[6])
        self.lin[nv.DL_IOU0], self.lin[nv.DL_IOU1], self.lin[nv.DL_IOU2] = L(iou_branch[0]), L(iou_branch[2]), L(iou_branch[4])
        for w, _, _, b in self.lin:
            if b is None or w.dtype != torch.float32:
                raise ValueError("bias / dtype")
        self.ln = [layer.norms[0], layer.norms[1], layer.norms[2], pe[1], pe[4], cls_branch[1], cls_branch[4]]
        if any(abs(n.eps - self.ln[0].eps) > 0 or not n.elementwise_affine or n.normalized_shape != (256,) for n in self.ln):
            raise ValueError("norms")
        self.attw, self.pe0 = ca.attention_weights, pe[0]
        self.ncls, self.code = cls_branch[6].out_features, reg_branch[4].out_features
        if self.ncls > 32 or self.code > 32 or iou_branch[4].out_features != 1 or pe[0].in_features != 3:
            raise ValueError("narrow widths")
        # dropout: one probability for the four residual / FFN sites, one for the attention weights
        ps = {float(sa.dropout_layer.p) if isinstance(sa.dropout_layer, nn.Dropout) else 0.0, float(ca.dropout.p),
              float(ffn.layers[0][2].p), float(ffn.layers[2].p)}
        if len(ps) != 1 or float(sa.proj_drop.p) != 0.0 or isinstance(ffn.dropout_layer, nn.Dropout):
            raise ValueError("dropout layout")
        self.p_drop, self.p_attn = ps.pop(), float(sa.attn_drop)
        self.modules = (sa, ca, ffn)

    def tensors(self):
        """Flat list of the parameter tensors handed to autograd (order fixed: see FusedLayerFn.backward)."""
        out = []
        for w, _, _, b in self.lin:
            out += [w, b]
        for n in self.ln:
            out += [n.weight, n.bias]
        out += [self.attw.weight, self.attw.bias, self.pe0.weight, self.pe0.bias]
        return out


class FusedDecoder:
    """bf16 weight copies + per-layer parameter blocks of one (decoder, head branches) pair."""

    def __init__(self, decoder, reg_branches, cls_branches, iou_branches):
        self.decoder = decoder
        self.specs = [LayerSpec(decoder, i, reg_branches[i], cls_branches[i], iou_branches[i]) for i in range(decoder.num_layers)]
        self._states = {}              # element type -> _State
        self.params = None
        self.rng = None
        self._dim_t = None

    # ---- weight copies ---------------------------------------------------------------------------------------------------
    def _build(self, dev, et):
        uniq = {}                      # (id(weight), first row) -> (w, r0, rows)
        for sp in self.specs:
            for i, (w, r0, rows, _) in enumerate(sp.lin):
                uniq.setdefault((id(w), r0), (w, r0, rows, i in _NARROW))
        total = 0
        plan = {}
        for key, (w, r0, rows, narrow) in uniq.items():
            k = w.shape[1]
            n_pad = 64 if narrow else rows
            n_pad_t = 32 if narrow else rows
            plan[key] = (total, total + n_pad * k, n_pad, n_pad_t)
            total += n_pad * k + k * n_pad_t
            total = (total + 127) // 128 * 128
        st = _State()
        st.wbuf = torch.zeros(total, dtype=et, device=dev)
        es = st.wbuf.element_size()
        descs = (nv.WPackDesc * len(uniq))()
        views = {}
        max_elems = 1
        for j, (key, (w, r0, rows, narrow)) in enumerate(uniq.items()):
            k = w.shape[1]
            o, ot, n_pad, n_pad_t = plan[key]
            d = descs[j]
            d.src = w.data_ptr() + r0 * k * 4
            d.dst = st.wbuf.data_ptr() + o * es
            d.dst_t = st.wbuf.data_ptr() + ot * es
            d.n, d.k, d.n_pad, d.n_pad_t = rows, k, n_pad, n_pad_t
            d.t_plain = 0                 # every transpose in fragment order (the narrow ones feed the MFMA narrow-dgrad step)
            views[key] = (d.dst, d.dst_t)
            max_elems = max(max_elems, n_pad * k, k * n_pad_t)
        st.ndesc, st.max_elems = len(uniq), max_elems
        st.descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        if self._dim_t is None or self._dim_t.device != dev:
            d = torch.arange(128, dtype=torch.float32, device=dev)
            self._dim_t = 10000 ** (2 * torch.div(d, 2, rounding_mode="floor") / 128)
        st.params = []
        for sp in self.specs:
            p = nv.DecLayerParams()
            for i, (w, r0, rows, b) in enumerate(sp.lin):
                dst, dst_t = views[(id(w), r0)]
                p.w[i], p.wt[i] = dst, dst_t
                p.b[i] = b.data_ptr() + r0 * 4
            for i, n in enumerate(sp.ln):
                p.ln_g[i], p.ln_b[i] = n.weight.data_ptr(), n.bias.data_ptr()
            p.attw_w, p.attw_b = sp.attw.weight.data_ptr(), sp.attw.bias.data_ptr()
            p.pe0_w, p.pe0_b = sp.pe0.weight.data_ptr(), sp.pe0.bias.data_ptr()
            p.dim_t = self._dim_t.data_ptr()
            st.params.append(p)
        if self.rng is None or self.rng.device != dev:
            self.rng = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFF], dtype=torch.int64, device=dev)
        return st

    def xyz_cols(self, dev):
        """code columns of the (x, y, z) offsets (ref :194-202), cached on the device: no host-to-device copy inside a graph capture"""
        c = getattr(self, "_cols", None)
        if c is None or c.device != dev:
            c = self._cols = torch.tensor([0, 1, 4], device=dev)
        return c

    def refresh(self, dev, et=torch.bfloat16):
        """Copies of the current master weights in the element type `et` (one launch); rebuilt when a parameter moved (e.g. into the
        trainer's flat buffer).  bf16: rounded copies; f32: the zero-padded / transposed layouts the kernels read."""
        key = (str(dev),) + tuple(t.data_ptr() for sp in self.specs for t in sp.tensors())
        st = self._states.get(et)
        if st is None or st.key != key:
            st = self._states[et] = self._build(dev, et)
            st.key = key
        self.params = st.params
        nv.wpack(st.descs_dev, st.ndesc, st.max_elems, et)
        return st


class _State:
    """weight copies + parameter blocks of one element type"""
    key = None


def element_type(decoder, query, value, reg_branches, head_branches):
    """torch dtype the fused kernels run this call in (bf16 under bf16 autocast, f32 for plain f32 tensors), or None when the fused
    path does not cover it."""
    if not (ENABLED and query.is_cuda):
        return None
    if reg_branches is None or head_branches is None or value.dim() != 5 or value.shape[1] != 256 or query.shape[-1] != 256:
        return None
    if len(reg_branches) != decoder.num_layers or len({id(m) for m in reg_branches}) != decoder.num_layers:
        return None
    if torch.is_autocast_enabled():
        return torch.bfloat16 if torch.get_autocast_dtype("cuda") == torch.bfloat16 else None
    return torch.float32 if (query.dtype == torch.float32 and value.dtype == torch.float32) else None


def eligible(decoder, query, value, reg_branches, head_branches):
    return element_type(decoder, query, value, reg_branches, head_branches) is not None


class FusedLayerFn(torch.autograd.Function):
    """One decoder layer.  Inputs: x f32 [M,256]; xc = x in the element type (bf16 copy, or None: made here; f32 mode: x itself);
    ref f32 [M,3] logits; rows [B*D*H*W,256] in the element type; meta = (FusedDecoder, layer, dims, accum[, element type])."""

    @staticmethod
    def forward(ctx, x, xc, ref, rows, meta, *ptensors):
        fd, lid, dims_t, accum = meta[:4]
        et = meta[4] if len(meta) > 4 else torch.bfloat16
        sp = fd.specs[lid]
        B, qps, nq, D, H, W = dims_t
        M = x.shape[0]
        x = x.contiguous()
        xc = x if et == torch.float32 else (x.to(et) if xc is None else xc)
        ref = ref.contiguous().float()
        assert rows.dtype == et and rows.is_contiguous()
        d = nv.DecLayerDims()
        d.m, d.nq, d.qps, d.batch, d.dz, d.dy, d.dx = M, nq, qps, B, D, H, W
        d.ncls, d.code, d.has_qs, d.need_dref, d.layer = sp.ncls, sp.code, int(lid > 0), 0, lid
        d.dtype = nv.dt_code(et)
        train = fd.decoder.training
        d.p_attn = sp.p_attn if train else 0.0
        d.p_drop = sp.p_drop if train else 0.0
        d.ln_eps = float(sp.ln[0].eps)
        so, go = nv.decoder_layer_slots(M, sp.ncls, sp.code, et)
        save = torch.empty(so["_total"], dtype=torch.uint8, device=x.device)
        if POISON:
            save.fill_(255)
        params = fd._states[et].params[lid]
        x_out, xc_out, reg, cls, iou = nv.decoder_layer_fwd(params, d, x, xc, ref, rows, fd.rng, save)
        ctx.save_for_backward(x, xc, ref, rows, xc_out, save)
        ctx.meta, ctx.dims, ctx.et, ctx.params = meta, d, et, params
        ctx.wids = []
        from .transformer import _Deferred
        for w, r0, rows_, b in sp.lin:
            if r0 == 0:
                _Deferred.uses[id(w)] = _Deferred.uses.get(id(w), 0) + 1
                _Deferred.fused_uses[id(w)] = _Deferred.fused_uses.get(id(w), 0) + 1
                _Deferred.params[id(w)] = (w, b)
        for t in (sp.attw.weight, sp.pe0.weight):
            _Deferred.uses[id(t)] = _Deferred.uses.get(id(t), 0) + 1
        for n in sp.ln:
            _Deferred.uses[id(n.weight)] = _Deferred.uses.get(id(n.weight), 0) + 1
        # f32 mode: the layer state IS its own compute copy (xc_out aliases x_out inside): hand autograd an empty stand-in instead of
        # two outputs over one storage; the next layer takes its x as xc
        xc_ret = xc_out if et == torch.bfloat16 else x_out.new_empty(0)
        ctx.mark_non_differentiable(xc_ret)
        return x_out, xc_ret, reg, cls, iou

    @staticmethod
    def backward(ctx, dx_out, _dxc, dreg, dcls, diou):
        from .transformer import _Deferred
        x, xc, ref, rows, xc_out, save = ctx.saved_tensors
        fd, lid, dims_t, accum = ctx.meta[:4]
        et = ctx.et
        sp, d = fd.specs[lid], ctx.dims
        M = x.shape[0]
        dev = x.device
        d.need_dref = int(ctx.needs_input_grad[2])
        so, go = nv.decoder_layer_slots(M, sp.ncls, sp.code, et)
        grad = torch.empty(go["_total"], dtype=torch.uint8, device=dev)
        if POISON:
            grad.fill_(255)
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else t.contiguous().float()
        dreg, dcls, diou = z(dreg, (M, sp.code)), z(dcls, (M, sp.ncls)), z(diou, (M,))
        dx_out = None if dx_out is None else dx_out.contiguous().float()
        want_dv = ctx.needs_input_grad[3]
        # bf16 kernels: the volume gradient accumulates in bf16 through packed atomics (two channels per atomic; no f32 volume to zero and
        # cast).  A cell collects a handful of contributions at most, each already rounded to bf16 on the way into the conv backward.
        acc_dt = torch.bfloat16 if (PK_SCATTER and rows.dtype == torch.bfloat16) else torch.float32
        if accum is not None:
            if accum.buf is None:
                accum.buf = torch.zeros(rows.shape, dtype=acc_dt, device=dev)
            dvalue = accum.buf
        else:
            dvalue = torch.zeros(rows.shape, dtype=acc_dt, device=dev)
        dx, dref = nv.decoder_layer_bwd(ctx.params, d, x, xc, ref, rows, fd.rng, xc_out, save, dx_out, dreg, dcls, diou, dvalue, grad)
        if DEBUG_KEEP is not None:
            DEBUG_KEEP.append(grad)
        drows = None
        if accum is not None:
            accum.remaining -= 1
            if accum.remaining == 0:
                drows = accum.buf.to(rows.dtype) if want_dv else None
                accum.buf = None
        elif want_dv:
            drows = dvalue.to(rows.dtype)
        f32 = torch.float32
        S = lambda name, cols, dt=et: nv.slot_view(save, so[name], M, cols, dt)
        Gv = lambda name, cols, dt=et: nv.slot_view(grad, go[name], M, cols, dt)
        if d.need_dref:      # the sine-embedding path of the reference-point gradient (ref_point_head's input)
            dref = dref + nv.sine_embed_bwd(ref, fd._dim_t, Gv("SINE", 384))
        # ---- parameter gradients: (dY, X) per linear --------------------------------------------------------------------------
        pairs = {
            nv.DL_RPH0: (Gv("RPH1", 256), S("SINE", 384)), nv.DL_RPH1: (Gv("RPH2", 256), S("RPH1", 256)),
            nv.DL_RPH2: (Gv("RAW", 256), S("RPH2", 256)),
            nv.DL_INQK: (Gv("DQK", 512), S("QKIN", 256)), nv.DL_INV: (Gv("DV", 256), xc),
            nv.DL_OUTP: (Gv("O2", 256), S("O", 256)), nv.DL_OPROJ: (Gv("OUT", 256), S("GATED", 256)),
            nv.DL_PE1: (Gv("UPE1", 256), S("PEH0", 256)), nv.DL_FFN0: (Gv("FFH", 512), S("X2C", 256)),
            nv.DL_FFN1: (Gv("F", 256), S("FFH", 512)), nv.DL_REG0: (Gv("R1", 256), xc_out), nv.DL_REG1: (Gv("R2", 256), S("R1", 256)),
            nv.DL_REG2: (Gv("REGO", sp.code), S("R2", 256)), nv.DL_CLS0: (Gv("C1U", 256), xc_out),
            nv.DL_CLS1: (Gv("C2U", 256), S("C1", 256)), nv.DL_CLS2: (Gv("CLSO", sp.ncls), S("C2", 256)),
            nv.DL_IOU0: (Gv("I1", 256), xc_out), nv.DL_IOU1: (Gv("I2", 256), S("I1", 256)), nv.DL_IOU2: (Gv("IOUO", 1), S("I2", 256)),
        }
        if lid > 0:
            pairs.update({nv.DL_QS0: (Gv("QS1", 256), xc), nv.DL_QS1: (Gv("QS2", 256), S("QS1", 256)), nv.DL_QS2: (Gv("QS", 256), S("QS2", 256))})
        # the deferred / batched parameter-gradient launches are bf16 kernels; parity mode computes every product right here on the
        # exact-f32 weight-gradient kernel (u3d_spconv_wgrad, one offset)
        deferred_ok = _Deferred.active and et == torch.bfloat16
        # `parity` precision (f32 layer, split-bf16 products elsewhere in the model): the wide dY^T X products as THREE bf16 products of
        # hi / lo planes (dyh.xh + dyl.xh + dyh.xl, f32 accumulation, ~2^-16 relative) in the batched bf16 weight-gradient launch -
        # consecutive batch slots with one output are summed by the kernel.  54 launches of the exact-f32 kernel (2.7 ms per step: it
        # walks 7 200 rows on one offset) become one launch per shape.
        split_wg = et == torch.float32 and getattr(fd.decoder, "split_f32_wgrad", False)
        planes = {}
        # every tensor that needs planes is split by ONE launch (u3d_split_rows_batch, strided slot views read in place) right before the
        # batched weight-gradient launch that consumes them: a copy + a u3d_split_rows_f32 launch per tensor was ~90 launches per step
        to_split = []
        ok = lambda t: t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.shape[1] % 4 == 0 and t.data_ptr() % 16 == 0      # noqa: E731

        def hi_lo(t):
            key = id(t)
            if key not in planes:
                t2 = t if ok(t) else t.contiguous()
                pl = torch.empty((2 * t2.shape[0], t2.shape[1]), dtype=torch.bfloat16, device=dev)
                to_split.append((t2, pl))
                planes[key] = (t, pl[:t2.shape[0]], pl[t2.shape[0]:])
            return planes[key][1], planes[key][2]
        grads = []
        wgrad_now = {}              # (n, k) -> list of (dy, x, out)
        sums_now = []               # (matrix, out vector)
        new = lambda *shape: torch.empty(shape, dtype=f32, device=dev)

        def slot(param, deferred):
            """Gradient tensor handed to autograd for `param`: under TrainStep, when the value is written later by the flush (so
            nothing else accumulates into it), the parameter's own slice of the flat gradient buffer - the step's packing copy then
            has nothing to move for it; else a fresh tensor."""
            v = getattr(param, "_u3d_grad_view", None) if deferred else None
            # (a fresh alias: AccumulateGrad keeps an incoming tensor as .grad only if nobody else holds that tensor object)
            return v.view(v.shape) if (v is not None and v.dtype == f32 and v.is_contiguous()) else new(*param.shape)

        def skinny_now(dy, xin, dw):
            """dW = dY^T X for a product with a side <= 16, written into dw ([n, k] view)"""
            if et == torch.bfloat16:
                sums_now.append((nv.skinny_wgrad_partial(dy, xin), dw.view(-1)))
            else:
                dw.copy_(_wgrad_f32(dy, xin))
        inproj_dw = inproj_db = None
        for i, (w, r0, rows_, b) in enumerate(sp.lin):
            if i not in pairs:                                   # query_scale in the first layer: unused
                grads += [None, None]
                continue
            dy, xin = pairs[i]
            n, k = rows_, w.shape[1]
            if i in (nv.DL_INQK, nv.DL_INV):
                if inproj_dw is None:
                    dfr = deferred_ok and _Deferred.uses.get(id(w), 0) == 1
                    inproj_dw, inproj_db = slot(w, dfr), slot(b, dfr)
                dw, db = inproj_dw[r0:r0 + rows_], inproj_db[r0:r0 + rows_]
            else:
                dfr = deferred_ok and _Deferred.uses.get(id(w), 0) == 1
                dw, db = slot(w, dfr), slot(b, dfr)
            single = _Deferred.uses.get(id(w), 0) == 1
            # a linear SHARED by several layers (ref_point_head, query_scale) whose every use is a fused layer: all its (dY, X) pairs are
            # queued and the flush sums them into ONE gradient (consecutive batch slots with one output, u3d_wgrad_batched_bf16) - the
            # first backward to get here hands autograd the placeholder, the others contribute None instead of a tensor to be added
            shared = (SHARED_DEFER and deferred_ok and not single and i not in _NARROW and i not in (nv.DL_INQK, nv.DL_INV)
                      and _Deferred.fused_uses.get(id(w), 0) == _Deferred.uses.get(id(w), 0))
            if shared:
                if id(w) in _Deferred.shared_seen:
                    dw = db = None
                else:
                    _Deferred.shared_seen.add(id(w))
                    dw, db = slot(w, True), slot(b, True)
                _Deferred.items.append((dy, xin, id(w), r0, r0 + rows_, True))
                grads += [dw, db]
                continue
            if i in _NARROW:
                if deferred_ok and single:
                    _Deferred.skinny.append((dy, xin, w))        # the product itself waits for the flush: one launch for all layers
                    _Deferred.sum_items.append((dy, b))
                else:
                    skinny_now(dy, xin, dw)
                    sums_now.append((dy, db))
            elif deferred_ok and single:
                _Deferred.items.append((dy, xin, id(w), r0, r0 + rows_, True))
            elif et == torch.bfloat16:
                wgrad_now.setdefault((n, k), []).append((dy, xin, dw))
                sums_now.append((dy, db))
            elif split_wg and n % 64 == 0 and k % 64 == 0:
                (dyh, dyl), (xh, xl) = hi_lo(dy), hi_lo(xin)
                wgrad_now.setdefault((n, k), []).extend([(dyh, xh, dw), (dyl, xh, dw), (dyh, xl, dw)])
                sums_now.append((dy, db))
            else:
                dw.copy_(_wgrad_f32(dy, xin))
                sums_now.append((dy, db))
            if i == nv.DL_INV:
                grads += []                                      # in_proj weight / bias were emitted with INQK
            elif i == nv.DL_INQK:
                grads += [inproj_dw, inproj_db]
            else:
                grads += [dw, db]
        nb = nv.decoder_blocks(M, et)
        lnp = nv.slot_view(grad, go["LNP"], nv.DL_NLN * 2 * nb, 256, f32)
        for j, nmod in enumerate(sp.ln):
            dfr = deferred_ok and _Deferred.uses.get(id(nmod.weight), 0) == 1
            dg, db = slot(nmod.weight, dfr), slot(nmod.bias, dfr)
            mg, mb = lnp[(2 * j) * nb:(2 * j + 1) * nb], lnp[(2 * j + 1) * nb:(2 * j + 2) * nb]
            if deferred_ok and _Deferred.uses.get(id(nmod.weight), 0) == 1:
                _Deferred.sum_items += [(mg, nmod.weight), (mb, nmod.bias)]
            else:
                sums_now += [(mg, dg), (mb, db)]
            grads += [dg, db]
        # attention_weights (256 -> 1) and the position encoder's first layer (3 -> 256): skinny products
        for dy, xin, mod in ((Gv("WL", 1), S("QP", 256), sp.attw), (Gv("P0", 256), ref.to(et), sp.pe0)):
            dfr = deferred_ok and _Deferred.uses.get(id(mod.weight), 0) == 1
            dw, db = slot(mod.weight, dfr), slot(mod.bias, dfr)
            if deferred_ok and _Deferred.uses.get(id(mod.weight), 0) == 1:
                _Deferred.params[id(mod.weight)] = (mod.weight, mod.bias)
                _Deferred.skinny.append((dy, xin.contiguous(), mod.weight))
                _Deferred.sum_items.append((dy, mod.bias))
            else:
                skinny_now(dy, xin.contiguous(), dw)
                sums_now.append((dy, db))
            grads += [dw, db]
        if to_split:
            nv.split_rows_batch_into([a for a, _ in to_split], [b_ for _, b_ in to_split])
        for (n, k), lst in wgrad_now.items():
            nv.wgrad_batched([a for a, _, _ in lst], [b_ for _, b_, _ in lst], [c for _, _, c in lst])
        groups = {}
        for mat, out in sums_now:
            groups.setdefault((mat.shape[0], mat.shape[1], str(mat.dtype)), []).append((mat, out))
        for lst in groups.values():
            nv.colsum_batched([a for a, _ in lst], [b_ for _, b_ in lst])
        # in_proj appears twice in sp.lin (INQK, INV) but once in the tensor list: the INV entry added nothing above
        return (dx, None, dref, drows, None) + tuple(grads)


def _wgrad_f32(dy, xin):
    """dY^T X for f32 [M, n] / [M, k] on the exact-f32 MFMA weight-gradient kernel (u3d_spconv_wgrad, one offset): f32 [n, k].
    The kernel wants channel counts that are multiples of 4: narrow sides (1, 3, 10 columns) are zero-padded."""
    n, k = dy.shape[1], xin.shape[1]
    pad = lambda t: t if t.shape[1] % 4 == 0 else torch.nn.functional.pad(t, (0, 4 - t.shape[1] % 4))
    dyp, xp = pad(dy).contiguous(), pad(xin).contiguous()
    md = nv.count_tensor(dy.shape[0], dy.device)
    return nv.spconv_wgrad(dyp, xp, None, md, 1).view(dyp.shape[1], xp.shape[1])[:n, :k]


def d_blocks(m, et=torch.bfloat16):
    return nv.decoder_blocks(m, et)


def tensor_list(sp):
    """Parameter tensors in the order FusedLayerFn.backward emits their gradients."""
    out = []
    for i, (w, r0, rows, b) in enumerate(sp.lin):
        if i == nv.DL_INV:
            continue
        out += [w, b]
    for n in sp.ln:
        out += [n.weight, n.bias]
    out += [sp.attw.weight, sp.attw.bias, sp.pe0.weight, sp.pe0.bias]
    return out


FUSED_REFINE_DECODE = True


class _RefineDecode(torch.autograd.Function):
    """The decoder loop's reference-point refinement and the head's box decode of one layer in one launch (u3d_refine_decode_fwd;
    ref: uni3detr_transformer.py:194-202, uni3detr_head.py:463-490).  reg f32 [M,code], ref_in f32 [M,3] logits (detached), ref_s
    [M,3] = the same point in sigmoid space (differentiable for the first layer: init_reference = sigmoid(refpoint_embed));
    ref_out / ref_sig: [M,3] slices of the stacked per-layer buffers the launch writes.  -> decoded box codes f32 [M,code]."""

    @staticmethod
    def forward(ctx, reg, ref_in, ref_s, pc_range, ref_out, ref_sig):
        reg, ref_s = reg.contiguous(), ref_s.contiguous()
        ctx.save_for_backward(reg, ref_s)
        ctx.pc_range = pc_range
        return nv.refine_decode_fwd(reg, ref_in.contiguous(), ref_s, pc_range, ref_out, ref_sig)

    @staticmethod
    def backward(ctx, dout):
        reg, ref_s = ctx.saved_tensors
        dtmp, dref = nv.box_decode_bwd(reg, ref_s, dout.contiguous().float(), ctx.pc_range, want_dref=ctx.needs_input_grad[2])
        return dtmp, None, dref, None, None, None


def run(fd, query, ref_logits, value, group, et=torch.bfloat16, pc_range=None, ref_sig=None):
    """query [B,N,256] f32, ref_logits [B,N,3], value [B,256,D,H,W] -> per-layer lists (states [B,N,256] f32, refs, reg, cls, iou).
    et: element type of the kernels (bf16: throughput mode; f32: parity mode).
    With pc_range (the head's point-cloud range) and ref_sig (sigmoid(ref_logits), the transformer's init_reference) the layer tail runs
    as ONE launch (_RefineDecode) and three more results come back: the decoded box codes per layer and the stacked reference points
    after each layer as logits / in sigmoid space ([L,B,N,3] buffers the launches wrote into)."""
    from .transformer import ValueGradAccum
    B, N, Cc = query.shape
    _, _, D, H, W = value.shape
    rows = value.permute(0, 2, 3, 4, 1).reshape(-1, Cc)
    rows = rows if rows.dtype == et else rows.to(et)
    rows = rows if rows.is_contiguous() else rows.contiguous()
    dev = query.device
    fd.refresh(dev, et)
    if fd.decoder.training:
        fd.rng.add_(0x9E3779B1)                         # new dropout stream every step (captured: advances on every graph replay)
    L = fd.decoder.num_layers
    accum = ValueGradAccum(L) if (value.requires_grad and torch.is_grad_enabled()) else None
    x = query.reshape(B * N, Cc).float()
    xc = None
    ref = ref_logits.reshape(B * N, 3).float()
    states, refs, regs, clss, ious = [], [], [], [], []
    cols = None
    fused_tail = FUSED_REFINE_DECODE and pc_range is not None and ref_sig is not None and ref_sig.dtype == torch.float32
    if fused_tail:
        refs_buf = torch.empty((L, B * N, 3), dtype=torch.float32, device=dev)
        sig_buf = torch.empty((L, B * N, 3), dtype=torch.float32, device=dev)
        rs = ref_sig.reshape(B * N, 3)
        coords = []
    for lid in range(L):
        sp = fd.specs[lid]
        meta = (fd, lid, (B, N, group, D, H, W), accum, et)
        x, xc, reg, cls, iou = FusedLayerFn.apply(x, xc, ref, rows, meta, *tensor_list(sp))
        if fused_tail:
            coords.append(_RefineDecode.apply(reg, ref.detach(), rs, pc_range, refs_buf[lid], sig_buf[lid]).view(B, N, -1))
            ref, rs = refs_buf[lid], sig_buf[lid]
        else:
            if cols is None:
                cols = fd.xyz_cols(dev)
            ref = (ref.detach() + reg.detach().index_select(-1, cols)).detach()
        states.append(x.view(B, N, Cc))
        refs.append(ref.view(B, N, 3))
        regs.append(reg.view(B, N, -1))
        clss.append(cls.view(B, N, -1))
        ious.append(iou.view(B, N, 1))
    if fused_tail:
        return states, refs, regs, clss, ious, coords, refs_buf.view(L, B, N, 3), sig_buf.view(L, B, N, 3)
    return states, refs, regs, clss, ious
