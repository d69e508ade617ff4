"""Per-step low-precision copies ("shadows") of the fp32 master parameters.

In bf16 mode every conv / linear used to cast its own weight (and bias) on every forward: ~170 cast launches per step of a few
microseconds each.  A ShadowSet keeps persistent bf16 tensors per parameter and refreshes ALL of them at the top of the training
step (one multi-tensor copy for the linears; the conv weights are re-laid-out at the same time); `compute_copy()` /
`conv_weights()` hand the shadows out while the set is active and fall back to an on-the-spot cast otherwise (eval, modules
called on their own, parameters modified since the refresh).

Conv weights get TWO shadows, both [K, ., .] with K = kD*kH*kW:
    kio [K, Cin, Cout]  - what dgrad reads (n-major for the input-gradient GEMM) and the k-major forward kernels
    koi [K, Cout, Cin]  - n-major for the forward GEMM: the LDS-DMA kernel (k_igemm_glds) stages both operands row-linearly
from either checkpoint layout: "dhwio" = [kD,kH,kW,Cin,Cout] (mmcv spconv-1.x, SparseEncoderHD) or "oidhw" = nn.Conv3d.
"""
import contextlib

import torch

_ACTIVE = [False]


def _conv_views(p, layout):
    """(kio_view, koi_view) of a conv weight in `layout`, as (possibly strided) views [kD,kH,kW,.,.]."""
    if layout == "dhwio":
        return p, p.transpose(-1, -2)
    if layout == "oidhw":
        return p.permute(2, 3, 4, 1, 0), p.permute(2, 3, 4, 0, 1)
    raise ValueError(layout)


class ShadowSet:
    def __init__(self, params, dtype, conv_layouts=None, flat=None):
        """params: parameters to shadow as they are; conv_layouts: {parameter: "dhwio" | "oidhw"} for 5-D conv weights.
        flat: (flat f32 buffer, {id(parameter): element offset}) when every parameter is a view of ONE flat buffer (the training
        step's layout, uni3detr_amd/trainer.py): the refresh is then two launches - a flat cast (the shadows are views of its
        output) and one batched re-layout for the conv-weight layouts that are not the checkpoint's own."""
        self.dtype = dtype
        conv_layouts = conv_layouts or {}
        self.flat = None
        if (flat is not None and dtype == torch.bfloat16 and flat[0].is_cuda
                and all(id(p) in flat[1] for p in list(params) + list(conv_layouts))):
            self._init_flat(params, conv_layouts, flat)
            return
        conv_ids = {id(p) for p in conv_layouts}
        self.params = [p for p in params if p.is_floating_point() and p.dtype != dtype and id(p) not in conv_ids]
        self.shadows = [torch.empty_like(p, dtype=dtype) for p in self.params]
        for p, s in zip(self.params, self.shadows):
            p._u3d_shadow = [s, -1]
        self.conv_params, self._conv_src, self._conv_dst = [], [], []
        for p, layout in conv_layouts.items():
            kio_v, koi_v = _conv_views(p.detach(), layout)
            kio = torch.empty(kio_v.shape, dtype=dtype, device=p.device)
            koi = torch.empty(koi_v.shape, dtype=dtype, device=p.device)
            self.conv_params.append(p)
            self._conv_src += [kio_v, koi_v]
            self._conv_dst += [kio, koi]
            p._u3d_conv_shadow = [kio.view(-1, kio.shape[-2], kio.shape[-1]), koi.view(-1, koi.shape[-2], koi.shape[-1]), -1, layout]

    def _init_flat(self, params, conv_layouts, flat):
        from . import native as nv
        src, offsets = flat
        self.flat = src
        self.flat_shadow = torch.empty(src.numel(), dtype=torch.bfloat16, device=src.device)
        self.params = [p for p in params if p.is_floating_point() and p.dtype != self.dtype and id(p) not in {id(q) for q in conv_layouts}]
        self.shadows = []
        for p in self.params:
            o = offsets[id(p)]
            sh = self.flat_shadow[o:o + p.numel()].view(p.shape)
            self.shadows.append(sh)
            p._u3d_shadow = [sh, -1]
        self.conv_params = list(conv_layouts)
        descs, n_perm = [], 0

        def add(src_off, n, rows, cols, sk, sr, sc):
            nonlocal n_perm
            d = dict(src_off=src_off, dst_off=n_perm, n=n, rows=rows, cols=cols, stride_k=sk, stride_r=sr, stride_c=sc)
            descs.append(d)
            n_perm += (n + 63) // 64 * 64
            return d

        pending = []
        for p, layout in conv_layouts.items():
            o, n = offsets[id(p)], p.numel()
            if layout == "dhwio":
                K, ci, co = p.shape[0] * p.shape[1] * p.shape[2], p.shape[3], p.shape[4]
                kio = self.flat_shadow[o:o + n].view(K, ci, co)                   # the checkpoint layout IS [K,Cin,Cout]
                pending.append((p, layout, kio, None, add(o, n, co, ci, ci * co, 1, co), (K, co, ci)))
            elif layout == "oidhw":
                co, ci, K = p.shape[0], p.shape[1], p.shape[2] * p.shape[3] * p.shape[4]
                pending.append((p, layout, None, add(o, n, ci, co, 1, K, ci * K), add(o, n, co, ci, 1, ci * K, K), (K, ci, co)))
            else:
                raise ValueError(layout)
        self.perm = torch.empty(max(n_perm, 64), dtype=torch.bfloat16, device=src.device)
        for p, layout, kio, d_kio, d_koi, dims in pending:
            if kio is None:
                K, ci, co = dims
                kio = self.perm[d_kio["dst_off"]:d_kio["dst_off"] + p.numel()].view(K, ci, co)
                koi = self.perm[d_koi["dst_off"]:d_koi["dst_off"] + p.numel()].view(K, co, ci)
            else:
                K, co, ci = dims
                koi = self.perm[d_koi["dst_off"]:d_koi["dst_off"] + p.numel()].view(K, co, ci)
            p._u3d_conv_shadow = [kio, koi, -1, layout]
        # 64 -> 64 channel, 27-offset weights (the SubM blocks of the stride-4 stage): MFMA-fragment-packed copies of both layouts
        # for the halo kernel (csrc/subm_halo.hip), all of them in ONE launch per refresh
        self._halo_plans = []
        # (kvol, channels, layout): the sparse encoder's 64- / 128-channel SubM weights
        for k, c, lay in ((27, 64, "dhwio"), (27, 128, "dhwio")):
            pairs = []
            for p in self.conv_params:
                kio, koi = p._u3d_conv_shadow[0], p._u3d_conv_shadow[1]
                if tuple(kio.shape) == (k, c, c) and conv_layouts[p] == lay:
                    pk = torch.empty((2, k, c, c), dtype=torch.bfloat16, device=src.device)
                    pairs += [(koi, pk[0]), (kio, pk[1])]
                    p._u3d_halo_pack = [pk[0], pk[1], -1]
            if pairs:
                self._halo_plans.append(nv.subm_halo_wpack_plan(pairs, src.device))
        self._plan = nv.permute_plan(descs, src.device)

    def refresh(self):
        if self.flat is not None:
            from . import native as nv
            nv.cast_bf16(self.flat, self.flat_shadow)
            nv.permute_bf16_batched(self.flat_shadow, self.perm, self._plan)
            for plan in self._halo_plans:
                nv.subm_halo_wpack_batched(plan)
            for p in self.params:
                p._u3d_shadow[1] = p._version
            for p in self.conv_params:
                p._u3d_conv_shadow[2] = p._version
                hp = getattr(p, "_u3d_halo_pack", None)
                if hp is not None:
                    hp[2] = p._version
            return
        with torch.no_grad():
            if self.shadows:
                torch._foreach_copy_(self.shadows, self.params)
            if self._conv_dst:
                torch._foreach_copy_(self._conv_dst, self._conv_src)        # strided sources: cast + re-layout in one pass per tensor
        for p in self.params:
            p._u3d_shadow[1] = p._version
        for p in self.conv_params:
            p._u3d_conv_shadow[2] = p._version

    @contextlib.contextmanager
    def active(self, refresh=True):
        """Refresh, then let compute_copy() / conv_weights() use the shadows for the duration of the block (one training forward)."""
        if refresh:
            self.refresh()
        prev = _ACTIVE[0]
        _ACTIVE[0] = True
        try:
            yield self
        finally:
            _ACTIVE[0] = prev


def halo_packs(p, kio, koi):
    """(packed koi, packed kio or None) of a [27, 64, 64] / [27, 128, 128] conv weight for the halo kernels: the refresh's copies while the shadow set is
    active and fresh, else packed on the spot (the transposed one only on demand: halo_pack_one)."""
    if _ACTIVE[0]:
        hp = getattr(p, "_u3d_halo_pack", None)
        if hp is not None and hp[2] == p._version:
            return hp[0], hp[1]
    from . import native as nv
    return nv.subm_halo_wpack(koi), None


def compute_copy(p, dtype):
    """`p` in the compute dtype: its shadow when fresh (no launch), else a cast."""
    if p is None or p.dtype == dtype:
        return p
    if _ACTIVE[0]:
        sh = getattr(p, "_u3d_shadow", None)
        if sh is not None and sh[0].dtype == dtype and sh[1] == p._version:
            return sh[0]
    return p.detach().to(dtype)


def conv_weights(p, layout, dtype, want_koi=True):
    """(kio [K,Cin,Cout], koi [K,Cout,Cin] or None) of conv weight `p` (checkpoint `layout`) in `dtype`, contiguous."""
    if _ACTIVE[0]:
        sh = getattr(p, "_u3d_conv_shadow", None)
        if sh is not None and sh[0].dtype == dtype and sh[2] == p._version and sh[3] == layout:
            return sh[0], sh[1]
    kio_v, koi_v = _conv_views(p.detach(), layout)
    kio = kio_v.to(dtype).contiguous()
    koi = koi_v.to(dtype).contiguous() if want_koi else None
    return kio.view(-1, kio.shape[-2], kio.shape[-1]), (None if koi is None else koi.view(-1, koi.shape[-2], koi.shape[-1]))
