"""Host side of the sparse levels: geometry (BitGrid / neighbour tables) and autograd wrappers around the HIP
sparse-conv / BatchNorm1d / dense() kernels.  Every op calls libu3d_hip.so; nothing here has a torch fallback.

Replaces spconv's SparseConvTensor / indice_key machinery used by the reference encoder
(ref: models/pts_encoder/sparse_encoder_hd.py:106-138).
"""
import contextlib
import os

import torch

from . import native as nv
from .shadow import conv_weights, halo_packs

K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)


class Level:
    """One sparse resolution: coordinates in block-major rank order + its SubM tables (built lazily, once per step —
    the reference rebuilds them for each of its 16 block convs because SparseBasicBlock carries no indice_key)."""

    def __init__(self, grid, coords, n, n_dev):
        self.grid, self.coords, self.n, self.n_dev = grid, coords, n, n_dev
        self.dims, self.batch = grid.dims, grid.batch
        self._subm = None
        self._halo = None

    def halo(self):
        """Distinct-row lists of the 128-row tiles (native.SubmHalo) for the 64 -> 64 convs of this level: built on first use."""
        if self._halo is None:
            h = nv.SubmHalo(self.subm_tables()[0], self.n_dev, self.n) if self.n <= nv.SubmHalo.MAX_ROWS else None
            self._halo = h if (h is not None and h.ok) else False      # False: too many rows for the build's LDS bitmap
        return self._halo or None

    def subm_tables(self):
        if self._subm is None:
            fwd = self.grid.nbr_table(self.coords, self.n_dev, K3, S1, P1, 0)
            # the transposed table of a SubM layer = the forward table with the offsets reversed (native.RevNbr): not built
            bwd = nv.RevNbr(fwd) if REV_SUBM_TABLE else self.grid.nbr_table(self.coords, self.n_dev, K3, S1, P1, 1)
            self._subm = (fwd, bwd)
        return self._subm


class ResidualToken:
    """Side channel between the two autograd nodes of a residual block (ref: mmdet3d SparseBasicBlock, `out += identity`): the
    BatchNorm that adds the identity deposits the identity branch's gradient here in ITS backward, and the block's first conv - whose
    backward runs later, and whose input IS the identity - sums it into its input gradient in the kernel epilogue
    (u3d_igemm_fwd_add_bf16).  Autograd would add the two gradients with a separate element-wise pass over both tensors."""

    def __init__(self):
        self.dres = None


class BnGradToken:
    """Side channel from a BatchNorm (+ ReLU) layer to the ONE conv that consumes its output (or, in a residual block, to the block's
    first conv, whose input gradient already includes the identity branch's: ResidualToken): that conv's input-gradient launch writes
    the BatchNorm's dy, so its epilogue reduces the BatchNorm-backward sums (sum g, sum g * xhat) per row tile on the way out
    (u3d_igemm_dgrad_bnstats_bf16 / the halo kernel) and leaves them here; the BatchNorm's backward then skips its own pass over dy
    and x (u3d_bn_bwd_stats).  The caller vouches that no other consumer of the BatchNorm's output contributes a gradient."""

    def __init__(self):
        self.epi = None          # native.BnEpi of the producing layer (+ the tensors it points at, kept alive)
        self.keep = None
        self.partial = None      # (f64 [tiles, 2, C], tile_rows) left by the consumer's backward
        self.c = 0

    def fill(self, x, y, mean, invstd, gamma, beta, relu):
        self.keep = (x, y, mean, invstd, gamma, beta)
        self.epi = nv.BnEpi.of(x, y, mean, invstd, gamma, beta, relu)
        self.c = x.shape[1]


class FanoutToken:
    """Side channel between the first convs of `n` branches that take the SAME input (SECOND3D with is_cascade=False): autograd
    would add their input gradients with n - 1 element-wise passes over the full tensor.  Instead each conv's backward adds what the
    branches before it left here (kernel epilogue, u3d_igemm_fwd_add_bf16) and keeps the partial sum; only the last one hands the
    total to autograd, the others return no gradient for the input."""

    def __init__(self, n):
        self.n = self.remaining = n
        self.acc = None

    def step(self, din):
        """din already contains self.acc.  -> what this conv returns to autograd for its input."""
        self.remaining -= 1
        if self.remaining > 0:
            self.acc = din
            return None
        self.acc, self.remaining = None, self.n
        return din


class ConvGeom:
    """Tables of one convolution instance: forward (output-stationary), transposed (input-stationary), sizes."""

    def __init__(self, nbr_fwd, nbr_bwd, n_in, n_in_dev, n_out, n_out_dev, kind="sparse", strided=False):
        self.strided = strided      # stride > 1: bf16 dgrad = per-offset products over the output rows + gather (u3d_tap_gather_sum)
        self.nbr_fwd, self.nbr_bwd = nbr_fwd, nbr_bwd
        self.n_in, self.n_in_dev, self.n_out, self.n_out_dev = n_in, n_in_dev, n_out, n_out_dev
        self.kind = kind            # "sparse" (encoder levels) or "dense" (SECOND3D/FPN lattice): bench.py tags timings with it
        self.level = None           # SubM convs: the Level (its halo() serves the 64 -> 64 / 128 -> 128 convs, subm_halo.hip)


def level_from_coors(coors, batch, dims):
    """coors int32 [N,4] (b,z,y,x), unique.  Returns (Level, rank) with rank[i] = internal row of input row i."""
    dev = coors.device
    g = nv.BitGrid(batch, dims, dev)
    g.mark(coors)
    g.scan()
    n = coors.shape[0]
    g.set_row_capacity(n)          # unique coors => count <= n always; guards static-shape buffers all the same
    rank = g.rank(coors)
    lvl = Level(g, g.coords(n), n, g.count_dev)
    return lvl, rank


def subm_geom(lvl):
    fwd, bwd = lvl.subm_tables()
    g = ConvGeom(fwd, bwd, lvl.n, lvl.n_dev, lvl.n, lvl.n_dev)
    g.level = lvl
    return g


def strided_level(lvl, ksize, stride, pad, capacity=None):
    """Active output set + tables of a SparseConv3d (ref: sparse_encoder_hd.py:181-192).
    capacity=None: exact row count, one host read (the reference's own flow syncs at models/detectors/uni3detr.py:153).
    capacity=int : static-shape mode for hipGraph capture — tensors are sized by the capacity, every kernel is bounded by the
    device-side count, no host read (overflow is checked by the caller after the step: Level.count_dev vs capacity)."""
    dims_out = tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(lvl.dims, ksize, stride, pad))
    g = nv.BitGrid(lvl.batch, dims_out, lvl.coords.device)
    g.mark_strided(lvl.coords, lvl.n_dev, ksize, stride, pad)
    g.scan()
    n_out = int(g.count_dev.item()) if capacity is None else int(capacity)
    if capacity is not None:
        g.set_row_capacity(n_out)
    out = Level(g, g.coords(n_out), n_out, g.count_dev)
    fwd = lvl.grid.nbr_table(out.coords, out.n_dev, ksize, stride, pad, 0)
    bwd = g.nbr_table(lvl.coords, lvl.n_dev, ksize, stride, pad, 1)
    return out, ConvGeom(fwd, bwd, lvl.n, lvl.n_dev, n_out, out.n_dev, strided=True)


REV_SUBM_TABLE = True
SUBM_HALO = os.environ.get("U3D_SUBM_HALO", "1") == "1"       # 64 -> 64 SubM convs out of per-tile staged distinct rows (subm_halo.hip)
HALO_WGRAD = True     # ... and their weight gradients (k_subm_halo_wgrad64)
HALO_128 = True         # ... and the 128 -> 128 SubM convs (k_subm_halo128: forward / input gradient)
STRIDED_DGRAD_SPLIT = True
NMAJOR_FWD = True
STRIDED_SPLIT_MIN_RATIO = 16
STRIDED_SPLIT_SPARSE_RATIO = 16      # same threshold for the sparse levels' strided convs


_CONV_USES = {}          # id(conv weight) -> forward uses since reset_conv_uses() (TrainStep resets it at the top of every step)


def reset_conv_uses():
    _CONV_USES.clear()
    _PLANES_BWD.clear()       # (gradient planes nobody picked up in the previous step: the 4 -> 16 input conv runs on the exact f32 kernel)


# ---- split-bf16 convolutions: f32-grade products on the bf16 matrix pipe (csrc/igemm_bf16.hip, u3d_igemm_fwd_split_bf16) -------------
# `mixed` precision = the REFERENCE's recipe (SparseEncoderHD + SECOND3D in fp32: sparse_encoder_hd.py:62-64, uni3detr.py:150-151).
# The exact f32 MFMA runs at 1/16 of the bf16 rate; inside split_scope() every f32 conv whose channel counts are multiples of 64 runs
# instead as three bf16 products (hi.wh + hi.wl + lo.wh, f32 accumulation) on the LDS-DMA kernels the bf16 mode is benchmarked on:
# activations travel as f32 rows, are split into hi / lo bf16 planes in front of each conv, and the three products are three sets of
# offsets of ONE launch (tripled neighbour table and weights).  Narrow levels (16 / 32 channels) stay on the exact f32 kernels.
SPLIT_BF16 = os.environ.get("U3D_SPLIT_BF16", "1") == "1"
SPLIT_FUSED_ADD = True      # residual / fan-out gradient sums in the split input gradient's epilogue
_SPLIT = [False]


# hi / lo planes that already exist for an f32 row matrix: an earlier conv split the same rows (SECOND3D's three branches read one
# input; a residual block's input feeds its first conv once).  Keyed by the tensor OBJECT (an entry holds the tensor, so the id
# cannot be reused while it lives) and its autograd version; the scope clears the table on entry and on exit, so an entry never
# outlives the forward that made it.  (Writing the planes from the BatchNorm apply that produces the rows - one pass less per layer -
# was built and measured time-neutral, 216.0 vs 217.2 scenes/s: the extra 4 B / element of stores cost what the saved read gains.)
_PLANES = {}
_PLANES_BWD = {}          # planes of a gradient tensor produced together with it (u3d_bn_bwd_apply_planes); popped by their one consumer
BN_PLANES = True          # test-only module attribute: False = every tensor is split by its own u3d_split_rows_f32 pass (the A/B formulation)


def _planes_of(feats, n_dev):
    ent = _PLANES.get(id(feats))
    if ent is not None and ent[0] is feats and ent[2] == feats._version:
        return ent[1]
    xs = nv.split_rows(feats.contiguous(), n_dev)
    if _SPLIT[0]:
        _PLANES[id(feats)] = (feats, xs, feats._version)
    return xs


def _give_planes(t, planes):
    """The BatchNorm apply pass wrote `planes` = split_rows(t) along with t: the convolution that consumes t finds them here."""
    if planes is not None and _SPLIT[0]:
        _PLANES[id(t)] = (t, planes, t._version)


def _give_bwd_planes(t, planes):
    if planes is not None:
        _PLANES_BWD[t.data_ptr()] = (t, planes, t._version)


def _bwd_planes_of(dout, n_dev):
    """bf16 planes of a convolution's output gradient: handed over by the BatchNorm backward that produced it (same storage, same
    version: the entry keeps the tensor alive, so the address cannot have been recycled), else split here."""
    ent = _PLANES_BWD.pop(dout.data_ptr(), None)
    if (ent is not None and ent[0].numel() == dout.numel() and ent[0].dtype == dout.dtype and ent[2] == ent[0]._version
            and dout.dim() == 2 and dout.is_contiguous() and ent[0].is_contiguous()):
        # (a reshaped view of the same rows - the FPN's transposed convolutions see [n, taps * C] where their BatchNorm saw
        #  [n * taps, C] - has the same planes: both are the tensor's elements in memory order, hi plane then lo plane)
        return ent[1].view(2 * dout.shape[0], dout.shape[1])
    return nv.split_rows(dout.float() if dout.dtype != torch.float32 else dout, n_dev)


@contextlib.contextmanager
def split_scope(on=True, split3=None):
    """split3: the native.Split3Set that owns the weight planes of the convolutions run inside (a detector passes its own: the job
    table and plane buffers a captured graph reads then live as long as the model); None: the module-wide default set."""
    prev, prev3 = _SPLIT[0], nv.SPLIT3_ACTIVE
    _SPLIT[0] = bool(on) and SPLIT_BF16
    _PLANES.clear()
    _PLANES_BWD.clear()
    if _SPLIT[0]:
        nv.SPLIT3_ACTIVE = split3 if split3 is not None else nv.SPLIT3_DEFAULT
        if nv.SPLIT3_BATCH:
            nv.SPLIT3_ACTIVE.refresh()          # every weight split of the step in one launch (the sets fill up during the first step)
    try:
        yield
    finally:
        _SPLIT[0] = prev
        nv.SPLIT3_ACTIVE = prev3
        _PLANES.clear()


def _split_serves(feats, cin, cout, kvol=None):
    """"wide": channels % 64 == 0 - the LDS-DMA kernels with tripled offsets; "narrow": the 27-offset 16 / 32 / 64-channel levels - three
    accumulating launches of the direct-operand kernel (kvol=None: the caller only asks about "wide"); None: the exact f32 kernels."""
    if not (_SPLIT[0] and feats.is_cuda and feats.dtype == torch.float32):
        return None
    if cin % 64 == 0 and cout % 64 == 0:
        return "wide"
    if kvol == 27 and cin in (16, 32, 64) and cout in (16, 32, 64):
        return "narrow"
    return None


def _split_table(geom, which, n_rows, plane):
    """int32 [3K, ld] = (t, t, t + plane): the table side - the third product reads the lo plane, `plane` rows below the hi plane.
    which: "fwd" / "bwd" / "id" (identity: a plain row product) / "wgrad" ([2K, ld] = (t, t + plane): the two products whose second
    operand is dy's hi plane).
    Cached on the geometry (static for the dense lattice; the sparse levels' geometries live for one step)."""
    owner = geom.level if (getattr(geom, "level", None) is not None and which != "id") else geom      # SubM: one set per Level, not per conv
    cache = owner.__dict__.setdefault("_split_tables", {})
    key = (which, plane)
    if key not in cache:
        t = geom.nbr_bwd if which == "bwd" else (None if which == "id" else geom.nbr_fwd)
        if t is None:                                   # 1x1x1 (or "id": a plain product over n_rows rows): the identity table
            t = torch.arange(n_rows, dtype=torch.int32, device=geom.n_out_dev.device).view(1, -1)
        elif isinstance(t, nv.RevNbr):
            t = t.t.flip(0)
        lo = torch.where(t >= 0, t + plane, t)
        cache[key] = torch.cat([t, lo] if which == "wgrad" else [t, t, lo], 0).contiguous()
    return cache[key]


class _SparseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, geom, layout, want_stats=False, res_token=None, fan_token=None, bn_in=None):
        # weight: the PARAMETER in its checkpoint layout: "dhwio" [kD,kH,kW,Cin,Cout] (mmcv spconv-1.x) or "oidhw" (nn.Conv3d)
        # want_stats: also return the per-row-tile BatchNorm statistics of the output (empty tensor when the kernel serving this
        # shape does not produce them) - second, non-differentiable output
        ctx.set_materialize_grads(False)        # else autograd zero-fills a gradient for the statistics output on every backward
        bf16 = feats.dtype == torch.bfloat16
        kio_shape = weight.shape if layout == "dhwio" else tuple(weight.shape[i] for i in (2, 3, 4, 1, 0))
        cin, cout = kio_shape[3], kio_shape[4]
        if feats.shape[1] != cin:               # the kernels take the channel counts from the weight: a mismatch would read out of bounds
            raise ValueError(f"sparse conv: input has {feats.shape[1]} channels, weight expects {cin}")
        # n-major forward weights: the LDS-DMA kernels (channels % 64) and the direct-operand kernel of the narrow 27-offset levels
        # (igemm_direct.hip stages [K][Cout][Cin] rows as they are; the k-major layout costs it a transposing prologue)
        kv = kio_shape[0] * kio_shape[1] * kio_shape[2]
        narrow = kv == 27 and cin in (16, 32, 64) and cout in (16, 32, 64) and not (cin == 64 and cout == 64)
        nmajor = NMAJOR_FWD and bf16 and ((cin % 64 == 0 and cout % 64 == 0) or narrow)
        split = _split_serves(feats, cin, cout, kv)
        if split == "narrow" and geom.nbr_fwd is None:
            split = None
        kio, koi = (None, None) if split else conv_weights(weight, layout, feats.dtype, want_koi=nmajor)
        ctx.geom, ctx.layout = geom, layout
        ctx.res_token = res_token
        ctx.fan_token = fan_token if (fan_token is not None and feats.requires_grad) else None
        ctx.bn_in = bn_in if (BN_GRAD_FUSION and bn_in is not None and bn_in.epi is not None and bf16 and fan_token is None) else None
        # TrainStep: this parameter's slice of the flat gradient buffer - written in place by the backward ONLY if this is the
        # weight's single use in the step (a weight used twice gets two gradients that autograd must add: it may not alias them)
        _CONV_USES[id(weight)] = _CONV_USES.get(id(weight), 0) + 1
        ctx.weight_id = id(weight)
        ctx.grad_view = getattr(weight, "_u3d_grad_view", None)
        ctx.kio_shape, ctx.wdtype = kio_shape, weight.dtype
        nv.CALL_KIND = geom.kind
        ctx.split = split
        if ctx.split:
            n_in = feats.shape[0]
            xs = _planes_of(feats, geom.n_in_dev)                                        # bf16 [2 * n_in, cin]: hi | lo planes
            ctx.save_for_backward(xs, weight)                                            # the weight gradient reads the planes
            ctx.halo = False
            ctx.split3 = nv.SPLIT3_ACTIVE                                                # the backward runs outside the scope: same set
            w3 = nv.split3_weights(weight, layout, nmajor=True)                          # [3K, cout, cin], straight from the parameter
            if split == "narrow":
                y = nv.spconv_fwd_split_direct(xs, w3, geom.nbr_fwd, geom.n_out_dev, geom.n_out, cout)
                res = (y, None, 0)
            else:
                t3 = _split_table(geom, "fwd", geom.n_out, n_in)
                res = nv.spconv_fwd_split(xs, w3, t3, geom.n_out_dev, geom.n_out, cout, want_stats=want_stats)
                if not want_stats:
                    return res
            if not want_stats:
                return res[0]
            y, stats, tr = res
            if stats is None:
                stats = torch.empty(0, dtype=torch.float64, device=feats.device)
            else:
                stats._u3d_tile_rows = tr
            ctx.mark_non_differentiable(stats)
            return y, stats
        ctx.halo = False
        ctx.save_for_backward(feats, kio)
        nbr = geom.nbr_fwd if kio.shape[0] > 1 else None
        ctx.halo = (SUBM_HALO and REV_SUBM_TABLE and nmajor and geom.level is not None and kv == 27 and cin == cout
                    and (cin == 64 or (cin == 128 and HALO_128)) and geom.n_out >= 4096
                    and geom.level.halo() is not None)
        ctx.halo_tab = geom.level.halo() if ctx.halo else None
        if ctx.halo:
            pk_fwd, ctx.pk_bwd = halo_packs(weight, kio, koi)
            if want_stats:
                y, stats, tr = nv.subm_halo_conv(feats, pk_fwd, ctx.halo_tab, want_stats=True)
                stats._u3d_tile_rows = tr
                ctx.mark_non_differentiable(stats)
                return y, stats
            return nv.subm_halo_conv(feats, pk_fwd, ctx.halo_tab)
        if want_stats:
            res = nv.spconv_fwd_stats(feats, koi, nbr, geom.n_out_dev, geom.n_out, cout) if nmajor else None
            if res is not None:
                y, stats, tr = res
                stats._u3d_tile_rows = tr
            else:
                y = (nv.spconv_fwd(feats, koi, nbr, geom.n_out_dev, geom.n_out, cout, transpose_w=True, tag="spconv_fwd") if nmajor
                     else nv.spconv_fwd(feats, kio, nbr, geom.n_out_dev, geom.n_out, cout))
                stats = torch.empty(0, dtype=torch.float64, device=feats.device)
            ctx.mark_non_differentiable(stats)
            return y, stats
        if nmajor:      # forward on the LDS-DMA kernel (n-major weight shadow)
            return nv.spconv_fwd(feats, koi, nbr, geom.n_out_dev, geom.n_out, cout, transpose_w=True, tag="spconv_fwd")
        return nv.spconv_fwd(feats, kio, nbr, geom.n_out_dev, geom.n_out, cout)

    @staticmethod
    def backward(ctx, dout, _dstats=None):
        feats, wc = ctx.saved_tensors
        g = ctx.geom
        if dout is None:
            return None, None, None, None, None, None, None, None
        dout = dout.contiguous()
        kvol = wc.shape[0]
        din = dw = None
        nv.CALL_KIND = g.kind
        if ctx.split:
            return _SparseConv._split_backward(ctx, feats, wc, dout)

        def weight_grad():
            nbr = g.nbr_fwd if kvol > 1 else None
            cin_w, cout_w = wc.shape[1], wc.shape[2]
            v2 = feats.dtype == torch.bfloat16 and nv.USE_IGEMM_V2 and cin_w % 16 == 0 and cout_w % 16 == 0 and ctx.wdtype == torch.float32
            # the reduction stage can write the parameter's own layout straight into its slice of the flat gradient buffer (out=):
            # autograd keeps that view as .grad and the step's packing copy has nothing to move for this parameter
            gv = ctx.grad_view if (v2 and ctx.grad_view is not None and ctx.grad_view.is_contiguous()
                                   and ctx.grad_view.dtype == torch.float32 and _CONV_USES.get(ctx.weight_id, 0) == 1) else None
            if ctx.layout == "oidhw" and v2:
                # nn.Conv3d's own [Cout,Cin,kD,kH,kW] layout: autograd keeps the tensor as the gradient
                ks = ctx.kio_shape
                dw = nv.spconv_wgrad(feats, dout, nbr, g.n_out_dev, kvol, out_oik=True, out=gv).view(ks[4], ks[3], ks[0], ks[1], ks[2])
            elif ctx.layout == "dhwio" and v2 and ctx.halo and HALO_WGRAD and cin_w == 64:
                dw = nv.subm_halo_wgrad(feats, dout, ctx.halo_tab, out=gv).view(ctx.kio_shape)
            elif ctx.layout == "dhwio" and v2:
                dw = nv.spconv_wgrad(feats, dout, nbr, g.n_out_dev, kvol, out=gv).view(ctx.kio_shape)
            else:
                dw = nv.spconv_wgrad(feats, dout, nbr, g.n_out_dev, kvol).reshape(ctx.kio_shape).to(ctx.wdtype)
                if ctx.layout == "oidhw":
                    dw = dw.permute(4, 3, 0, 1, 2)
            return dw

        if ctx.needs_input_grad[1]:
            dw = weight_grad()
        if ctx.needs_input_grad[0]:
            nbr = g.nbr_bwd if kvol > 1 else None
            cin, cout = wc.shape[1], wc.shape[2]
            fan = ctx.fan_token
            facc = fan.acc if fan is not None else None          # partial sum of the branches that ran before this one
            if (STRIDED_DGRAD_SPLIT and g.strided and kvol > 1 and dout.dtype == torch.bfloat16 and cout % 64 == 0
                    and (kvol * cin) % 64 == 0
                    and g.n_out * (STRIDED_SPLIT_SPARSE_RATIO if g.kind == "sparse" else STRIDED_SPLIT_MIN_RATIO) <= g.n_in):      # stride 4: 15/16 of the direct dgrad's MFMAs hit zero rows
                prod = nv.linear_bf16(dout, wc.view(kvol * cin, cout), None, False)      # [n_out, K*Cin]
                fa = facc if (facc is not None and facc.dtype == prod.dtype and facc.is_contiguous() and facc.shape == (g.n_in, cin)) else None
                din = nv.tap_gather_sum(prod, nbr, g.n_in_dev, g.n_in, cin, kvol, addend=fa)      # the other branches' sum rides the gather
                if facc is not None and fa is None:
                    din += facc
            else:
                tok = ctx.res_token
                add = tok.dres if tok is not None else None
                if tok is not None:
                    tok.dres = None
                if add is not None and facc is not None:
                    add = add + facc
                elif facc is not None:
                    add = facc
                bt = ctx.bn_in
                if bt is not None and (bt.c != cin or (add is not None and not (add.dtype == torch.bfloat16 and add.is_contiguous()))):
                    bt = None
                if ctx.halo and cin != 64:
                    bt = None                    # (the 128-channel halo kernel has no BatchNorm-backward epilogue)
                if ctx.halo:
                    pk = ctx.pk_bwd if ctx.pk_bwd is not None else nv.subm_halo_wpack(wc)
                    if bt is not None:
                        din, st, tr = nv.subm_halo_conv(dout, pk, ctx.halo_tab, krev=True, addend=add, want_stats=True, tag="spconv_dgrad",
                                                        bn_epi=bt.epi)
                        bt.partial = (st, tr)
                    else:
                        din = nv.subm_halo_conv(dout, pk, ctx.halo_tab, krev=True, addend=add, tag="spconv_dgrad")
                else:
                    r = nv.spconv_dgrad_bnstats(dout, wc, nbr, g.n_in_dev, g.n_in, cin, add, bt.epi) if (bt is not None and kvol > 1) else None
                    if r is not None:
                        din, bt.partial = r[0], (r[1], r[2])
                    else:
                        din = nv.spconv_fwd(dout, wc, nbr, g.n_in_dev, g.n_in, cin, transpose_w=True, addend=add)
            if fan is not None:
                din = fan.step(din)
        elif ctx.res_token is not None:
            ctx.res_token.dres = None
        return din, dw, None, None, None, None, None, None


def _split_backward_impl(ctx, xs, wc, dout):
    """Backward of a split-bf16 conv: dy is split into planes once; dx = three products as in the forward (transposed tables /
    weights); dW = x^T dy ~ xh^T dyh + xl^T dyh + xh^T dyl: two launches of the bf16 weight-gradient kernel (offsets doubled for the
    two products against dy's hi plane), summed in f32."""
    g = ctx.geom
    ks = ctx.kio_shape
    kvol, cin, cout = ks[0] * ks[1] * ks[2], ks[3], ks[4]
    n_in, n_out = xs.shape[0] // 2, dout.shape[0]
    dys = _bwd_planes_of(dout, g.n_out_dev)                                              # bf16 [2 * n_out, cout]
    din = dw = None
    if ctx.split == "narrow":
        if ctx.needs_input_grad[1]:
            # three launches of the narrow weight-gradient kernels on plane views (the table indexes rows of a plane): no doubled table
            xh, xl, dyh, dyl = xs[:n_in], xs[n_in:], dys[:n_out], dys[n_out:]
            dwk = nv.sum3(nv.spconv_wgrad(xh, dyh, g.nbr_fwd, g.n_out_dev, kvol), nv.spconv_wgrad(xl, dyh, g.nbr_fwd, g.n_out_dev, kvol),
                          nv.spconv_wgrad(xh, dyl, g.nbr_fwd, g.n_out_dev, kvol)).reshape(ctx.kio_shape).to(ctx.wdtype)
            dw = dwk.permute(4, 3, 0, 1, 2) if ctx.layout == "oidhw" else dwk
        if ctx.needs_input_grad[0]:
            din = nv.spconv_fwd_split_direct(dys, nv.split3_weights(wc, ctx.layout, nmajor=False, cache=getattr(ctx, "split3", None)), g.nbr_bwd, g.n_in_dev, g.n_in, cin,
                                             tag="spconv_dgrad")
            fan, tok = ctx.fan_token, ctx.res_token
            if fan is not None:
                if fan.acc is not None:
                    din += fan.acc
                din = fan.step(din)
            if tok is not None and tok.dres is not None:
                din = din + tok.dres
                tok.dres = None
        elif ctx.res_token is not None:
            ctx.res_token.dres = None
        return din, dw, None, None, None, None, None, None
    if ctx.needs_input_grad[1]:
        ta = _split_table(g, "wgrad", n_out, n_in)                                         # [2K, ld]: (nbr, nbr + n_in)
        tb = ta[:kvol]
        a = nv.spconv_wgrad(xs, dys[:n_out], ta, g.n_out_dev, 2 * kvol)                    # [2K, cin, cout]: xh^T dyh | xl^T dyh
        b = nv.spconv_wgrad(xs, dys[n_out:], tb, g.n_out_dev, kvol)                        # [K, cin, cout]: xh^T dyl
        dwk = nv.sum3(a[:kvol], a[kvol:], b).reshape(ctx.kio_shape).to(ctx.wdtype)
        dw = dwk.permute(4, 3, 0, 1, 2) if ctx.layout == "oidhw" else dwk
    if ctx.needs_input_grad[0]:
        if (STRIDED_DGRAD_SPLIT and g.strided and kvol > 1 and (kvol * cin) % 64 == 0
                and n_out * (STRIDED_SPLIT_SPARSE_RATIO if g.kind == "sparse" else STRIDED_SPLIT_MIN_RATIO) <= g.n_in):
            # strided conv: per-offset products over the (few) OUTPUT rows - one split product with the identity table, [n_out, K * Cin]
            # f32 - then the gather over the offsets that reach each input row (as the bf16 path, u3d_tap_gather_sum; the
            # output-stationary form runs all K x 3 products for every input row, 15/16 of them on absent neighbours at stride 4)
            tid = _split_table(g, "id", n_out, n_out)
            wt3 = nv.split3_weights(wc, ctx.layout, nmajor=False, cache=getattr(ctx, "split3", None)).view(3, kvol * cin, cout)      # [K, Cin, Cout] as ONE [K * Cin, Cout] matrix
            prod = nv.spconv_fwd_split(dys, wt3, tid, g.n_out_dev, n_out, kvol * cin, tag="spconv_dgrad")
            din = nv.tap_gather_sum(prod, g.nbr_bwd, g.n_in_dev, g.n_in, cin, kvol)
            fused_add = False
        else:
            # the residual branch's / the other fan-out branches' gradient rides the launch's epilogue (f32 addend), as in bf16 mode
            fan, tok = ctx.fan_token, ctx.res_token
            add = tok.dres if (tok is not None and tok.dres is not None) else None
            facc = fan.acc if fan is not None else None
            if add is not None and facc is not None:
                add = add + facc
            elif facc is not None:
                add = facc
            fused_add = SPLIT_FUSED_ADD and (add is None or (add.dtype == torch.float32 and add.is_contiguous() and tuple(add.shape) == (g.n_in, cin)))
            t3 = _split_table(g, "bwd", g.n_in, n_out)
            din = nv.spconv_fwd_split(dys, nv.split3_weights(wc, ctx.layout, nmajor=False, cache=getattr(ctx, "split3", None)), t3, g.n_in_dev, g.n_in, cin, tag="spconv_dgrad",
                                      addend=add if fused_add else None)
        fan = ctx.fan_token
        tok = ctx.res_token
        if fused_add:
            if tok is not None:
                tok.dres = None
            if fan is not None:
                din = fan.step(din)
        else:
            if fan is not None:
                if fan.acc is not None:
                    din += fan.acc
                din = fan.step(din)
            if tok is not None and tok.dres is not None:
                din = din + tok.dres
                tok.dres = None
    elif ctx.res_token is not None:
        ctx.res_token.dres = None
    return din, dw, None, None, None, None, None, None


_SparseConv._split_backward = staticmethod(_split_backward_impl)


def sparse_conv(feats, weight, geom, layout="dhwio", fan_token=None):
    """weight: conv PARAMETER in checkpoint layout ("dhwio" = [kD,kH,kW,Cin,Cout]; "oidhw" = nn.Conv3d's [Cout,Cin,kD,kH,kW])."""
    return _SparseConv.apply(feats, weight, geom, layout, False, None, fan_token)


# BatchNorm-backward sums out of the consumer's dgrad epilogue (BnGradToken).  Measured in the captured step (profiles/
# r03_bn_grad_fusion_diff.txt): 24 of the 45 statistics passes disappear (-0.47 ms of k_col_stats_vec), but the conv launches that
# take them over grow by +0.44 ms (per-tile sums in a tail that runs at one or two workgroups per CU: its x loads and ~8 VALU ops per
# element are exposed, where the separate pass streams at full occupancy), and k_bn_bwd_apply loses the L2 hits the statistics pass
# left behind (+0.05 ms): 19.77 -> 19.85 ms per step.  Kept (parity-tested) but off.
BN_GRAD_FUSION = False
FUSED_CONV_STATS = True


def conv_bn(feats, weight, geom, bn, n_dev, residual=None, relu=True, layout="dhwio", post_add=None, res_take=None, res_give=None,
            fan_token=None, bn_in=None, bn_out=None):
    """conv -> BatchNorm rows (+ residual) (+ ReLU).  In training the conv's epilogue already reduces the BatchNorm statistics per row
    tile where its kernel supports it (bf16, channels % 64 == 0): the separate statistics pass over the conv output disappears."""
    if FUSED_CONV_STATS and bn.training and feats.is_cuda and (feats.dtype == torch.bfloat16 or _split_serves(feats, feats.shape[1], bn.num_features) == "wide"):
        # res_take: this conv's input is the identity of a residual block - its backward sums the token's gradient into the input
        # gradient; res_give: this BatchNorm adds that identity - its backward leaves the identity's gradient in the token
        # fan_token: this conv is one of several that take the same input (FanoutToken)
        # bn_in: BnGradToken of the BatchNorm that produced `feats`, when this conv's input gradient is that layer's whole dy;
        # bn_out: token this call's BatchNorm fills for ITS consumer
        y, stats = _SparseConv.apply(feats, weight, geom, layout, True, res_take, fan_token, bn_in)
        if stats.numel():
            tr = getattr(stats, "_u3d_tile_rows", None)
            if tr is None:                      # attribute lost on the way through autograd: recover it from the shape
                nb = stats.shape[0]
                tr = next((t for t in (128, 256, 192) if (y.shape[0] + t - 1) // t == nb), 0)
            return _BNRows.apply(y, bn.weight, bn.bias, residual, n_dev, bn, relu, True, stats, tr, None, post_add, res_give, bn_out)
        return _BNRows.apply(y, bn.weight, bn.bias, residual, n_dev, bn, relu, bn.training, None, 0, None, post_add, res_give, bn_out)
    return bn_rows(sparse_conv(feats, weight, geom, layout, fan_token), bn, n_dev, residual, relu, None, post_add)


class _BNRows(torch.autograd.Function):
    """BatchNorm1d over active rows (+ residual) (+ ReLU), training or eval statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, n_dev, bn, relu, training, stats=None, tile_rows=0, row_map=None, post_add=None, res_token=None,
                bn_tok=None):
        n = x.shape[0]
        if training and stats is not None:
            mean, invstd = nv.bn_finalize_partials(stats, tile_rows, n_dev, n, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                                   bn.running_mean, bn.running_var, bn.num_batches_tracked)
        elif training:
            mean, invstd = nv.bn_forward_stats(x, n_dev, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                               bn.running_mean, bn.running_var, bn.num_batches_tracked)
        else:
            mean = bn.running_mean
            invstd = torch.rsqrt(bn.running_var + bn.eps)
        g32, b32 = gamma.float(), beta.float()
        if post_add is not None:
            assert relu and residual is None and post_add.shape == x.shape and post_add.dtype == x.dtype
            post_add = post_add.contiguous()
        # split-bf16 scope, f32 rows: the pass also writes y's hi / lo bf16 planes for the convolution that reads y next, and the backward
        # will write the planes of dx for the convolution that produced x (u3d_bn_apply_planes / u3d_bn_bwd_apply_planes)
        ctx.planes = bool(BN_PLANES and _SPLIT[0] and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 16 == 0)
        if ctx.planes:
            y, ypl = nv.bn_apply(x, mean, invstd, g32, b32, residual, relu, n_dev, row_map, post_add, want_planes=True)
            _give_planes(y, ypl)
        else:
            y = nv.bn_apply(x, mean, invstd, g32, b32, residual, relu, n_dev, row_map, post_add)
        # ReLU without residual: the backward recomputes the mask from x (same expression) instead of streaming y again
        ctx.remask = bool(relu and residual is None)
        ctx.save_for_backward(x, None if ctx.remask else y, mean, invstd, g32, b32)
        ctx.n_dev, ctx.relu, ctx.training, ctx.has_res = n_dev, relu, training, residual is not None
        ctx.pdtype = gamma.dtype
        ctx.row_map = row_map
        ctx.has_post = post_add is not None
        ctx.res_token = res_token if residual is not None else None
        # BatchNorm-backward sums from the consumer's input-gradient launch (BnGradToken): training statistics, plain row order, bf16
        ctx.bn_tok = None
        if bn_tok is not None and training and row_map is None and post_add is None and x.dtype == torch.bfloat16 and x.is_cuda:
            bn_tok.fill(x, None if ctx.remask else y, mean, invstd, g32, b32, relu)
            ctx.bn_tok = bn_tok
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        tok, part = ctx.bn_tok, None
        if tok is not None:
            part, tok.partial, tok.epi, tok.keep = tok.partial, None, None, None
        if part is not None:      # the conv that consumed this layer's output already reduced them per row tile on writing dy
            sums, s32 = nv.bn_bwd_finalize_partials(part[0], part[1], ctx.n_dev, x.shape[0])
        else:
            sums, s32 = nv.bn_bwd_stats(dy, y, x, mean, invstd, ctx.relu, ctx.n_dev, gamma, beta, ctx.row_map, want_f32=True)
        use = sums if ctx.training else torch.zeros_like(sums)       # eval statistics are constants: dx = gamma*invstd*g
        if getattr(ctx, "planes", False):
            dx, dres, dpl = nv.bn_bwd_apply(dy, y, x, mean, invstd, gamma, use, ctx.relu, ctx.n_dev, ctx.has_res, beta, ctx.row_map, want_planes=True)
            _give_bwd_planes(dx, dpl)
        else:
            dx, dres = nv.bn_bwd_apply(dy, y, x, mean, invstd, gamma, use, ctx.relu, ctx.n_dev, ctx.has_res, beta, ctx.row_map)
        if ctx.pdtype != torch.float32:
            s32 = s32.to(ctx.pdtype)
        if ctx.res_token is not None and dres is not None:
            ctx.res_token.dres, dres = dres, None            # picked up by the block's first conv (ResidualToken)
        # post_add enters y by a plain sum: its gradient is dy itself (no launch)
        return dx, s32[1], s32[0], dres, None, None, None, None, None, None, None, (dy if ctx.has_post else None), None, None


def bn_rows(x, bn, n_dev, residual=None, relu=True, row_map=None, post_add=None):
    """row_map: the output (and its gradient) use a permuted row order, y[row_map[r]] = bn(x[r]) - see u3d_bn_apply."""
    return _BNRows.apply(x, bn.weight, bn.bias, residual, n_dev, bn, relu, bn.training, None, 0, row_map, post_add, None)


class _ToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, lvl):
        ctx.lvl = lvl
        return nv.to_dense(feats, lvl.coords, lvl.n_dev, lvl.batch, lvl.dims)

    @staticmethod
    def backward(ctx, dvol):
        lvl = ctx.lvl
        cl = dvol.permute(0, 2, 3, 4, 1).contiguous()      # no copy when dvol is channels_last_3d
        return nv.from_dense(cl, lvl.coords, lvl.n_dev, lvl.n), None


def to_dense(feats, lvl):
    """rows -> [B,C,D,H,W] (channels_last_3d memory) (ref: sparse_encoder_hd.py:133)."""
    return _ToDense.apply(feats, lvl)


class _PermuteRows(torch.autograd.Function):
    """out[rank[i]] = in[i] (API row order -> internal block-major order)."""

    @staticmethod
    def forward(ctx, feats, rank, n_out):
        ctx.save_for_backward(rank)
        return nv.scatter_rows(feats.contiguous(), rank, n_out)

    @staticmethod
    def backward(ctx, dout):
        (rank,) = ctx.saved_tensors
        return nv.gather_rows(dout.contiguous(), rank), None, None


def permute_rows(feats, rank, n_out):
    return _PermuteRows.apply(feats, rank, n_out)
