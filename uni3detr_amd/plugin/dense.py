"""SECOND3D backbone and SECOND3DFPN neck behind the reference's registry names and constructors (ref:
projects/mmdet3d_plugin/models/backbones/second_3d.py:11-114, projects/mmdet3d_plugin/models/necks/second3d_fpn.py:11-143).

Execution is NOT torch/MIOpen convolution (on this stack MIOpen resolves NDHWC 3-D convolutions to naive kernels:
profiles/r01_a_*): a dense volume is the special case "every cell active" of the sparse levels, so every Conv3d /
ConvTranspose3d + BatchNorm3d + ReLU here runs on the same HIP implicit-GEMM, weight-gradient and BatchNorm-rows
kernels as the sparse encoder, over channels-last rows [B*D*H*W, C] with a static neighbour table of the lattice.
The nn.Conv3d / nn.BatchNorm3d modules are kept as parameter holders so that state_dict names and shapes are the
reference checkpoints' (`blocks.{i}.{3j}.weight` [Cout,Cin,kd,kh,kw], `deblocks.{i}.0.weight`, `extra_blocks.{3j}.weight`).
"""

import numpy as np
import torch
from torch import nn

from .. import native as nv
from .. import sparse as sp
from ..registry import BACKBONES, NECKS


def _conv(cfg, cin, cout, kernel, stride=1, padding=0):
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t != "Conv3d":
        raise NotImplementedError("the HIP dense path implements the Conv3d variant used by every shipped config")
    if cfg.get("bias", True):
        raise NotImplementedError("shipped configs build these convolutions with bias=False")
    return nn.Conv3d(cin, cout, kernel, stride=stride, padding=padding, **cfg)


def _norm(cfg, c):
    cfg = dict(cfg)
    t = cfg.pop("type")
    assert t == "BN3d", "shipped configs use BN3d"
    cfg.pop("requires_grad", None)
    return nn.BatchNorm3d(c, **cfg)


_BRANCH_STREAMS = []
PARALLEL_BRANCHES = True      # eager mode: run the independent SECOND3D branches on separate streams


class Lattice:
    """Static geometry of dense volumes on one device: neighbour tables keyed by (B, dims, kernel, stride, pad)."""

    _cache = {}

    @classmethod
    def conv(cls, device, batch, dims_in, ksize, stride, pad):
        key = ("c", str(device), batch, tuple(dims_in), tuple(ksize), tuple(stride), tuple(pad))
        g = cls._cache.get(key)
        if g is None:
            dims_out = tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(dims_in, ksize, stride, pad))
            n_in = batch * dims_in[0] * dims_in[1] * dims_in[2]
            n_out = batch * dims_out[0] * dims_out[1] * dims_out[2]
            one = all(k == 1 for k in ksize) and all(s == 1 for s in stride)
            fwd = None if one else nv.dense_nbr_table(batch, dims_out, dims_in, ksize, stride, pad, 0, device)
            bwd = None if one else nv.dense_nbr_table(batch, dims_in, dims_out, ksize, stride, pad, 1, device)
            n_in_dev = torch.tensor([n_in], dtype=torch.int32, device=device)
            n_out_dev = torch.tensor([n_out], dtype=torch.int32, device=device)
            g = (sp.ConvGeom(fwd, bwd, n_in, n_in_dev, n_out, n_out_dev, kind="dense", strided=any(s > 1 for s in stride)), dims_out)
            cls._cache[key] = g
        return g

    @classmethod
    def upsample_index(cls, device, batch, dims_in, s):
        """Row map of a (1,s,s)/(1,s,s) transposed conv: output row -> row of Y.view(N_in*s*s, Cout), Y = X @ [W_tap...]."""
        key = ("u", str(device), batch, tuple(dims_in), s)
        v = cls._cache.get(key)
        if v is None:
            D, H, W = dims_in
            b = torch.arange(batch, device=device).view(-1, 1, 1, 1)
            z = torch.arange(D, device=device).view(1, -1, 1, 1)
            y = torch.arange(H * s, device=device).view(1, 1, -1, 1)
            x = torch.arange(W * s, device=device).view(1, 1, 1, -1)
            in_row = ((b * D + z) * H + y // s) * W + x // s
            tap = (y % s) * s + (x % s)
            idx = (in_row * (s * s) + tap).reshape(-1).int().contiguous()
            inv = torch.empty_like(idx)
            inv[idx.long()] = torch.arange(idx.numel(), device=device, dtype=torch.int32)
            v = (idx, inv, (D, H * s, W * s))
            cls._cache[key] = v
        return v


FUSED_LEVEL_SUM = True
FUSED_UPSAMPLE_ORDER = True


class _GatherBijection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, idx, inv):
        ctx.save_for_backward(inv)
        return nv.gather_rows(rows.contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        (inv,) = ctx.saved_tensors
        return nv.gather_rows(dout.contiguous(), inv), None, None


def to_rows(x):
    """[B,C,D,H,W] (any strides) -> (rows [B*D*H*W, C] contiguous, B, dims); free for channels_last_3d tensors."""
    B, C, D, H, W = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(-1, C), B, (D, H, W)


def to_volume(rows, B, dims):
    """rows -> logical [B,C,D,H,W] view in channels_last_3d memory."""
    return rows.view(B, *dims, rows.shape[1]).permute(0, 4, 1, 2, 3)


def conv_bn_relu(rows, B, dims, conv, bn, post_add=None, fan_token=None, bn_in=None, bn_out=None):
    ks, st, pd = tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)
    geom, dims_out = Lattice.conv(rows.device, B, dims, ks, st, pd)
    # nn.Conv3d layout [Cout,Cin,kd,kh,kw] (re-laid-out by the shadow set); BatchNorm statistics come out of the conv's epilogue
    return sp.conv_bn(rows, conv.weight, geom, bn, geom.n_out_dev, None, True, "oidhw", post_add, fan_token=fan_token, bn_in=bn_in,
                      bn_out=bn_out), dims_out


def deconv_bn_relu(rows, B, dims, deconv, bn, post_add=None):
    s = deconv.stride[1]
    assert tuple(deconv.kernel_size) == (1, s, s) and tuple(deconv.stride) == (1, s, s), "non-overlapping (1,s,s) upsampling only"
    cin, cout = deconv.weight.shape[0], deconv.weight.shape[1]
    n_in = rows.shape[0]
    geom, _ = Lattice.conv(rows.device, B, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    w = deconv.weight.permute(0, 2, 3, 4, 1).reshape(1, 1, 1, cin, s * s * cout)     # columns = (tap, cout)
    y = sp.sparse_conv(rows, w, geom).view(n_in * s * s, cout)
    idx, inv, dims_out = Lattice.upsample_index(rows.device, B, dims, s)
    n_dev = Lattice.conv(rows.device, B, dims_out, (1, 1, 1), (1, 1, 1), (0, 0, 0))[0].n_out_dev
    if FUSED_UPSAMPLE_ORDER:
        # the GEMM leaves the rows tap-major; BatchNorm statistics do not care about row order, and its apply kernel writes
        # row r to its lattice position inv[r] (the backward reads dy there) - no separate gather pass over the upsampled volume
        return sp.bn_rows(y, bn, n_dev, None, True, row_map=inv, post_add=post_add), dims_out
    y = _GatherBijection.apply(y, idx, inv)
    return sp.bn_rows(y, bn, n_dev, None, True, post_add=post_add), dims_out


FANOUT_FUSION = True


@BACKBONES.register_module()
class SECOND3D(nn.Module):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 is_cascade=True, norm_cfg=dict(type="BN3d", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv3d", bias=False),
                 init_cfg=None, pretrained=None):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        in_filters = list(in_channels) if isinstance(in_channels, (list, tuple)) else [in_channels, *out_channels[:-1]]
        self.is_cascade = is_cascade
        self.kernel_type = conv_cfg.type if hasattr(conv_cfg, "type") else conv_cfg["type"]
        kernel = tuple(conv_cfg.pop("kernel")) if "kernel" in conv_cfg else (1, 3, 3)      # mutates the cfg like the reference (:47-48)
        padding = tuple((k - 1) // 2 for k in kernel)
        blocks = []
        for i, n in enumerate(layer_nums):
            s = layer_strides[i]
            layers = [_conv(conv_cfg, in_filters[i], out_channels[i], kernel, (1, s, s), padding), _norm(norm_cfg, out_channels[i]),
                      nn.ReLU(inplace=True)]
            for _ in range(n):
                layers += [_conv(conv_cfg, out_channels[i], out_channels[i], kernel, 1, padding), _norm(norm_cfg, out_channels[i]),
                           nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(blocks)

    @staticmethod
    def _run_block(blk, rows, B, dims, fan=None):
        mods = list(blk)
        prev = None                     # inside a block every conv -> BN -> ReLU output has one consumer, the next conv: its input-gradient
        for j in range(0, len(mods), 3):      # launch reduces that BatchNorm's backward sums (sp.BnGradToken); the block's output goes elsewhere
            cur = sp.BnGradToken() if j + 3 < len(mods) else None
            rows, dims = conv_bn_relu(rows, B, dims, mods[j], mods[j + 1], fan_token=fan if j == 0 else None, bn_in=prev, bn_out=cur)
            prev = cur
        return rows, dims

    def forward(self, x):
        rows, B, dims = to_rows(x)
        outs = []
        if self.is_cascade:
            for blk in self.blocks:
                rows, dims = self._run_block(blk, rows, B, dims)
                outs.append(to_volume(rows, B, dims))
            return tuple(outs)
        # non-cascade: the blocks are independent branches on the same input.  The strided branches have few rows (12 000 /
        # 48 000 at B=8 -> 94 / 188 workgroups for 256 CUs), so each branch runs on its own stream and the small ones fill the
        # CUs the large one leaves idle; autograd replays the same fork/join in backward; inside a hipGraph capture the
        # event waits become graph edges.
        if not PARALLEL_BRANCHES or torch.cuda.is_current_stream_capturing():
            # measured: inside the captured step the fork/join costs more than it gains (56.1 vs 54.4 ms) — the 256x256-tile
            # kernels own a CU's LDS, so branches cannot co-reside; streams only pay off against eager-mode launch gaps
            # the branches' input gradients are summed by the first convs' own backward launches (sp.FanoutToken), not by autograd
            fan = sp.FanoutToken(len(self.blocks)) if (FANOUT_FUSION and rows.requires_grad and len(self.blocks) > 1) else None
            seq = [self._run_block(blk, rows, B, dims, fan) for blk in self.blocks]
            return tuple(to_volume(r, B, d) for r, d in seq)
        cur = torch.cuda.current_stream()
        while len(_BRANCH_STREAMS) < len(self.blocks) - 1:
            _BRANCH_STREAMS.append(torch.cuda.Stream())
        results = [None] * len(self.blocks)
        for i in range(len(self.blocks) - 1, 0, -1):              # launch the small branches first
            st = _BRANCH_STREAMS[i - 1]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                results[i] = self._run_block(self.blocks[i], rows, B, dims)
        results[0] = self._run_block(self.blocks[0], rows, B, dims)
        for i in range(1, len(self.blocks)):
            rows.record_stream(_BRANCH_STREAMS[i - 1])
            cur.wait_stream(_BRANCH_STREAMS[i - 1])
        return tuple(to_volume(r, B, d) for r, d in results)


@NECKS.register_module()
class SECOND3DFPN(nn.Module):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type="BN3d", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv3d", bias=False),
                 conv_cfg=dict(type="Conv3d", bias=False), extra_conv=None, use_conv_for_no_stride=False, use_for_distill=False,
                 init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.fp16_enabled = False
        if use_for_distill:
            raise NotImplementedError("use_for_distill is not used by any shipped Uni3DETR config")
        assert upsample_cfg["type"] == "deconv3d" and not upsample_cfg.get("bias", True)
        deblocks = []
        for i, oc in enumerate(out_channels):
            s = upsample_strides[i]
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose3d(in_channels[i], oc, (1, s, s), stride=(1, s, s), bias=False)
            else:
                s2 = int(np.round(1 / s))
                up = _conv(conv_cfg, in_channels[i], oc, (1, s2, s2), (1, s2, s2))
            deblocks.append(nn.Sequential(up, _norm(norm_cfg, oc), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        self.extra_conv = extra_conv
        if extra_conv is not None:
            extra = dict(extra_conv)
            self.layer_num = extra.pop("num_conv")
            kernel = tuple(extra.pop("kernel")) if "kernel" in extra else (3, 3, 3)
            if "sep_kernel" in extra:
                raise NotImplementedError("sep_kernel is not used by any shipped Uni3DETR config")
            padding = tuple((k - 1) // 2 for k in kernel)
            layers = []
            for _ in range(self.layer_num):
                layers += [_conv(extra, out_channels[-1], out_channels[-1], kernel, 1, padding), _norm(norm_cfg, out_channels[-1]),
                           nn.ReLU(inplace=True)]
            self.extra_blocks = nn.Sequential(*layers)

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        out, odims, B = None, None, None
        for i, d in enumerate(self.deblocks):
            rows, B, dims = to_rows(x[i])
            # the running sum of the levels rides in each level's BatchNorm apply kernel (y = relu(bn(x)) + sum so far)
            fuse = FUSED_LEVEL_SUM and out is not None and rows.is_cuda and out.dtype == rows.dtype
            if isinstance(d[0], nn.ConvTranspose3d):
                r, dd = deconv_bn_relu(rows, B, dims, d[0], d[1], out if fuse else None)
            else:
                r, dd = conv_bn_relu(rows, B, dims, d[0], d[1], out if fuse else None)
            assert odims is None or odims == dd, "FPN levels must land on one lattice"
            out, odims = (r if (out is None or fuse) else out + r), dd
        if self.extra_conv is not None:
            mods = list(self.extra_blocks)
            for j in range(0, len(mods), 3):
                out, odims = conv_bn_relu(out, B, odims, mods[j], mods[j + 1])
        return to_volume(out, B, odims)
