"""SECOND3D backbone and SECOND3DFPN neck (ref: projects/mmdet3d_plugin/models/backbones/second_3d.py:11-114,
projects/mmdet3d_plugin/models/necks/second3d_fpn.py:11-143).  Dense 3-D convolutions run through PyTorch-ROCm
(MIOpen) in channels_last_3d; module/parameter names are the reference's (`blocks.{i}.{3j}.weight`, `deblocks.{i}.0.weight`,
`extra_blocks.{3j}.weight`)."""
import numpy as np
import torch
from torch import nn

from ..registry import BACKBONES, NECKS


def _conv(cfg, cin, cout, kernel, stride=1, padding=0):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cls = {"Conv3d": nn.Conv3d, "Conv2d": nn.Conv2d}[t]
    return cls(cin, cout, kernel, stride=stride, padding=padding, **cfg)


def _norm(cfg, c):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cls = {"BN3d": nn.BatchNorm3d, "BN2d": nn.BatchNorm2d, "BN": nn.BatchNorm2d}[t]
    cfg.pop("requires_grad", None)
    return cls(c, **cfg)


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
            stride = (1, s, s) if len(padding) == 3 else (s, s)
            layers = [_conv(conv_cfg, in_filters[i], out_channels[i], kernel, stride, padding), _norm(norm_cfg, out_channels[i]),
                      nn.ReLU(inplace=True)]
            for _ in range(n):
                layers += [_conv(conv_cfg, out_channels[i], out_channels[i], kernel, 1, padding), _norm(norm_cfg, out_channels[i]),
                           nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        outs = []
        batch = x.shape[0]
        if self.kernel_type == "Conv2d":
            x = x.transpose(1, 2).flatten(0, 1)
        for blk in self.blocks:
            if self.is_cascade:
                x = blk(x)
                outs.append(x)
            else:
                outs.append(blk(x))
        if self.kernel_type == "Conv2d":
            outs = [o.reshape(batch, -1, *o.shape[-3:]).transpose(1, 2) for o in outs]
        return tuple(outs)


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
        self.use_for_distill = use_for_distill
        up3d = "3d" in upsample_cfg["type"]
        deblocks = []
        for i, oc in enumerate(out_channels):
            s = upsample_strides[i]
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                cfg = dict(upsample_cfg)
                cfg.pop("type")
                k = (1, s, s) if up3d else (s, s)
                up = (nn.ConvTranspose3d if up3d else nn.ConvTranspose2d)(in_channels[i], oc, k, stride=k, **cfg)
            else:
                s2 = int(np.round(1 / s))
                c3d = "3d" in conv_cfg["type"]
                k = (1, s2, s2) if c3d else (s2, s2)
                up = _conv(conv_cfg, in_channels[i], oc, k, k)
            deblocks.append(nn.Sequential(up, _norm(norm_cfg, oc), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        self.extra_conv = extra_conv
        if extra_conv is not None:
            extra = dict(extra_conv)
            self.layer_num = extra.pop("num_conv")
            kernel = tuple(extra.pop("kernel")) if "kernel" in extra else (3, 3, 3)
            padding = tuple((k - 1) // 2 for k in kernel)
            sep_kernel = tuple(extra.pop("sep_kernel")) if "sep_kernel" in extra else None
            layers = []
            for _ in range(self.layer_num):
                layers.append(_conv(extra, out_channels[-1], out_channels[-1], kernel, 1, padding))
                if sep_kernel:
                    layers.append(_conv(extra, out_channels[-1], out_channels[-1], sep_kernel, 1, tuple((k - 1) // 2 for k in sep_kernel)))
                layers += [_norm(norm_cfg, out_channels[-1]), nn.ReLU(inplace=True)]
            self.extra_blocks = nn.Sequential(*layers)

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        out = ups[0]
        for u in ups[1:]:
            out = out + u
        if self.extra_conv is not None:
            if self.use_for_distill:
                final, before = out, []
                for i in range(self.layer_num):
                    mid = self.extra_blocks[i * 3:(i + 1) * 3 - 1](final)
                    before.append(mid.clone())
                    final = self.extra_blocks[(i + 1) * 3 - 1](mid)
                return {"final": final, "before_relu": before}
            out = self.extra_blocks(out)
        return out
