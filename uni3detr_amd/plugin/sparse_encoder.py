"""SparseEncoderHD behind the reference's registry name and constructor (ref:
projects/mmdet3d_plugin/models/pts_encoder/sparse_encoder_hd.py:14-214), running on the HIP sparse-conv path.

Parameter names follow the reference checkpoints: `conv_input.0.weight`,
`encoder_layers.encoder_layer{i}.{j}.conv1.weight / bn1.*`, `encoder_layers.encoder_layer{i}.{j}.0.weight / 1.*`,
`conv_out.0.weight / 1.*`; sparse weights are stored [kD,kH,kW,Cin,Cout] (mmcv spconv-1.x layout).
"""
import math


import torch
from torch import nn

from .. import sparse as sp
from ..registry import MIDDLE_ENCODERS


class SparseConvWeight(nn.Module):
    """Holds the weight of one SubMConv3d / SparseConv3d (bias-free, as make_sparse_convmodule builds them)."""

    def __init__(self, cin, cout, ksize, stride=(1, 1, 1), padding=(0, 0, 0), subm=False):
        super().__init__()
        self.cin, self.cout, self.ksize, self.stride, self.padding, self.subm = cin, cout, tuple(ksize), tuple(stride), tuple(padding), subm
        self.weight = nn.Parameter(torch.empty(*self.ksize, cin, cout))
        self.reset_parameters()

    def reset_parameters(self):
        # spconv's own init: kaiming_uniform(a=sqrt(5)) with fan_in = Cin * prod(k)
        fan_in = self.cin * self.ksize[0] * self.ksize[1] * self.ksize[2]
        bound = 1.0 / math.sqrt(fan_in)
        nn.init.uniform_(self.weight, -bound, bound)


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _bn(norm_cfg, c):
    cfg = dict(norm_cfg)
    assert cfg.pop("type") in ("BN1d", "BN"), "sparse rows are normalised by BatchNorm1d"
    return nn.BatchNorm1d(c, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1))


class SparseConvModule(nn.Sequential):
    """conv + BN1d + ReLU (upstream make_sparse_convmodule, order ('conv','norm','act')): children `0` and `1`."""

    def __init__(self, cin, cout, ksize, norm_cfg, stride=1, padding=0, subm=False):
        super().__init__(SparseConvWeight(cin, cout, _triple(ksize), _triple(stride), _triple(padding), subm), _bn(norm_cfg, cout))


class SparseBasicBlock(nn.Module):
    """conv1-bn1-relu-conv2-bn2-(+identity)-relu with two SubMConv3d k3 (upstream mmdet3d SparseBasicBlock)."""

    def __init__(self, planes, norm_cfg):
        super().__init__()
        self.conv1 = SparseConvWeight(planes, planes, (3, 3, 3), subm=True, padding=(1, 1, 1))
        self.bn1 = _bn(norm_cfg, planes)
        self.conv2 = SparseConvWeight(planes, planes, (3, 3, 3), subm=True, padding=(1, 1, 1))
        self.bn2 = _bn(norm_cfg, planes)


RESIDUAL_FUSION = True


@MIDDLE_ENCODERS.register_module()
class SparseEncoderHD(nn.Module):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"), norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01),
                 base_channels=16, output_channels=128, encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), encoder_strides=(2, 2, 2, 1),
                 block_type="conv_module", keep_depth=True, fp16_enabled=False):
        super().__init__()
        assert block_type in ["conv_module", "basicblock"]
        assert isinstance(order, tuple) and len(order) == 3 and set(order) == {"conv", "norm", "act"}
        if order[0] != "conv":
            raise NotImplementedError("pre-activation order is not used by any shipped Uni3DETR config")
        self.sparse_shape = tuple(sparse_shape)
        self.in_channels, self.base_channels, self.output_channels = in_channels, base_channels, output_channels
        self.encoder_channels, self.encoder_paddings, self.encoder_strides = encoder_channels, encoder_paddings, encoder_strides
        self.stage_num = len(encoder_channels)
        self.keep_depth = keep_depth
        if fp16_enabled:
            self.fp16_enabled = fp16_enabled       # attribute only exists when enabled (ref :63-64)
        self.compute_dtype = torch.float32          # set to torch.bfloat16 for the throughput mode
        # static-shape mode (hipGraph capture): row capacities of the strided levels, in order; None = exact sizes (host reads)
        self.level_capacities = None
        self.last_level_counts = []                 # device scalars (occupied rows per level) of the latest forward
        self.cin_pad = (in_channels + 7) // 8 * 8 if in_channels % 4 else in_channels
        self.conv_input = SparseConvModule(in_channels, base_channels, 3, norm_cfg, padding=1, subm=True)
        self.encoder_layers = nn.Sequential()
        cin = base_channels
        for i, blocks in enumerate(encoder_channels):
            stage = []
            for j, cout in enumerate(tuple(blocks)):
                padding = tuple(encoder_paddings[i])[j]
                if i != 0 and j == 0 and block_type == "conv_module":
                    stage.append(SparseConvModule(cin, cout, 3, norm_cfg, stride=encoder_strides[i], padding=padding))
                elif block_type == "basicblock":
                    if j == len(blocks) - 1 and i != len(encoder_channels) - 1:
                        stage.append(SparseConvModule(cin, cout, 3, norm_cfg, stride=encoder_strides[i], padding=padding))
                    else:
                        stage.append(SparseBasicBlock(cout, norm_cfg))
                else:
                    stage.append(SparseConvModule(cin, cout, 3, norm_cfg, padding=padding, subm=True))
                cin = cout
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", nn.Sequential(*stage))
        self.conv_out = SparseConvModule(cin, output_channels, (1, 1, 1), norm_cfg, stride=1, padding=0)

    # ------------------------------------------------------------------------------------------
    def _module(self, m, x, lvl, bn_in=None, bn_out=None):
        """bn_in / bn_out (sp.BnGradToken): the BatchNorm that produced x hands its backward sums to this conv's input-gradient launch;
        this module's BatchNorm fills bn_out for the next module.  Every module's output has exactly one consumer in forward()."""
        conv, bn = m[0], m[1]
        if conv.subm or conv.ksize == (1, 1, 1):
            geom = sp.subm_geom(lvl) if conv.ksize != (1, 1, 1) else sp.ConvGeom(None, None, lvl.n, lvl.n_dev, lvl.n, lvl.n_dev)
            new = lvl
        else:
            cap = None
            if self.level_capacities is not None:
                cap = self.level_capacities[len(self.last_level_counts) - 1]
            new, geom = sp.strided_level(lvl, conv.ksize, conv.stride, conv.padding, capacity=cap)
            self.last_level_counts.append(new.n_dev)
        return sp.conv_bn(x, conv.weight, geom, bn, new.n_dev, None, True, bn_in=bn_in, bn_out=bn_out), new

    def _block(self, blk, x, lvl, bn_in=None, bn_out=None):
        geom = sp.subm_geom(lvl)
        # bf16 training: the identity's gradient is summed into conv1's input gradient by that kernel's epilogue (sp.ResidualToken)
        tok = sp.ResidualToken() if (RESIDUAL_FUSION and x.dtype == torch.bfloat16 and x.requires_grad and torch.is_grad_enabled()
                                     and blk.bn1.training) else None
        # x feeds conv1 AND the identity: conv1's input gradient is x's whole gradient only when the identity's is summed in by its
        # epilogue (tok) - else autograd adds the two and the BatchNorm behind x keeps its own statistics pass
        mid = sp.BnGradToken()
        o = sp.conv_bn(x, blk.conv1.weight, geom, blk.bn1, lvl.n_dev, None, True, res_take=tok, bn_in=bn_in if tok is not None else None,
                       bn_out=mid)
        return sp.conv_bn(o, blk.conv2.weight, geom, blk.bn2, lvl.n_dev, x, True, res_give=tok, bn_in=mid, bn_out=bn_out)

    def forward(self, voxel_features, coors, batch_size):
        """voxel_features [N,C], coors int [N,4] (b,z,y,x), batch_size -> [B, C_out, D, H, W] (ref :106-138)."""
        coors = coors.int().contiguous()
        batch_size = int(batch_size)
        lvl, rank = sp.level_from_coors(coors, batch_size, self.sparse_shape)
        self.last_level_counts = [lvl.n_dev]
        x = voxel_features.float()
        w_in = self.conv_input[0].weight
        if x.shape[1] % 4:      # e.g. nuScenes' 5 point features: zero-pad channels (and weight rows) to the kernel's granule
            pad = (x.shape[1] + 3) // 4 * 4 - x.shape[1]
            x = torch.nn.functional.pad(x, (0, pad))
            w_in = torch.nn.functional.pad(w_in, (0, 0, 0, pad))
        x = sp.permute_rows(x, rank, lvl.n)
        if self.compute_dtype != torch.float32:
            if x.shape[1] % 8:
                pad = (x.shape[1] + 7) // 8 * 8 - x.shape[1]
                x = torch.nn.functional.pad(x, (0, pad))
                w_in = torch.nn.functional.pad(w_in, (0, 0, 0, pad))
            x = x.to(self.compute_dtype)
        y = sp.sparse_conv(x, w_in, sp.subm_geom(lvl))
        x = sp.bn_rows(y, self.conv_input[1], lvl.n_dev, None, True)
        prev = None                     # BnGradToken of the BatchNorm that produced x (each x below has ONE consumer: the next module)
        for stage in self.encoder_layers:
            for m in stage:
                cur = sp.BnGradToken()
                if isinstance(m, SparseBasicBlock):
                    x = self._block(m, x, lvl, prev, cur)
                else:
                    x, lvl = self._module(m, x, lvl, prev, cur)
                prev = cur
        x, lvl = self._module(self.conv_out, x, lvl, prev, None)
        dense = sp.to_dense(x, lvl)
        if not self.keep_depth:
            dense = dense.sum(dim=2)
        return dense
