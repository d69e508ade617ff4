"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement (numpy / torch-CPU) of the un-vendored third-party ops the Uni3DETR hot path reaches:
hard voxelization + mean VFE, sparse-conv active-set / rulebook rules, sparse convolution, BatchNorm1d on
sparse rows, dense().  The reference repo contains none of this code (it lives in mmcv / mmdet3d / spconv,
which are not installed and not under /root/reference) — PARITY UNPINNED against upstream binaries; the
semantics restated here are those written down in SURVEY.md §8a (a-2, a-4) and Appendix A2-A4, and they are
pinned by self-checking property tests (tests/test_oracle_cpu.py): sparse conv == F.conv3d on the densified
input restricted to the active output set, BN == F.batch_norm, etc.

Reference call sites:
  voxelize      projects/mmdet3d_plugin/models/detectors/uni3detr.py:148 (MVXTwoStageDetector.voxelize)
  VFE           uni3detr.py:149 (HardSimpleVFE), cfg projects/configs/uni3detr/uni3detr_sunrgbd.py:28-31
  sparse convs  projects/mmdet3d_plugin/models/pts_encoder/sparse_encoder_hd.py:71-104,140-214
  dense()       sparse_encoder_hd.py:133
"""
import numpy as np
import torch


# --------------------------------------------------------------------------------------------------
# hard voxelization (SURVEY.md App. A2 sequential semantics)
# --------------------------------------------------------------------------------------------------
def grid_size(voxel_size, pc_range):
    vs = np.asarray(voxel_size, np.float32)
    pr = np.asarray(pc_range, np.float32)
    return np.round((pr[3:] - pr[:3]) / vs).astype(np.int64)  # (x, y, z)


def voxelize_hard(points, voxel_size, pc_range, max_points, max_voxels):
    """points f32 [N,F] -> voxels [V,max_points,F] f32, coors [V,3] i32 (z,y,x), num [V] i32.  Point order matters."""
    points = np.asarray(points, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    g = grid_size(voxel_size, pc_range)
    q = (points[:, :3] - lo) / vs            # fp32 subtract, fp32 IEEE divide
    ok = np.all((q >= 0) & (q < g.astype(np.float32)), axis=1)
    c = np.floor(np.where(ok[:, None], q, 0)).astype(np.int64)
    table = {}
    F = points.shape[1]
    voxels = np.zeros((max_voxels, max_points, F), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    nv = 0
    for i in range(points.shape[0]):
        if not ok[i]:
            continue
        key = (int(c[i, 2]), int(c[i, 1]), int(c[i, 0]))
        v = table.get(key, -1)
        if v == -1:
            if nv >= max_voxels:
                continue
            v = nv
            nv += 1
            table[key] = v
            coors[v] = key
        if num[v] < max_points:
            voxels[v, num[v]] = points[i]
            num[v] += 1
    return voxels[:nv], coors[:nv], num[:nv]


def vfe_mean(voxels, num, num_features):
    """HardSimpleVFE: voxels[:, :, :nf].sum(1) / num."""
    v = torch.from_numpy(voxels[:, :, :num_features])
    return (v.sum(1) / torch.from_numpy(num).to(v.dtype).view(-1, 1)).numpy()


def voxelize_dynamic(points_list, voxel_size, pc_range):
    """Dynamic mode (max_num_points=-1): per-point (b,z,y,x) or (b,-1,-1,-1); then DynamicScatter mean:
    unique rows in lexicographic order + mean of member points (ref uni3detr.py:155-171)."""
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    g = grid_size(voxel_size, pc_range)
    coors = []
    for b, p in enumerate(points_list):
        p = np.asarray(p, np.float32)
        q = (p[:, :3] - lo) / vs
        ok = np.all((q >= 0) & (q < g.astype(np.float32)), axis=1)
        c = np.floor(np.where(ok[:, None], q, 0)).astype(np.int32)[:, ::-1]
        c = np.where(ok[:, None], c, -1)
        coors.append(np.concatenate([np.full((p.shape[0], 1), b, np.int32), c], 1))
    coors = np.concatenate(coors)
    pts = np.concatenate([np.asarray(p, np.float32) for p in points_list])
    valid = coors[:, 1] >= 0
    uniq, inv, cnt = np.unique(coors[valid], axis=0, return_inverse=True, return_counts=True)
    sums = np.zeros((uniq.shape[0], pts.shape[1]), np.float64)
    np.add.at(sums, inv.reshape(-1), pts[valid].astype(np.float64))
    return coors, (sums / cnt[:, None]).astype(np.float32), uniq.astype(np.int32)


def voxelize_batch(points_list, voxel_size, pc_range, max_points, max_voxels):
    """MVXTwoStageDetector.voxelize: per scene then concat with leading batch index."""
    vox, coo, num = [], [], []
    for b, p in enumerate(points_list):
        v, c, n = voxelize_hard(p, voxel_size, pc_range, max_points, max_voxels)
        vox.append(v)
        coo.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
        num.append(n)
    return np.concatenate(vox), np.concatenate(coo), np.concatenate(num)


# --------------------------------------------------------------------------------------------------
# active sets and neighbour tables
# --------------------------------------------------------------------------------------------------
def _key(coors, dims):
    c = np.asarray(coors, np.int64)
    return ((c[:, 0] * dims[0] + c[:, 1]) * dims[1] + c[:, 2]) * dims[2] + c[:, 3]


def conv_out_dims(dims, ksize, stride, pad):
    return tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(dims, ksize, stride, pad))


def strided_out_coords(coors, dims_in, ksize, stride, pad):
    """SparseConv3d active-set rule: output o is active iff some kappa has o*s - p + kappa == an active input.
    Returns lexicographically sorted unique (b,z,y,x) int32 and the output dims."""
    dims_out = conv_out_dims(dims_in, ksize, stride, pad)
    c = np.asarray(coors, np.int64)
    outs = []
    for kz in range(ksize[0]):
        for ky in range(ksize[1]):
            for kx in range(ksize[2]):
                t = c[:, 1:] + np.array(pad) - np.array([kz, ky, kx])
                okm = np.all((t >= 0) & (t % np.array(stride) == 0), axis=1)
                o = t // np.array(stride)
                okm &= np.all(o < np.array(dims_out), axis=1)
                outs.append(np.concatenate([c[okm, :1], o[okm]], 1))
    allo = np.concatenate(outs)
    keys = np.unique(_key(allo, dims_out))
    out = np.zeros((keys.shape[0], 4), np.int64)
    k = keys.copy()
    out[:, 3] = k % dims_out[2]; k //= dims_out[2]
    out[:, 2] = k % dims_out[1]; k //= dims_out[1]
    out[:, 1] = k % dims_out[0]; k //= dims_out[0]
    out[:, 0] = k
    return out.astype(np.int32), dims_out


def nbr_table(q_coors, t_coors, t_dims, ksize, stride, pad, mode):
    """[K, Nq] int64 index into t_coors rows (or -1).  mode 0: target = q*s - p + kappa; mode 1: target = (q+p-kappa)/s."""
    q = np.asarray(q_coors, np.int64)
    tk = _key(t_coors, t_dims)
    order = np.argsort(tk)
    tks = tk[order]
    K = ksize[0] * ksize[1] * ksize[2]
    out = np.full((K, q.shape[0]), -1, np.int64)
    k = 0
    s = np.array(stride)
    p = np.array(pad)
    for kz in range(ksize[0]):
        for ky in range(ksize[1]):
            for kx in range(ksize[2]):
                kap = np.array([kz, ky, kx])
                if mode == 0:
                    t = q[:, 1:] * s - p + kap
                    ok = np.ones(q.shape[0], bool)
                else:
                    u = q[:, 1:] + p - kap
                    ok = np.all((u >= 0) & (u % s == 0), axis=1)
                    t = u // s
                ok &= np.all((t >= 0) & (t < np.array(t_dims)), axis=1)
                key = _key(np.concatenate([q[:, :1], np.where(ok[:, None], t, 0)], 1), t_dims)
                pos = np.searchsorted(tks, key)
                pos = np.clip(pos, 0, max(len(tks) - 1, 0))
                hit = ok & (len(tks) > 0) & (tks[pos] == key if len(tks) else False)
                out[k, hit] = order[pos[hit]]
                k += 1
    return out


def block_major_order(coors, dims):
    """Row order used internally by the HIP path: sort by (4x4x4 block index, bit index) — see include/u3d_hip.h."""
    c = np.asarray(coors, np.int64)
    bz, by, bx = [(d + 3) // 4 for d in dims]
    word = ((c[:, 0] * bz + c[:, 1] // 4) * by + c[:, 2] // 4) * bx + c[:, 3] // 4
    bit = (c[:, 1] % 4) * 16 + (c[:, 2] % 4) * 4 + (c[:, 3] % 4)
    return np.argsort(word * 64 + bit, kind="stable")


# --------------------------------------------------------------------------------------------------
# sparse conv / BN / dense  (torch CPU, differentiable)
# --------------------------------------------------------------------------------------------------
def sparse_conv(feats, weight, nbr):
    """out[m] = sum_k feats[nbr[k][m]] @ weight[k]; feats [Nin,Cin], weight [K,Cin,Cout], nbr [K,Nout] (-1 none)."""
    nbr = torch.as_tensor(nbr)
    out = feats.new_zeros((nbr.shape[1], weight.shape[2]))
    zero_row = feats.new_zeros((1, feats.shape[1]))
    fz = torch.cat([feats, zero_row], 0)
    for k in range(weight.shape[0]):
        idx = nbr[k].clone()
        if not bool((idx >= 0).any()):
            continue
        idx[idx < 0] = feats.shape[0]
        out = out + fz.index_select(0, idx) @ weight[k]
    return out


def dense_conv_reference(feats, coors, dims_in, weight, ksize, stride, pad, batch):
    """Ground truth for property tests: densify, F.conv3d, return dense output [B,Cout,*dims_out]."""
    cin = feats.shape[1]
    vol = feats.new_zeros((batch, cin) + tuple(dims_in))
    c = torch.as_tensor(coors, dtype=torch.long)
    vol[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    w = weight.view(ksize[0], ksize[1], ksize[2], cin, -1).permute(4, 3, 0, 1, 2).contiguous()
    return torch.nn.functional.conv3d(vol, w, None, stride, pad)


def bn_train(x, gamma, beta, eps, residual=None, relu=True):
    y = torch.nn.functional.batch_norm(x, None, None, gamma, beta, True, 0.0, eps)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def to_dense(feats, coors, batch, dims):
    vol = feats.new_zeros((batch, feats.shape[1]) + tuple(dims))
    c = torch.as_tensor(coors, dtype=torch.long)
    vol[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    return vol
