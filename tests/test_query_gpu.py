"""GPU parity: FPS, match cost, device Hungarian, aligned rotated IoU vs the CPU oracle / golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import geometry as og
from oracle import model as om
from uni3detr_amd import native as nv
from uni3detr_amd.synth import room_scene, SUNRGBD_RANGE, SUNRGBD_VOXEL

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_fps_matches_oracle_packed_and_voxel(cuda):
    pl = [room_scene(i, 20000 - 3000 * i)[0] for i in range(2)]
    _, rc, _ = og.voxelize_batch(pl, SUNRGBD_VOXEL, SUNRGBD_RANGE, 5, 16000)
    sets, offs, ns, o = [], [], [], 0
    for p in pl:                       # (i) the packed-triple view of the raw [N,4] buffer
        sets.append(p.reshape(-1)); offs.append(o); ns.append(p.shape[0]); o += p.size
    for b in range(2):                 # (ii) float-cast integer voxel coords (z,y,x): massive ties
        vc = rc[rc[:, 0] == b][:, 1:].astype(np.float32)
        sets.append(vc.reshape(-1)); offs.append(o); ns.append(vc.shape[0]); o += vc.size
    base = torch.from_numpy(np.concatenate(sets)).to(cuda)
    idx = nv.fps(base, torch.tensor(offs, dtype=torch.int64, device=cuda), torch.tensor(ns, dtype=torch.int32, device=cuda),
                 max(ns), 300).cpu().numpy()
    for s in range(4):
        ref = om.fps_packed(sets[s], ns[s], 300)
        assert np.array_equal(idx[s], ref), s


def test_fps_small_and_mixed_sets_in_one_launch(cuda):
    """One launch over sets of very different sizes: below 1024 points the upstream tie rule's T is smaller than the workgroup (general
    two-key compare), from 1024 on it degenerates to first-maximum-wins inside a thread (k_fps' two round loops); integer-lattice points
    make ties the rule; an empty set yields zeros; fewer points than samples repeats as the reference does."""
    rng = np.random.default_rng(4)
    ns = [1, 37, 300, 1023, 1024, 1025, 5000, 0]
    sets = [rng.integers(0, 6, (n, 3)).astype(np.float32) for n in ns]
    offs, o = [], 0
    for p in sets:
        offs.append(o); o += p.size
    base = torch.from_numpy(np.concatenate([p.reshape(-1) for p in sets] + [np.zeros(3, np.float32)])).to(cuda)
    idx = nv.fps(base, torch.tensor(offs, dtype=torch.int64, device=cuda), torch.tensor(ns, dtype=torch.int32, device=cuda), max(ns), 64).cpu().numpy()
    for s, n in enumerate(ns):
        if n == 0:
            assert not idx[s].any()
        else:
            assert np.array_equal(idx[s], om.fps_packed(sets[s].reshape(-1), n, 64)), (s, n)


def test_fps_large_set_streaming_path(cuda):
    p = np.random.default_rng(0).random((50000, 3)).astype(np.float32)
    base = torch.from_numpy(p.reshape(-1)).to(cuda)
    idx = nv.fps(base, torch.tensor([0], dtype=torch.int64, device=cuda), torch.tensor([50000], dtype=torch.int32, device=cuda),
                 50000, 64).cpu().numpy()
    assert np.array_equal(idx[0], om.fps_packed(p.reshape(-1), 50000, 64))


def test_fps_large_sets_split_over_workgroups(cuda):
    """Sets above 20 480 points run on several resident workgroups that exchange their round winners (k_fps_multi): one launch over
    large and small sets (slice boundaries at 20 480 / 20 481 points, an empty set, a small set with the general tie rule), integer-
    lattice points so that ties across slices and workgroups are the rule - indices identical to the oracle's sequential FPS."""
    rng = np.random.default_rng(7)
    ns = [50000, 20480, 20481, 1000, 0, 70000, 300]
    sets = [rng.integers(0, 40, (n, 3)).astype(np.float32) for n in ns]
    offs, o = [], 0
    for p in sets:
        offs.append(o); o += p.size
    base = torch.from_numpy(np.concatenate([p.reshape(-1) for p in sets] + [np.zeros(3, np.float32)])).to(cuda)
    idx = nv.fps(base, torch.tensor(offs, dtype=torch.int64, device=cuda), torch.tensor(ns, dtype=torch.int32, device=cuda), max(ns), 96).cpu().numpy()
    for s, n in enumerate(ns):
        if n == 0:
            assert not idx[s].any()
        else:
            assert np.array_equal(idx[s], om.fps_packed(sets[s].reshape(-1), n, 96)), (s, n)
    # random (tie-free) coordinates, twice in a row on the same workspace-sized problem: the round tags of the first launch must not leak
    p = rng.random((120000, 3)).astype(np.float32)
    b2 = torch.from_numpy(p.reshape(-1)).to(cuda)
    ref = om.fps_packed(p.reshape(-1), 120000, 48)
    err = nv.fps_err_buffer(cuda)
    for _ in range(2):
        got = nv.fps(b2, torch.tensor([0], dtype=torch.int64, device=cuda), torch.tensor([120000], dtype=torch.int32, device=cuda), 120000, 48, err=err).cpu().numpy()
        assert np.array_equal(got[0], ref)
    assert err.tolist() == [0, 0]                        # no workgroup gave up waiting for a sibling
    # the resident-workgroup budget: max_wg below the 6 workgroups this set needs -> the single-workgroup streaming kernel, same indices
    got = nv.fps(b2, torch.tensor([0], dtype=torch.int64, device=cuda), torch.tensor([120000], dtype=torch.int32, device=cuda), 120000, 48, max_wg=4).cpu().numpy()
    assert np.array_equal(got[0], ref)


def test_fps_time_out_is_flagged_and_every_workgroup_leaves(cuda):
    """poll_ticks = 1 (10 ns): the first workgroup to publish its round winner cannot see its siblings' in time -> err[0] = 1, err[1]
    counts the call, ALL workgroups of the call return promptly (no per-round wait: the whole call finishes in well under the 0.5 s
    default limit), indices stay valid point indices; the next call with the default limit clears err[0] and is index-identical to
    the oracle again."""
    import time
    rng = np.random.default_rng(3)
    n, m = 100000, 64
    p = rng.random((n, 3)).astype(np.float32)
    base = torch.from_numpy(p.reshape(-1)).to(cuda)
    off = torch.tensor([0, 0], dtype=torch.int64, device=cuda)
    cnt = torch.tensor([n, n], dtype=torch.int32, device=cuda)
    err = nv.fps_err_buffer(cuda)
    nv.fps(base, off, cnt, n, m, err=err)
    torch.cuda.synchronize()
    assert err.tolist() == [0, 0]
    t0 = time.perf_counter()
    idx = nv.fps(base, off, cnt, n, m, err=err, poll_ticks=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert err.tolist() == [1, 1], err.tolist()
    assert dt < 0.25, dt
    got = idx.cpu().numpy()
    assert got.min() >= 0 and got.max() < n
    idx = nv.fps(base, off, cnt, n, m, err=err)
    assert err.tolist() == [0, 1]
    ref = om.fps_packed(p.reshape(-1), n, m)
    assert np.array_equal(idx[0].cpu().numpy(), ref) and np.array_equal(idx[1].cpu().numpy(), ref)


def test_fps_large_sets_replayed_from_a_graph(cuda):
    """The multi-workgroup FPS inside a captured hipGraph, replayed on changing data: the candidate slots are cleared by a kernel of the
    graph (a memset NODE does not leave zeros on replay on this stack: HISTORY.md), so no round tag of the previous replay survives."""
    rng = np.random.default_rng(11)
    n, m = 60000, 40
    data = [rng.integers(0, 50, (n, 3)).astype(np.float32) for _ in range(3)]
    base = torch.zeros(n * 3, device=cuda)
    off = torch.tensor([0], dtype=torch.int64, device=cuda)
    cnt = torch.tensor([n], dtype=torch.int32, device=cuda)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        base.copy_(torch.from_numpy(data[0].reshape(-1)))
        nv.fps(base, off, cnt, n, m)                     # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    err = nv.fps_err_buffer(cuda)
    with torch.cuda.graph(g, stream=side):
        idx = nv.fps(base, off, cnt, n, m, err=err)
    for d in data:
        base.copy_(torch.from_numpy(d.reshape(-1)))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(idx[0].cpu().numpy(), om.fps_packed(d.reshape(-1), n, m))
        assert err.tolist() == [0, 0]


def test_fps_many_large_sets_go_out_in_groups(cuda):
    """More large sets than fit the chip at once (80 sets x 3 workgroups > 192): several launches of k_fps_multi, one result."""
    rng = np.random.default_rng(9)
    ns = [45000 + 13 * i for i in range(80)]
    pts = [rng.integers(0, 64, (n, 3)).astype(np.float32) for n in ns[:3]]
    sets = [pts[i % 3][: ns[i]] if ns[i] <= pts[i % 3].shape[0] else np.concatenate([pts[i % 3], pts[(i + 1) % 3]])[: ns[i]] for i in range(80)]
    offs, o = [], 0
    for p in sets:
        offs.append(o); o += p.size
    base = torch.from_numpy(np.concatenate([p.reshape(-1) for p in sets])).to(cuda)
    idx = nv.fps(base, torch.tensor(offs, dtype=torch.int64, device=cuda), torch.tensor(ns, dtype=torch.int32, device=cuda), max(ns), 24).cpu().numpy()
    for s in (0, 1, 40, 63, 64, 65, 79):
        assert np.array_equal(idx[s], om.fps_packed(sets[s].reshape(-1), ns[s], 24)), s


def test_fps_streaming_path_for_sets_beyond_the_split_limit(cuda):
    """More than 16 x 20 480 points: the single-workgroup streaming kernel (min-distances through the workspace)."""
    p = np.random.default_rng(1).random((350000, 3)).astype(np.float32)
    base = torch.from_numpy(p.reshape(-1)).to(cuda)
    idx = nv.fps(base, torch.tensor([0], dtype=torch.int64, device=cuda), torch.tensor([350000], dtype=torch.int32, device=cuda),
                 350000, 12).cpu().numpy()
    assert np.array_equal(idx[0], om.fps_packed(p.reshape(-1), 350000, 12))


def test_fused_fps_queries_match_oracle_bit_exact(cuda):
    """u3d_fps_prep | u3d_fps2 | u3d_fps_points (the detector's three-launch form of uni3detr.py:178-189) against the oracle's
    fps_queries: sampled indices identical, unit-cube points bit-identical (one subtraction and one correctly rounded division each)."""
    pl = [room_scene(i, 20000 - 3000 * i)[0] for i in range(3)]
    _, rc, _ = og.voxelize_batch(pl, SUNRGBD_VOXEL, SUNRGBD_RANGE, 5, 16000)
    ref = om.fps_queries(pl, rc, dict(num_query=300, fps_packed_quirk=True)).numpy()
    cat = torch.from_numpy(np.concatenate(pl)).to(cuda)
    lens = [p.shape[0] for p in pl]
    scene_off = torch.tensor(np.cumsum([0] + lens), dtype=torch.int32, device=cuda)
    vlens = [int((rc[:, 0] == b).sum()) for b in range(3)]
    voxel_off = torch.tensor(np.cumsum([0] + vlens), dtype=torch.int32, device=cuda)
    coors = torch.from_numpy(np.concatenate([rc, -np.ones((100, 4), rc.dtype)])).int().to(cuda)      # capacity rows past the count: (-1, ...)
    out, idx = nv.fps_queries(cat, coors, scene_off, voxel_off, 3, max(max(lens), 16000), 300)
    assert np.array_equal(out.cpu().numpy(), ref)
    for b in range(3):
        assert np.array_equal(idx[b].cpu().numpy(), om.fps_packed(pl[b].reshape(-1), lens[b], 300))


@pytest.mark.parametrize("groups", [3, 4])
def test_fused_query_embed_matches_torch_formulation(cuda, groups):
    """u3d_query_embed_fwd / _bwd against the cat / expand / inverse_sigmoid formulation of uni3detr_head.py:436-455 (train: 3 groups,
    eval: 4 with a random-point group), forward bit-exact up to the log, gradients of both embeddings to f32 summation order."""
    from uni3detr_amd.plugin.head import _QueryEmbed
    from uni3detr_amd.plugin.transformer import inverse_sigmoid
    torch.manual_seed(0)
    B, nq, c = 4, 300, 256
    tgt = torch.randn(2 * nq, c, device=cuda, requires_grad=True)
    anchor = torch.randn(nq, 3, device=cuda, requires_grad=True)
    fps = torch.rand(B, 2 * nq, 3, device=cuda)
    fps[0, 0] = torch.tensor([0.0, 1.0, 1e-7], device=cuda)                                # the clamps' corner cases
    rnd = torch.rand(B, nq, 3, device=cuda) if groups == 4 else None
    qe, q, r, rs = _QueryEmbed.apply(tgt, anchor, fps, rnd, groups)
    refs = [anchor.unsqueeze(0).expand(B, -1, -1), inverse_sigmoid(fps)] + ([inverse_sigmoid(rnd)] if groups == 4 else [])
    tgts = [tgt[:nq]] + [tgt[nq:]] * (groups - 1)
    exp = torch.cat([torch.cat(tgts).unsqueeze(0).expand(B, -1, -1), torch.cat(refs, 1)], -1)
    assert torch.allclose(qe, exp, rtol=1e-6, atol=1e-6)
    assert torch.equal(q, qe[..., :c]) and torch.equal(r, qe[..., c:])
    assert torch.allclose(rs, exp[..., c:].sigmoid(), rtol=1e-6, atol=1e-7)                # init_reference
    w1, w2, w3, w4 = torch.randn_like(qe), torch.randn_like(q), torch.randn_like(r), torch.randn_like(r)
    g = torch.autograd.grad((qe * w1).sum() + (q * w2).sum() + (r * w3).sum() + (rs * w4).sum(), [tgt, anchor], retain_graph=True)
    ge = torch.autograd.grad((exp * w1).sum() + (exp[..., :c] * w2).sum() + (exp[..., c:] * w3).sum() + (exp[..., c:].sigmoid() * w4).sum(),
                             [tgt, anchor])
    for a, b_ in zip(g, ge):
        assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5)
    g2 = torch.autograd.grad((q * w2).sum(), [tgt, anchor], allow_unused=True)             # only one output used: the others' gradients are None
    assert torch.allclose(g2[0], torch.autograd.grad((exp[..., :c] * w2).sum(), tgt)[0], rtol=1e-5, atol=1e-5)


def _fixture():
    z = np.load(os.path.join(G, "head_train_b2.npz"))
    gts, labels, o = [], [], 0
    for n in z["gt_lens"]:
        g = torch.from_numpy(z["gts"][o:o + n])
        gts.append(torch.cat([g[:, :2], g[:, 2:3] + g[:, 5:6] * 0.5, g[:, 3:]], 1))   # gravity centre
        labels.append(torch.from_numpy(z["labels"][o:o + n]))
        o += n
    return z, gts, labels


def test_match_cost_and_assignment_match_reference_golden(cuda):
    z, gts, labels = _fixture()
    cfg = om.sunrgbd_cfg()
    cls, box = torch.from_numpy(z["cls"]), torch.from_numpy(z["box"])
    L, B, Q, _ = cls.shape
    gt_off = torch.tensor(np.cumsum([0] + [g.shape[0] for g in gts]), dtype=torch.int32, device=cuda)
    gmax = max(g.shape[0] for g in gts)
    cost = nv.match_cost(cls.to(cuda), box.to(cuda), torch.cat(gts).to(cuda), torch.cat(labels).int().to(cuda), gt_off, gmax,
                         cfg["cost_cls"], cfg["cost_reg"], cfg["cost_iou"])
    for l in range(L):
        for b in range(B):
            ref = om.match_cost(cls[l, b], box[l, b], gts[b], labels[b], cfg)          # [Q,G]
            got = cost[l * B + b, : gts[b].shape[0]].t().cpu()
            assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()) * 0.05
    asg = nv.lsa(cost, gt_off, L, B, Q, 300, gmax).cpu().numpy()
    assert np.array_equal(asg.astype(np.int16), z["assigned"])                       # == reference assigner (scipy)


def test_lsa_random_and_degenerate_vs_scipy(cuda):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(3)
    for G, nq, integer in [(1, 300, False), (7, 300, False), (40, 300, False), (64, 300, True), (300, 300, False), (13, 900, True)]:
        B = 3
        c = rng.random((B, G, nq)).astype(np.float32)
        if integer:
            c = rng.integers(0, 4, (B, G, nq)).astype(np.float32)                    # massive ties: tie rule must equal scipy's
        gt_off = torch.tensor([0, G, 2 * G, 3 * G], dtype=torch.int32, device=cuda)
        asg = nv.lsa(torch.from_numpy(c).to(cuda), gt_off, 1, B, nq, nq, G).cpu().numpy()[0]
        for b in range(B):
            r, col = linear_sum_assignment(c[b].T)                                   # scipy on [nq, G] as the reference calls it
            exp = np.zeros(nq, np.int32)
            exp[r] = col + 1
            assert np.array_equal(asg[b], exp), (G, nq, integer, b)


def test_lsa_empty_scene(cuda):
    gt_off = torch.tensor([0, 0, 5], dtype=torch.int32, device=cuda)
    c = torch.rand(2, 5, 300, device=cuda)
    asg = nv.lsa(c, gt_off, 1, 2, 300, 300, 5).cpu().numpy()[0]
    assert asg[0].sum() == 0 and (asg[1] > 0).sum() == 5


def test_rotated_iou_aligned(cuda):
    rng = np.random.default_rng(5)
    n = 4000
    a = np.concatenate([rng.uniform(-3, 3, (n, 3)), rng.uniform(0.2, 2.5, (n, 3)), rng.uniform(-4, 4, (n, 1))], 1).astype(np.float32)
    b = a + np.concatenate([rng.normal(0, 0.4, (n, 3)), rng.normal(0, 0.2, (n, 3)), rng.normal(0, 0.5, (n, 1))], 1).astype(np.float32)
    b[:, 3:6] = np.abs(b[:, 3:6]) + 0.05
    b[:10] = 0.0                                   # unmatched rows of the loss: all-zero targets
    a[10:20, 6] = b[10:20, 6] = 0.0                # axis aligned
    got = nv.iou3d_rotated_aligned(torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)).cpu()
    ref = ob.bbox_overlaps_3d_aligned(torch.from_numpy(a), torch.from_numpy(b))
    assert (got - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trilinear_sampler_matches_grid_sample(cuda, dtype):
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, C, D, H, W, N = 2, 256, 15, 40, 40, 900
    vol = torch.randn(B, C, D, H, W)
    grid = torch.rand(B, N, 3) * 2.4 - 1.2                       # includes out-of-range samples (zero padding)
    gy = torch.randn(B, N, C)
    vr = vol.clone().requires_grad_(True)
    gr = grid.clone().requires_grad_(True)
    ref = F.grid_sample(vr, gr.view(B, 1, 1, N, 3), mode="bilinear", padding_mode="zeros", align_corners=False)
    ref = ref.view(B, C, N).transpose(1, 2)
    (ref * gy).sum().backward()
    rows = vol.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous().to(cuda).to(dtype)
    out = nv.trilinear_fwd(rows, grid.to(cuda), B, (D, H, W))
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert (out.float().cpu() - ref.detach()).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    dv, dg = nv.trilinear_bwd(rows, grid.to(cuda), gy.to(cuda).to(dtype), B, (D, H, W))
    dvr = vr.grad.permute(0, 2, 3, 4, 1).reshape(-1, C)
    assert (dv.cpu() - dvr).abs().max().item() <= (1e-4 if dtype == torch.float32 else 5e-2) * max(1.0, dvr.abs().max().item())
    assert (dg.cpu() - gr.grad).abs().max().item() <= (1e-3 if dtype == torch.float32 else 8e-2) * max(1.0, gr.grad.abs().max().item())


def _nms_case(seed, n=500, thr=0.3):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-3, 3, (40, 2))
    base = centers[rng.integers(0, 40, n)] + rng.normal(0, 0.25, (n, 2))
    boxes = np.concatenate([base, rng.uniform(-1, 0, (n, 1)), rng.uniform(0.4, 1.6, (n, 3)), rng.uniform(-3.2, 3.2, (n, 1))], 1).astype(np.float32)
    scores = rng.random(n).astype(np.float32)
    labels = rng.integers(0, 4, n).astype(np.int64)
    exp = []
    for c in range(4):
        idx = np.nonzero(labels == c)[0]
        idx = idx[np.argsort(-scores[idx], kind="stable")]
        kept = []
        for i in idx:
            ok = True
            for k in kept:
                a, b = boxes[k], boxes[i]
                inter = ob.rotated_intersection_area((a[0], a[1], a[3], a[4], a[6]), (b[0], b[1], b[3], b[4], b[6]))
                iou = inter / max(a[3] * a[4] + b[3] * b[4] - inter, 1e-8)
                if abs(iou - thr) < 1e-4:
                    return None        # too close to the threshold to demand agreement between f32 and f64 clipping
                if iou > thr:
                    ok = False
                    break
            if ok:
                kept.append(i)
        exp += kept
    return boxes, scores, labels, exp


def test_classwise_rotated_nms_matches_sequential_oracle(cuda):
    case = None
    for seed in range(9, 40):
        case = _nms_case(seed)
        if case is not None:
            break
    assert case is not None
    boxes, scores, labels, exp = case
    keep = nv.nms3d_classwise(torch.from_numpy(boxes).to(cuda), torch.from_numpy(scores).to(cuda), torch.from_numpy(labels).to(cuda), 0.3).cpu().numpy()
    assert len(exp) < len(scores) and keep.tolist() == exp


@pytest.mark.parametrize("m,k,n,relu", [(7200, 256, 256, True), (7200, 384, 256, False), (1800, 256, 512, True), (333, 512, 256, False)])
def test_linear_bf16_matches_torch(cuda, m, k, n, relu):
    """u3d_linear_bf16 (out = act(x W^T + b) on the implicit-GEMM kernels; the strided input gradients' per-offset products run on it,
    sparse._SparseConv.backward) against F.linear on the same bf16-rounded operands."""
    torch.manual_seed(m + k)
    lin = torch.nn.Linear(k, n).to(cuda)
    x = torch.randn(m, k, device=cuda).bfloat16()
    w = lin.weight.detach().bfloat16()
    ref = torch.nn.functional.linear(x.float(), w.float(), lin.bias)
    ref = torch.relu(ref) if relu else ref
    out = nv.linear_bf16(x, w, lin.bias.detach().float(), relu)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert (out.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,c,dtype", [(7200, 256, torch.bfloat16), (21600, 10, torch.float32), (333, 1024, torch.bfloat16), (1, 8, torch.float32),
                                       (0, 16, torch.float32), (4097, 36, torch.bfloat16)])
def test_colsum_matches_float64_sum(cuda, n, c, dtype):
    torch.manual_seed(n + c)
    x = torch.randn(n, c, device=cuda).to(dtype)
    got = nv.colsum(x)
    exp = x.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (c,)
    assert (got.double() - exp).abs().max().item() <= 1e-4 * max(1.0, n ** 0.5) if n else float(got.abs().sum()) == 0.0
    assert torch.equal(got, nv.colsum(x))                      # fixed summation order


@pytest.mark.parametrize("amp,defer", [(True, False), (False, False), (True, True)])
def test_safe_linear_and_in_proj_backward_match_torch(cuda, amp, defer):
    """fast_linear (shadow weights, HIP wgrad + colsum backward) and the packed in-projection against plain F.linear autograd."""
    from uni3detr_amd.plugin import transformer as T
    from uni3detr_amd.shadow import ShadowSet
    torch.manual_seed(3)
    C = 256
    lin = torch.nn.Linear(C, 512).to(cuda)
    mha = torch.nn.MultiheadAttention(C, 8).to(cuda)
    x = torch.randn(8, 900, C, device=cuda)
    pos = torch.randn(8, 900, C, device=cuda)
    q = lambda t: t.detach().bfloat16().float() if amp else t.detach().clone()
    # reference: fp32 math on (bf16-rounded when amp) operands
    xr, pr = q(x).requires_grad_(True), q(pos).requires_grad_(True)
    wl, bl = q(lin.weight).requires_grad_(True), q(lin.bias).requires_grad_(True)
    wi, bi = q(mha.in_proj_weight).requires_grad_(True), q(mha.in_proj_bias).requires_grad_(True)
    qk_r = q(xr + pr) if amp else xr + pr
    y_ref = torch.nn.functional.linear(xr, wl, bl)
    qkp_ref = torch.nn.functional.linear(xr + pr, wi[: 2 * C], bi[: 2 * C])
    v_ref = torch.nn.functional.linear(xr, wi[2 * C:], bi[2 * C:])
    gy, gq, gv = torch.randn_like(y_ref), torch.randn_like(qkp_ref), torch.randn_like(v_ref)
    if amp:
        gy, gq, gv = gy.bfloat16().float(), gq.bfloat16().float(), gv.bfloat16().float()
    (y_ref * gy).sum().backward(retain_graph=True)
    ((qkp_ref * gq).sum() + (v_ref * gv).sum()).backward()
    x2, p2 = x.detach().clone().requires_grad_(True), pos.detach().clone().requires_grad_(True)
    shadows = ShadowSet([lin.weight, lin.bias, mha.in_proj_weight, mha.in_proj_bias], torch.bfloat16)
    import contextlib
    T.reset_param_uses()
    with (shadows.active() if amp else contextlib.nullcontext()), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = T.fast_linear(x2, lin)
        qkp, v = T._InProjFn.apply(x2 + p2, x2, mha.in_proj_weight, mha.in_proj_bias, torch.bfloat16 if amp else None)
    assert y.dtype == (torch.bfloat16 if amp else torch.float32)
    tol = 3e-2 if amp else 2e-3
    rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(1.0, b.abs().max().item())
    assert rel(y, y_ref) <= tol and rel(qkp, qkp_ref) <= tol and rel(v, v_ref) <= tol
    # defer=True: parameter gradients are queued during backward and produced by the batched kernels at the end of the block
    with (T.deferred_param_grads() if defer else contextlib.nullcontext()):
        ((y.float() * gy).sum() + (qkp.float() * gq).sum() + (v.float() * gv).sum()).backward()
        if defer:
            assert len(T._Deferred.items) == 3                 # linear + the two halves of the packed in-projection
    assert lin.weight.grad.dtype == torch.float32 and mha.in_proj_weight.grad.shape == (3 * C, C)
    assert rel(lin.weight.grad, wl.grad) <= tol and rel(lin.bias.grad, bl.grad) <= tol
    assert rel(mha.in_proj_weight.grad, wi.grad) <= tol and rel(mha.in_proj_bias.grad, bi.grad) <= tol
    assert rel(x2.grad, xr.grad) <= tol and rel(p2.grad, pr.grad) <= tol
    # stale shadows are never used: modify the parameter after the refresh -> the cast path must pick up the new value
    if amp:
        with torch.no_grad():
            lin.weight.mul_(2.0)
        with shadows.active():
            pass
        with torch.no_grad():
            lin.weight.mul_(0.5)
        from uni3detr_amd import shadow as S
        S._ACTIVE[0] = True
        try:
            w_now = S.compute_copy(lin.weight, torch.bfloat16)
        finally:
            S._ACTIVE[0] = False
        assert torch.equal(w_now, lin.weight.detach().bfloat16())


@pytest.mark.parametrize("m,n,k,cnt", [(7200, 256, 256, 5), (7200, 512, 256, 2), (7200, 256, 384, 1), (1000, 64, 128, 3), (333, 256, 512, 50)])
def test_batched_weight_and_bias_gradients(cuda, m, n, k, cnt):
    """u3d_wgrad_batched_bf16 / u3d_colsum_batched: every product of the batch against its own f32 reference (NaN-prefilled outputs)."""
    torch.manual_seed(m + n + k + cnt)
    ins = [torch.randn(m, n, device=cuda).bfloat16() for _ in range(cnt)]
    dos = [torch.randn(m, k, device=cuda).bfloat16() for _ in range(cnt)]
    dws = [torch.full((n, k), float("nan"), device=cuda) for _ in range(cnt)]
    dbs = [torch.full((n,), float("nan"), device=cuda) for _ in range(cnt)]
    nv.wgrad_batched(ins, dos, dws)
    nv.colsum_batched(ins, dbs)
    for i in range(cnt):
        ref = ins[i].float().t() @ dos[i].float()
        assert (dws[i] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
        assert (dbs[i] - ins[i].float().sum(0)).abs().max().item() <= 1e-3


@pytest.mark.parametrize("n,c,xdt,ydt,relu", [(7200, 256, torch.float32, torch.float32, False), (7200, 256, torch.bfloat16, torch.bfloat16, True),
                                              (999, 384, torch.float32, torch.bfloat16, True), (33, 1024, torch.bfloat16, torch.float32, False),
                                              (5, 100, torch.float32, torch.float32, True)])
def test_fused_layer_norm_matches_torch(cuda, n, c, xdt, ydt, relu):
    from uni3detr_amd.plugin import transformer as T
    torch.manual_seed(n + c)
    ln = torch.nn.LayerNorm(c).to(cuda)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
    x = (torch.randn(n, c, device=cuda) * 2 + 0.3).to(xdt)
    xr = x.float().detach().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (c,), ln.weight, ln.bias, ln.eps)
    ref = torch.relu(ref) if relu else ref
    gy = torch.randn_like(ref).to(ydt).float()
    gw_ref, gb_ref = torch.autograd.grad(ref, [ln.weight, ln.bias], gy, retain_graph=True)
    (gx_ref,) = torch.autograd.grad(ref, [xr], gy)
    xq = x.detach().clone().requires_grad_(True)
    T.reset_param_uses()
    y = T.fused_layer_norm(xq.view(1, n, c), ln, relu=relu, out_dtype=ydt).view(n, c)
    assert y.dtype == ydt
    tol = 2e-2 if (xdt == torch.bfloat16 or ydt == torch.bfloat16) else 2e-5
    assert (y.float() - ref.detach()).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    ln.weight.grad = ln.bias.grad = None
    y.backward(gy.to(ydt))
    assert (xq.grad.float() - gx_ref).abs().max().item() <= tol * max(1.0, gx_ref.abs().max().item())
    edge = 2e-2 if relu else 1.0       # bf16 rounding of x can flip a few ReLU edges: compare the sums loosely then
    assert (ln.weight.grad - gw_ref).abs().max().item() <= max(tol, 1e-4) * max(1.0, gw_ref.abs().max().item()) * (1 + (edge < 1) * 5)
    assert (ln.bias.grad - gb_ref).abs().max().item() <= max(tol, 1e-4) * max(1.0, gb_ref.abs().max().item()) * (1 + (edge < 1) * 5)
    # deferred parameter gradients give the same values
    if n >= 33:
        T.reset_param_uses()
        ln.weight.grad = ln.bias.grad = None
        xq2 = x.detach().clone().requires_grad_(True)
        y2 = T.fused_layer_norm(xq2, ln, relu=relu, out_dtype=ydt)
        gw_now, gb_now = None, None
        with T.deferred_param_grads():
            y2.backward(gy.to(ydt))
        y3 = T.fused_layer_norm(x.detach().clone().requires_grad_(True), ln, relu=relu, out_dtype=ydt)
        w_def, b_def = ln.weight.grad.clone(), ln.bias.grad.clone()
        ln.weight.grad = ln.bias.grad = None
        y3.backward(gy.to(ydt))
        assert torch.allclose(w_def, ln.weight.grad, rtol=1e-5, atol=1e-5) and torch.allclose(b_def, ln.bias.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k,n", [(256, 10), (256, 1), (256, 8), (3, 256), (16, 512)])
@pytest.mark.parametrize("defer", [False, True])
def test_skinny_linear_parameter_gradients(cuda, k, n, defer):
    """Linear layers with <= 16 input or output features: u3d_skinny_wgrad_bf16 (+ deferred column sums) against torch autograd."""
    import contextlib
    from uni3detr_amd.plugin import transformer as T
    torch.manual_seed(k * 1000 + n)
    lin = torch.nn.Linear(k, n).to(cuda)
    x = torch.randn(8, 900, k, device=cuda)
    xr = x.detach().bfloat16().float().requires_grad_(True)
    wr, br = lin.weight.detach().bfloat16().float().requires_grad_(True), lin.bias.detach().bfloat16().float().requires_grad_(True)
    y_ref = torch.nn.functional.linear(xr, wr, br)
    gy = torch.randn_like(y_ref).bfloat16().float()
    y_ref.backward(gy)
    xq = x.detach().clone().requires_grad_(True)
    T.reset_param_uses()
    old = T.SKINNY_WGRAD
    T.SKINNY_WGRAD = True                       # opt-in path (off by default: see transformer.SKINNY_WGRAD)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = T.fast_linear(xq, lin)
        with (T.deferred_param_grads() if defer else contextlib.nullcontext()):
            y.backward(gy.bfloat16())
    finally:
        T.SKINNY_WGRAD = old
    rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(1.0, b.abs().max().item())
    assert lin.weight.grad.shape == (n, k) and rel(lin.weight.grad, wr.grad) <= 3e-2 and rel(lin.bias.grad, br.grad) <= 3e-2
    assert rel(xq.grad, xr.grad) <= 3e-2


@pytest.mark.parametrize("amp", [False, True])
def test_linear_relu_epilogue_matches_linear_then_relu(cuda, amp):
    """fast_linear(relu=True) (bias + ReLU in the GEMM epilogue, mask re-applied in the backward) == F.relu(F.linear(.)) with autograd:
    output, input gradient and parameter gradients."""
    import contextlib
    from uni3detr_amd.plugin import transformer as T
    from uni3detr_amd.shadow import ShadowSet
    torch.manual_seed(9)
    lin = torch.nn.Linear(256, 512).to(cuda)
    x = torch.randn(8, 900, 256, device=cuda)
    q = lambda t: t.detach().bfloat16().float() if amp else t.detach().clone()
    xr, wr, br = q(x).requires_grad_(True), q(lin.weight).requires_grad_(True), q(lin.bias).requires_grad_(True)
    ref = torch.relu(torch.nn.functional.linear(xr, wr, br))
    gy = q(torch.randn_like(ref))
    (ref * gy).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    shadows = ShadowSet([lin.weight, lin.bias], torch.bfloat16)
    T.reset_param_uses()
    assert T.RELU_EPILOGUE
    with (shadows.active() if amp else contextlib.nullcontext()), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = T.fast_linear(x2, lin, relu=True)
    assert y.dtype == (torch.bfloat16 if amp else torch.float32) and (y >= 0).all()
    (y.float() * gy).sum().backward()
    tol = 3e-2 if amp else 2e-3
    rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(1.0, b.abs().max().item())
    assert rel(y, ref) <= tol and rel(x2.grad, xr.grad) <= tol
    assert rel(lin.weight.grad, wr.grad) <= tol and rel(lin.bias.grad, br.grad) <= tol


@pytest.mark.parametrize("nout,kin", [(10, 256), (1, 256), (256, 3)])
def test_narrow_linear_bias_sum_is_deferred_and_correct(cuda, nout, kin):
    """Linears whose weight gradient cannot ride the batched kernel (N or K not a multiple of 64) still queue their BIAS column sum:
    inside deferred_param_grads() the bias gradient is produced by the batched sums at the end of the block and equals torch's."""
    import contextlib
    from uni3detr_amd.plugin import transformer as T
    from uni3detr_amd.shadow import ShadowSet
    torch.manual_seed(nout + kin)
    lin = torch.nn.Linear(kin, nout).to(cuda)
    x = torch.randn(8, 900, kin, device=cuda)
    q = lambda t: t.detach().bfloat16().float()
    xr, wr, br = q(x).requires_grad_(True), q(lin.weight).requires_grad_(True), q(lin.bias).requires_grad_(True)
    ref = torch.nn.functional.linear(xr, wr, br)
    gy = q(torch.randn_like(ref))
    (ref * gy).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    shadows = ShadowSet([lin.weight, lin.bias], torch.bfloat16)
    T.reset_param_uses()
    with shadows.active(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = T.fast_linear(x2, lin)
    with T.deferred_param_grads():
        (y.float() * gy).sum().backward()
        assert any(p is lin.bias for _, p in T._Deferred.sum_items)          # queued, not computed on the spot
    rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(1.0, b.abs().max().item())
    assert rel(lin.bias.grad, br.grad) <= 3e-2 and rel(lin.weight.grad, wr.grad) <= 3e-2 and rel(x2.grad, xr.grad) <= 3e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_sine_embedding_matches_torch_formulation(cuda, dtype):
    """u3d_sine_embed_fwd/_bwd == get_sine_pos_embed(ref_logits.sigmoid()) (ref uni3detr_transformer.py:33-65,181): values in the
    requested dtype and the gradient w.r.t. the reference-point logits."""
    from uni3detr_amd.plugin import transformer as T
    torch.manual_seed(4)
    logits = (torch.randn(8, 900, 3, device=cuda) * 2.0)
    a = logits.clone().requires_grad_(True)
    out = T.sine_embed_of_logits(a, dtype)
    assert out.shape == (8, 900, 384) and out.dtype == dtype
    b = logits.clone().requires_grad_(True)
    ref = T.get_sine_pos_embed(b.sigmoid())
    g = torch.randn_like(ref)
    tol = 2e-5 if dtype == torch.float32 else 8e-3
    assert (out.float() - ref).abs().max().item() <= tol
    out.backward(g.to(dtype))
    ref.backward(g.to(dtype).float())
    # sin/cos of arguments up to 2*pi with slopes up to 2*pi: f32 device sin/cos vs torch's agree to a few ulp
    assert (a.grad - b.grad).abs().max().item() <= 2e-3 * max(1.0, b.grad.abs().max().item())
