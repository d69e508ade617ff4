"""On-device data path through the C ABI vs oracle/datapath.py: flip / rotate / scale of points and boxes,
PointsRangeFilter, PointSample, and the shipped SUN RGB-D train pipeline end to end (ref: uni3detr_sunrgbd.py:150-174)."""
import numpy as np
import pytest
import torch

from oracle import datapath as od

pytestmark = pytest.mark.gpu

RANGE = [-3.2, -0.2, -2.0, 3.2, 6.2, 0.56]


def _scenes(rng, lens, feat=4, spread=4.0):
    out = []
    for n in lens:
        p = (rng.random((n, feat)).astype(np.float32) - 0.5) * 2 * spread
        p[:, 1] += 3.0
        out.append(p)
    return out


def _boxes(rng, counts, dim=7):
    out = []
    for g in counts:
        b = rng.random((g, dim)).astype(np.float32)
        b[:, :3] = (b[:, :3] - 0.5) * 6
        b[:, 3:6] = 0.3 + b[:, 3:6] * 2
        b[:, 6] = (b[:, 6] - 0.5) * 2 * np.pi
        out.append(b)
    return out


@pytest.mark.parametrize("coord", [od.DEPTH, od.LIDAR])
@pytest.mark.parametrize("dim", [7, 9])
def test_augment_points_and_boxes_match_oracle(cuda, coord, dim):
    from uni3detr_amd import datapath as dp
    from uni3detr_amd import native as nv
    rng = np.random.default_rng(5 + coord + dim)
    lens, gcounts = [1500, 0, 37, 4096, 1], [3, 0, 1, 12, 0]          # ragged, an empty scene, a one-point scene, scenes without boxes
    pts, boxes = _scenes(rng, lens), _boxes(rng, gcounts, dim)
    B = len(lens)
    fh = np.array([1, 0, 1, 0, 1], bool)
    fv = np.array([0, 1, 1, 0, 0], bool)
    ang = rng.uniform(-0.5236, 0.5236, B).astype(np.float32)
    sc = rng.uniform(0.85, 1.15, B).astype(np.float32)
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts], [torch.from_numpy(b).cuda() for b in boxes],
                          box_type_3d="LiDAR" if coord == od.LIDAR else "Depth")
    tr = rng.normal(0, 0.1, (B, 3)).astype(np.float32)                # GlobalRotScaleTrans translation_std (ScanNet configs)
    tab = np.concatenate([np.stack([fh, fv, np.sin(ang), np.cos(ang), ang, sc], 1), tr], 1).astype(np.float32)
    tab = torch.from_numpy(tab).cuda()
    nv.points_augment(batch["points"], batch["scene_off"], tab, coord, 3)
    nv.boxes_augment(batch["gt_bboxes_3d"], batch["gt_off"], tab, coord)
    got_p, got_b = batch["points"].cpu().numpy(), batch["gt_bboxes_3d"].cpu().numpy()
    exp_p = np.concatenate([od.augment_points(p, fh[i], fv[i], ang[i], sc[i], coord, 3, tr[i]) for i, p in enumerate(pts)])
    exp_b = np.concatenate([od.augment_boxes(b, fh[i], fv[i], ang[i], sc[i], coord, tr[i]) for i, b in enumerate(boxes)])
    # f32 elementwise arithmetic on both sides; the device may contract a*b+c into an fma: 1 ulp of a coordinate of size <= 10
    np.testing.assert_allclose(got_p, exp_p, rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_b, exp_b, rtol=0, atol=2e-6)


def test_range_filter_is_order_preserving_and_strict(cuda):
    from uni3detr_amd import datapath as dp
    from uni3detr_amd import native as nv
    rng = np.random.default_rng(11)
    lens = [5000, 0, 1023, 1024, 1025, 3, 20000]
    pts = _scenes(rng, lens)
    pts[5][:] = 100.0                                   # a scene that loses every point
    pts[0][7, 0] = RANGE[0]                             # exactly on the lower bound: dropped (strict)
    pts[0][8, 2] = RANGE[5]                             # exactly on the upper bound: dropped
    pts[0][9, 1] = np.nan                               # NaN compares false: dropped
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts])
    src = batch["points"].clone()
    out, count = nv.points_range_filter(src, batch["scene_off"], RANGE)
    off = batch["scene_off"].cpu().numpy()
    for i, p in enumerate(pts):
        exp = od.range_filter(p, RANGE)
        assert int(count[i]) == len(exp)
        np.testing.assert_array_equal(out[off[i]:off[i] + len(exp)].cpu().numpy(), exp)
    # in place (out aliases the input): same result
    out2, count2 = nv.points_range_filter(src, batch["scene_off"], RANGE, out=src)
    assert torch.equal(count, count2)
    for i in range(len(lens)):
        n = int(count[i])
        assert torch.equal(out2[off[i]:off[i] + n], out[off[i]:off[i] + n])


def test_point_sample_distinct_rows_replacement_and_empty(cuda):
    from uni3detr_amd import datapath as dp
    from uni3detr_amd import native as nv
    rng = np.random.default_rng(3)
    lens = [5000, 300, 0, 1000, 1001]
    pts = _scenes(rng, lens, feat=6)
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts])
    count = torch.tensor([4000, 300, 0, 1000, 1001], dtype=torch.int32, device="cuda")      # scene 0: only the first 4000 rows are live
    seed = torch.tensor([12345], dtype=torch.int64, device="cuda")
    num = 1000
    out, idx = nv.point_sample(batch["points"], batch["scene_off"], count, num, seed, want_idx=True)
    out, idx = out.cpu().numpy().reshape(len(lens), num, 6), idx.cpu().numpy().reshape(len(lens), num)
    for b, n in enumerate([4000, 300, 0, 1000, 1001]):
        if n == 0:
            assert (idx[b] == -1).all() and (out[b] == 0).all()
            continue
        assert idx[b].min() >= 0 and idx[b].max() < n
        np.testing.assert_array_equal(out[b], pts[b][idx[b]])
        if n >= num:
            assert len(np.unique(idx[b])) == num                # without replacement
        else:
            assert len(np.unique(idx[b])) < num                 # with replacement (1000 draws from 300)
    assert sorted(idx[3]) == list(range(1000))                  # n == num_points: a permutation
    # another seed, another sample; the same seed, the same sample
    seed2 = torch.tensor([12346], dtype=torch.int64, device="cuda")
    _, idx2 = nv.point_sample(batch["points"], batch["scene_off"], count, num, seed2, want_idx=True)
    _, idx3 = nv.point_sample(batch["points"], batch["scene_off"], count, num, seed, want_idx=True)
    assert not np.array_equal(idx2.cpu().numpy().reshape(len(lens), num)[0], idx[0])
    assert np.array_equal(idx3.cpu().numpy().reshape(len(lens), num), idx)
    # every row is (about) equally likely: 4000 rows, 1000 picks, 400 seeds -> expected 100 hits per row
    hits = np.zeros(4000)
    for s in range(400):
        sd = torch.tensor([1000 + s], dtype=torch.int64, device="cuda")
        _, ii = nv.point_sample(batch["points"], batch["scene_off"], count, num, sd, want_idx=True)
        hits += np.bincount(ii[:num].cpu().numpy(), minlength=4000)
    assert hits.min() > 55 and hits.max() < 150 and abs(hits.mean() - 100) < 1e-9      # binomial(400, 0.25): sd 8.7


def test_shipped_sunrgbd_train_pipeline_end_to_end(cuda):
    """The device half of uni3detr_sunrgbd.py:150-174 with recorded draws == the oracle chain, scene by scene."""
    from uni3detr_amd import datapath as dp
    rng = np.random.default_rng(21)
    point_cloud_range = RANGE
    train_pipeline = [
        dict(type="LoadPointsFromFile", coord_type="DEPTH", shift_height=True, load_dim=6, use_dim=[0, 1, 2]),
        dict(type="LoadAnnotations3D"),
        dict(type="RandomFlip3D", sync_2d=False, flip_ratio_bev_horizontal=0.5),
        dict(type="GlobalRotScaleTrans", rot_range=[-0.523599, 0.523599], scale_ratio_range=[0.85, 1.15], shift_height=True),
        dict(type="PointsRangeFilter", point_cloud_range=point_cloud_range),
        dict(type="PointSample", num_points=2000),
        dict(type="DefaultFormatBundle3D", class_names=["bed"]),
        dict(type="Collect3D", keys=["points", "gt_bboxes_3d", "gt_labels_3d"]),
    ]
    pipe = dp.DevicePipeline(train_pipeline)
    assert [type(t).__name__ for t in pipe.transforms] == ["RandomFlip3D", "GlobalRotScaleTrans", "PointsRangeFilter", "PointSample"]
    lens, gcounts = [6000, 2500, 900], [4, 0, 2]
    pts, boxes = _scenes(rng, lens), _boxes(rng, gcounts)
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts], [torch.from_numpy(b).cuda() for b in boxes])
    np.random.seed(77)
    batch = pipe(batch)
    fh, ang, sc = batch["pcd_horizontal_flip"], batch["pcd_rotation_angle"], batch["pcd_scale_factor"]
    assert not np.asarray(batch["pcd_vertical_flip"]).any() and np.all(np.abs(ang) <= 0.523599) and np.all((sc >= 0.85) & (sc <= 1.15))
    got = batch["points"].cpu().numpy().reshape(3, 2000, 4)
    gb = batch["gt_bboxes_3d"].cpu().numpy()
    exp_b = np.concatenate([od.augment_boxes(b, fh[i], False, ang[i], sc[i], od.DEPTH) for i, b in enumerate(boxes)])
    np.testing.assert_allclose(gb, exp_b, rtol=0, atol=2e-6)
    for i, p in enumerate(pts):
        kept = od.range_filter(od.augment_points(p, fh[i], False, ang[i], sc[i], od.DEPTH, 3), point_cloud_range)
        # every sampled row is one of the scene's surviving augmented rows (the device and numpy draw different samples)
        keys = {r.tobytes() for r in np.round(kept, 5)}
        rows = np.round(got[i], 5)
        # rounding at 1e-5 against the 2e-6 arithmetic tolerance: compare through nearest neighbours instead of hashing when it misses
        miss = [r for r in rows if r.tobytes() not in keys]
        for r in miss[:50]:
            assert np.abs(kept - r).max(1).min() < 2e-5
        assert len(miss) < len(rows) // 2
        if len(kept) >= 2000:
            assert len(np.unique(got[i], axis=0)) == 2000
    assert batch["scene_off"].tolist() == [0, 2000, 4000, 6000]
    m = batch["uni_rot_aug"][0]
    np.testing.assert_allclose(m, od.uni_rot_aug(fh[0], False, ang[0], sc[0]), atol=1e-6)


def test_full_size_batch_properties(cuda):
    """8 scenes x 100 000 points (the config's PointSample size): counts equal the oracle's, every output row lies inside the range."""
    from uni3detr_amd import datapath as dp
    from uni3detr_amd import native as nv
    rng = np.random.default_rng(2)
    pts = _scenes(rng, [100000] * 8, feat=4, spread=3.6)
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts])
    out, count = nv.points_range_filter(batch["points"], batch["scene_off"], RANGE)
    exp = [len(od.range_filter(p, RANGE)) for p in pts]
    assert count.tolist() == exp
    seed = torch.tensor([9], dtype=torch.int64, device="cuda")
    s = nv.point_sample(out, batch["scene_off"], count, 20000, seed).cpu().numpy()
    lo, hi = np.array(RANGE[:3], np.float32), np.array(RANGE[3:], np.float32)
    assert s.shape == (160000, 4) and (s[:, :3] > lo).all() and (s[:, :3] < hi).all()
    assert len(np.unique(s[:20000], axis=0)) == 20000


def test_device_pipeline_feeds_the_captured_training_step(cuda):
    """The flow INTEGRATION.md describes: raw scenes -> device pipeline (flip / rot-scale / range filter / PointSample) -> set_batch ->
    graph replay, with a different augmented batch every iteration (ref: extra_tools/train.py:204-254 with the config's train_pipeline)."""
    import copy
    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd import datapath as dp
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.synth import room_scene
    from uni3detr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    model = build_model(copy.deepcopy(MODEL_CFG)).to(dev).train().set_precision("bf16")
    pipe = dp.DevicePipeline([
        dict(type="RandomFlip3D", sync_2d=False, flip_ratio_bev_horizontal=0.5),
        dict(type="GlobalRotScaleTrans", rot_range=[-0.523599, 0.523599], scale_ratio_range=[0.85, 1.15], shift_height=True),
        dict(type="PointsRangeFilter", point_cloud_range=RANGE),
        dict(type="PointSample", num_points=12000),
    ])

    def raw(i):                                            # two raw scenes of different sizes (30 k / 42 k points) + their boxes
        P, G, L = [], [], []
        for j, n in enumerate((30000, 42000)):
            p, g, l = room_scene(100 * i + j, n)
            gb = torch.from_numpy(g).clone()
            gb[:, 2] -= gb[:, 5] / 2
            P.append(torch.from_numpy(p).to(dev)); G.append(gb.to(dev)); L.append(torch.from_numpy(l).to(dev))
        return P, G, L

    def batch(i):
        P, G, L = raw(i)
        b = pipe(dp.pack_batch(P, G))
        return dp.unpack_batch(b, L), b

    (p0, g0, l0), b0 = batch(0)
    assert [int(p.shape[0]) for p in p0] == [12000, 12000] and all(int(g.tensor.shape[0]) == 8 for g in g0)
    ts = TrainStep(model, p0, g0, l0, graph=True, lr=1e-4)
    ts.capture(batches=[(p0, g0, l0), batch(1)[0]])
    losses, angles = [], [b0["pcd_rotation_angle"].copy()]
    for i in range(2, 5):
        (p, g, l), b = batch(i)
        angles.append(b["pcd_rotation_angle"].copy())
        ts.set_batch(p, g, l)
        losses.append(float(ts.step()))
    assert all(np.isfinite(v) and 0 < v < 1e4 for v in losses), losses
    assert len({tuple(np.round(a, 6)) for a in angles}) == len(angles)            # a new draw every iteration
    assert ts.recaptures == 0



def test_object_range_filter_on_device_and_all_shipped_pipelines_build(cuda):
    """(a) ObjectRangeFilter runs after flip / rotation / scale in the KITTI / nuScenes pipelines, so it is a device transform
    here: survivors (strict BEV range on the centres) keep their order, labels follow, yaw is wrapped into [-pi, pi) - against
    oracle/datapath.py; (b) DevicePipeline builds the train AND test pipeline of every shipped configuration (CollectUnified3D,
    ScanNet's translation_std, ...)."""
    from uni3detr_amd import datapath as dp
    rng = np.random.default_rng(77)
    pc_range = [0, -40, -3, 70.4, 40, 1]
    gcounts = [9, 0, 70, 1]                                            # a scene without boxes, one longer than a wave
    boxes, labels = [], []
    for n in gcounts:
        b = np.concatenate([rng.uniform(-10, 80, (n, 1)), rng.uniform(-50, 50, (n, 1)), rng.uniform(-2, 0, (n, 1)), rng.uniform(0.5, 4, (n, 3)),
                            rng.uniform(-7, 7, (n, 1)), rng.normal(0, 2, (n, 2))], 1).astype(np.float32)
        boxes.append(b)
        labels.append(rng.integers(0, 3, n).astype(np.int64))
    pts = _scenes(rng, [100, 50, 80, 10])
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts], [torch.from_numpy(b).cuda() for b in boxes], box_type_3d="LiDAR",
                          gt_labels_3d=[torch.from_numpy(l).cuda() for l in labels])
    batch = dp.ObjectRangeFilter(pc_range)(batch)
    _, gb, gl = dp.unpack_batch(batch)
    for i in range(len(gcounts)):
        eb, el = od.object_range_filter(boxes[i], labels[i], pc_range)
        assert gb[i].tensor.shape[0] == eb.shape[0] and 0 <= eb.shape[0] <= gcounts[i]
        if eb.shape[0]:
            np.testing.assert_allclose(gb[i].tensor.cpu().numpy(), eb, rtol=0, atol=2e-6)
        assert np.array_equal(gl[i].cpu().numpy(), el)
        if eb.shape[0]:
            assert float(np.abs(eb[:, 6]).max()) <= np.pi + 1e-6
    assert sum(int(c) for c in batch["gt_count"].tolist()) < sum(gcounts)          # something really was filtered
    from uni3detr_amd.configs import pipelines as P
    for name, cfg in P.SHIPPED.items():
        for key in ("train_pipeline", "test_pipeline"):
            pipe = dp.DevicePipeline(cfg[key])
            assert pipe.transforms or key == "test_pipeline", (name, key)


def test_object_range_filter_on_a_batch_without_any_box(cuda):
    """a batch whose scenes hold no GT box at all (plausible for KITTI / nuScenes at 2-4 scenes per GPU after filtering) must
    pass through ObjectRangeFilter: zero live boxes per scene, nothing raised (an empty tensor has no device pointer to hand over)."""
    from uni3detr_amd import datapath as dp
    rng = np.random.default_rng(3)
    pts = _scenes(rng, [100, 50])
    batch = dp.pack_batch([torch.from_numpy(p).cuda() for p in pts], [torch.zeros((0, 7)).cuda(), torch.zeros((0, 7)).cuda()], box_type_3d="LiDAR",
                          gt_labels_3d=[torch.zeros((0,), dtype=torch.int64).cuda(), torch.zeros((0,), dtype=torch.int64).cuda()])
    batch = dp.ObjectRangeFilter([0, -40, -3, 70.4, 40, 1])(batch)
    assert batch["gt_count"].tolist() == [0, 0]
    _, gb, gl = dp.unpack_batch(batch)
    assert all(b.tensor.shape[0] == 0 for b in gb) and all(l.numel() == 0 for l in gl)
