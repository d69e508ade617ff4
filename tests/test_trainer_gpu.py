"""GPU: the hipGraph-captured training step (static-shape sparse levels, 3 graphs) follows the same loss trajectory as the
eager step from identical initial weights, and capacity overflow is detected."""
import copy

import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.plugin.structures import Boxes3D
from uni3detr_amd.registry import build_model
from uni3detr_amd.synth import room_scene
from uni3detr_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def _data(dev, B=2, n=12000):
    pts, gts, labels = [], [], []
    for i in range(B):
        p, g, l = room_scene(i, n)
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(dev)); gts.append(Boxes3D(gb).to(dev)); labels.append(torch.from_numpy(l).to(dev))
    return pts, gts, labels


def _model(dev, sd=None):
    torch.manual_seed(5)
    m = build_model(copy.deepcopy(MODEL_CFG))
    if sd is not None:
        m.load_state_dict(sd)
    for mod in m.modules():                       # dropout off: the two runs must be comparable
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "attn_drop"):
            mod.attn_drop = 0.0
    return m.to(dev).train().set_precision("bf16")


def test_graph_step_matches_eager_step(cuda):
    pts, gts, labels = _data(cuda)
    ref = _model(cuda)
    sd = copy.deepcopy(ref.state_dict())
    eager = TrainStep(ref, pts, gts, labels, graph=False)
    le = [float(eager.step()) for _ in range(4)]
    m2 = _model(cuda, sd)
    ts = TrainStep(m2, pts, gts, labels, graph=True)
    snap = ts.snapshot()
    counts, caps = ts.capture()
    ts.restore(snap)
    lg = [float(ts.step()) for _ in range(4)]
    ts.check_capacities()
    assert all(c <= cap for c, cap in zip(counts[1:], caps))
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * abs(a), (le, lg)
    assert le[-1] < le[0]                         # and it trains


@pytest.mark.parametrize("mode", ["mixed", "parity"])
def test_mixed_precision_graph_step_matches_eager_and_trains(cuda, mode):
    """`mixed` (fp32 encoder + backbone as split-bf16 products, 16-bit neck + head) and `parity` (f32 storage everywhere, all convolutions
    as split-bf16 products, exact-f32 decoder): the captured step - staged stage-1 graphs, FPS graph on its second stream - follows the
    eager step's loss trajectory from the same weights, and the loss goes down."""
    pts, gts, labels = _data(cuda)
    ref = _model(cuda).set_precision(mode)
    sd = copy.deepcopy(ref.state_dict())
    eager = TrainStep(ref, pts, gts, labels, graph=False)
    le = [float(eager.step()) for _ in range(4)]
    m2 = _model(cuda, sd).set_precision(mode)
    ts = TrainStep(m2, pts, gts, labels, graph=True)
    snap = ts.snapshot()
    ts.capture()
    ts.restore(snap)
    assert isinstance(ts._graphs[0], tuple) and len(ts.fps_stream_calibration_ms) == 8       # voxelize | FPS | features | head graphs
    lg = [float(ts.step()) for _ in range(4)]
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * abs(a), (le, lg)
    assert le[-1] < le[0] and lg[-1] < lg[0]


def test_dynamic_voxelization_step_is_captured(cuda):
    """ScanNet-large style configuration (dynamic voxelization + DynamicSimpleVFE, ref: uni3detr.py:155-171,
    uni3detr_scannet_large.py:28-31): the voxel list is capacity-sized with its count on the device, so the whole step captures
    (no host read of the voxel count) and follows the eager step; a batch that outgrows the list is reported by check_capacities."""
    from uni3detr_amd.configs import variants
    from uni3detr_amd.synth import room_scene
    cfg = copy.deepcopy(variants.scannet_large)
    rng_range = tuple(cfg["pts_voxel_layer"]["point_cloud_range"])
    ncls = cfg["pts_bbox_head"]["num_classes"]

    def data(seed0, n):
        pts, gts, labels = [], [], []
        for i in range(2):
            p, g, l = room_scene(seed0 + i, n, pc_range=rng_range)
            gb = torch.from_numpy(g).clone()
            gb[:, 2] -= gb[:, 5] / 2
            pts.append(torch.from_numpy(p).to(cuda)); gts.append(Boxes3D(gb).to(cuda)); labels.append((torch.from_numpy(l) % ncls).to(cuda))
        return pts, gts, labels

    def model(sd=None):
        torch.manual_seed(5)
        m = build_model(copy.deepcopy(cfg))
        if sd is not None:
            m.load_state_dict(sd)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, "attn_drop"):
                mod.attn_drop = 0.0
        return m.to(cuda).train().set_precision("bf16")

    pts, gts, labels = data(0, 30000)
    ref = model()
    assert ref.dynamic_voxelization
    sd = copy.deepcopy(ref.state_dict())
    eager = TrainStep(ref, pts, gts, labels, graph=False)
    le = [float(eager.step()) for _ in range(3)]
    m2 = model(sd)
    ts = TrainStep(m2, pts, gts, labels, graph=True)
    snap = ts.snapshot()
    ts.capture()
    ts.restore(snap)
    assert ts._graphs is not None and m2.pts_voxel_encoder.capacity is not None
    lg = [float(ts.step()) for _ in range(3)]
    ts.check_capacities()
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * abs(a), (le, lg)
    m2.pts_voxel_encoder.capacity = 256                         # a voxel list that is too short: reported, not silently truncated
    ts.eager_step()
    with pytest.raises(RuntimeError, match="overflow"):
        ts.check_capacities()


def test_fps_time_out_holds_the_update_and_falls_back_to_one_workgroup(cuda):
    """VERDICT r4 item 4 / ADVICE r4: the several-workgroup FPS (sets above 20 480 points: ScanNet-large, nuScenes) waits for sibling
    workgroups; a time-out must not train on the invalid samples until somebody looks.  The launch's flag (model.fps_err[0]) is a
    "count over capacity 0" of the step's capacity flag, i.e. part of the collective HOLD message: the update of THAT step is skipped
    (parameters, moments, step count untouched, held_steps() == 1), and at the next check the step is re-captured with the
    single-workgroup streaming FPS (fps_max_wg = 1).  The time-out is forced through the C ABI's wait limit (poll_ticks = 1 = 10 ns)."""
    from uni3detr_amd.configs import variants
    cfg = copy.deepcopy(variants.scannet_large)
    rng_range = tuple(cfg["pts_voxel_layer"]["point_cloud_range"])
    ncls = cfg["pts_bbox_head"]["num_classes"]
    pts, gts, labels = [], [], []
    for i in range(2):
        p, g, l = room_scene(i, 30000, pc_range=rng_range)
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(cuda)); gts.append(Boxes3D(gb).to(cuda)); labels.append((torch.from_numpy(l) % ncls).to(cuda))
    torch.manual_seed(5)
    m = build_model(cfg).to(cuda).train().set_precision("bf16")
    ts = TrainStep(m, pts, gts, labels, graph=True, lr=1e-3, check_every=2)
    assert ts._fps_can_time_out()
    w = m.pts_bbox_head.cls_branches[0][0].weight
    # (a) eager step with the forced time-out: held
    ts.eager_step()
    torch.cuda.synchronize()
    assert ts.held_steps() == 0 and ts.fps_timeouts() == 0 and float(ts.opt_state[0]) == 1.0
    w0, m0, v0 = w.detach().clone(), ts.exp_avg.clone(), ts.exp_avg_sq.clone()
    m.fps_poll_ticks = 1
    ts.eager_step()
    torch.cuda.synchronize()
    assert m.fps_err.tolist() == [1, 1]
    assert ts.held_steps() == 1 and float(ts.opt_state[0]) == 1.0
    assert torch.equal(w.detach(), w0) and torch.equal(ts.exp_avg, m0) and torch.equal(ts.exp_avg_sq, v0)
    # (b) captured with the forced time-out baked in: every replay is held; the periodic check re-captures on the one-workgroup path
    ts.opt_state[11:13].zero_()
    m.fps_err.zero_()
    ts.capture()
    m.fps_err.zero_()
    ts.step()
    assert ts.held_steps() == 1 and ts.fps_timeouts() == 1 and torch.equal(w.detach(), w0) and torch.equal(ts.exp_avg, m0)
    ts.step()
    assert ts.held_steps() == 2 and ts.recaptures == 0
    m.fps_poll_ticks = 0
    ts.step()                                              # check_every reached: held steps + FPS time-outs -> re-capture, then a real step
    assert ts.recaptures == 1 and m.fps_max_wg == 1 and ts.fps_timeouts_seen >= 1
    assert ts.held_steps() == 0 and ts.fps_timeouts() == 0 and not torch.equal(w.detach(), w0)
    assert not ts._fps_can_time_out()


def test_capacity_overflow_is_reported(cuda):
    pts, gts, labels = _data(cuda, n=6000)
    m = _model(cuda)
    ts = TrainStep(m, pts, gts, labels, graph=True, capacity_margin=1.0)
    ts.measure_capacities()
    m.pts_middle_encoder.level_capacities = [c // 2 // 256 * 256 for c in m.pts_middle_encoder.level_capacities]
    ts.eager_step()
    with pytest.raises(RuntimeError, match="overflow"):
        ts.check_capacities()


def test_graph_replay_gradients_match_eager_and_stay_finite(cuda):
    """Regression for the HIP-graph reduction hazard (tools/reduce_probe.py): every parameter gradient of a replayed step must
    equal the eager one (dropout off), over several replays — torch column reductions replayed from a graph return garbage on
    this stack, so the step routes them through GEMV (transformer.colsum / head.layer_sums)."""
    pts, gts, labels = _data(cuda, B=2, n=10000)
    ref = _model(cuda)
    sd = copy.deepcopy(ref.state_dict())
    eager = TrainStep(ref, pts, gts, labels, graph=False, lr=0.0)          # lr 0: weights frozen, every step sees the same problem
    eager.step()
    g_ref = eager.flat_grad.clone()
    names = [n for n, p in ref.named_parameters() if p.requires_grad]
    sizes = [p.numel() for p in ref.parameters() if p.requires_grad]
    m2 = _model(cuda, sd)
    ts = TrainStep(m2, pts, gts, labels, graph=True, lr=0.0)
    snap = ts.snapshot(); ts.capture(); ts.restore(snap)
    for it in range(6):
        ts.step()
        torch.cuda.synchronize()
        assert torch.isfinite(ts.flat_grad).all(), f"non-finite gradient at replay {it}"
        o = 0
        for n_, k in zip(names, sizes):
            a, b = ts.flat_grad[o:o + k], g_ref[o:o + k]
            o += k
            err = float((a - b).norm())
            assert err <= 3e-2 * float(b.norm()) + 1e-4, (it, n_, err, float(b.norm()))


@pytest.mark.parametrize("n,max_norm", [(1000003, 10.0), (4096, 0.0), (37, 1e-3)])
def test_flat_adamw_with_clipping_matches_torch(cuda, n, max_norm):
    """u3d_adamw_step == torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (ref optimizer config: uni3detr_sunrgbd.py:234-235)."""
    from uni3detr_amd import native as nv
    torch.manual_seed(n)
    p0 = torch.randn(n, device=cuda)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=3e-3, weight_decay=0.01)
    p = p0.clone()
    m, v, st = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(16, device=cuda)
    for it in range(6):
        g = torch.randn(n, device=cuda) * (10.0 if it % 2 else 0.01)
        pr.grad = g.clone()
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_([pr], max_norm)
        opt.step()
        nv.adamw_step(p, g, m, v, st, 3e-3, (0.9, 0.999), 1e-8, 0.01, max_norm)
        assert float(st[0]) == it + 1
        assert abs(float(st[4]) - float(g.double().norm())) <= 1e-5 * float(g.double().norm())
        assert (p - pr.detach()).abs().max().item() <= 2e-6 * max(1.0, pr.detach().abs().max().item()), it


def test_flat_update_step_matches_torch_optimizer_step(cuda):
    pts, gts, labels = _data(cuda)
    ref = _model(cuda)
    sd = copy.deepcopy(ref.state_dict())
    a = TrainStep(ref, pts, gts, labels, graph=False, flat_update=False)
    la = [float(a.step()) for _ in range(3)]
    m2 = _model(cuda, sd)
    b = TrainStep(m2, pts, gts, labels, graph=False, flat_update=True)
    lb = [float(b.step()) for _ in range(3)]
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-2 * abs(x), (la, lb)
    # parameters stay addressable under their reference names / shapes after being re-homed into the flat buffer
    assert set(m2.state_dict().keys()) == set(sd.keys())
    assert all(m2.state_dict()[k].shape == sd[k].shape for k in sd)


def test_flat_parameter_shadows_equal_per_tensor_casts(cuda):
    """u3d_cast_bf16 + u3d_permute_bf16_batched (shadow refresh over the flat parameter buffer): every linear shadow and both conv
    layouts ([K,Cin,Cout], [K,Cout,Cin]) of every conv weight, from both checkpoint layouts, bit-equal to the per-tensor cast."""
    from uni3detr_amd import shadow as S
    pts, gts, labels = _data(cuda, B=1, n=4000)
    m = _model(cuda)
    ts = TrainStep(m, pts, gts, labels, graph=False, flat_update=True)
    with torch.no_grad():
        ts.flat_param.add_(torch.randn_like(ts.flat_param) * 1e-3)          # parameters differ from what any earlier cast saw
    with m.shadow_scope():
        sh = m._shadows
        assert sh.flat is ts.flat_param
        n_lin = n_conv = 0
        for p in sh.params:
            assert torch.equal(S.compute_copy(p, torch.bfloat16), p.detach().to(torch.bfloat16)) and p._u3d_shadow[0].data_ptr() != p.data_ptr()
            n_lin += 1
        layouts = set()
        for p in sh.conv_params:
            layout = p._u3d_conv_shadow[3]
            layouts.add(layout)
            kio, koi = S.conv_weights(p, layout, torch.bfloat16)
            kio_v, koi_v = S._conv_views(p.detach(), layout)
            assert torch.equal(kio, kio_v.to(torch.bfloat16).reshape(kio.shape)) and torch.equal(koi, koi_v.to(torch.bfloat16).reshape(koi.shape))
            assert kio.is_contiguous() and koi.is_contiguous()
            n_conv += 1
    assert n_lin > 50 and n_conv > 30 and layouts == {"dhwio", "oidhw"}
    # nn.Conv3d's weights (k-contiguous source) go through the LDS-tiled kernel, the rest through the element-wise one
    assert sh._plan[4] > 0 and sh._plan[2] > 0


@pytest.mark.parametrize("buckets", [2, 3])
@pytest.mark.parametrize("graph", [False, True])
def test_two_phase_backward_matches_single_backward(cuda, graph, buckets):
    """overlap_reduce=True: backward cut at the sparse encoder's dense() output (phase A: losses/head/decoder/dense stack, phase B: the
    encoder) - and, with three buckets (the default since round 6), also at SECOND3D's outputs (phase A: losses/head/decoder/FPN, phase
    M: SECOND3D) - must leave the same flat gradient as the single backward, eager and captured; the flat buffer is
    [encoder | SECOND3D | neck + head], reduced last to first."""
    pts, gts, labels = _data(cuda)
    ref = _model(cuda)
    sd = copy.deepcopy(ref.state_dict())
    a = TrainStep(ref, pts, gts, labels, graph=False, lr=0.0)
    a.step()
    ga = a.flat_grad.clone()
    m2 = _model(cuda, sd)
    b = TrainStep(m2, pts, gts, labels, graph=graph, lr=0.0, overlap_reduce=True, reduce_buckets=buckets)
    assert b.overlap and 0 < b.n_enc < len(b.params) and b.enc_end == b.offsets[b.n_enc]
    assert all(n.startswith("pts_middle_encoder.") for n, _ in list(m2.named_parameters())[:b.n_enc])
    assert b.three_phase == (buckets == 3)
    if b.three_phase:
        assert b.enc_end < b.bb_end < b.flat_grad.numel() and b.bb_end == b.offsets[b.n_bb_end]
        assert all(n.startswith("pts_backbone.") for n, _ in list(m2.named_parameters())[b.n_enc:b.n_bb_end])
        assert list(m2.named_parameters())[b.n_bb_end][0].startswith("pts_neck.")
    if graph:
        snap = b.snapshot()
        b.capture()
        b.restore(snap)
    b.step()
    lb = float(b.step())
    gb = b.flat_grad
    assert abs(lb - float(a.loss)) <= 2e-2 * abs(float(a.loss))
    # bf16 kernels, identical math: the schedules differ in the order the three SECOND3D branches' gradients are summed into the
    # bf16 encoder-output gradient (AccumulateGrad on the cut leaf vs the engine's input buffer) and in capacity padding under capture
    rel = (ga - gb).norm().item() / ga.norm().item()
    assert rel <= (5e-2 if graph else 5e-3), rel
    enc = slice(0, b.enc_end)
    assert gb[enc].abs().sum() > 0 and gb[b.enc_end:].abs().sum() > 0
    if b.three_phase:                      # every bucket on its own, so a bucket packed into the wrong slice cannot hide in the total
        for lo, hi in ((b.enc_end, b.bb_end), (b.bb_end, gb.numel())):
            r = (ga[lo:hi] - gb[lo:hi]).norm().item() / ga[lo:hi].norm().item()
            assert r <= (5e-2 if graph else 5e-3), (lo, hi, r)


def test_adamw_state_entry_follows_hyper_parameters_and_skip_mask(cuda):
    """u3d_adamw_step_state reads lr / betas / weight decay from the device state (u3d_adamw_set_hyper) - what lets a captured graph
    follow the step / cyclic schedules - and leaves chunks flagged in the skip mask untouched (torch.optim.AdamW skips .grad is None)."""
    from uni3detr_amd import native as nv
    torch.manual_seed(3)
    n = 4096
    p0 = torch.randn(n, device=cuda)
    pr = torch.nn.Parameter(p0[:2048].clone())                     # second half: "no gradient" -> skipped
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.95, 0.99), weight_decay=0.05)
    p, m, v, st = p0.clone(), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda), torch.zeros(16, device=cuda)
    skip = torch.zeros(n // 64, dtype=torch.uint8, device=cuda)
    skip[2048 // 64:] = 1
    for it in range(5):
        lr = 1e-3 * (1 + it)                                        # a schedule
        for g_ in opt.param_groups:
            g_["lr"] = lr
        g = torch.randn(n, device=cuda)
        g[2048:] = 0
        pr.grad = g[:2048].clone()
        opt.step()
        nv.adamw_set_hyper(st, lr, (0.95, 0.99), 1e-8, 0.05, 0.0)
        nv.adamw_step_state(p, g, m, v, st, skip)
        assert (p[:2048] - pr.detach()).abs().max().item() <= 2e-6 * max(1.0, pr.detach().abs().max().item()), it
        assert torch.equal(p[2048:], p0[2048:]) and float(m[2048:].abs().max()) == 0.0


def test_captured_step_follows_lr_schedule_and_new_batches(cuda):
    """A captured TrainStep (a) takes its learning rate from device state: lr = 0 freezes the weights, a later set_hyper moves them
    again, with no re-capture; (b) trains on batches loaded with set_batch: the loss of a replay equals the eager loss on that batch;
    (c) capture() leaves weights / optimizer state where they were (the warm-up iterations do not count as training)."""
    pts, gts, labels = _data(cuda)
    pts2, gts2, labels2 = [], [], []
    from uni3detr_amd.synth import room_scene
    for i in range(2):
        p_, g_, l_ = room_scene(40 + i, 12000)
        gb = torch.from_numpy(g_).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts2.append(torch.from_numpy(p_).to(cuda)); gts2.append(Boxes3D(gb[:5]).to(cuda)); labels2.append(torch.from_numpy(l_[:5]).to(cuda))
    m = _model(cuda)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ts = TrainStep(m, pts, gts, labels, graph=True, lr=1e-3)
    ts.capture(batches=[(pts, gts, labels), (pts2, gts2, labels2)])
    for k, v in m.state_dict().items():                                         # (c)
        assert torch.equal(v, sd0[k]), k
    assert float(ts.opt_state[0]) == 0.0
    ts.set_hyper(lr=0.0, weight_decay=0.0)
    w = m.pts_bbox_head.cls_branches[0][0].weight
    w0 = w.detach().clone()
    ts.step()
    assert torch.equal(w.detach(), w0)                                          # (a) lr = 0: nothing moves
    ts.set_batch(pts2, gts2, labels2)
    l_replay = float(ts.step())
    ref = _model(cuda, {k: v.clone() for k, v in m.state_dict().items()})
    eager = TrainStep(ref, pts2, gts2, labels2, graph=False, lr=0.0, weight_decay=0.0)
    l_eager = float(eager.step())
    assert abs(l_replay - l_eager) <= 2e-2 * abs(l_eager), (l_replay, l_eager)   # (b)
    ts.set_hyper(lr=1e-3)
    ts.step()
    assert not torch.equal(w.detach(), w0)                                      # (a) and now they move
    with pytest.raises(ValueError):
        ts.set_batch([pts[0][:100], pts[1]], gts, labels)


def test_capacity_overflow_holds_the_update_and_recaptures_collectively(cuda):
    """VERDICT r2 / ADVICE: a sparse level that outgrows its captured capacity must neither train on truncated levels nor be handled
    by one rank alone.  The device computes the flag inside G1, it rides in the positive-count all-reduce, G3 holds the update when it
    is set; every `check_every` steps the (rank-identical) held-step counter triggers recapture(), which tears the process group down
    through pg_hooks, captures, and creates it again.  One rank + a gloo group exercises exactly that code path."""
    import os
    import socket
    import torch.distributed as dist
    pts, gts, labels = _data(cuda)
    # the same scenes squeezed to a fifth of their extent: far fewer occupied voxels on every strided level -> small capacities
    ctr = torch.tensor([0.0, 3.0, -0.7, 0.0], device=cuda)
    small = [(p - ctr) * torch.tensor([0.2, 0.2, 0.2, 1.0], device=cuda) + ctr for p in pts]
    m = _model(cuda)
    gen = [0]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()

    def setup():
        store = dist.TCPStore("127.0.0.1", port + gen[0], 1, True)
        gen[0] += 1
        dist.init_process_group("gloo", store=store, rank=0, world_size=1)

    def teardown():
        dist.destroy_process_group()
    ts = TrainStep(m, small, gts, labels, graph=True, lr=1e-3, capacity_margin=1.0, check_every=3, pg_hooks=(teardown, setup))
    ts.capture(batches=[(small, gts, labels)])
    setup()
    try:
        ts.enable_dist()
        with pytest.raises(RuntimeError):
            ts.capture()                                   # a capture under a live process group is refused, not attempted
        w = m.pts_bbox_head.cls_branches[0][0].weight
        ts.step()
        assert ts.held_steps() == 0
        w0 = w.detach().clone()
        ts.set_batch(pts, gts, labels)                     # the real scenes overflow the squeezed scenes' capacities
        ts.step(); ts.step()
        assert ts.held_steps() == 2 and torch.equal(w.detach(), w0)      # both updates were held: nothing trained on truncated levels
        with pytest.raises(RuntimeError):
            ts.check_capacities()
        ts.step()                                          # check_every reached: collective re-capture, then a real step
        assert ts.recaptures == 1 and ts.dist_on and dist.is_initialized()
        assert ts.held_steps() == 0 and not torch.equal(w.detach(), w0)
        ts.check_capacities()
        assert float(ts.opt_state[0]) == 2.0               # optimizer step count: 1 before the overflow + 1 after; held steps not counted
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_overflow_without_flat_update_recaptures_and_keeps_the_bound_batch(cuda):
    """graph=True with flat_update=False has no device-side hold flag (torch.optim): step() must notice a level that
    outgrew its captured capacity on the host BEFORE the backward / update, re-capture, and then train on the batch the caller had
    bound - not on the last of the capture batches that recapture() cycles through the static buffers."""
    pts, gts, labels = _data(cuda)
    ctr = torch.tensor([0.0, 3.0, -0.7, 0.0], device=cuda)
    small = [(p - ctr) * torch.tensor([0.2, 0.2, 0.2, 1.0], device=cuda) + ctr for p in pts]
    m = _model(cuda)
    ts = TrainStep(m, small, gts, labels, graph=True, flat_update=False, capacity_margin=1.0)
    ts.capture(batches=[(small, gts, labels)])
    ts.step()
    assert ts.recaptures == 0
    ts.set_batch(pts, gts, labels)                         # the real scenes overflow the squeezed scenes' capacities
    bound = ts.pts["cat"].clone()
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    loss = float(ts.step())
    assert ts.recaptures == 1 and torch.isfinite(torch.tensor(loss))
    assert torch.equal(ts.pts["cat"], bound)               # the static input buffers hold the caller's batch again
    ts.check_capacities()                                  # and the new capacities cover it
    # the step that ran was a step on `pts` from the weights it started with (the re-capture's own warm-up steps were rolled back):
    # its loss is the eager loss on that batch from those weights
    eager = TrainStep(_model(cuda, before), pts, gts, labels, graph=False, flat_update=False, lr=0.0)
    l_eager = float(eager.step())
    assert abs(l_eager - loss) <= 2e-2 * abs(loss), (l_eager, loss)


def test_optimizer_state_round_trip(cuda):
    """resume_from (ref: extra_tools/train.py:141-142): flat AdamW moments / step count exported and re-imported bit for bit."""
    pts, gts, labels = _data(cuda)
    m = _model(cuda)
    ts = TrainStep(m, pts, gts, labels, graph=False, lr=1e-3)
    for _ in range(2):
        ts.step()
    sd = ts.optimizer_state_dict()
    msd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    l3 = float(ts.step())
    m2 = _model(cuda, msd)
    ts2 = TrainStep(m2, pts, gts, labels, graph=False, lr=1e-3)
    ts2.load_optimizer_state_dict(sd)
    assert float(ts2.opt_state[0]) == 2.0
    l3b = float(ts2.step())
    assert abs(l3 - l3b) <= 1e-3 * abs(l3), (l3, l3b)
    a = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    b = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
    assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
