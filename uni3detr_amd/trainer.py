"""Training step driver: eager (DDP) or hipGraph-captured.

Graph mode captures the step as three HIP graphs on static buffers — the MI355X answer to the ~3500 short launches a
step consists of (host enqueue time ≈ GPU time in eager mode):
    G1  voxelize -> sparse encoder -> dense stack -> FPS -> decoder/head -> match cost + Hungarian -> targets
        [eager] all-reduce of the per-layer positive counts (3 floats; skipped for world size 1)
    G2  losses + backward into ONE flat gradient buffer
        [eager] all-reduce of the flat gradient buffer over RCCL/xGMI (world size > 1)
        (overlap_reduce=True: the backward is cut at SECOND3D's outputs and at the sparse encoder's dense() output - THREE graphs and
         three gradient buckets in reverse layer order: G2a = losses + head / decoder / FPN backward -> asynchronous
         all-reduce of the neck + head slice; G2m = SECOND3D's backward underneath it -> asynchronous all-reduce of the backbone
         slice; G2b = the sparse encoder's backward underneath both; then the all-reduce of the encoder's small slice)
    G3  gradient clipping + fused AdamW
The sparse levels run in static-shape mode (capacity-sized tensors, device-side row counts; uni3detr_amd/sparse.py), so the
captured launches are valid for any batch whose level sizes fit the capacities.  A batch that does NOT fit is handled on the device and
collectively: G1 ends with a capacity flag (u3d_capacity_flag), the flag travels in the positive-count all-reduce (so every rank sees
the job-wide value) and G3's update is held when it is set (u3d_adamw_step_hold) - no rank ever trains on truncated levels, and the
ranks cannot diverge.  Every `check_every` steps the host reads the held-step counter (identical on all ranks) and re-captures with more
room: on one rank directly; with a live process group through `pg_hooks` (tear the group down, capture, create it again - capture and
RCCL do not mix on this stack), or it raises on ALL ranks when no hooks were given.
"""
import os

import torch
import torch.distributed as dist

from . import native as nv
from . import sparse as _sp
from .plugin import transformer as _T


class FpsTimeout(RuntimeError):
    """check_capacities(): the step's FPS launch did not finish (the caller re-captures with the single-workgroup FPS)."""


class TrainStep:
    def __init__(self, model, points, gt_bboxes_3d, gt_labels_3d, lr=1e-4, weight_decay=0.01, max_norm=10.0, graph=True,
                 capacity_margin=1.25, flat_update=True, overlap_reduce=False, betas=(0.9, 0.999), eps=1e-8, gt_capacity=64,
                 check_every=50, pg_hooks=None, fps_graph=None, grad_comm_dtype=torch.float32, reduce_buckets=None):
        """pg_hooks: (teardown, setup) callables that destroy / re-create the default process group; needed only for a collective
        re-capture after a capacity overflow on a multi-rank run (bench.py passes them)."""
        self.model = model
        # gradient exchange dtype (N > 1): float32 (default: the all-reduced gradient is the exact mean) or bfloat16 - half the xGMI
        # bytes (63 instead of 127 MB per step), the mean rounded to 8 mantissa bits per hop of the ring (what torch's bf16 DDP
        # compression hook trades too); the master gradient / moments stay f32 either way
        assert grad_comm_dtype in (torch.float32, torch.bfloat16)
        self.grad_comm_dtype = grad_comm_dtype
        self._comm = None
        # captured step: the FPS rounds as their own graph on a second stream next to the encoder / dense stack (see capture())
        self.fps_graph = (os.environ.get("U3D_FPS_GRAPH", "1") == "1") if fps_graph is None else bool(fps_graph)
        self._fps_stream = None
        self.fps_timeouts_seen = 0
        self.pg_hooks = pg_hooks
        self._capture_batches = None
        self._msg = None
        self.dev = next(model.parameters()).device
        # time-out record of the several-workgroup FPS (sets above 20 480 points; native.fps_err_buffer): [0] = this step's launch gave
        # up waiting for a sibling workgroup - ORed into the step's collective HOLD flag (_stage1_head), so NO rank applies an update
        # computed from such samples; [1] = how often that happened, read with the held-step counter (held_steps / step)
        model.fps_err = nv.fps_err_buffer(self.dev)
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.dist_on else 1
        self.max_norm = max_norm
        self.graph = graph
        self.params = [p for p in model.parameters() if p.requires_grad]
        # one flat gradient buffer; every .grad is a view of it (static addresses for capture, one all-reduce message)
        # every parameter starts on a 256-byte boundary of the flat buffers: torch picks its vectorised kernels (LayerNorm, bias
        # adds) by pointer alignment, and packed offsets had silently moved LayerNorm to the scalar RowwiseMoments path
        ALIGN = 64
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=self.dev)
        self.views = []
        for p, o in zip(self.params, self.offsets):
            self.views.append(self.flat_grad[o:o + p.numel()].view_as(p))
            p.grad = self.views[-1]
            # conv weight gradients are reduced straight into this view (sparse._SparseConv.backward): the packing copy below then
            # only moves the small tensors
            p._u3d_grad_view = self.views[-1]
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, tuple(betas), eps
        self.flat_update = flat_update
        self.check_every, self._steps_since_check, self.recaptures = check_every, 0, 0
        self._grad_missing = [False] * len(self.params)
        # overlap_reduce: the backward runs in two phases cut at the sparse encoder's dense() output.  Phase A (losses, head, decoder,
        # dense stack: ~90 % of the gradient bytes) is followed by an ASYNCHRONOUS all-reduce of its slice of the flat buffer, which then
        # rides under phase B (the sparse encoder's backward); only the encoder's small slice is reduced after it.  The encoder's
        # parameters must form a prefix of the flat buffer (they do: it is the first parameterised child of the detector).
        enc = getattr(model, "pts_middle_encoder", None)
        enc_ids = {id(p) for p in enc.parameters()} if enc is not None else set()
        n_enc = sum(1 for p in self.params if id(p) in enc_ids)
        prefix = n_enc > 0 and all(id(p) in enc_ids for p in self.params[:n_enc]) and n_enc < len(self.params)
        self.overlap = bool(overlap_reduce and prefix)
        model.cut_encoder_backward = self.overlap
        self.n_enc = n_enc
        self.enc_end = self.offsets[n_enc] if self.overlap else 0          # flat offset where phase A's slice starts
        # second cut (round 6): SECOND3D's parameters follow the encoder's in the flat buffer (module order of the detector); the phase
        # between the two cuts is the backbone's backward, and the slice behind it (neck, head, decoder) leaves first
        bb = getattr(model, "pts_backbone", None)
        bb_ids = {id(p) for p in bb.parameters()} if bb is not None else set()
        n_bb = sum(1 for p in self.params if id(p) in bb_ids)
        mid = (self.overlap and n_bb > 0 and all(id(p) in bb_ids for p in self.params[n_enc:n_enc + n_bb]) and n_enc + n_bb < len(self.params)
               and getattr(model, "pts_neck", None) is not None
               and int(reduce_buckets if reduce_buckets is not None else os.environ.get("U3D_REDUCE_BUCKETS", "3")) >= 3)
        self.three_phase = bool(mid)
        model.cut_backbone_backward = self.three_phase
        self.n_bb_end = n_enc + n_bb if self.three_phase else n_enc
        self.bb_end = self.offsets[self.n_bb_end] if self.three_phase else self.enc_end
        self._works = []                                                   # asynchronous bucket reductions in flight: (work, staging, lo, hi)
        self.comm_diag, self._comm_events = False, []
        if flat_update:
            # parameters re-homed into ONE flat buffer (each p.data becomes a view; names/shapes/state_dict unchanged), moments flat:
            # clip + AdamW is u3d_adamw_step - three launches that stream the 7 arrays once (torch: ~30 multi-tensor launches)
            self.flat_param = torch.zeros(n, dtype=torch.float32, device=self.dev)
            with torch.no_grad():
                for p, o in zip(self.params, self.offsets):
                    v = self.flat_param[o:o + p.numel()].view_as(p)
                    v.copy_(p.data)
                    p.data = v
            # the bf16 parameter shadows of the model are (re)built over the flat buffer: refresh = cast + one re-layout launch
            model._flat_params = (self.flat_param, {id(p): o for p, o in zip(self.params, self.offsets)})
            if hasattr(model, "_shadows"):
                model._shadows = None
            self.exp_avg = torch.zeros_like(self.flat_param)
            self.exp_avg_sq = torch.zeros_like(self.flat_param)
            self.opt_state = torch.zeros(16, dtype=torch.float32, device=self.dev)
            self._opt_ws = torch.empty(int(nv.lib().u3d_adamw_workspace(n)), dtype=torch.uint8, device=self.dev)
            self.opt = None
            self._skip = None                 # uint8 per 64-element chunk: parameters without a gradient (set after the first backward)
            self._skip_known = False
            self.set_hyper()
        else:
            self.opt = torch.optim.AdamW(self.params, lr=lr, betas=self.betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=graph)
        self.pts = model.pack_points(points) if not isinstance(points, dict) else points
        self.gt_capacity = gt_capacity
        self.gts = self._pack_gts_static(gt_bboxes_3d, gt_labels_3d) if not isinstance(gt_bboxes_3d, dict) else gt_bboxes_3d
        self.labels = gt_labels_3d
        self.capacity_margin = capacity_margin
        self._graphs = None
        self.loss = None

    # ---- batches --------------------------------------------------------------------------------------------------------
    def _pack_gts_static(self, gt_bboxes_3d, gt_labels_3d):
        """GT buffers with a fixed per-scene capacity (`gt_capacity` boxes): the captured graphs size the cost matrix by the capacity
        and read the real per-scene counts from the device-side `gt_off`, so the next batch may hold any number of boxes up to it."""
        d = self.model.pts_bbox_head.pack_gts(gt_bboxes_3d, gt_labels_3d, self.dev)
        B = d["gt_off"].numel() - 1
        cap = max(int(d["gmax"]), int(self.gt_capacity))
        gt = torch.zeros((B * cap, d["gt"].shape[1]), dtype=d["gt"].dtype, device=self.dev)
        labels = torch.zeros((B * cap,), dtype=d["labels"].dtype, device=self.dev)
        n = d["gt"].shape[0]
        gt[:n].copy_(d["gt"])
        labels[:n].copy_(d["labels"])
        return dict(gt=gt, labels=labels, gt_off=d["gt_off"].clone(), gmax=cap)

    def set_batch(self, points, gt_bboxes_3d, gt_labels_3d):
        """Load the next batch INTO the static input buffers the captured graphs read (device-to-device copies on the current
        stream; ref: the runner's data loader feeding train_step, extra_tools/train.py:204-254).  Shapes the graphs were captured
        with must hold: same number of scenes, the same number of points per scene, at most `gmax` boxes per scene - anything else
        raises (build a new TrainStep / call capture() again for a different shape)."""
        pts = self.model.pack_points(points) if not isinstance(points, dict) else points
        if list(pts["lens"]) != list(self.pts["lens"]) or pts["cat"].shape != self.pts["cat"].shape:
            raise ValueError(f"set_batch: points per scene {list(pts['lens'])} differ from the captured layout {list(self.pts['lens'])}")
        d = self.model.pts_bbox_head.pack_gts(gt_bboxes_3d, gt_labels_3d, self.dev) if not isinstance(gt_bboxes_3d, dict) else gt_bboxes_3d
        if d["gt_off"].numel() != self.gts["gt_off"].numel():
            raise ValueError("set_batch: number of scenes differs from the captured batch")
        n = d["gt"].shape[0]
        if int(d["gmax"]) > int(self.gts["gmax"]) or n > self.gts["gt"].shape[0]:
            raise ValueError(f"set_batch: {int(d['gmax'])} boxes in one scene exceed the captured capacity {int(self.gts['gmax'])}")
        self.pts["cat"].copy_(pts["cat"])
        self.pts["scene_off"].copy_(pts["scene_off"])
        self.gts["gt"][:n].copy_(d["gt"])
        self.gts["labels"][:n].copy_(d["labels"])
        self.gts["gt_off"].copy_(d["gt_off"])

    # ---- optimizer hyper-parameters (device state: a captured step follows a schedule) -----------------------------------
    def set_hyper(self, lr=None, betas=None, weight_decay=None):
        """Learning rate / betas / weight decay of the NEXT steps (ref: the `step` and `cyclic` lr + momentum policies of the shipped
        configs).  Written into the device-side optimizer state by a one-thread launch, so it also steers replayed graphs."""
        if lr is not None:
            self.lr = float(lr)
        if betas is not None:
            self.betas = (float(betas[0]), float(betas[1]))
        if weight_decay is not None:
            self.weight_decay = float(weight_decay)
        if self.flat_update:
            nv.adamw_set_hyper(self.opt_state, self.lr, self.betas, self.eps, self.weight_decay, self.max_norm)
        elif self.opt is not None:
            if self._graphs is not None and (lr is not None or betas is not None or weight_decay is not None):
                # torch's capturable AdamW bakes lr / betas / decay into the captured launches as host scalars
                raise RuntimeError("graph=True with flat_update=False cannot follow a schedule: the captured torch.optim step ignores "
                                   "param_groups edits; use flat_update=True (device-side hyper-parameters)")
            for g in self.opt.param_groups:
                g["lr"], g["betas"], g["weight_decay"] = self.lr, self.betas, self.weight_decay

    # ---- the three stages (same code eager or captured) -------------------------------------------------------
    def _stage1(self):
        m = self.model
        _T.reset_param_uses()
        _sp.reset_conv_uses()
        with m.shadow_scope():
            feat, fps = m.extract_pts_feat(self.pts)
            self._stage1_head(feat, fps)

    # ---- stage 1 in pieces (captured step with fps_graph): voxelize | FPS (own graph, second stream) || features | head + matching ----
    def _stage1a(self):
        _T.reset_param_uses()
        _sp.reset_conv_uses()
        self._v = self.model.stage_voxelize(self.pts)

    def _stage1_fps(self):
        self._fps = self.model.stage_fps(self._v)

    def _stage1b(self):
        m = self.model
        with m.shadow_scope():
            self._feat = m.stage_features(self._v)

    def _stage1c(self):
        with self.model.shadow_scope(refresh=False):
            self._stage1_head(self._feat, self._fps)

    def _stage1_head(self, feat, fps):
        """Decoder / head forward, matching and targets (inside the caller's shadow scope)."""
        m = self.model
        amp = m.amp_dtype
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            self._outs = m.pts_bbox_head(feat, None, fps)
        self._T = m.pts_bbox_head.loss_targets(self.gts, None, self._outs)
        # the step's collective message: [L positive counts | capacity flag].  The flag (levels over capacity, this rank) is computed on
        # the device from the counts the encoder just produced: no host read
        npos = self._T["num_pos"]
        L = npos.numel()
        self._msg = torch.empty(L + 1, dtype=torch.float32, device=npos.device)
        self._msg[:L].copy_(npos)
        enc = getattr(m, "pts_middle_encoder", None)
        caps = getattr(enc, "level_capacities", None) if enc is not None else None
        cnts, cps = [], []
        if caps is not None and m.static_shapes:
            cnts, cps = list(enc.last_level_counts[1:1 + len(caps)]), list(caps)
            vfe = getattr(m, "pts_voxel_encoder", None)
            if getattr(vfe, "capacity", None) is not None and vfe.last_count_dev is not None:      # dynamic voxelization: the voxel list too
                cnts, cps = [vfe.last_count_dev] + cnts, [int(vfe.capacity)] + cps
        if self._fps_can_time_out():
            cnts, cps = cnts + [m.fps_err[:1]], cps + [0]        # the FPS time-out flag of this step's launch as a "count" with capacity 0
        if cnts:
            nv.capacity_flag(cnts, cps, self._msg[L:])
        else:
            self._msg[L:].zero_()
        self._num_pos = self._msg[:L]

    def _fps_can_time_out(self):
        """Only FPS over sets above native.FPS_REG_MAX points runs on several workgroups that wait for each other."""
        m = self.model
        if getattr(m, "fps_err", None) is None or getattr(m, "fps_max_wg", 0) == 1:
            return False
        vl = getattr(m, "pts_voxel_layer", None)
        mv = 0 if (vl is None or m.dynamic_voxelization) else int(vl.max_voxels[0 if m.training else 1])
        return max(max(self.pts["lens"]), mv) > nv.FPS_REG_MAX

    def _reduce_num_pos(self):
        """ONE small all-reduce per step: the mean positive counts (ref: reduce_mean in uni3detr_head.py:658-664, 680-681) and, in the
        same message, the job-wide capacity flag (> 0 on every rank iff any rank overflowed)."""
        if self.dist_on:
            self._msg.div_(self.world)
            dist.all_reduce(self._msg)

    def _stage2(self):
        # .grad = None: autograd hands each parameter its gradient tensor as is (no per-parameter "grad += new" launch, ~270 of
        # them); one multi-tensor copy then packs them into the flat buffer the all-reduce / clip / AdamW stages work on
        for p in self.params:
            p.grad = None
        losses = self.model.pts_bbox_head.loss_from_targets(self._outs, self._T, self._num_pos)
        self._losses = losses
        loss = getattr(self.model.pts_bbox_head, "_loss_total", None)          # = the sum below, reduced in one launch (fused loss path)
        if loss is None or not all("loss" in k for k in losses):
            loss = sum(v for k, v in losses.items() if "loss" in k)
        self.model.pts_bbox_head._loss_total = None
        with _T.deferred_param_grads():      # dW / db of the decoder + head linears: queued, then one batched launch per shape
            loss.backward()
        self.loss = loss.detach()
        dst, src, missing = [], [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if p.grad is None:
                missing.append(v)
                self._grad_missing[i] = True
            elif p.grad.data_ptr() != v.data_ptr():          # (already in place: written there by the producing kernel)
                dst.append(v); src.append(p.grad)
            p.grad = v
        torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        self._note_missing(0, len(self.params))

    def _note_missing(self, lo, hi):
        """Parameters that received no gradient keep weights AND moments untouched, as torch.optim.AdamW skips `.grad is None`
        (the reference trains with find_unused_parameters=True).  The set is fixed by the model's structure: recorded once, from the
        first eager backward, as a per-64-element skip mask for u3d_adamw_step_state."""
        if not self.flat_update or self._skip_known:
            return
        seen = getattr(self, "_seen_ranges", set())
        seen.add((lo, hi))
        self._seen_ranges = seen
        miss = getattr(self, "_missing_idx", set())
        for i in range(lo, hi):
            if self._grad_missing[i]:
                miss.add(i)
        self._missing_idx = miss
        if sum(h - l for l, h in seen) >= len(self.params):
            self._skip_known = True
            if miss:
                mask = torch.zeros((self.flat_param.numel() + 63) // 64, dtype=torch.uint8, device=self.dev)
                for i in miss:
                    o, n = self.offsets[i], self.params[i].numel()
                    mask[o // 64:(o + n + 63) // 64] = 1
                self._skip = mask

    def _pack(self, lo, hi):
        """Gradients of params[lo:hi] (as autograd left them in .grad) -> their views of the flat buffer; .grad = the view again."""
        dst, src, missing = [], [], []
        for i, (p, v) in enumerate(zip(self.params[lo:hi], self.views[lo:hi])):
            if p.grad is None:
                missing.append(v)
                self._grad_missing[lo + i] = True
            elif p.grad.data_ptr() != v.data_ptr():          # (already in place: written there by the producing kernel)
                dst.append(v); src.append(p.grad)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        self._note_missing(lo, hi)

    def _stage2a(self):
        """Losses + backward down to the sparse encoder's dense() output (phase A of the two-phase backward)."""
        for p in self.params:
            p.grad = None
        losses = self.model.pts_bbox_head.loss_from_targets(self._outs, self._T, self._num_pos)
        self._losses = losses
        loss = getattr(self.model.pts_bbox_head, "_loss_total", None)
        if loss is None or not all("loss" in k for k in losses):
            loss = sum(v for k, v in losses.items() if "loss" in k)
        self.model.pts_bbox_head._loss_total = None
        cut = self.model._encoder_cut                # detached leaf the dense stack / head were fed with (detector.extract_pts_feat)
        cut.grad = None
        bcut = self.model._backbone_cut if self.three_phase else None
        if bcut is not None:
            for t in bcut:
                t.grad = None
        with _T.deferred_param_grads():
            loss.backward()
        self.loss = loss.detach()
        self._gbb = None
        if bcut is not None:                         # the backward stopped at SECOND3D's outputs: phase M continues from their gradients
            self._pack(self.n_bb_end, len(self.params))
            self._gbb = [t.grad for t in bcut]
            for t in bcut:
                t.grad = None
            return
        self._pack(self.n_enc, len(self.params))
        self._gx = cut.grad
        cut.grad = None

    def _stage2m(self):
        """Phase M (three-phase backward): SECOND3D's backward from the gradients of its three outputs down to the encoder cut."""
        m = self.model
        if getattr(self, "_gbb", None) is None or m._backbone_out is None:
            return                       # the forward made no second cut (nothing behind SECOND3D required a gradient): phase A did it all
        cut = m._encoder_cut
        outs, gs = [], []
        for o, g in zip(m._backbone_out, self._gbb):
            if g is not None:
                outs.append(o); gs.append(g)
        torch.autograd.backward(outs, gs)
        self._pack(self.n_enc, self.n_bb_end)
        self._gx = cut.grad
        cut.grad = None
        self._gbb = None
        m._backbone_out = m._backbone_cut = None

    def _stage2b(self):
        """Phase B: the sparse encoder's backward from the gradient of its output."""
        x = self.model._encoder_out
        x.backward(self._gx)
        self._pack(0, self.n_enc)
        self._gx = None
        self.model._encoder_out = self.model._encoder_cut = None

    def _comm_view(self, lo, hi):
        """bf16 staging slice for flat_grad[lo:hi] (allocated once: static addresses, one cast launch each way)."""
        if self._comm is None or self._comm.numel() != self.flat_grad.numel():
            self._comm = torch.empty(self.flat_grad.numel(), dtype=torch.bfloat16, device=self.flat_grad.device)
        return self._comm[lo:hi]

    def _all_reduce_slice(self, lo, hi, async_op=False):
        """Mean over ranks of flat_grad[lo:hi], in place; -> (work or None, staging slice or None)."""
        a = self.flat_grad[lo:hi]
        a.div_(self.world)
        if self.grad_comm_dtype == torch.float32:
            return dist.all_reduce(a, async_op=async_op), None
        buf = self._comm_view(lo, hi)
        buf.copy_(a)
        w = dist.all_reduce(buf, async_op=async_op)
        if not async_op:
            a.copy_(buf)
            return w, None
        return w, buf

    def _launch_bucket(self, lo, hi):
        """Asynchronous mean over ranks of flat_grad[lo:hi]; collected by _reduce_grads_b()."""
        if self.dist_on and hi > lo:
            w, buf = self._all_reduce_slice(lo, hi, async_op=True)
            self._works.append((w, buf, lo, hi))

    def _reduce_grads_a(self):
        """First bucket on the wire: neck + head + decoder (three-phase) or everything behind the encoder (two-phase)."""
        self._launch_bucket(self.bb_end if getattr(self, "three_phase", False) else self.enc_end, self.flat_grad.numel())

    def _reduce_grads_m(self):
        """Second bucket (three-phase): SECOND3D's slice, in flight underneath the encoder's backward."""
        self._launch_bucket(self.enc_end, self.bb_end)

    def _reduce_grads_b(self):
        if self.dist_on:
            # comm_diag: event pairs around the waits of the compute stream - how much of each asynchronous bucket is NOT hidden under
            # the backward that follows it (the stream stalls in work.wait()) and what the encoder's bucket (never overlapped) costs;
            # read by comm_exposed_ms()
            diag = getattr(self, "comm_diag", False)
            ev = []
            if diag:
                ev.append(torch.cuda.Event(enable_timing=True)); ev[-1].record()
            for w, buf, lo, hi in self._works:
                w.wait()
                if buf is not None:
                    self.flat_grad[lo:hi].copy_(buf)
                if diag:
                    ev.append(torch.cuda.Event(enable_timing=True)); ev[-1].record()
            n_async = len(self._works)
            del self._works[:]
            self._all_reduce_slice(0, self.enc_end)
            if diag:
                ev.append(torch.cuda.Event(enable_timing=True)); ev[-1].record()
                self._comm_events.append((n_async, ev))
                del self._comm_events[:-256]

    def comm_exposed_ms(self):
        """Mean stall of the compute stream in the gradient waits over the steps recorded since comm_diag was switched on:
        reduce_a_exposed_ms (first asynchronous bucket: neck + head, or everything behind the encoder in the two-phase plan),
        reduce_m_exposed_ms (the backbone's bucket, three-phase only), reduce_b_exposed_ms (the encoder's bucket, never overlapped),
        bucket sizes, steps - synchronises with the device."""
        if not self._comm_events:
            return None
        torch.cuda.synchronize()
        three = bool(getattr(self, "three_phase", False))
        rows = [[e[i].elapsed_time(e[i + 1]) for i in range(len(e) - 1)] for _, e in self._comm_events]
        mean = lambda k: sum(r[k] for r in rows if len(r) > k) / max(1, sum(1 for r in rows if len(r) > k))      # noqa: E731
        bpe = 4 if self.grad_comm_dtype == torch.float32 else 2
        n = self.flat_grad.numel()
        out = dict(reduce_a_exposed_ms=mean(0), reduce_b_exposed_ms=mean(2 if three else 1), steps=len(rows), buckets=3 if three else 2,
                   bucket_a_MB=(n - (self.bb_end if three else self.enc_end)) * bpe / 1e6, bucket_b_MB=self.enc_end * bpe / 1e6)
        if three:
            out.update(reduce_m_exposed_ms=mean(1), bucket_m_MB=(self.bb_end - self.enc_end) * bpe / 1e6)
        return out

    def _reduce_grads(self):
        if self.dist_on:
            self._all_reduce_slice(0, self.flat_grad.numel())

    def _stage3(self):
        if self.flat_update:
            nv.adamw_step_state(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.opt_state, self._skip, self._opt_ws,
                                 hold=None if self._msg is None else self._msg[-1:])
            return
        torch.nn.utils.clip_grad_norm_(self.params, self.max_norm, foreach=True)
        self.opt.step()

    def _reset_opt_state(self):
        if self.flat_update:
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.opt_state[:5].zero_()            # step count and derived values; the hyper-parameter slots stay
            self.opt_state[11:13].zero_()         # hold flag / held-step counter
            return
        for st in self.opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()

    # ---- optimizer state export / import (ref: the runner's resume_from, extra_tools/train.py:141-142) -------------------------------
    def optimizer_state_dict(self):
        """Flat AdamW state as a dict of CPU tensors (per-parameter views are recoverable through `offsets`)."""
        if not self.flat_update:
            return dict(kind="torch", state=self.opt.state_dict())
        return dict(kind="flat", exp_avg=self.exp_avg.cpu(), exp_avg_sq=self.exp_avg_sq.cpu(), opt_state=self.opt_state.cpu(),
                    offsets=list(self.offsets), numels=[p.numel() for p in self.params],
                    skip=None if self._skip is None else self._skip.cpu())

    def load_optimizer_state_dict(self, sd):
        if sd.get("kind") == "torch":
            self.opt.load_state_dict(sd["state"])
            return
        if list(sd["offsets"]) != list(self.offsets) or list(sd["numels"]) != [p.numel() for p in self.params]:
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.opt_state.copy_(sd["opt_state"])
        self.opt_state[11:13].zero_()
        if sd.get("skip") is not None:
            self._skip = sd["skip"].to(self.dev)
            self._skip_known = True
        self.set_hyper()

    def eager_step(self, stage1_done=False):
        if not stage1_done:
            self._stage1()
        self._reduce_num_pos()
        if self.overlap:
            self._stage2a(); self._reduce_grads_a()
            if self.three_phase:
                self._stage2m(); self._reduce_grads_m()
            self._stage2b(); self._reduce_grads_b()
        else:
            self._stage2(); self._reduce_grads()
        self._stage3()
        return self.loss

    def enable_dist(self):
        """Call AFTER dist.init_process_group.  hipGraph capture and an RCCL process group do not mix on this stack (the
        process-group watchdog polls events while a capture is open -> hipErrorCapturedEvent), so bench.py captures first
        and creates the process group afterwards; the warm-up/capture iterations have touched weights and optimizer state
        with UNREDUCED gradients, so here every rank is reset to rank 0's model and a fresh optimizer state (in place: the
        captured graphs hold these addresses)."""
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.dist_on else 1
        if not self.dist_on:
            return
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, 0)
            self._reset_opt_state()
        torch.cuda.synchronize()

    def snapshot_full(self):
        """Model state + optimizer state (moments, step count); restore_full() puts all of it back in place."""
        snap = dict(model=self.snapshot())
        if self.flat_update:
            snap.update(m=self.exp_avg.clone(), v=self.exp_avg_sq.clone(), st=self.opt_state.clone())
        else:       # torch.optim.AdamW: moments and step counters of every parameter that has state already
            snap["opt"] = {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state.get(p, {}).items()}
                           for i, p in enumerate(self.params)}
        return snap

    def restore_full(self, snap):
        with torch.no_grad():
            for t, s_ in zip(list(self.model.parameters()) + list(self.model.buffers()), snap["model"]):
                t.copy_(s_)
            if self.flat_update:
                self.exp_avg.copy_(snap["m"]); self.exp_avg_sq.copy_(snap["v"]); self.opt_state.copy_(snap["st"])
            else:
                # in place (a captured optimizer step holds these addresses); state created after the snapshot goes back to zero
                for i, p in enumerate(self.params):
                    cur, old = self.opt.state.get(p, {}), snap["opt"].get(i, {})
                    for k, v in cur.items():
                        if torch.is_tensor(v):
                            if k in old:
                                v.copy_(old[k])
                            else:
                                v.zero_()

    def snapshot(self):
        return [t.detach().clone() for t in list(self.model.parameters()) + list(self.model.buffers())]

    def restore(self, snap):
        with torch.no_grad():
            for t, s_ in zip(list(self.model.parameters()) + list(self.model.buffers()), snap):
                t.copy_(s_)
            self._reset_opt_state()

    # ---- capture ------------------------------------------------------------------------------------------------
    def measure_capacities(self, batches=None):
        """Run exact-size eager steps (the bound batch, or every batch of `batches`) and size the strided sparse levels from the
        largest counts seen (x margin, multiple of 256)."""
        m = self.model
        m.static_shapes = False
        m.pts_middle_encoder.level_capacities = None
        vfe = getattr(m, "pts_voxel_encoder", None) if getattr(m, "dynamic_voxelization", False) else None
        if vfe is not None:
            vfe.capacity = None
        counts = None
        for b in (batches if batches else [None]):
            if b is not None:
                self.set_batch(*b)
            self.eager_step()
            c = [int(c.item()) for c in m.pts_middle_encoder.last_level_counts]
            counts = c if counts is None else [max(a, b_) for a, b_ in zip(counts, c)]
        caps = [((int(c * self.capacity_margin) + 255) // 256) * 256 for c in counts[1:]]
        m.pts_middle_encoder.level_capacities = caps
        if vfe is not None:
            # dynamic voxelization: the voxel list itself (level 0) is capacity-sized too, with the count kept on the device
            vfe.capacity = ((int(counts[0] * self.capacity_margin) + 255) // 256) * 256
        m.static_shapes = True
        return counts, caps

    def check_capacities(self):
        enc = self.model.pts_middle_encoder
        counts = [int(c.item()) for c in enc.last_level_counts]
        for c, cap in zip(counts[1:], enc.level_capacities):
            if c > cap:
                raise RuntimeError(f"sparse level overflow: {c} active rows > capacity {cap}; re-capture with a larger margin")
        vfe = getattr(self.model, "pts_voxel_encoder", None)
        if getattr(vfe, "capacity", None) is not None and vfe.last_count_dev is not None:
            c = int(vfe.last_count_dev.item())
            if c > int(vfe.capacity):
                raise RuntimeError(f"sparse level overflow: {c} voxels > capacity {vfe.capacity} of the dynamic voxel list; re-capture with a larger margin")
        # the several-workgroup FPS's time-out record: [0] != 0 = THIS step's launch gave up waiting for a sibling workgroup, i.e. the
        # samples behind the decoder's queries are invalid (index 0 for the unfinished rounds) - as fatal for the step as a truncated level
        e = getattr(self.model, "fps_err", None)
        if e is not None and getattr(self.model, "fps_max_wg", 0) != 1 and int(e[0].item()) != 0:
            raise FpsTimeout("several-workgroup FPS timed out waiting for a sibling workgroup: this step's query samples are invalid")
        return counts

    def capture(self, warmup=3, keep_state=True, batches=None, remember_batches=True):
        """Measure the sparse-level capacities (over `batches`, a list of (points, gts, labels), when given), warm up, capture.
        keep_state: weights, BatchNorm statistics and optimizer state are restored afterwards - the measuring / warm-up iterations
        are real optimizer steps and must not count as training."""
        assert self.graph
        if self.dist_on:
            raise RuntimeError("capture() with a live process group: hipGraph capture and RCCL do not mix on this stack (see "
                               "enable_dist); use recapture(), which tears the group down through pg_hooks first")
        if batches is not None and remember_batches:
            self._capture_batches = batches
        snap = self.snapshot_full() if keep_state else None
        counts, caps = self.measure_capacities(batches)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):                      # static-shape eager warm-up on the capture stream
                self.eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # drop the eager iteration's activations / autograd graph BEFORE capturing: releasing them from inside a capture
        # (when the attributes are re-assigned) tears down autograd nodes mid-capture and crashes hipStreamEndCapture
        self._outs = self._T = self._num_pos = self._msg = self._losses = self.loss = None
        self._gx = self._v = self._fps = self._feat = None
        self.model._encoder_out = self.model._encoder_cut = None
        self.model.pts_bbox_head._loss_total = None
        dec = getattr(getattr(self.model.pts_bbox_head, "transformer", None), "decoder", None)
        if dec is not None:
            dec._reg_outputs = None
            dec._states_c = None
            dec._cls_outputs = dec._iou_outputs = None
            dec._coord_outputs = dec._refs_sig = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        g2b = torch.cuda.CUDAGraph() if self.overlap else None
        g2m = torch.cuda.CUDAGraph() if (self.overlap and self.three_phase) else None
        pool = torch.cuda.graph_pool_handle()
        self._v = self._fps = self._feat = None
        if self.fps_graph and not getattr(self.model, "dynamic_voxelization", False):
            # stage 1 as FOUR graphs: voxelize | FPS || encoder + dense stack | head + matching.  Every node of one hipGraph runs on a
            # single hardware queue on this stack, so the forked FPS branch of a monolithic stage-1 graph sat alone on the device for
            # its ~1.1 ms (rocprofv3 timeline, profiles/r04a_timeline.txt); as its own graph, replayed on a second stream, it runs
            # underneath the encoder.  The FPS graph has its OWN memory pool: it is the one graph that runs concurrently with another,
            # and graphs sharing a pool may reuse each other's freed blocks.
            g1a, gf, g1b, g1c = (torch.cuda.CUDAGraph() for _ in range(4))
            if self._fps_stream is None:
                self._fps_stream = torch.cuda.Stream()
            with torch.cuda.graph(g1a, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage1a()
            with torch.cuda.graph(gf, stream=s, capture_error_mode="thread_local"):
                self._stage1_fps()
            with torch.cuda.graph(g1b, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage1b()
            with torch.cuda.graph(g1c, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage1c()
            g1 = (g1a, gf, g1b, g1c)
        else:
            with torch.cuda.graph(g1, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage1()
        self._reduce_num_pos()
        if self.overlap:
            with torch.cuda.graph(g2, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage2a()
            self._reduce_grads_a()
            if g2m is not None:
                with torch.cuda.graph(g2m, pool=pool, stream=s, capture_error_mode="thread_local"):
                    self._stage2m()
                self._reduce_grads_m()
            with torch.cuda.graph(g2b, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage2b()
            self._reduce_grads_b()
        else:
            with torch.cuda.graph(g2, pool=pool, stream=s, capture_error_mode="thread_local"):
                self._stage2()
            self._reduce_grads()
        with torch.cuda.graph(g3, pool=pool, stream=s, capture_error_mode="thread_local"):
            self._stage3()
        torch.cuda.synchronize()
        self._graphs = (g1, g2, g2b, g3)
        self._g2m = g2m
        if isinstance(g1, tuple) and getattr(self, "fps_stream_calibration_ms", None) is None:
            # once per TrainStep: a re-capture keeps the stream the first capture chose (every rank then keeps ITS choice for the
            # whole job, and a re-capture costs no calibration replays)
            bufs = [(b, b.clone()) for b in self.model.buffers()]
            self._pick_fps_stream(g1)
            with torch.no_grad():
                for b, v in bufs:                    # the ~40 calibration replays ran BatchNorm in training mode on one batch
                    b.copy_(v)
        if snap is not None:
            self.restore_full(snap)
        self._steps_since_check = 0
        return counts, caps

    def held_steps(self):
        """Steps whose update was held because a sparse level overflowed - or the several-workgroup FPS timed out - somewhere in the job
        (one small device-to-host read; identical on every rank: the flag is all-reduced before the update)."""
        if not self.flat_update:
            return 0
        return int(self.opt_state[12].item())

    def fps_timeouts(self):
        """FPS launches of THIS rank that timed out since the last check (host read of model.fps_err[1])."""
        e = getattr(self.model, "fps_err", None)
        return int(e[1].item()) if e is not None else 0

    def recapture(self):
        """Collective re-capture with more room (every rank calls it at the same step: the decision comes from the all-reduced
        counter).  With a live process group the group is torn down first and re-created afterwards through `pg_hooks`: capture and
        RCCL do not mix (enable_dist).  Weights, BatchNorm statistics and optimizer state are put back (capture keep_state=True), so
        replicas stay identical without a broadcast."""
        was_dist = self.dist_on
        if was_dist:
            if self.pg_hooks is None:
                raise RuntimeError("sparse level overflow on a multi-rank run and no pg_hooks=(teardown, setup) to re-capture with: "
                                   "capture with a larger capacity_margin or more representative `batches`")
            for w, _buf, _lo, _hi in self._works:
                w.wait()
            del self._works[:]
            torch.cuda.synchronize()
            dist.barrier()
            self.pg_hooks[0]()
            self.dist_on = False
        held = self.held_steps()
        # WHY steps were held is a per-rank fact, what to change must be a job-wide decision: step() exchanged the maximum over ranks
        # before the group was torn down (_job_fps_timeouts); a direct caller on one rank falls back to its own count
        n_to = max(self.fps_timeouts(), getattr(self, "_job_fps_timeouts", 0))
        self._job_fps_timeouts = 0
        self.recaptures += 1
        import sys
        if n_to > 0:
            # the several-workgroup FPS needs all its workgroups resident together and waited in vain: from here on every rank samples
            # with the single-workgroup streaming kernel (slower rounds, no cross-workgroup wait) - same indices
            self.fps_timeouts_seen += n_to
            self.model.fps_max_wg = 1
            self.model.fps_err.zero_()
            print(f"[TrainStep] {held} step(s) held, {n_to} FPS launch(es) timed out waiting for a sibling workgroup; re-capturing with the "
                  f"single-workgroup FPS (re-capture #{self.recaptures})", file=sys.stderr, flush=True)
        else:
            self.capacity_margin *= 1.5
            print(f"[TrainStep] {held} step(s) held by a sparse-level capacity overflow; re-capturing with margin "
                  f"{self.capacity_margin:.2f} (re-capture #{self.recaptures})", file=sys.stderr, flush=True)
        # the batch the caller has bound (the one that overflowed, or its successor) is (a) measured together with the capture batches -
        # capacities must cover what actually ran - and (b) put back into the static input buffers afterwards: measure_capacities()
        # cycles every capture batch through them, and the replay that follows must train on the caller's batch, not on the last of those
        cur_pts = dict(self.pts, cat=self.pts["cat"].clone(), scene_off=self.pts["scene_off"].clone())
        cur_gts = dict(self.gts, gt=self.gts["gt"].clone(), labels=self.gts["labels"].clone(), gt_off=self.gts["gt_off"].clone())
        bound = (cur_pts, cur_gts, None)
        self.capture(batches=(list(self._capture_batches) + [bound]) if self._capture_batches else [bound], remember_batches=False)
        self.set_batch(*bound)
        if self.flat_update:
            self.opt_state[11:13].zero_()
        if was_dist:
            self.pg_hooks[1]()
            self.dist_on = dist.is_available() and dist.is_initialized()
            self.world = dist.get_world_size() if self.dist_on else 1

    def _pick_fps_stream(self, g1, candidates=8, reps=3):
        """Two streams overlap only if their hardware queues sit on different pipes of the command processor, and which pipe a stream
        lands on is the runtime's business (measured on this stack: with an unlucky pair the FPS graph and the encoder graph run one
        after the other although nothing orders them - profiles/r04e_timeline.txt -, with a lucky pair the FPS rounds disappear under
        the encoder).  So the side stream is CHOSEN by measurement: stage 1 is replayed with each candidate stream and the fastest one
        is kept.  Runs inside capture(), before the state snapshot is put back (the replays touch BatchNorm running statistics)."""
        import time
        cur = torch.cuda.current_stream()

        def wall(side):
            self._fps_stream = side
            self._replay_stage1(g1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self._replay_stage1(g1)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        best, best_t, seen = self._fps_stream, None, []
        for i in range(candidates):
            side = self._fps_stream if i == 0 else torch.cuda.Stream(priority=-1 if i % 2 else 0)
            t = wall(side)
            seen.append(round(t * 1e3, 3))
            if best_t is None or t < best_t:
                best, best_t = side, t
        self._fps_stream = best
        self.fps_stream_calibration_ms = seen          # stage-1 wall time per candidate (bench.py reports it)
        # two reference points for the numbers above: stage 1 with the FPS graph on the MAIN stream (no overlap possible) and without
        # the FPS graph at all (what full overlap would give; g1c then reads the previous replay's samples - timing only)
        g1a, gf, g1b, g1c = g1

        def wall2(with_fps):
            def once():
                g1a.replay()
                if with_fps:
                    gf.replay()
                g1b.replay(); g1c.replay()
            once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                once()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / reps * 1e3, 3)
        self.fps_overlap_reference_ms = {"serial": wall2(True), "without_fps": wall2(False), "chosen": round(best_t * 1e3, 3)}

    def _replay_stage1(self, g1):
        if not isinstance(g1, tuple):
            g1.replay()
            return
        g1a, gf, g1b, g1c = g1
        cur, side = torch.cuda.current_stream(), self._fps_stream
        g1a.replay()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gf.replay()                       # the FPS rounds: second stream, underneath ...
        g1b.replay()                          # ... the sparse encoder and the dense stack
        cur.wait_stream(side)
        g1c.replay()

    def _eager_check(self):
        """Eager mode has no graph to re-capture, but the device-side HOLD still skips updates (level overflow cannot happen without
        captured capacities; a persistent FPS time-out can): every `check_every` steps read the counters and, on a time-out, fall back
        to the single-workgroup FPS - otherwise every step would be a silent no-op that still returns a finite loss."""
        self._steps_since_check = 0
        held, n_to = self.held_steps(), self.fps_timeouts()
        if n_to > 0:
            import sys
            self.fps_timeouts_seen += n_to
            self.model.fps_max_wg = 1
            self.model.fps_err.zero_()
            if self.flat_update:
                self.opt_state[11:13].zero_()
            print(f"[TrainStep] eager mode: {held} step(s) held, {n_to} FPS launch(es) timed out waiting for a sibling workgroup; "
                  "continuing with the single-workgroup FPS", file=sys.stderr, flush=True)

    def step(self):
        if self._graphs is None:
            if self.check_every and self._steps_since_check >= self.check_every:
                self._eager_check()
            self._steps_since_check += 1
            if not self.flat_update and getattr(self.model, "fps_err", None) is not None and getattr(self.model, "fps_max_wg", 0) != 1:
                # torch.optim path: no device-side hold -> the time-out is caught on the host (one small read per step) between the
                # forward and the backward; the forward is repeated with the single-workgroup FPS, nothing trains on invalid samples
                self._stage1()
                if int(self.model.fps_err[0].item()) != 0:
                    import sys
                    self.fps_timeouts_seen += 1
                    self.model.fps_max_wg = 1
                    self.model.fps_err.zero_()
                    print("[TrainStep] eager mode: the several-workgroup FPS timed out; repeating the forward with the single-workgroup "
                          "FPS", file=sys.stderr, flush=True)
                    self._stage1()
                return self.eager_step(stage1_done=True)
            return self.eager_step()
        if self.check_every and self._steps_since_check >= self.check_every:
            # held steps = a level outgrew its capacity on some rank (the device already refused to train on it, on every rank):
            # one small device-to-host read every `check_every` steps, then a collective re-capture with room to spare
            self._steps_since_check = 0
            if self.held_steps() > 0:
                # the reason (level overflow vs FPS time-out) is per rank; the decision what to change must not be: job-wide maximum
                self._job_fps_timeouts = self.fps_timeouts()
                if self.dist_on:
                    t = torch.tensor([float(self._job_fps_timeouts)], device=self.dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    self._job_fps_timeouts = int(t.item())
                self.recapture()
        self._steps_since_check += 1
        g1, g2, g2b, g3 = self._graphs
        self._replay_stage1(g1)
        if not self.flat_update:
            # torch.optim has no device-side hold flag (u3d_adamw_step_hold is the flat path's): without it a level that outgrew its
            # captured capacity would be truncated and trained on silently, so this configuration pays one host read per step and
            # re-captures BEFORE the backward / update of the overflowing batch
            try:
                self.check_capacities()        # (also reads the FPS time-out flag: this path has no device-side hold to catch it)
            except RuntimeError as e:
                if self.dist_on:
                    raise                      # a per-rank decision cannot drive a collective re-capture: flat_update=True does that
                if isinstance(e, FpsTimeout):
                    self._job_fps_timeouts = max(1, self.fps_timeouts())       # recapture() switches to the single-workgroup FPS
                self.recapture()
                g1, g2, g2b, g3 = self._graphs
                self._replay_stage1(g1)
        self._reduce_num_pos()
        g2.replay()
        if g2b is not None:
            self._reduce_grads_a()
            if getattr(self, "_g2m", None) is not None:
                self._g2m.replay()
                self._reduce_grads_m()
            g2b.replay()
            self._reduce_grads_b()
        else:
            self._reduce_grads()
        g3.replay()
        return self.loss
