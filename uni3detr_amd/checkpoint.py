"""Checkpoint I/O with the reference's conventions (ref: extra_tools/test.py:197 `load_checkpoint(model, args.checkpoint,
map_location='cpu')`, extra_tools/train.py:141-142 `resume_from` / `load_from`; upstream mmcv.runner.load_checkpoint):

* a checkpoint is a dict with a 'state_dict' entry (or a bare state_dict); keys may carry a 'module.' prefix (DDP wrapper);
* parameter names / shapes are the reference's (tests/test_plugin_cpu.py pins them);
* sparse-conv weights: this package keeps the mmcv spconv-1.x layout [kD,kH,kW,Cin,Cout] (what SparseEncoderHD falls back to when
  spconv 2.x is absent, sparse_encoder_hd.py:9-12); checkpoints written with spconv 2.x carry [Cout,kD,kH,kW,Cin] - converted here
  by shape, in both directions (`to_spconv2=True` when saving for a spconv-2.x consumer).
"""
import collections

import torch


def _is_sparse_weight(name, t, model_shape):
    return t.dim() == 5 and "pts_middle_encoder" in name and name.endswith("weight")


def convert_spconv_layout(name, t, want_shape):
    """incoming 5-D sparse conv weight -> the layout with `want_shape`; returns None when neither layout fits."""
    if tuple(t.shape) == tuple(want_shape):
        return t
    if t.dim() != 5:
        return None
    a = t.permute(1, 2, 3, 4, 0)          # spconv 2.x [Cout,kD,kH,kW,Cin] -> 1.x [kD,kH,kW,Cin,Cout]
    if tuple(a.shape) == tuple(want_shape):
        return a.contiguous()
    b = t.permute(4, 0, 1, 2, 3)          # 1.x -> 2.x
    if tuple(b.shape) == tuple(want_shape):
        return b.contiguous()
    return None


def _load_file(path, map_location, trusted=False):
    """Tensors-only unpickling (a checkpoint from an untrusted source must not run code).  Checkpoints whose `meta` holds arbitrary
    python objects (older mmcv runners pickled more than strings) need the full unpickler, which EXECUTES what the file says: that is
    an explicit opt-in - `trusted=True` or U3D_TRUST_CHECKPOINTS=1 - never a silent fall-back (a malicious pickle fails the restricted
    unpickler by construction, so a fall-back would always hand it the full one).  I/O errors propagate as they are."""
    import os
    import pickle
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if not (trusted or os.environ.get("U3D_TRUST_CHECKPOINTS") == "1"):
            raise RuntimeError(f"{path}: holds more than tensors and plain containers ({e}); loading it runs the code it contains. If "
                               f"you trust the file pass trusted=True to load_checkpoint (or set U3D_TRUST_CHECKPOINTS=1)") from e
        return torch.load(path, map_location=map_location, weights_only=False)


def load_checkpoint(model, checkpoint, map_location="cpu", strict=False, trusted=False):
    """checkpoint: path | dict.  Returns the checkpoint dict (like mmcv).  strict=False is mmcv.runner.load_checkpoint's default (what
    extra_tools/test.py:197 gets): missing / unexpected keys are REPORTED (meta['missing_keys'], meta['unexpected_keys'] and a
    warning), not raised; shape mismatches always raise.  trusted: allow the full (code-executing) unpickler for files the
    tensors-only one rejects (see _load_file)."""
    ck = _load_file(checkpoint, map_location, trusted) if isinstance(checkpoint, (str, bytes)) else checkpoint
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    own = model.state_dict()
    new, converted, bad_shape = collections.OrderedDict(), [], []
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        if k in own and tuple(v.shape) != tuple(own[k].shape):
            c = convert_spconv_layout(k, v, own[k].shape) if _is_sparse_weight(k, v, own[k].shape) else None
            if c is None:
                bad_shape.append((k, tuple(v.shape), tuple(own[k].shape)))
                continue
            v = c
            converted.append(k)
        new[k] = v
    if bad_shape:
        raise RuntimeError(f"load_checkpoint: shape mismatch for {bad_shape[:5]}{' ...' if len(bad_shape) > 5 else ''}")
    missing = [k for k in own if k not in new]
    unexpected = [k for k in new if k not in own]
    if strict and (missing or unexpected):
        raise RuntimeError(f"load_checkpoint: missing keys {missing[:5]} unexpected keys {unexpected[:5]}")
    if missing or unexpected:
        import warnings
        warnings.warn(f"load_checkpoint: {len(missing)} missing key(s) {missing[:5]}, {len(unexpected)} unexpected key(s) {unexpected[:5]}")
    model.load_state_dict(new, strict=False)
    if isinstance(ck, dict):
        meta = ck.setdefault("meta", {})
        meta["converted_sparse_weights"] = converted
        meta["missing_keys"], meta["unexpected_keys"] = missing, unexpected
    return ck


def save_checkpoint(model, path, meta=None, optimizer_state=None, to_spconv2=False):
    """{'meta', 'state_dict' [, 'optimizer']} as the reference's runner writes it; to_spconv2 permutes the sparse conv weights to the
    spconv 2.x layout.  optimizer_state: TrainStep.optimizer_state_dict() (flat AdamW moments + step count) for `resume_from`;
    TrainStep.load_optimizer_state_dict(ck['optimizer']) puts it back."""
    sd = collections.OrderedDict()
    for k, v in model.state_dict().items():
        v = v.detach().cpu()
        if to_spconv2 and _is_sparse_weight(k, v, None):
            v = v.permute(4, 0, 1, 2, 3).contiguous()
        sd[k] = v
    ck = {"meta": dict(meta or {}), "state_dict": sd}
    if optimizer_state is not None:
        ck["optimizer"] = optimizer_state
    torch.save(ck, path)
    return ck
