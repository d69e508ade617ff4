"""ORACLE (test infrastructure).  Name-seeded parameters: every tensor is a pure function of (key, shape, seed), so
the reference modules (under oracle/refshim.py), the oracle restatement and the HIP product can all be loaded
with bit-identical weights without committing a multi-MB state_dict.  Uses only numpy's PCG64 `random()`
(uniform doubles), whose stream is stable across numpy versions."""
import zlib

import numpy as np
import torch


def _uniform(key, shape, seed):
    rng = np.random.default_rng([zlib.crc32(key.encode()), seed])
    return rng.random(int(np.prod(shape)) if len(shape) else 1).reshape(shape if len(shape) else ())


def seeded_tensor(key, shape, seed=0):
    shape = tuple(int(s) for s in shape)
    u = _uniform(key, shape, seed) * 2.0 - 1.0                       # U(-1,1)
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("running_mean",):
        v = 0.05 * u
    elif leaf in ("running_var",):
        v = 1.0 + 0.1 * u
    elif leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    elif leaf == "code_weights":
        return torch.ones(shape, dtype=torch.float32)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:])) if "tgt_embed" not in key and "refpoint_embed" not in key else 1
        if "middle_encoder" in key and len(shape) == 5:              # sparse conv weight [kD,kH,kW,Cin,Cout]
            fan_in = int(np.prod(shape[:4]))
        v = u * np.sqrt(3.0 / max(fan_in, 1))                        # unit-gain uniform
        if "refpoint_embed" in key:
            v = u * 2.0                                              # logits spread over the room
    elif leaf == "weight":                                           # norm scales
        v = 1.0 + 0.1 * u
    else:                                                            # biases
        v = 0.1 * u
    return torch.from_numpy(np.asarray(v, np.float32))


def seeded_state_dict(named_shapes, seed=0):
    """named_shapes: iterable of (key, shape) e.g. from module.state_dict().items()."""
    return {k: seeded_tensor(k, tuple(s.shape) if hasattr(s, "shape") else tuple(s), seed) for k, s in named_shapes}


def seeded_input(key, shape, seed=0, lo=-1.0, hi=1.0):
    u = _uniform("input:" + key, tuple(shape), seed)
    return torch.from_numpy(np.asarray(lo + (hi - lo) * u, np.float32))
