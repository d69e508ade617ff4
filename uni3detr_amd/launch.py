"""One process per GPU, started by the program itself (ref: extra_tools/dist_train.sh:7-9, which wraps train.py in
`python -m torch.distributed.launch --nproc_per_node=$GPUS`).  `bench.py --gpus N` calls `spawn_ranks` when it was NOT started by
torch.distributed.run (no WORLD_SIZE in the environment): N children of the same command line, each with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (env:// rendezvous on 127.0.0.1), rank 0 owning stdout.  Fails loudly - non-zero exit,
nothing launched - when the node has fewer than N GPUs."""
import os
import socket
import subprocess
import sys
import time


class LaunchError(RuntimeError):
    pass


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_envs(n, base_env=None, port=None):
    """-> the N environments of the ranks (one node): what torch.distributed.run would have exported."""
    port = free_port() if port is None else int(port)
    envs = []
    for r in range(n):
        e = dict(os.environ if base_env is None else base_env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), U3D_SELF_LAUNCHED="1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
        e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        envs.append(e)
    return envs


def spawn_ranks(n, argv, device_count, popen=subprocess.Popen, poll_s=0.2, timeout_s=None):
    """Start `argv` N times (rank r on GPU r), wait for all, return the worst exit code.  A rank that fails takes the others down
    (a hung collective must not outlive its peer).  `device_count`: GPUs visible on this node (torch.cuda.device_count())."""
    if n < 1:
        raise LaunchError(f"--gpus {n}: need at least one rank")
    if device_count < n:
        raise LaunchError(f"--gpus {n} asked for, but this node exposes {device_count} GPU(s): refusing to run a smaller job under "
                          f"a larger job's name")
    procs = []
    for r, env in enumerate(rank_envs(n)):
        # rank 0 inherits stdout (the ONE JSON line); the other ranks' stdout goes to stderr so that nothing else lands on it
        procs.append(popen(list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    t0 = time.time()
    codes = [None] * n
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            bad = [c for c in codes if c not in (None, 0)]
            if bad or (timeout_s is not None and time.time() - t0 > timeout_s):
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            codes[i] = p.wait(timeout=10)
                        except Exception:
                            p.kill()
                            codes[i] = p.wait()
                break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()
        raise
    worst = max((abs(c) for c in codes if c is not None), default=1)
    return worst, codes
