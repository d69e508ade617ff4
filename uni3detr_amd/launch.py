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


# ---- NUMA placement of a rank: the host threads of rank r (launch loop, RCCL proxy threads, the data path's staging copies) next to GPU r --
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0,1,2,3,8,10,11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_for_pci(bdf, sysfs="/sys"):
    """CPUs of the NUMA node the PCI device `bdf` ('0000:c1:00.0') hangs off, or None when the platform does not say (single-node hosts
    report numa_node = -1)."""
    try:
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read().strip())
        if node < 0:
            return None
        cpus = parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        return (node, cpus) if cpus else None
    except (OSError, ValueError):
        return None


def bind_to_gpu_numa(device_index, sysfs="/sys", props=None, setaffinity=None, task_dir="/proc/self/task"):
    """Pin THIS process to the CPUs of the NUMA node of GPU `device_index` (first-touch then places its host allocations there too).
    Intersects with the CPUs the process may already run on; leaves everything alone when the topology is unknown or the intersection
    is empty.  -> dict(node=, cpus=) describing what was done, or None."""
    if props is None:
        import torch
        props = torch.cuda.get_device_properties(device_index)
    try:
        bdf = f"{int(getattr(props, 'pci_domain_id', 0)):04x}:{int(props.pci_bus_id):02x}:{int(props.pci_device_id):02x}.0"
    except (AttributeError, TypeError, ValueError):
        return None
    found = numa_cpus_for_pci(bdf, sysfs)
    if found is None:
        return None
    node, cpus = found
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(cpus)
    use = sorted(allowed & set(cpus))
    if not use:
        return None
    # sched_setaffinity(0, ...) pins only the CALLING thread: by the time a rank knows its GPU the HIP runtime and torch's intra-op
    # pool have already started threads with the old mask -> every thread of the process (/proc/self/task) gets the new one.
    # (Pages those threads touched earlier stay where they are; bench.py therefore binds before it allocates its host buffers.)
    setaff = setaffinity or os.sched_setaffinity
    try:
        tids = sorted(int(t) for t in os.listdir(task_dir))
    except OSError:
        tids = [0]
    done = 0
    for tid in (tids or [0]):
        try:
            setaff(tid, use)
            done += 1
        except OSError:                                     # a thread that exited meanwhile, or a container that does not allow it:
            continue                                        # placement is an optimisation, never an error
    if done == 0:
        return None
    return dict(node=node, cpus=len(use), pci=bdf, threads=done)
