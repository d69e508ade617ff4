"""CPU: `bench.py --gpus N` starts its own ranks (uni3detr_amd/launch.py; ref: extra_tools/dist_train.sh:7-9).
Argument -> rank fan-out, the environment each rank gets, a real 2-rank gloo rendezvous through it, failure propagation, and the
loud refusal when the node has fewer GPUs than asked for (this container has none: the CLI itself must exit non-zero)."""
import os
import subprocess
import sys

import pytest

from uni3detr_amd.launch import LaunchError, rank_envs, spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_envs_are_what_a_launcher_exports():
    envs = rank_envs(4, base_env={"PATH": os.environ.get("PATH", "")}, port=29999)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_refuses_more_ranks_than_gpus():
    started = []
    with pytest.raises(LaunchError, match="exposes 1 GPU"):
        spawn_ranks(2, ["true"], device_count=1, popen=lambda *a, **k: started.append(a))
    assert not started                                  # nothing was launched


WORKER = r"""
import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
open(os.path.join(sys.argv[1], f"rank{dist.get_rank()}.txt"), "w").write(f"{dist.get_world_size()} {os.environ['LOCAL_RANK']} {t.item()}")
dist.destroy_process_group()
"""


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    worst, codes = spawn_ranks(2, [sys.executable, "-c", WORKER, str(tmp_path)], device_count=2, timeout_s=240)
    assert worst == 0 and codes == [0, 0]
    for r in range(2):
        world, local, total = open(tmp_path / f"rank{r}.txt").read().split()
        assert (world, local, float(total)) == ("2", str(r), 3.0)


def test_failing_rank_takes_the_job_down():
    code = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(120)\n"
    worst, codes = spawn_ranks(2, [sys.executable, "-c", code], device_count=2, timeout_s=60)
    assert worst != 0 and codes[1] == 7 and codes[0] not in (None, 0)


def test_bench_cli_exits_nonzero_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "GPU(s)" in p.stderr and p.stdout.strip() == ""


def test_numa_binding_reads_the_gpus_node_from_sysfs(tmp_path):
    """launch.bind_to_gpu_numa: PCI address of the rank's GPU -> numa_node -> that node's cpulist -> sched_setaffinity.
    Driven against a fake sysfs tree; unknown topology (numa_node = -1, missing files) leaves the process alone."""
    from types import SimpleNamespace
    from uni3detr_amd import launch as L
    assert L.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    allowed = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(f"{allowed[0]},{allowed[-1]},100000\n")
    seen = []
    props = SimpleNamespace(pci_domain_id=0, pci_bus_id=0xC1, pci_device_id=0)
    # every thread of the process is pinned (sched_setaffinity(0, ...) moves only the caller; the HIP runtime's and torch's
    # worker threads exist by the time a rank knows its GPU): a fake /proc/self/task with three thread ids
    tasks = tmp_path / "task"
    for tid in (4242, 4243, 4250):
        (tasks / str(tid)).mkdir(parents=True)
    r = L.bind_to_gpu_numa(0, sysfs=str(tmp_path), props=props, setaffinity=lambda pid, cpus: seen.append((pid, list(cpus))),
                           task_dir=str(tasks))
    want = sorted({allowed[0], allowed[-1]})
    assert r == dict(node=1, cpus=len(want), pci="0000:c1:00.0", threads=3)
    assert seen == [(4242, want), (4243, want), (4250, want)]

    def flaky(pid, cpus):                 # a thread that exited between the listing and the call is skipped, not an error
        if pid == 4243:
            raise ProcessLookupError(pid)
        seen.append((pid, list(cpus)))
    del seen[:]
    assert L.bind_to_gpu_numa(0, sysfs=str(tmp_path), props=props, setaffinity=flaky, task_dir=str(tasks))["threads"] == 2
    # no readable task list: the calling thread alone
    del seen[:]
    assert L.bind_to_gpu_numa(0, sysfs=str(tmp_path), props=props, setaffinity=lambda pid, cpus: seen.append(pid),
                              task_dir=str(tmp_path / "no_such_dir"))["threads"] == 1 and seen == [0]
    n = len(seen)
    (dev / "numa_node").write_text("-1\n")
    assert L.bind_to_gpu_numa(0, sysfs=str(tmp_path), props=props, setaffinity=lambda *a: seen.append(a)) is None
    assert L.bind_to_gpu_numa(0, sysfs=str(tmp_path / "nope"), props=props, setaffinity=lambda *a: seen.append(a)) is None
    assert len(seen) == n
