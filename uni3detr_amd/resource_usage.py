"""Print VGPR / spill / LDS / occupancy per kernel of one csrc file: python -m uni3detr_amd.resource_usage query.hip"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(name):
    src = os.path.join(HERE, "csrc", name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(HERE, "..", "include"),
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            if cur:
                print(cur)
            cur = {"fn": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        else:
            cur[k.split(" [")[0]] = v
    if cur:
        print(cur)


if __name__ == "__main__":
    main(sys.argv[1])
