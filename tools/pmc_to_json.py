"""gpurun_out/<tag>_{fetch,write,tcc,sq}_pmc.csv (tools/pmc_step.sh) -> profiles/<tag>_*.csv copies + profiles/pmc_traffic.json:
per (kernel symbol, grid) HBM bytes read / written, L2 hit rate, MFMA utilisation of the build the counters were taken on.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: wide streaming reads are tallied at half on gfx950); WRITE_SIZE as reported (KiB).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32) - the normalisation used since round 2.
source_sha16 records the kernel sources the counters were taken on: bench.py refuses an entry whose source file has changed since."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    return {(r["kernel"].replace("void ", ""), int(r["grid"])): r for r in csv.DictReader(open(path))} if os.path.exists(path) else {}


def main(tag):
    g = os.path.join(ROOT, "gpurun_out")
    files = {k: os.path.join(g, f"{tag}_{k}_pmc.csv") for k in ("fetch", "write", "tcc", "sq")}
    fetch, write, tcc, sq = (load(files[k]) for k in ("fetch", "write", "tcc", "sq"))
    out = {}
    for key in sorted(set(fetch) | set(write) | set(sq)):
        e = {}
        if key in fetch:
            e["hbm_read_bytes"] = 2.0 * 1024.0 * float(fetch[key]["FETCH_SIZE_per_dispatch"])
            e["dispatches"] = int(fetch[key]["dispatches"])
        if key in write:
            e["hbm_write_bytes"] = 1024.0 * float(write[key]["WRITE_SIZE_per_dispatch"])
        if key in tcc:
            h, m = float(tcc[key]["TCC_HIT_sum_per_dispatch"]), float(tcc[key]["TCC_MISS_sum_per_dispatch"])
            e["l2_hit_rate"] = h / (h + m) if h + m else None
        if key in sq:
            b, mf = float(sq[key]["SQ_BUSY_CYCLES_per_dispatch"]), float(sq[key]["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"])
            e["mfma_util"] = mf / (b * 32.0) if b else None
        out[f"{key[0]}@{key[1]}"] = e
    for k, p in files.items():
        if os.path.exists(p):
            shutil.copy(p, os.path.join(ROOT, "profiles", os.path.basename(p)))
    ts = os.path.join(g, f"{tag}_trace_kernel_stats.csv")
    if os.path.exists(ts):
        shutil.copy(ts, os.path.join(ROOT, "profiles", f"{tag}_eager_kernel_stats.csv"))
    import hashlib
    src_sha = {f: hashlib.sha256(open(os.path.join(ROOT, "uni3detr_amd", "csrc", f), "rb").read()).hexdigest()[:16]
               for f in sorted(os.listdir(os.path.join(ROOT, "uni3detr_amd", "csrc"))) if f.endswith((".hip", ".h", ".inc"))}
    doc = {"tag": tag, "source_sha16": src_sha, "source": f"profiles/{tag}_{{fetch,write,tcc,sq}}_pmc.csv (tools/pmc_step.sh: separate rocprofv3 --pmc passes over "
                                 "`bench.py --steps 2 --warmup 1 --no-graph`, per (kernel, grid) averages)",
           "note": "hbm_read_bytes = 2 x FETCH_SIZE (gfx950 tallies wide streaming reads at half: MI355X_MICROARCH.md); key = '<symbol>@<grid threads>'",
           "kernels": out}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(f"{len(out)} (kernel, grid) entries -> profiles/pmc_traffic.json")


if __name__ == "__main__":
    main(sys.argv[1])
