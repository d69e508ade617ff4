#!/bin/bash
# One rocprofv3 counter pass over any command; keeps a per-kernel summary.  usage: tools/pmc_any.sh <tag> "<COUNTER ...>" "<kernel regex>" <command...>
# -> gpurun_out/<tag>_pmc.csv   (counters in their own pass, with --kernel-trace only: see the gpurun note on --pmc)
set -u
TAG=$1; CTRS=$2; RE=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=/tmp/pmc_$TAG; rm -rf "$D"
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$D" -o r -- "$@" > gpurun_out/${TAG}.log 2>&1
f=$(find "$D" -name '*counter_collection.csv' | head -1)
if [ -z "$f" ]; then echo "no counter csv" >> gpurun_out/${TAG}.log; find "$D" | head >> gpurun_out/${TAG}.log; exit 0; fi
python - "$f" "$RE" > gpurun_out/${TAG}_pmc.csv <<'PY'
import csv, re, sys, collections
f, rx = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0]
    if not rx.search(name):
        continue
    key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
ctrs = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "dispatches"] + [c + "_per_dispatch" for c in ctrs])
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = max(1, len(cnt[key]))
    w.writerow([key[0], key[1], n] + [f"{v.get(c, 0.0) / n:.6g}" for c in ctrs])
PY
cat gpurun_out/${TAG}_pmc.csv | cut -c1-300
