#!/bin/bash
# usage: tools/prof_trace_summary.sh <tag> <kernel-name-substring> <command...>
# rocprofv3 kernel trace of the command; keeps a per-(kernel, grid size) duration summary of the matching kernel (the raw trace is too
# large for gpurun's return channel) -> gpurun_out/<tag>_<kernel>_by_grid.csv
set -u
TAG=$1; KERN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=/tmp/proft_$TAG; rm -rf "$D"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o r -- "$@" > gpurun_out/${TAG}_trace.log 2>&1
f=$(find "$D" -name '*kernel_trace.csv' | head -1)
python - "$f" "$KERN" > gpurun_out/${TAG}_${KERN}_by_grid.csv <<'PY'
import csv, sys, collections
f, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if kern in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("Grid_Size_Y", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)  # kernel names carry template commas: quote them
w.writerow("kernel,grid_x,grid_y,launches,avg_us,min_us,max_us,sorted_durations_us_of_the_last_step".split(","))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    per_step = max(1, len(v) // 17)
    last = ";".join(f"{d/1e3:.0f}" for d in v[-per_step:])
    w.writerow([k[0], k[1], k[2], len(v), f"{sum(v)/len(v)/1e3:.1f}", f"{min(v)/1e3:.1f}", f"{max(v)/1e3:.1f}", last])
PY
cat gpurun_out/${TAG}_${KERN}_by_grid.csv | head -12
