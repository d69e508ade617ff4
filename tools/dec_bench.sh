#!/bin/bash
# tools/dec_bench.sh <tag> [lib.so]: per-kernel average durations of the fused decoder layer (rocprofv3 kernel trace) -> gpurun_out/dec_<tag>.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1
[ -n "$2" ] && export U3D_LIB_PATH=$REPO/$2
export TMPDIR=/tmp
OUT=/tmp/decprof_$TAG
rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $REPO/tools/dec_bench.py 30 > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
{ echo "== $TAG"; grep dec_bench $OUT/run.log | tail -1; [ -z "$F" ] && tail -5 $OUT/run.log; [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = r["Name"]
    if n.startswith(("void k_dec", "void k_mha", "k_dec", "k_mha")):
        short = n.split("(")[0].replace("void ", "")
        print(f"{short:28s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:8.1f} us")
        tot += float(r["AverageNs"]) / 1e3
print(f"{'sum of the 7 kernels':28s}            {tot:8.1f} us per layer fwd+bwd")
PY
} | tee $REPO/gpurun_out/dec_$TAG.txt
