#!/bin/bash
# same-box A/B of the round-6 `parity`-mode changes: BatchNorm-written planes + batched weight splits on (default) / off
cd "$GRAFT_REPO_ROOT"
run() { python - "$@" <<'PY'
import sys, json, io, contextlib
import uni3detr_amd.sparse as sp, uni3detr_amd.native as nv
on = sys.argv[1] == "on"
sp.BN_PLANES = on; nv.SPLIT3_BATCH = on
import bench
sys.argv = ["bench.py", "--precision", sys.argv[2], "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(sys.argv[2], "new paths", "on " if on else "off", round(d["value"], 2), "scenes/s", round(d["ms_per_step"], 3), "ms")
PY
}
for i in 1 2; do for m in parity mixed; do run off $m 2>/dev/null; run on $m 2>/dev/null; done; done
