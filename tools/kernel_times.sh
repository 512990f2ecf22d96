#!/bin/bash
# Kernel-trace only (no counters): per-kernel average duration of a short default bench run.
set -u
TAG=${1:-kt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
python - "$OUT" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print("%-56s calls %4s avg_us %9.1f pct %5s" % (n[:56], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
