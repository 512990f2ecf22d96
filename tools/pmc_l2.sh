set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/l2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc" -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 2
ls -R "$OUT/pmc" | head
