#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command, then HBM
# traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are
# never combined with other trace domains).  Results land under gpurun_out/<tag>/.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err" || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || exit 2
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || exit 3
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_sq.json" 2> "$OUT/bench_sq.err" || exit 5
timeout -s KILL 240 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $BENCH > "$OUT/bench_l2.json" 2> "$OUT/bench_l2.err" || exit 6
python "$ROOT/tools/summarize_profile.py" "$OUT" > "$OUT/summary.json" || exit 4
cat "$OUT/summary.json" | head -c 3000
