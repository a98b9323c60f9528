#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC
# passes of the bench command.  Usage: bash scripts/profile.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/...  (copy the summaries you keep into profiles/)
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="--steps 30 --warmup 3 --no-cpu-baseline $*"
echo "== kernel trace + stats" 
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || echo "stats pass failed"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $* > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc pass $pass failed"
done
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
