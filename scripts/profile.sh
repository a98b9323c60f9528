#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC
# passes of the bench command.  Usage: bash scripts/profile.sh <tag> [bench args...]
# (--sustained off --only-timed: the continuous legs and the moving-camera / tile-order latency passes of the default run would add
# frames of OTHER cameras and launch orders to a pass whose per-frame figures are averages over every launch of the process)
# Output: gpurun_out/prof_<tag>/...  (scripts/summarise_profile.py turns it into
# profiles/<tag>_*.{csv,json}, which are committed)
set -u
# the hardware queues bench.py asks for: under rocprofv3 --pmc the profiler has initialised the GPU before bench.py can set it
export GPU_MAX_HW_QUEUES=8
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --configs none --sustained off --only-timed "$@" > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || echo "stats pass failed"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
            "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $pass"
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --configs none --sustained off --only-timed "$@" > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.err" || echo "pmc pass $i failed"
done
du -sh "$OUT"
