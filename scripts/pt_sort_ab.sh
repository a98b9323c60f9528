#!/bin/bash
# A/B of the coherent bounce levels (SDFHIP_PT_SORT=R): per-kernel time (rocprofv3 --kernel-trace --stats) and HBM bytes
# (separate --pmc FETCH_SIZE / WRITE_SIZE / TCC hit passes) of cfg-5, R = 0 (off) and the values given.  Runs on the GPU box:
#   bash scripts/pt_sort_ab.sh 0 3     -> gpurun_out/pt_sort/<R>/...
set -u
# the hardware queues bench.py asks for: under rocprofv3 --pmc the profiler has initialised the GPU before bench.py can set it
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
for SPEC in "$@"; do
  R=${SPEC%%:*}; FROM=0; [[ "$SPEC" == *:* ]] && FROM=${SPEC##*:}
  export SDFHIP_PT_SORT_FROM=$FROM
  OUT=$ROOT/gpurun_out/pt_sort/$SPEC
  mkdir -p "$OUT"
  export SDFHIP_PT_SORT=$R
  ARGS="--size 3840x2160 --spp 16 --steps 4 --warmup 1 --no-cpu-baseline --configs none --frames-in-flight 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/stats.err" || echo "stats pass failed"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_$tag.err" || echo "pmc pass $tag failed"
  done
done
python3 scripts/pt_sort_summary.py "$@" > "$ROOT/gpurun_out/pt_sort/summary.txt"
cat "$ROOT/gpurun_out/pt_sort/summary.txt"
