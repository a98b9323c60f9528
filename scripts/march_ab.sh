#!/bin/bash
# A/B of the k_march variants of ROUND 4 (MARCH_AB=0..3 in raymarch_device.h find_units as of commit 818aed4; round 5 took the variants out of the
# product header -- the numbers are profiles/r04_k_march_ab.txt (history: commit 53ee955)): bench lines per build, A/B/A/B, then PMC.  Kept as the recipe of an A/B.
#   bash scripts/march_ab.sh        (on the GPU box)  -> gpurun_out/march_ab/
set -u
# the hardware queues bench.py asks for: under rocprofv3 --pmc the profiler has initialised the GPU before bench.py can set it
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/march_ab
mkdir -p "$OUT"
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1 2 3; do
    export SDFHIP_LIB=$ROOT/sdfbox_amd/libsdfhip_ab$v.so
    python3 bench.py --no-cpu-baseline --configs none > "$OUT/b1080_v${v}_$rep.json" 2> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --steps 20 --warmup 5 > "$OUT/b1080d_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --size 3840x2160 --steps 200 --warmup 20 > "$OUT/b4k_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
  done
done
for v in 0 1 2 3; do
  export SDFHIP_LIB=$ROOT/sdfbox_amd/libsdfhip_ab$v.so
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_v${v}_$tag" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --configs none > /dev/null 2>> "$OUT/err.txt" || echo "pmc $tag v$v failed"
  done
done
python3 scripts/march_ab_summary.py > "$OUT/summary.txt"
cat "$OUT/summary.txt"
