#!/bin/bash
# the flat-run form of k_march loops (MARCH_FLAT_RUN in raymarch_kernels.h as of commit 972d54c; removed since: it lost) against
# the default kernel.  Build first (at that commit):   cd sdfbox_amd/csrc && for v in 1 2; do rm -rf obj; make -s -j8 product EXTRA_HIPFLAGS=-DMARCH_FLAT_RUN=$v OUT=../libsdfhip_fr$v.so; done; rm -rf obj; make -s -j8
# On the GPU box: bash scripts/flat_run_ab.sh "0 1 2"   -> gpurun_out/flat_run_ab/summary.txt
#   parity of every variant first (the whole parity suite + 300 fuzz seeds through SDFHIP_LIB), then A/B/A/B bench lines at 1080p (400 steps and the
#   driver's 20), 4K and the depth-10 stand-in, then PMC passes (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_WAIT_ANY, ...).
set -u
export GPU_MAX_HW_QUEUES=8
VARIANTS=${1:-"0 1"}
PHASE=${2:-all}          # parity | time | pmc | all (a gpurun call is at most 20 minutes: the phases fit one each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT=$ROOT/gpurun_out/flat_run_ab; [ "$PHASE" = all ] && rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
lib() { if [ "$1" = 0 ]; then echo $ROOT/sdfbox_amd/libsdfhip.so; else echo $ROOT/sdfbox_amd/libsdfhip_fr$1.so; fi; }
if [ "$PHASE" = all ] || [ "$PHASE" = parity ]; then
for v in $VARIANTS; do
  [ "$v" = 0 ] && continue
  echo "parity of variant $v" >> $OUT/progress.txt
  SDFHIP_LIB=$(lib $v) SDFHIP_FUZZ_SEEDS=300 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "product and not native_library" > $OUT/parity_v$v.log 2>&1 || { echo "PARITY FAILED for variant $v" | tee -a $OUT/progress.txt; tail -5 $OUT/parity_v$v.log; exit 1; }
  tail -1 $OUT/parity_v$v.log >> $OUT/progress.txt
done
fi
if [ "$PHASE" = all ] || [ "$PHASE" = time ]; then
for rep in 1 2; do
  for v in $VARIANTS; do
    export SDFHIP_LIB=$(lib $v)
    python3 bench.py --no-cpu-baseline --configs none > "$OUT/b1080_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --steps 20 --warmup 5 > "$OUT/b1080d_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --size 3840x2160 --steps 200 --warmup 20 > "$OUT/b4k_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --depth 10 --steps 200 --warmup 20 > "$OUT/bd10_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    echo "rep $rep variant $v timed" >> $OUT/progress.txt
  done
done
fi
if [ "$PHASE" = all ] || [ "$PHASE" = pmc ]; then
for v in $VARIANTS; do
  export SDFHIP_LIB=$(lib $v)
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_WAVES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    for wl in "1080:" "4k:--size 3840x2160" "d10:--depth 10"; do
      name=${wl%%:*}; args=${wl#*:}
      rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_v${v}_${name}_$tag" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --configs none $args > /dev/null 2>> "$OUT/err.txt" || echo "pmc $tag v$v $name failed"
    done
  done
  echo "variant $v counted" >> $OUT/progress.txt
done
fi
python3 scripts/flat_run_ab_summary.py "$VARIANTS" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
