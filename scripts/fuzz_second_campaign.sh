#!/bin/bash
# Runs on the GPU box (via gpurun): a second correctness campaign on the final build over seeds the first one (VERDICT r5 item 6:
# seeds 0-2999, clouds 0-149) did not see: seeds 3000-8999 in two halves (a progress line between them), clouds 150-599.
# Output: gpurun_out/fuzz2/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=$ROOT/gpurun_out/fuzz2; mkdir -p "$OUT"
python3 -c "import bench_report; print(bench_report.kernel_source_hash())" > "$OUT/sha.txt"
for first in 3000 6000; do
  SDFHIP_FUZZ_FIRST=$first SDFHIP_FUZZ_SEEDS=3000 timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k test_fuzz_random_trees_and_on_grid_cameras > "$OUT/fuzz_$first.log" 2>&1
  rc=$?; echo "fuzz seeds $first .. $((first + 2999)): exit $rc: $(tail -1 "$OUT/fuzz_$first.log")" | tee -a "$OUT/progress.txt"
  [ $rc -eq 0 ] || exit 1
done
timeout -k 10 900 python3 scripts/sdfgen_fuzz.py 450 150 > "$OUT/sdfgen_fuzz.log" 2>&1
rc=$?; echo "sdfgen_fuzz exit $rc: $(tail -1 "$OUT/sdfgen_fuzz.log")" | tee -a "$OUT/progress.txt"
exit $rc
