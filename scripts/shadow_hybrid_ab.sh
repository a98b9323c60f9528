#!/bin/bash
# Laboratory A/B (GPU): the shadow-ray queue with a threshold -- waves holding at least T shadow rays march them in place, sparser
# waves hand theirs to the queue (ballot + prefix) for k_shadow -- against the default kernel.  T = 65 queues every ray (the
# form measured in rounds 2-3), T = 1 queues none.  usage: bash scripts/shadow_hybrid_ab.sh   -> gpurun_out/shadow_hybrid/
set -u
# the hardware queues bench.py asks for: under rocprofv3 --pmc the profiler has initialised the GPU before bench.py can set it
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=$ROOT/gpurun_out/shadow_hybrid; mkdir -p "$OUT"
# parity first: the queued form's tests with a threshold in the middle
SDFHIP_SHADOW_MIN_LANES=16 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "oracle or golden or fuzz or edge or bench_frame" > "$OUT/parity_T16.log" 2>&1; echo "parity T=16 rc=$?"
tail -2 "$OUT/parity_T16.log"
common="--lab --no-cpu-baseline --configs none"
for rep in 1 2; do
  python3 bench.py $common > "$OUT/1080_default_$rep.json" 2>> "$OUT/err.txt"
  python3 bench.py $common --size 3840x2160 --steps 200 --warmup 20 > "$OUT/4k_default_$rep.json" 2>> "$OUT/err.txt"
  for T in 65 48 32 16 8 4; do
    SDFHIP_SHADOW_MIN_LANES=$T python3 bench.py $common --shadow-queue > "$OUT/1080_T${T}_$rep.json" 2>> "$OUT/err.txt"
    SDFHIP_SHADOW_MIN_LANES=$T python3 bench.py $common --shadow-queue --size 3840x2160 --steps 200 --warmup 20 > "$OUT/4k_T${T}_$rep.json" 2>> "$OUT/err.txt"
  done
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for size in ("1080", "4k"):
    for name in ["default"] + [f"T{t}" for t in (65, 48, 32, 16, 8, 4)]:
        ms = []
        for rep in (1, 2):
            try:
                d = json.loads(open(f"{out}/{size}_{name}_{rep}.json").read().strip().splitlines()[-1]); ms.append((d["ms_per_step"], d["latency_ms"]))
            except Exception:
                ms.append(None)
        print(size, name, ms)
PY
