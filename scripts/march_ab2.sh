#!/bin/bash
# A/B of k_march scheduling variants (MARCH_AB=1 the product, 4 / 5 wave priority round the cell load; raymarch_device.h find_units):
#   for v in 1 4 5; do (cd sdfbox_amd/csrc && rm -rf obj && make -j8 product EXTRA_HIPFLAGS=-DMARCH_AB=$v OUT=../libsdfhip_ab$v.so); done
#   bash scripts/march_ab2.sh "1 4 5"       (on the GPU box)  -> gpurun_out/march_ab2/
set -u
# the hardware queues bench.py asks for: under rocprofv3 --pmc the profiler has initialised the GPU before bench.py can set it
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
VARS=${1:-"1 4 5"}
OUT=$ROOT/gpurun_out/march_ab2
mkdir -p "$OUT"
for rep in 1 2; do
  for v in $VARS; do
    export SDFHIP_LIB=$ROOT/sdfbox_amd/libsdfhip_ab$v.so
    python3 bench.py --no-cpu-baseline --configs none > "$OUT/b1080_v${v}_$rep.json" 2> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --steps 20 --warmup 5 > "$OUT/b1080d_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --size 3840x2160 --steps 200 --warmup 20 > "$OUT/b4k_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
    python3 bench.py --no-cpu-baseline --configs none --depth 10 --steps 200 --warmup 20 > "$OUT/bd10_v${v}_$rep.json" 2>> "$OUT/err.txt" || echo "bench v$v failed"
  done
done
python3 - "$OUT" "$VARS" <<'PY'
import json, sys, os
out, vs = sys.argv[1], sys.argv[2].split()
for name in ("b1080", "b1080d", "b4k", "bd10"):
    row = []
    for v in vs:
        ms = []
        for rep in (1, 2):
            try:
                ms.append(json.loads(open(f"{out}/{name}_v{v}_{rep}.json").read().strip().splitlines()[-1])["ms_per_step"])
            except Exception as e:
                ms.append(None)
        row.append(f"v{v}: " + " / ".join(str(m) for m in ms))
    print(f"{name:7s} " + "   ".join(row))
PY
