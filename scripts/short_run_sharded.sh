#!/bin/bash
# Dev script (GPU): what a SHORT sharded run measures -- the driver's scaling bench times 20 steps -- against the steady state,
# through the one-process-per-GPU path with one NCCL rank (bands + sparse shares + gather + expansion on this GPU).
# usage: scripts/short_run_sharded.sh <tag>     -> gpurun_out/<tag>/*.json
set -u
out=gpurun_out/${1:-short}; mkdir -p "$out"
common="--gpus 1 --exercise-gather --no-cpu-baseline --configs none"
for g in 4 8; do
  for k in "20 5" "24 5" "400 40"; do
    set -- $k
    python bench.py $common --gather-every $g --steps $1 --warmup $2 > "$out/1080p_G${g}_K$1.json" 2> "$out/1080p_G${g}_K$1.err" || exit 1
    python bench.py $common --gather-every $g --steps $1 --warmup $2 --size 3840x2160 > "$out/4k_G${g}_K$1.json" 2> "$out/4k_G${g}_K$1.err" || exit 1
  done
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d["ms_per_step"], d["value"], d.get("assembled_frame_equals_whole_frame_render"))
PY
