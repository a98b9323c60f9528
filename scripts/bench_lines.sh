#!/bin/bash
# The round's bench lines (GPU box, via gpurun): gpurun_out/bench_lines/<tag>_<name>.json -> copy to profiles/bench_lines/
R=${1:-r02}; O=gpurun_out/bench_lines; mkdir -p $O
run() { n=$1; shift; python bench.py "$@" > $O/${R}_$n.json 2> $O/${R}_$n.err || echo "$n failed"; }
run 1080p_default
run 1080p_driver_style --steps 20 --warmup 5
run 1080p_one_frame_in_flight --no-cpu-baseline --frames-in-flight 1
run 1080p_shadow_queue --no-cpu-baseline --shadow-queue
run 1080p_shadow_queue_single --no-cpu-baseline --shadow-queue --frames-in-flight 1
run 1080p_one_kernel --no-cpu-baseline --one-kernel
run 1080p_one_kernel_single --no-cpu-baseline --one-kernel --frames-in-flight 1
run 1080p_display --no-cpu-baseline --display
run 1080p_depth10 --no-cpu-baseline --depth 10
run 1080p_compact --no-cpu-baseline --compact 1
run 1080p_orbit --no-cpu-baseline --orbit 90
run 4k_default --no-cpu-baseline --size 3840x2160
run 4k_shadow_queue --no-cpu-baseline --size 3840x2160 --shadow-queue
run 4k_one_kernel --no-cpu-baseline --size 3840x2160 --one-kernel
run 4k_compact --no-cpu-baseline --size 3840x2160 --compact 1
run cfg5_4k_spp16 --no-cpu-baseline --size 3840x2160 --spp 16 --steps 8 --warmup 2
run cfg5_4k_spp16_one_kernel --no-cpu-baseline --size 3840x2160 --spp 16 --steps 8 --warmup 2 --one-kernel
run sharded_1rank_nccl_4k --no-cpu-baseline --exercise-gather --check --size 3840x2160
run sharded_2rank_gloo_1080p --no-cpu-baseline --gpus 2 --backend gloo --check --steps 40 --warmup 8
for f in $O/${R}_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], "Mray/s", d["ms_per_step"], "ms/step, latency", d["latency_ms"], "frac", d["roofline"]["frac"], d["roofline"]["binding"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
