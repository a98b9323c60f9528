#!/bin/bash
# The round's bench lines (GPU box, via gpurun): gpurun_out/bench_lines/<tag>_<name>.json -> copy to profiles/bench_lines/
R=${1:-r05}; O=gpurun_out/bench_lines; mkdir -p $O
run() { n=$1; shift; python bench.py "$@" > $O/${R}_$n.json 2> $O/${R}_$n.err || echo "$n failed"; }
run driver_line_with_configs --steps 20 --warmup 5
run 1080p_default
run 1080p_one_frame_in_flight --no-cpu-baseline --frames-in-flight 1
run 1080p_shadow_queue --no-cpu-baseline --lab --shadow-queue
run 1080p_shadow_queue_single --no-cpu-baseline --lab --shadow-queue --frames-in-flight 1
run 1080p_one_kernel --no-cpu-baseline --lab --one-kernel
run 1080p_one_kernel_single --no-cpu-baseline --lab --one-kernel --frames-in-flight 1
run 1080p_display --no-cpu-baseline --display
run 1080p_depth10 --no-cpu-baseline --depth 10
run 1080p_compact --no-cpu-baseline --compact 1
run 1080p_orbit --no-cpu-baseline --orbit 90
run 4k_default --no-cpu-baseline --size 3840x2160
run 4k_shadow_queue --no-cpu-baseline --size 3840x2160 --lab --shadow-queue
run 4k_one_kernel --no-cpu-baseline --size 3840x2160 --lab --one-kernel
run 4k_compact --no-cpu-baseline --size 3840x2160 --compact 1
run cfg5_4k_spp16 --no-cpu-baseline --size 3840x2160 --spp 16 --steps 8 --warmup 2
run cfg5_4k_spp16_one_kernel --no-cpu-baseline --size 3840x2160 --spp 16 --steps 8 --warmup 2 --lab --one-kernel
run sharded_1rank_nccl_4k --no-cpu-baseline --exercise-gather --check --size 3840x2160
run sharded_2rank_gloo_1080p --no-cpu-baseline --gpus 2 --backend gloo --check --steps 40 --warmup 8
# the library's own multi-device entry points (sdfhip_multi_*), the ranks played by the one GPU
run multi_4k_groups_1rank --single-process --devices 0 --size 3840x2160 --check --steps 64 --warmup 16
run multi_4k_groups_2ranks --single-process --devices 0,0 --size 3840x2160 --check --steps 64 --warmup 16
run multi_4k_groups_4ranks --single-process --devices 0,0,0,0 --size 3840x2160 --check --steps 64 --warmup 16
run multi_4k_groups_8ranks --single-process --devices 0,0,0,0,0,0,0,0 --size 3840x2160 --check --steps 64 --warmup 16
run multi_4k_frame_4ranks --single-process --devices 0,0,0,0 --size 3840x2160 --check --steps 64 --warmup 16 --multi-mode frame
run multi_4k_frame_4ranks_moving --single-process --devices 0,0,0,0 --size 3840x2160 --check --steps 64 --warmup 16 --multi-mode frame --orbit 90
run multi_1080p_groups_4ranks --single-process --devices 0,0,0,0 --check --steps 128 --warmup 32
run multi_1080p_frame_4ranks --single-process --devices 0,0,0,0 --check --steps 128 --warmup 32 --multi-mode frame
run multi_tiny_frame_1rank --single-process --devices 0 --size 64x64 --depth 6 --steps 300 --warmup 30 --multi-mode frame
run multi_tiny_frame_4ranks --single-process --devices 0,0,0,0 --size 64x64 --depth 6 --steps 300 --warmup 30 --multi-mode frame
run multi_cfg5_4ranks --single-process --devices 0,0,0,0 --size 3840x2160 --spp 16 --check --steps 4 --warmup 1
for f in $O/${R}_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "Mray/s", d["ms_per_step"], "ms/step, latency", d["latency_ms"], "hbm_frac", r.get("hbm_frac"), "of measured", r.get("hbm_frac_of_measured"), "valu of spec", r.get("valu_frac_of_spec"), "limiting", r.get("limiting"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
