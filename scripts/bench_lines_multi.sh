R=r04; O=gpurun_out/bench_lines; mkdir -p $O
run() { n=$1; shift; python bench.py "$@" > $O/${R}_$n.json 2> $O/${R}_$n.err || echo "$n failed"; }
run sharded_1rank_nccl_4k --no-cpu-baseline --exercise-gather --check --size 3840x2160
run sharded_2rank_gloo_1080p --no-cpu-baseline --gpus 2 --backend gloo --check --steps 40 --warmup 8
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
for f in $O/${R}_multi*.json $O/${R}_sharded*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], "Mray/s", d["ms_per_step"], "ms/step, latency", d["latency_ms"], d["config"].get("assembled_frame_equals_whole_frame_render"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
python scripts/rank_emulation.py > gpurun_out/rank_emul3.log 2>&1; python scripts/rank_emulation.py 3840x2160 >> gpurun_out/rank_emul3.log 2>&1
