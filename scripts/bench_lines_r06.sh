#!/bin/bash
# Round 6's bench lines (GPU box, via gpurun): gpurun_out/bench_lines/r06_<name>.json -> copy to profiles/bench_lines/
# Run AFTER scripts/summarise_all.py r06 + compulsory_bytes.py r06 --install of the same build, so that the lines find their counters.
R=${1:-r06}; O=gpurun_out/bench_lines; mkdir -p $O
run() { n=$1; shift; python bench.py "$@" > $O/${R}_$n.json 2> $O/${R}_$n.err || echo "$n failed"; }
run driver_line_with_configs --steps 20 --warmup 5
run 1080p_default --no-cpu-baseline --configs none
run 1080p_orbit --no-cpu-baseline --configs none --orbit 90
run 4k_default --no-cpu-baseline --size 3840x2160
run cfg5_4k_spp16 --no-cpu-baseline --size 3840x2160 --spp 16 --steps 8 --warmup 2
# the N > 1 machinery rehearsed on the one GPU: one NCCL rank through the sharded path, two gloo ranks (link check, per_rank_ms,
# steady_state), and the library's own multi-device entry points over the device list [0, 0]
run sharded_1rank_nccl_1080p --no-cpu-baseline --exercise-gather --check --steps 20 --warmup 5
run sharded_2rank_gloo_1080p --no-cpu-baseline --gpus 2 --backend gloo --check --steps 20 --warmup 5
run multi_1080p_groups_2ranks --single-process --devices 0,0 --check --steps 64 --warmup 16
for f in $O/${R}_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], "Mray/s", d["ms_per_step"], "ms/step; hbm_frac", r.get("hbm_frac"), "valu", r.get("valu_frac_of_spec"),
          "traffic/compulsory", r.get("traffic_over_compulsory"), r.get("traffic_over_compulsory_per_xcd"), "| check", d["config"].get("assembled_frame_equals_whole_frame_render"))
    if d.get("sustained"): print("   sustained:", {k: (v.get("ms_per_step"), (v.get("telemetry") or {}).get("sclk_mhz")) for k, v in d["sustained"].items() if isinstance(v, dict)})
    if d.get("steady_state"): print("   steady_state:", d["steady_state"]["ms_per_step"], "per_rank_ms", d["config"].get("per_rank_ms"), "rank0_assemble_ms", d["config"].get("rank0_assemble_ms"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
