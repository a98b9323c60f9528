#!/bin/bash
# memory-path counters of the default kernel (one frame in flight: PMC serialises launches anyway)
# usage: bash scripts/pmc_mem.sh <tag> [bench args]     (few counters per pass: the TA / TCP blocks have few slots)
TAG=${1:-mem}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_mem_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
A="--steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $*"
i=0
for pass in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM" \
            "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); echo "pass $i: $pass" >> $OUT/progress.txt
  timeout -k 5 150 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 bench.py $A > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/progress.txt
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_march" in n and "true" not in n.split("k_march")[1][:14]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-40s %.5g   (%d launches)"%(k, sum(v)/len(v), len(v)))
PY
