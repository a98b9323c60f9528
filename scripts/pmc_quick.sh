#!/bin/bash
# quick PMC comparison: bash scripts/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmcq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > $OUT/p2.json 2> $OUT/p2.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ("k_plain" in r["Kernel_Name"] or "k_compact" in r["Kernel_Name"]) and ", true" not in r["Kernel_Name"].split("<")[1][:12].replace("true, true","X") :
            agg[(r["Kernel_Name"][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "%.4g"%(sum(v)/len(v)), len(v))
PY
