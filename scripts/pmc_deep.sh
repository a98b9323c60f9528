#!/bin/bash
# deeper SQ counters of the default kernel (one frame in flight: PMC serialises launches anyway)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_deep; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
A="--steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $*"
rocprofv3 --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 bench.py $A > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 bench.py $A > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p3 -- python3 bench.py $A > $OUT/p3.json 2> $OUT/p3.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/p4 -- python3 bench.py $A > $OUT/p4.json 2> $OUT/p4.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if ("k_plain" in n or "k_compact" in n or "k_path" in n) and "true, true" not in n:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "%.5g"%(sum(v)/len(v)), len(v))
PY
