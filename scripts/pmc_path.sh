#!/bin/bash
# PMC passes of the path-traced kernel (1080p, 4 spp): HBM traffic, cache hit rates, waits, lane utilisation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_path; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
A="--spp 4 --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 bench.py $A > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -- python3 bench.py $A > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 bench.py $A > $OUT/p3.json 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $OUT/p4 -- python3 bench.py $A > $OUT/p4.json 2> $OUT/p4.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_path<" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "%.4g"%(sum(v)/len(v)), len(v))
PY
tail -1 $OUT/p1.json | cut -c1-200
