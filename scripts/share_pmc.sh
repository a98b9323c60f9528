#!/bin/bash
# What does a rank's share cost beside the whole frame, by the counters?  scripts/burst_trace.py (a peer's launches at 8 ranks: k_march<..., OUT_SPARSE>,
# 8 frame-shares per launch) under separate --pmc passes, against the whole-frame launches of the same script's first render.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/share_pmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 scripts/burst_trace.py --whole > $OUT/p$i.txt 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"]:
            key = ("whole frame (OUT_RGBA32F)" if "Li0E" in r["Kernel_Name"] or ", 0, false" in r["Kernel_Name"] else "share (OUT_SPARSE)") + " grid " + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, "launches", len(next(iter(agg[k].values()))))
    for c, v in sorted(agg[k].items()):
        print("   %-26s %14.6g per launch" % (c, sum(v) / len(v)))
PY
