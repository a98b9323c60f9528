#!/bin/bash
# rocprofv3 evidence for the GPU SdfGen builder on the 1 M-point knot, depth 9 and 10: kernel trace + stats, then
# separate FETCH_SIZE / WRITE_SIZE passes.  usage (GPU box): bash scripts/profile_sdfgen.sh <tag>
TAG=${1:-r03}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_${TAG}_sdfgen; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
for d in 9 10; do
  echo "depth $d: stats" >> $OUT/progress.txt
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_d$d -- python3 scripts/sdfgen_profile.py $d > $OUT/stats_d$d.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
    n=$(echo $c | cut -d' ' -f1); echo "depth $d: pmc $n" >> $OUT/progress.txt
    timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_d${d}_$n -- python3 scripts/sdfgen_profile.py $d > $OUT/pmc_d${d}_$n.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in (9, 10):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"$OUT/pmc_d{d}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sdfhip::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["launches_" + r["Counter_Name"]] += 1
    out[f"depth_{d}"] = {k: dict(v) for k, v in agg.items()}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels:", {d: len(v) for d, v in out.items()})
PY
ls $OUT
