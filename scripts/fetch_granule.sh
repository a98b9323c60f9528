#!/bin/bash
# What FETCH_SIZE counts for scattered 16-byte loads, and the L2's fetch granule: scripts/micro/fetch_granule.hip plainly and
# under separate --pmc passes (build the binary first: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/fetch_granule scripts/micro/fetch_granule.hip)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/fetch_granule; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
BIN=$ROOT/scripts/micro/fetch_granule
$BIN > $OUT/plain.txt || exit 1
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1)); echo "pass $i: $pass" >> $OUT/progress.txt
  timeout -k 5 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- $BIN > $OUT/p$i.txt 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/progress.txt
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open("$OUT/plain.txt").read())
LINES=float(1<<25)
for k in sorted(agg):
    print(k)
    for c,v in sorted(agg[k].items()):
        m=sum(v)/len(v)
        print("   %-28s %14.6g per launch = %8.3f per line%s   (%d launches)"%(c, m, m/LINES, "  (x 1024 B: %.1f B per line)"%(m*1024/LINES) if c in ("FETCH_SIZE","WRITE_SIZE") else "", len(v)))
PY
cat $OUT/summary.txt; cat $OUT/progress.txt
