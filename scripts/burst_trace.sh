#!/bin/bash
# kernel trace of a peer's bursts (scripts/burst_trace.py): when does each launch of a 20-step burst start and end?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/burst_trace; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8; cd $ROOT
for v in "" "--order"; do
  tag=${v:+order}; tag=${tag:-default}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -- python3 scripts/burst_trace.py $v > $OUT/$tag.txt 2> $OUT/$tag.err
  cat $OUT/$tag.txt
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/$tag/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_Y", r.get("Grid_Size", "0")) or 0)))
rows.sort()
# bursts: groups of launches whose starts are within 2 ms of each other and that are 3 launches long (8 + 8 + 4 frames)
i = 0; n = 0
while i < len(rows):
    j = i
    while j + 1 < len(rows) and rows[j + 1][0] - rows[i][0] < 1_000_000: j += 1
    grp = rows[i:j + 1]
    if len(grp) == 3:
        t0 = grp[0][0]
        print("$tag burst:", "; ".join(f"launch {k}: start {(a - t0) / 1e3:.1f} end {(b - t0) / 1e3:.1f} us (grid y {gy})" for k, (a, b, gy) in enumerate(grp)), f"| total {(max(b for _, b, _ in grp) - t0) / 1e3:.1f} us")
        n += 1
    i = j + 1
PY
done
