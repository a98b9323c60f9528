#!/bin/bash
# scripts/pt_xcd_order_ab.py under the clock, then per-kernel HBM bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of three
# settings -> gpurun_out/pt_xcd/summary.txt
set -u
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pt_xcd
mkdir -p "$OUT"
python3 scripts/pt_xcd_order_ab.py time > "$OUT/time.txt" 2> "$OUT/time.err" || { echo "timing pass failed"; tail -5 "$OUT/time.err"; exit 1; }
cat "$OUT/time.txt"
for SPEC in 0:0 3:0 3:1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${SPEC/:/_}_$C" -- python3 scripts/pt_xcd_order_ab.py one $SPEC 2 > /dev/null 2> "$OUT/pmc_${SPEC/:/_}_$C.err" || echo "pmc pass $SPEC $C failed"
  done
done
for SPEC in 0:0 3:0 3:1; do
  PT_AB_FRAMES_IN_FLIGHT=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${SPEC/:/_}" -- python3 scripts/pt_xcd_order_ab.py one $SPEC 3 > /dev/null 2> "$OUT/trace_${SPEC/:/_}.err" || echo "trace pass $SPEC failed"
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, glob, os, re, sys
out = sys.argv[1]
print(open(os.path.join(out, "time.txt")).read())
for spec in ("0_0", "3_0", "3_1"):
    row = []
    for c, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        fs = glob.glob(f"{out}/pmc_{spec}_{c}/*/*_counter_collection.csv")
        if not fs:
            continue
        agg, n = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            m = re.match(r"(?:void )?(?:sdfhip::)?(k_pt_\w+)", r["Kernel_Name"])
            if m and r["Counter_Name"] == c:
                agg[m.group(1)] += float(r["Counter_Value"]); n[m.group(1)] += 1
        frames = n.get("k_pt_primary", 1) or 1
        row.append(f"{c}" + (" (doubled)" if mult == 2 else "") + ": " + ", ".join(f"{k} {v * 1024 * mult / frames / 1e9:.2f} GB" for k, v in sorted(agg.items())))
    print(f"SORT={spec.replace('_', ' XCD=')}: per frame: " + "; ".join(row))
    fs = glob.glob(f"{out}/trace_{spec}/*/*_kernel_trace.csv")
    if fs:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            m = re.match(r"(?:void )?(?:sdfhip::)?(k_pt_\w+)", r["Kernel_Name"])
            if m:
                per[m.group(1)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        frames = len(per.get("k_pt_primary", [])) or 1
        parts = []
        for k, v in sorted(per.items()):
            v.sort()
            parts.append(f"{k} {sum(x[1] for x in v) / frames / 1e6:.3f}")
            if k == "k_pt_bounce":
                lv = len(v) // frames
                parts.append("(levels " + " ".join(f"{sum(x[1] for x in v[i::lv]) / frames / 1e6:.3f}" for i in range(lv)) + ")")
        print(f"    kernel ms per frame (one frame in flight under the tracer): " + ", ".join(parts))
PY
