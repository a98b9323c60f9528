"""gpurun_out/pt_sort/<R>/ -> per-kernel time per frame and HBM bytes per frame of cfg-5 (scripts/pt_sort_ab.sh)."""
import collections, csv, glob, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"(?:void )?(?:sdfhip::)?(k_\w+)", name)
    return m.group(1) if m else None


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


for R in sys.argv[1:]:
    d = os.path.join(REPO, "gpurun_out", "pt_sort", R)
    print(f"== SDFHIP_PT_SORT={R}")
    try:
        line = json.loads([l for l in open(os.path.join(d, "bench.json")) if l.startswith("{")][-1])
        print("bench (under the tracer, one frame in flight): ms_per_step", line["ms_per_step"])
    except Exception as e:  # noqa: BLE001
        print("no bench line", e)
    f = newest(f"{d}/stats/*/*_kernel_trace.csv")
    frames = 0
    if f:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and k.startswith("k_pt_"):
                per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        frames = len(per.get("k_pt_primary", [])) or 1
        for k, v in sorted(per.items()):
            v.sort()
            tot = sum(x[1] for x in v) / 1e6
            print(f"  {k:16s} launches {len(v):4d}  total {tot:9.3f} ms  per frame {tot / frames:8.3f} ms")
            if k == "k_pt_bounce":
                lv = len(v) // frames
                for i in range(lv):
                    print(f"      level {i}: {sum(x[1] for x in v[i::lv]) / frames / 1e6:8.3f} ms per frame")
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
        f = newest(f"{d}/pmc_{tag}/*/*_counter_collection.csv")
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and k.startswith("k_pt_"):
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
                    n[k] += 1
        fr = n.get("k_pt_primary", 1) or 1
        for k, cs in sorted(agg.items()):
            if tag == "TCC_HIT_sum":
                hit, miss = cs.get("TCC_HIT_sum", 0), cs.get("TCC_MISS_sum", 0)
                print(f"  {k:16s} L2 hit {hit / max(1.0, hit + miss):.3f}  requests per frame {cs.get('TCC_REQ_sum', 0) / fr / 1e6:10.1f} M")
            else:
                mult = 2.0 if tag == "FETCH_SIZE" else 1.0
                print(f"  {k:16s} {tag} per frame {cs[tag] * 1024 * mult / fr / 1e9:8.2f} GB" + (" (doubled)" if mult == 2 else ""))
