"""gpurun_out/flat_run_ab/ -> the A/B table of k_march's flat-run forms (scripts/flat_run_ab.sh)."""
import collections, csv, glob, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(REPO, "gpurun_out", "flat_run_ab")
NAMES = {0: "the default kernel (flat and non-flat lanes share an iteration)", 1: "flat-run form of the primary march", 2: "flat-run form of the primary and the shadow march",
         3: "primary march, the trilinear block as soon as 8 lanes wait for it", 4: "primary march, the trilinear block as soon as 16 lanes wait for it"}
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0 1").split()]
# (SQ_INSTS_VMEM_RD = vector load instructions of the waves: 1.3 per wave-iteration of the default kernel -- the count of iterations)


def ms(f):
    try:
        return json.loads([l for l in open(f) if l.startswith("{")][-1])["ms_per_step"]
    except Exception:  # noqa: BLE001
        return None


def counters(v, wl):
    pm = {}
    for d in glob.glob(f"{D}/pmc_v{v}_{wl}_*"):
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if re.search(r"k_march", r["Kernel_Name"]):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, vals in agg.items():
                pm[c] = sum(vals) / len(vals)
    return pm


print("ms per frame, A/B/A/B (two runs each): 1080p 400 steps | 1080p driver style, 20 steps | 4K, 200 steps | depth-10 stand-in 1080p, 200 steps")
for v in variants:
    row = [f"{v}: {NAMES.get(v, '?')}"]
    for kind in ("b1080", "b1080d", "b4k", "bd10"):
        row.append(" / ".join(str(ms(os.path.join(D, f"{kind}_v{v}_{r}.json"))) for r in (1, 2)))
    print(" | ".join(row))
print()
print("counters per frame of k_march (rocprofv3 --pmc, one frame in flight while counting)")
for wl in ("1080", "4k", "d10"):
    for v in variants:
        pm = counters(v, wl)
        if not pm:
            continue
        lanes = pm.get("SQ_THREAD_CYCLES_VALU", 0) / (pm["SQ_INSTS_VALU"] * 64) if pm.get("SQ_INSTS_VALU") else None
        print(f"{wl:5s} variant {v}: " + ", ".join(f"{k} {pm[k] / (1e6 if k != 'FETCH_SIZE' else 1024):.2f}{' M' if k != 'FETCH_SIZE' else ' MB (raw)'}" for k in
              ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_THREAD_CYCLES_VALU", "FETCH_SIZE") if k in pm) +
              (f", lanes on in {lanes:.3f} of the VALU thread-cycles" if lanes else ""))
