"""gpurun_out/march_ab/ -> the A/B table of the k_march variants (scripts/march_ab.sh)."""
import collections, csv, glob, json, os, re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(REPO, "gpurun_out", "march_ab")
NAMES = {0: "round 2's kernel", 1: "lattice branch with __builtin_expect", 2: "cell load issued before the lattice test", 3: "all-inside-the-cube waves skip the clamps (removed)"}


def ms(f):
    try:
        return json.loads([l for l in open(f) if l.startswith("{")][-1])["ms_per_step"]
    except Exception:  # noqa: BLE001
        return None


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


print("variant | 1080p 400 steps (two runs) | 1080p driver style 20 steps | 4K 200 steps | per frame: VALU M, SALU M, wave-cycles M, wait-any M, wait-inst M, FETCH_SIZE MB (raw), TA busy, L1 lookups M")
for v in range(4):
    row = [f"{v} {NAMES[v]}"]
    for kind in ("b1080", "b1080d", "b4k"):
        row.append(" / ".join(str(ms(os.path.join(D, f"{kind}_v{v}_{r}.json"))) for r in (1, 2)))
    pm = {}
    for tag in ("SQ_INSTS_VALU", "FETCH_SIZE", "TA_TA_BUSY_sum"):
        f = newest(f"{D}/pmc_v{v}_{tag}/*/*_counter_collection.csv")
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if re.search(r"k_march<3, false, 0, false>|k_marchILi3ELb0ELi0ELb0", r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, vals in agg.items():
            pm[c] = sum(vals) / len(vals)
    row.append(", ".join(f"{k} {pm[k] / (1e6 if k != 'FETCH_SIZE' else 1024):.2f}" for k in
                         ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "TA_TA_BUSY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum") if k in pm))
    print(" | ".join(row))
