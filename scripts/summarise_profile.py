"""gpurun_out/prof_<tag>/ -> profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json
(per-launch means of every counter for the ray-march kernel) and the HBM traffic
entry of profiles/hbm_traffic.json that bench.py reports as roofline.traffic.

HBM bytes per launch (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KB;
WRITE_SIZE is exact for 16-B-per-lane stores (ours); FETCH_SIZE = TCC_EA0_RDREQ x 64 B
under-reports wide coalesced reads by 2x on gfx950 and is uncalibrated for other shapes:
we report the doubled figure as the (upper) estimate and keep the raw one beside it."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(f"{src}/stats/*/*_kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
bench = None
try:
    bench = json.loads([l for l in open(f"{src}/stats_bench.json") if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
pmc = {}
kname = None
for f in sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_plain" in r["Kernel_Name"] or "k_compact" in r["Kernel_Name"]:
            if ", true>" in r["Kernel_Name"]:       # the one counting launch
                continue
            kname = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc["VGPR_Count"] = int(r["VGPR_Count"]); pmc["SGPR_Count"] = int(r["SGPR_Count"]); pmc["LDS_Block_Size"] = int(r["LDS_Block_Size"])
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
pmc["kernel"] = kname
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    pmc["hbm_read_bytes_raw"] = pmc["FETCH_SIZE"] * 1024
    pmc["hbm_read_bytes_x2"] = pmc["FETCH_SIZE"] * 2048
    pmc["hbm_write_bytes"] = pmc["WRITE_SIZE"] * 1024
    pmc["hbm_bytes_per_launch"] = pmc["hbm_read_bytes_x2"] + pmc["hbm_write_bytes"]
if bench:
    pmc["bench_line"] = bench
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in pmc.items() if k != "bench_line"}, indent=1, sort_keys=True))
if bench and "hbm_bytes_per_launch" in pmc and len(sys.argv) > 2:
    key = sys.argv[2]            # e.g. "1920x1080:dragon_standin_d9"
    p = "profiles/hbm_traffic.json"
    t = json.load(open(p)) if os.path.exists(p) else {}
    t[key] = int(pmc["hbm_bytes_per_launch"])
    json.dump(t, open(p, "w"), indent=1, sort_keys=True)
