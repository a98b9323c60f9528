"""gpurun_out/lat_<tag>/p*/ (scripts/pmc_latency_study.sh) -> per kernel, per launch: the counters' means and what they say about
the memory path.  k_pt_bounce's launches are split by bounce level (its launches come in fours: level 0 .. 3 of a frame).
usage: python scripts/pmc_latency_summary.py <tag> [n_levels]"""
import collections, csv, glob, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def short(name):
    m = re.match(r"(?:void )?(?:sdfhip::)?(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


vals = collections.defaultdict(lambda: collections.defaultdict(list))       # kernel (+ level) -> counter -> per launch
for d in sorted(glob.glob(os.path.join(REPO, "gpurun_out", f"lat_{tag}", "p*"))):
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    rows = collections.defaultdict(dict)                                     # dispatch id -> {kernel, counters}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        if not k or not k.startswith(("k_march", "k_shadow", "k_pt_")) or re.search(r"<\d+, true", k) or k.endswith("<true>"):
            continue
        rows[int(r["Dispatch_Id"])]["k"] = k
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    nth = collections.Counter()
    for did in sorted(rows):
        e = rows[did]; k = e.pop("k")
        name = k.split("<")[0]
        if name == "k_pt_bounce":
            name += f" level {nth[name] % levels}"; nth["k_pt_bounce"] += 1
        for c, v in e.items():
            vals[name][c].append(v)
out = {}
for name in sorted(vals):
    m = {c: sum(v) / len(v) for c, v in vals[name].items()}
    m["launches"] = max(len(v) for v in vals[name].values())

    def ratio(a, b):
        return round(m[a] / m[b], 2) if m.get(a) is not None and m.get(b) else None
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (a 0.156 ms frame reads 2.99 M): the kernel's own cycles are an eighth
    clk = m.get("GRBM_GUI_ACTIVE") / 8.0 if m.get("GRBM_GUI_ACTIVE") else None
    d = {"launches": m["launches"], "kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)": clk, "ms at 2.4 GHz": (round(clk / 2.4e6, 3) if clk else None),
         "L1->L2 read latency (cycles) = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ": ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
         "L2->fabric read latency (cycles) = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ": ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"),
         "VMEM latency as the SQ sees it (cycles) = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD": ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD"),
         "L2 reads outstanding at the fabric, chip-wide average = TCC_EA0_RDREQ_LEVEL / cycles": (round(m["TCC_EA0_RDREQ_LEVEL_sum"] / clk, 1) if m.get("TCC_EA0_RDREQ_LEVEL_sum") and clk else None),
         "L1->L2 reads outstanding per CU, average = TCP_TCC_READ_REQ_LATENCY / cycles / 256": (round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / clk / 256, 1) if m.get("TCP_TCC_READ_REQ_LATENCY_sum") and clk else None),
         "fabric read rate (TB/s) = 128 B x TCC_EA0_RDREQ / time at 2.4 GHz": (round(128 * m["TCC_EA0_RDREQ_sum"] / (clk / 2.4e9) / 1e12, 2) if m.get("TCC_EA0_RDREQ_sum") and clk else None),
         "fabric reads: DRAM share = RDREQ_DRAM / RDREQ": ratio("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_sum"),
         "fabric reads by size 32/64/128 B": [m.get("TCC_EA0_RDREQ_32B_sum"), m.get("TCC_EA0_RDREQ_64B_sum"), m.get("TCC_EA0_RDREQ_128B_sum")],
         "DRAM credit stall cycles per fabric read": ratio("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_sum"),
         "GMI credit stall cycles per fabric read": ratio("TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_RDREQ_sum"),
         "L1 pending-miss stall cycles / (256 CUs x cycles)": (round(m["TCP_PENDING_STALL_CYCLES_sum"] / (256 * clk), 3) if m.get("TCP_PENDING_STALL_CYCLES_sum") is not None and clk else None),
         "L1 stalled by the L2's return path / (256 CUs x cycles)": (round(m["TCP_TCR_TCP_STALL_CYCLES_sum"] / (256 * clk), 3) if m.get("TCP_TCR_TCP_STALL_CYCLES_sum") is not None and clk else None),
         "UTCL1 miss rate = MISS / REQUEST": ratio("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
         "UTCL1 requests per L1->L2 read": ratio("TCP_UTCL1_REQUEST_sum", "TCP_TCC_READ_REQ_sum"),
         "TA busy / (256 CUs x cycles)": (round(m["TA_TA_BUSY_sum"] / (256 * clk), 3) if m.get("TA_TA_BUSY_sum") and clk else None),
         "TA address stalled by TC / TA busy": ratio("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum"),
         "L2 tag stall / L2 busy": ratio("TCC_TAG_STALL_sum", "TCC_BUSY_sum"),
         "L2 hit rate": (round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3) if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") else None),
         "waves waiting on an instruction / wave cycles": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
         "raw": {c: round(v, 1) for c, v in sorted(m.items()) if c != "launches"}}
    out[name] = d
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_latency_counters.json"), "w"), indent=1)
for name, d in out.items():
    print(name)
    for k, v in d.items():
        if k != "raw":
            print("   ", k, "=", v)
