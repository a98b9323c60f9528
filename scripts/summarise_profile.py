"""gpurun_out/prof_<tag>/ -> profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json: per-launch means
of every PMC counter for each ray-march kernel of the run (k_march, k_shadow, k_plain, k_compact, k_path; the
one counting launch of a bench run is left out), the bench line of the stats pass, and the HBM traffic per
frame that bench.py reports as roofline.traffic (profiles/hbm_traffic.json, keyed by workload, with the
profile tag and the hash of the kernel sources it was measured on).

HBM bytes (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KB, collected in separate passes;
WRITE_SIZE is exact for 16-B-per-lane stores; FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports half the bytes of
wide reads on gfx950 and is uncalibrated for other shapes: the doubled figure is reported (an upper
estimate for our 12-byte gathers) and the raw one is kept beside it.

usage: python scripts/summarise_profile.py <tag> [workload-key]"""
import collections, csv, glob, hashlib, json, os, re, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    sys.path.insert(0, REPO)
    import bench                       # one definition of "the kernel sources": bench.KERNEL_SOURCES
    return bench.kernel_source_hash()


def short(name):
    m = re.match(r"(?:void )?(?:sdfhip::)?(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    tag = sys.argv[1]
    src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    # gpurun merges a call's files into gpurun_out/ and keeps those of earlier calls (rocprofv3 names them by process id): of every
    # pass take the NEWEST file only -- the others belong to earlier builds
    def newest(pattern):
        files = glob.glob(pattern)
        return max(files, key=os.path.getmtime) if files else None
    f = newest(f"{src}/stats/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))
    bench = None
    try:
        bench = json.loads([l for l in open(f"{src}/stats_bench.json") if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        print("no bench line:", e)
    kernels = collections.defaultdict(dict)
    per_frame = collections.defaultdict(float)        # counter -> sum over the frame's kernels, per frame
    FIRST = ("k_march", "k_pt_primary", "k_plain", "k_compact", "k_path")   # one launch of these per frame (or per batch of frames)
    for f in [g for g in (newest(f"{d}/*/*_counter_collection.csv") for d in sorted(glob.glob(f"{src}/pmc_*")) if os.path.isdir(d)) if g]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k or not k.startswith(("k_march", "k_shade", "k_shadow", "k_plain", "k_compact", "k_path", "k_pt_")):
                continue
            if re.search(r"<\d+, true", k) or k.endswith("<true>"):     # the one counting launch
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernels[k]["VGPR_Count"] = int(r["VGPR_Count"]); kernels[k]["SGPR_Count"] = int(r["SGPR_Count"])
            kernels[k]["LDS_Block_Size"] = int(r["LDS_Block_Size"])
        counters = {c for cs in agg.values() for c in cs}
        for c in counters:
            frames = sum(len(cs[c]) for k, cs in agg.items() if k.startswith(FIRST) and c in cs)
            if frames:
                per_frame[c] = sum(sum(cs[c]) for cs in agg.values() if c in cs) / frames
        for k, cs in agg.items():
            for c, v in cs.items():
                kernels[k][c] = sum(v) / len(v)          # mean per launch of this kernel
                kernels[k]["launches_" + c] = len(v)
    out = {"tag": tag, "kernel_source_sha": kernel_source_hash(), "kernels": kernels, "per_frame": dict(per_frame)}
    rd = per_frame.get("FETCH_SIZE", 0.0) * 1024
    wr = per_frame.get("WRITE_SIZE", 0.0) * 1024
    if rd or wr:
        # per frame = the frame's whole pipeline (k_march + k_shadow; k_pt_primary + every k_pt_bounce level + k_pt_resolve; ...)
        fpl = float(bench["roofline"].get("frames_per_launch", 1.0)) if bench else 1.0
        out["hbm_read_bytes_raw_per_frame"] = rd / fpl
        out["hbm_read_bytes_x2_per_frame"] = 2 * rd / fpl
        out["hbm_write_bytes_per_frame"] = wr / fpl
        out["hbm_bytes_per_frame"] = (2 * rd + wr) / fpl
    if bench:
        out["bench_line"] = bench
    json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in out.items() if k != "bench_line"}, indent=1, sort_keys=True))
    if "hbm_bytes_per_frame" in out and len(sys.argv) > 2:
        p = os.path.join(REPO, "profiles", "hbm_traffic.json")
        t = json.load(open(p)) if os.path.exists(p) else {}
        fpl = float(bench["roofline"].get("frames_per_launch", 1.0)) if bench else 1.0
        valu = per_frame.get("SQ_INSTS_VALU", 0.0) / fpl
        salu = per_frame.get("SQ_INSTS_SALU", 0.0) / fpl
        busy = per_frame.get("SQ_ACTIVE_INST_VALU", 0.0) / fpl      # quad-cycles in which a SIMD's VALU was executing (x 4 = cycles)
        t[sys.argv[2]] = {"hbm_bytes_per_frame": int(out["hbm_bytes_per_frame"]), "read_x2": int(out["hbm_read_bytes_x2_per_frame"]),
                          "write": int(out["hbm_write_bytes_per_frame"]), "valu_insts_per_frame": int(valu), "salu_insts_per_frame": int(salu),
                          "valu_active_quad_cycles_per_frame": int(busy),
                          "profile": f"profiles/{tag}_pmc.json", "kernel_source_sha": out["kernel_source_sha"]}
        json.dump(t, open(p, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
