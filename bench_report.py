"""bench_report.py -- what bench.py's line says ABOUT a measurement: the roofline object from the rocprofv3 counter passes (the committed
ones: load_pmc, load_compulsory; the headline's own live ones: bench_live_pmc.py; roofline), the line's last key (configs_summary) and the box's own HBM rate (measured_hbm_bandwidth).
The CPU baseline -- the one leg that may touch oracle/ -- stays in bench.py itself.  No GPU work is timed here
except the bandwidth kernels of the library.  Split out of bench.py in round 6 without a change of behaviour."""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))


# the kernel sources of the frame's pipeline (device code only: host-side edits do not change what the counters measured)
KERNEL_SOURCES = ("raymarch_device.h", "raymarch_kernels.h", "upload_kernels.h", "tile_order_kernels.h",
                  #  what else decides the counters: COMPACT_MIN_LANES and march_grid (scene.h), the launch shapes, hit_cap
                  # and shade_grid (render.hip), the sparse-share kernels (gather_kernels.h)
                  "scene.h", "render.hip", "gather_kernels.h")


def kernel_source_hash():
    """What the PMC figures in profiles/hbm_traffic.json were measured on (scripts/summarise_profile.py)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(REPO, "sdfbox_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc(key):
    """The COMMITTED rocprofv3 PMC figures of this workload (HBM bytes, issued VALU / SALU wave instructions and VALU-busy quad-cycles per
    frame), or a dict {"dropped": reason}.  scripts/profile.sh collects the separate --pmc passes of this very command and
    scripts/summarise_profile.py writes profiles/hbm_traffic.json together with the hash of the kernel sources they were measured
    on; figures of another build are not reported, and the line says so.  (The headline's command also measures the same counters
    in its own run, through child processes: bench_live_pmc.py; this record then stands beside the live one.)"""
    try:
        with open(os.path.join(REPO, "profiles", "hbm_traffic.json")) as f:
            e = json.load(f).get(key)
    except (OSError, ValueError):
        return {"dropped": "profiles/hbm_traffic.json is missing or unreadable"}
    if not isinstance(e, dict):
        return {"dropped": f"no PMC pass of workload '{key}' under profiles/ (scripts/profile.sh)"}
    here = kernel_source_hash()
    if e.get("kernel_source_sha") != here:
        return {"dropped": f"the PMC pass of '{key}' ({e.get('profile')}) was measured on kernel sources {e.get('kernel_source_sha')}; "
                           f"this build is {here}: figures of another build are not reported"}
    return e


def load_compulsory(key):
    """The compulsory bytes of this design for workload `key` (profiles/compulsory_bytes.json, written by
    scripts/compulsory_bytes.py from the laboratory library's line-touch bitmaps: every distinct 128-byte line of the lookup grid
    that the frame's find() touches, once, + the frame it stores + -- path-traced mode -- the queues and per-path results between
    its launches), or {"dropped": reason}.  Like the PMC figures it belongs to a build: other kernel sources, no figure."""
    try:
        with open(os.path.join(REPO, "profiles", "compulsory_bytes.json")) as f:
            e = json.load(f).get(key)
    except (OSError, ValueError):
        return {"dropped": "profiles/compulsory_bytes.json is missing or unreadable"}
    if not isinstance(e, dict):
        return {"dropped": f"no line count of workload '{key}' under profiles/ (scripts/compulsory_bytes.py)"}
    here = kernel_source_hash()
    if e.get("kernel_source_sha") != here:
        return {"dropped": f"the line count of '{key}' was taken on kernel sources {e.get('kernel_source_sha')}; this build is {here}"}
    return e


def compulsory_fields(traffic, comp):
    """-> the keys a roofline object / a `configs` entry carries about the compulsory bytes: what the frame MUST move (chip-wide: one
    ideal cache in front of HBM; per XCD: eight ideal L2s that share nothing, the bound of the L2s' fabric-side counters given which
    XCD renders what) and the counter traffic over each."""
    if not isinstance(comp, dict) or "dropped" in comp or not comp.get("compulsory_bytes"):
        return {"compulsory_bytes": None, "traffic_over_compulsory": None, "compulsory_bytes_per_xcd": None, "traffic_over_compulsory_per_xcd": None,
                "compulsory_dropped": (comp or {}).get("dropped") if isinstance(comp, dict) else "no line count"}
    c, cx = int(comp["compulsory_bytes"]), int(comp["compulsory_bytes_per_xcd"])
    return {"compulsory_bytes": c, "traffic_over_compulsory": round(traffic / c, 3) if traffic else None,
            "compulsory_bytes_per_xcd": cx, "traffic_over_compulsory_per_xcd": round(traffic / cx, 3) if traffic else None,
            "compulsory_lines": comp.get("distinct_lines"), "compulsory_lines_summed_over_xcds": comp.get("distinct_lines_summed_over_xcds")}


# VALU issue ceilings, in wave64 instructions per second chip-wide:
#   spec      256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles: a SIMD retires 32 lanes per clock (the 157.3 TFLOP/s fp32 vector figure
#             = 1024 SIMDs x 2.4 GHz x 32 lanes x 2 flops), so a wave64 instruction takes two
#   measured  / 2.35 cycles: the cheapest instruction on this chip with 8 waves per SIMD (v_mov_b32; v_and / v_add / v_sub /
#             v_mul / v_fmac 2.4-2.6; shifts, compares, conversions, v_fma_f32 (VOP3), min3 / med3 4.0-4.4; a packed fp32
#             instruction 4.4-4.7 for two results: scripts/micro/valu_mix.hip, profiles/r03_micro_valu_mix.txt in the history, commit 53ee955)
# Both bound ANY instruction mix from above; `valu_busy` below is the measured utilisation.
VALU_PEAK_SPEC_GINSTR = 256 * 4 * 2.4 / 2.0
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.35
HBM_PEAK_GBS = 8000.0                    # HBM3E spec (MI355X_MICROARCH.md)
N_SIMD, CLOCK_GHZ = 256 * 4, 2.4


def roofline(sec_per_frame, own_bytes, ref_bytes, pmc, measured, compulsory=None, latency_ms=None, orbit=None):
    """The frame against its two roofs, both from the rocprofv3 counters of THIS build and workload (profiles/hbm_traffic.json;
    the PMC passes serialise launches -- one frame in flight while they count -- which changes times, not counts), over the
    steady-state time per frame (the driver-verifiable ms_per_step; with frames in flight the launch durations overlap):
      frac = hbm_frac   HBM bytes per frame (2 x FETCH_SIZE + WRITE_SIZE, separate passes, the guide's gfx950 correction) / time
                        / 8 TB/s.  THIS is the fraction SURVEY.md 8d and BASELINE's "% of HBM roofline" ask for, so it is the
                        object's `frac`, with bound = "hbm", achieved / peak in GB/s and `traffic` the counter bytes.
                        hbm_frac_of_measured: the same over `measured` -- the box's own streaming rate (the library's float4
                        copy / triad, sdfhip_device_bandwidth), SURVEY 8d's "measured device bandwidth" denominator.
      valu_frac_of_spec issued VALU wave instructions per frame (SQ_INSTS_VALU) / time against the chip's SPEC issue rate, 1 228.8 G
                        wave64 instructions per second (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles); valu_frac_of_measured_ceiling:
                        against the cheapest instruction as measured here (2.35 cycles, scripts/micro/valu_mix.hip).  The PEER of
                        hbm_frac, not a substitute: `limiting` names the larger of the two -- "valu" on the primary-ray frames
                        (the march is instruction-issue bound: its lookups mostly hit L1 / L2), "hbm" on cfg-5 and the depth-10 scene.
      valu_busy         SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles: the instruction count priced at 4 cycles each, NOT a utilisation.
    pmc_stale = true: the committed PMC pass was measured on other kernel sources (or there is none): no fraction, only `demand`.
    `demand`: the bytes this kernel's own algorithm asks of the MEMORY SYSTEM per frame (16 B per cell a LANE loads + the pixel
    store, by the counting build) and the bytes the REFERENCE algorithm would read for the same pixels (8 B per node
    visit of find(), Compute.hlsl:88-108, 8 B per sample, the store), each over 8 TB/s x time.  Both exceed 1 on the bench
    frames: they are ratios, not fractions of a roof -- a lookup grid built at upload replaces the descent (1.1 loads per step
    instead of 8.2), and L1 / L2 serve most of the kernel's own loads."""
    dropped = pmc.get("dropped") if isinstance(pmc, dict) else None
    if dropped:
        pmc = None
    best_gbs = measured.get("best_gbs") if isinstance(measured, dict) else measured
    traffic = hbm = valu = valu_busy = None
    if pmc:
        traffic = int(pmc["hbm_bytes_per_frame"])
        ach = traffic / sec_per_frame / 1e9
        hbm = {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "per_frame": traffic, "frac": round(ach / HBM_PEAK_GBS, 4),
               "frac_of_measured": round(ach / best_gbs, 4) if best_gbs else None}
        if pmc.get("valu_insts_per_frame"):
            ach = pmc["valu_insts_per_frame"] / sec_per_frame / 1e9
            valu = {"achieved": round(ach, 1), "peak": round(VALU_PEAK_SPEC_GINSTR, 1),
                    "peak_is": "spec: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction",
                    "measured_ceiling": round(VALU_PEAK_GINSTR, 1),
                    "measured_ceiling_is": "the cheapest VALU instruction on this chip, 2.35 cycles (scripts/micro/valu_mix.hip)",
                    "frac": round(ach / VALU_PEAK_SPEC_GINSTR, 4), "frac_of_measured_ceiling": round(ach / VALU_PEAK_GINSTR, 4),
                    "unit": "G wave-instr/s", "per_frame": int(pmc["valu_insts_per_frame"])}
        if pmc.get("valu_active_quad_cycles_per_frame"):
            valu_busy = round(pmc["valu_active_quad_cycles_per_frame"] * 4.0 / (N_SIMD * CLOCK_GHZ * 1e9 * sec_per_frame), 4)
    cands = {k: v for k, v in (("hbm-traffic", hbm), ("valu", valu)) if v}
    limiting = max(cands, key=lambda k: cands[k]["frac"]) if cands else None
    own_over, ref_over = own_bytes / sec_per_frame / 1e9 / HBM_PEAK_GBS, ref_bytes / sec_per_frame / 1e9 / HBM_PEAK_GBS
    extra = compulsory_fields(traffic, compulsory)
    # one frame at a time (what the reference's loop does): the same counter bytes over the LATENCY of a frame alone on the chip --
    # the PMC passes serialise launches, so this is the condition the bytes were counted under
    extra["hbm_frac_latency"] = round(traffic / (latency_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and latency_ms) else None
    extra["valu_frac_of_spec_latency"] = (round(pmc["valu_insts_per_frame"] / (latency_ms * 1e-3) / 1e9 / VALU_PEAK_SPEC_GINSTR, 4)
                                          if (pmc and pmc.get("valu_insts_per_frame") and latency_ms) else None)
    extra["pmc_serialises_launches"] = ("rocprofv3 --pmc runs one launch at a time: the counts are a frame's, whatever is in flight in the timed "
                                        "run; hbm_frac divides them by the pipelined time per frame, hbm_frac_latency by one frame's own time")
    # the same passes with a camera that moves every frame (bench.py --orbit 90; profiles/*_orbit_pmc.json), over that run's time
    if isinstance(orbit, dict) and orbit.get("pmc") and "dropped" not in orbit["pmc"] and orbit.get("ms_per_step"):
        op, osec = orbit["pmc"], orbit["ms_per_step"] * 1e-3
        extra["orbit"] = {"cameras": orbit.get("cameras"), "ms_per_step": orbit["ms_per_step"], "traffic": int(op["hbm_bytes_per_frame"]),
                          "hbm_frac": round(op["hbm_bytes_per_frame"] / osec / 1e9 / HBM_PEAK_GBS, 4),
                          "valu_insts_per_frame": int(op.get("valu_insts_per_frame") or 0),
                          "valu_frac_of_spec": (round(op["valu_insts_per_frame"] / osec / 1e9 / VALU_PEAK_SPEC_GINSTR, 4)
                                                if op.get("valu_insts_per_frame") else None),
                          "profile": op.get("profile")}
        extra["hbm_frac_orbit"] = extra["orbit"]["hbm_frac"]
        extra["valu_frac_of_spec_orbit"] = extra["orbit"]["valu_frac_of_spec"]
    elif isinstance(orbit, dict):
        extra["orbit"] = {"dropped": (orbit.get("pmc") or {}).get("dropped", "no orbit pass")}
    return {**{
        # the contract's object: the HBM roof, which is what SURVEY 8d / BASELINE ask the fraction of
        "bound": "hbm" if hbm else None,
        "achieved": hbm["achieved"] if hbm else None, "peak": hbm["peak"] if hbm else None, "unit": hbm["unit"] if hbm else None,
        "frac": hbm["frac"] if hbm else None,
        "frac_is": "hbm_frac: counter HBM bytes / time / 8 TB/s -- the fraction SURVEY.md 8d asks for; valu_frac_of_spec is its peer, "
                   "`limiting` the larger of the two",
        "traffic": traffic,
        "pmc_stale": not cands,
        "hbm_frac": hbm["frac"] if hbm else None,
        "hbm_frac_of_measured": hbm["frac_of_measured"] if hbm else None,
        "valu_frac_of_spec": valu["frac"] if valu else None,
        "valu_frac_of_measured_ceiling": valu["frac_of_measured_ceiling"] if valu else None,
        "limiting": None if limiting is None else ("hbm" if limiting.startswith("hbm") else "valu"),
        "limiting_is": "the larger of the two fractions -- a ranking, not a proof of the bound (cfg-5 ranks 'hbm' at 0.9 of the measured rate and is "
                       "bound by the latency of its dependent lookups: a third fewer bytes bought no time, profiles/r06_cfg5_xcd_order_ab.txt)",
        "valu_busy": valu_busy,
        "valu_busy_is": "SQ_ACTIVE_INST_VALU x 4 cycles over the SIMD-cycles of the frame: every instruction is charged one quad-cycle, "
                        "so this is the instruction count at 4 cycles each, not a measured utilisation",
        "traffic_source": ({"profile": pmc.get("profile"), "kernel_source_sha": pmc.get("kernel_source_sha"),
                            "frames_in_flight_while_counting": 1} if pmc else
                           (dropped or "no PMC pass of this build and workload under profiles/ (scripts/profile.sh)")),
        "candidates": cands,
        "demand": {"own_bytes_per_frame": int(own_bytes), "own_demand_over_hbm_peak": round(own_over, 3),
                   "reference_bytes_per_frame": int(ref_bytes), "reference_demand_over_hbm_peak": round(ref_over, 3),
                   "note": "requests to the memory system over 8 TB/s x time, NOT roofline fractions: the lookup grid built at upload replaces "
                           "the reference's descent and L1 / L2 serve most of the kernel's own loads (see roofline() in bench.py)"},
        "measured_hbm_gbs": measured,
    }, **extra}


def configs_summary(out, cfgs):
    """<= 600 characters that carry every configuration's time, rays and fractions (+ <= 380 for the sustained legs): the LAST key of
    the line, so that the last 2 000 characters of it (what the driver's record keeps) hold all of them.  ms per frame / Mray/s / hbm_frac (of 8 TB/s) /
    hbm_frac_of_measured / valu_frac_of_spec; '-' where there is no PMC pass of this build."""
    names = {"cfg3_4k": "cfg3", "cfg3_4k_compact": "cfg3c", "cfg5_4k_spp16": "cfg5", "cfg2_depth10": "d10", "cfg2_mesh_knot_d10": "mesh"}

    def f(v, nd):
        return "-" if v is None else f"{v:.{nd}f}"

    def one(tag, e, r):
        if "error" in e:
            return f"{tag} ERR {str(e['error'])[:40]}"
        return f"{tag} {f(e.get('ms_per_step'), 4)}/{f(e.get('value'), 0)}/{f(r.get('hbm_frac'), 2)}/{f(r.get('hbm_frac_of_measured'), 2)}/{f(r.get('valu_frac_of_spec'), 2)}"
    parts = [one("cfg2", out, out.get("roofline") or {})]
    if isinstance(cfgs, dict):
        if "error" in cfgs and not any(k in cfgs for k in names):
            parts.append("configs ERR " + str(cfgs["error"])[:60])
        for k, e in cfgs.items():
            if isinstance(e, dict) and k != "error":
                parts.append(one(names.get(k, k[:10]), e, e))
    text = ("ms/Mray/hbm/hbm_meas/valu: " + "; ".join(parts))[:600]
    # continuous operation (bench_sustained.py): ms per frame over the leg / its seconds / frames, first 20 and last 1000 frames,
    # the shader clock (min-mean MHz) and the mean socket power while it ran, VALU issue at the observed clock -- <= 360 characters more
    def sus(tag, r):
        if not isinstance(r, dict) or "ms_per_step" not in r:
            return None
        t = r.get("telemetry") or {}
        clk, pw = t.get("sclk_mhz") or {}, t.get("power_w") or {}
        return (f"{tag} {f(r['ms_per_step'], 4)}ms/{f(r.get('seconds'), 1)}s/{r.get('frames')}f first20 {f(r.get('ms_per_step_first_20'), 4)} "
                f"last1000 {f(r.get('ms_per_step_last_1000'), 4)} sclk {f(clk.get('min'), 0)}-{f(clk.get('mean'), 0)}MHz {f(pw.get('mean'), 0)}W "
                f"valu@clk {f(r.get('valu_frac_at_observed_clock'), 2)}")
    legs = []
    for tag, key in (("cfg2", "cfg2"), ("orbit", "cfg2_orbit")):
        legs.append(sus(tag, (out.get("sustained") or {}).get(key)))
    if isinstance(cfgs, dict) and isinstance(cfgs.get("cfg3_4k"), dict):
        legs.append(sus("cfg3", cfgs["cfg3_4k"].get("sustained")))
    legs = [l for l in legs if l]
    if legs:
        text += " | sustained: " + "; ".join(legs)
    # traffic over the compulsory bytes of this design (scripts/compulsory_bytes.py): one ideal cache / eight unshared ideal L2s
    ratios = []
    r0 = out.get("roofline") or {}
    if r0.get("traffic_over_compulsory"):
        ratios.append(f"cfg2 {f(r0['traffic_over_compulsory'], 2)}/{f(r0.get('traffic_over_compulsory_per_xcd'), 2)}")
    if isinstance(cfgs, dict):
        for k, e in cfgs.items():
            if isinstance(e, dict) and e.get("traffic_over_compulsory"):
                ratios.append(f"{names.get(k, k[:10])} {f(e['traffic_over_compulsory'], 2)}/{f(e.get('traffic_over_compulsory_per_xcd'), 2)}")
    if ratios:
        text += " | traffic/compulsory (chip/per-XCD): " + "; ".join(ratios)
    # the headline's counters measured in this run (bench_live_pmc.py) against the committed pass of the same build
    lp = r0.get("live_pmc")
    if isinstance(lp, dict):
        if lp.get("used") == "live":
            text += (f" | live pmc (this run, {f(lp.get('seconds'), 0)}s): cfg2 {f((lp.get('hbm_bytes_per_frame') or 0) / 1e6, 1)}MB "
                     f"{f((lp.get('valu_insts_per_frame') or 0) / 1e6, 2)}M valu = {f(lp.get('live_over_committed'), 4)} x committed bytes")
            # --live-pmc all: the moving-camera pass and the `configs` rows counted in this run too -- live bytes over committed bytes
            more = []
            if isinstance(lp.get("orbit"), dict):
                more.append("orbit " + (f(lp["orbit"].get("live_over_committed"), 4) if lp["orbit"].get("used") == "live" else "dropped"))
            for k, e in (cfgs.items() if isinstance(cfgs, dict) else ()):
                if isinstance(e, dict) and isinstance(e.get("live_pmc"), dict):
                    more.append(f"{names.get(k, k[:10])} " + (f(e["live_pmc"].get("live_over_committed"), 4) if e["live_pmc"].get("used") == "live" else "dropped"))
            if more:
                text += "; " + "; ".join(more)
        else:
            text += " | live pmc dropped: " + str(lp.get("dropped"))[:80]
    return text[:1500]


def measured_hbm_bandwidth(sb, device=0, nbytes=2 << 30, reps=10):
    """SURVEY.md 8d's denominator: what this box's HBM delivers to a streaming kernel -- the library's own float4 copy, STREAM
    triad and read-only sum over 2 GiB arrays (sdfhip_device_bandwidth, csrc/bandwidth.hip; the kernel shapes chosen by
    scripts/micro/bw_variants.hip), bytes read + written per second in GB/s; `best_gbs` = the largest of the three, the roof no
    kernel of this repository beats.  -> dict or None.  (Until round 5 this was a torch `copy_` of 1 GiB, which reaches
    4.8-5.3 TB/s -- below what the frame's own kernels sustain on cfg-5, so fractions of it exceeded 1.)"""
    try:
        c, t, r = sb.device_bandwidth(device, nbytes, reps)
    except Exception as e:                       # (out of memory beside a large scene: the line goes on without the figure)
        print(f"[bench] measured_hbm_bandwidth: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    return {"copy_gbs": round(c, 1), "triad_gbs": round(t, 1), "read_gbs": round(r, 1), "best_gbs": round(max(c, t, r), 1), "array_bytes": nbytes,
            "reps": reps,
            "is": "sdfhip_device_bandwidth: float4 copy (2 x array bytes moved), triad a = b + s c (3 x) and a read-only sum (1 x) over arrays far "
                  "larger than the 256 MiB Infinity Cache, HIP-event time; best_gbs = the largest of the three"}
