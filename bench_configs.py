"""bench_configs.py -- the workloads bench.py times besides its headline: the cfg-2 camera and its orbit, and BASELINE.json's other
single-GPU configurations timed in the same process (run_configs -> the line's `configs` block).  Split out of bench.py in round 6
without a change of behaviour."""
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))

from bench_report import HBM_PEAK_GBS, VALU_PEAK_SPEC_GINSTR, compulsory_fields, kernel_source_hash, load_compulsory, load_pmc
from bench_live_pmc import merged as merge_pmc
from bench_sustained import at_observed_clock, sustained_leg


def load_package(args):
    """sdfbox_amd against the product library, or -- for the A/B forms -- against the experiments flavour"""
    if args.lab or args.one_kernel or args.shadow_queue:
        if not args.lab:
            raise SystemExit("--one-kernel / --shadow-queue are A/B forms of the experiments build: add --lab")
        import sdfbox_amd.lab
        return sdfbox_amd.lab.load()
    import sdfbox_amd
    return sdfbox_amd


def orbit_cameras(sb, W, H, n, step_deg=1.0):
    """n cameras, step_deg (one degree) apart, on the horizontal circle through the cfg-2 camera position about the scene's
    centre, each turned by its angle (yaw convention of Matrix4x4.CreateFromYawPitchRoll: forward = (sin yaw, ., cos yaw))."""
    import math
    out = []
    r = 0.5 + 0.35
    for k in range(n):
        phi = math.radians(k * step_deg)
        c = sb.Logic(W, H)
        c.Position = (0.5 - r * math.sin(phi), 0.5, 0.5 - r * math.cos(phi))
        c.Heading = (-0.2, 0.35 + phi)
        out.append(c)
    return out


def bench_camera(sb, W, H):
    """SURVEY.md 8d cfg-2's camera"""
    cam = sb.Logic(W, H)
    cam.Position = (0.5, 0.5, -0.35)
    cam.Heading = (-0.2, 0.35)            # (X = pitch, Y = yaw), Logic.cs:53
    return cam


def grid_suffix(scene, pt=None):
    """The part of a PMC key that names the scene's grid: "grid9" dense, "grid8+blocks" split (SDFHIP_TOP_GRID_LEVEL / _SPLIT change it)"""
    lvl, gbytes = scene.top_grid_level, scene.top_grid_bytes
    return f":grid{lvl}" + ("+blocks" if lvl and gbytes > (16 << (3 * lvl)) and pt is None else "")


def run_configs(sb, torch, scene, scene_name, copy_gbs, scale=1, depth=9, shared_streams=None, sustained_4k_seconds=0.0, telemetry=None, live=None):
    """BASELINE.json's other single-GPU configurations, timed in this process behind the headline: the same
    clock (host time around `steps` frames, synchronised on both sides, frames in flight on their own streams), the HIP-event
    time of a launch beside it, and the counter fractions from the committed PMC pass of the SAME command line
    (profiles/hbm_traffic.json -> profiles/<tag>_pmc.json, formulas in profiles/README.md) when it was measured on this build's
    kernel sources -- else pmc_stale and no fraction.  live (bench.py --live-pmc all): a function that runs a configuration's command
    line as child processes under rocprofv3 --pmc (bench_live_pmc.live_pmc); the fractions then come from THIS run's counts, the
    committed pass beside them (`live_pmc`)."""
    out = {}
    suffix9 = grid_suffix(scene)                      # (before the path-traced mode adds its second grid to the byte count)

    def measure(name, sc, sname, W, H, mode, suffix, flags=0, pt=None, steps=60, warmup=12, nbuf=4, note=None, sustained=0.0, live_args=None):
        cam = bench_camera(sb, W, H)
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        # the headline's streams again: which hardware queue a stream gets is the runtime's business, and these are known to have
        # queues of their own (fresh streams for every configuration: the 4K frame took 0.328 ms where the same run alone takes 0.313)
        streams = (list(shared_streams[:nbuf]) if shared_streams and len(shared_streams) >= nbuf else []) or [torch.cuda.Stream() for _ in range(nbuf)]

        torch.cuda.synchronize()           # (the buffers' fills run on torch's stream, the renders on their own: not ordered by themselves)

        def launch(k):
            s = streams[k % nbuf].cuda_stream
            if pt is not None:
                sc.DrawPathDevice(cam, W, H, bufs[k % nbuf].data_ptr(), pt=pt, flags=flags, stream=s)
            else:
                sc.DrawDevice(cam, W, H, bufs[k % nbuf].data_ptr(), flags=flags, stream=s)
        for k in range(warmup):
            launch(k)
        torch.cuda.synchronize()
        ev = []
        t0 = time.perf_counter()
        for k in range(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[k % nbuf])
            launch(k)
            e1.record(streams[k % nbuf])
            ev.append((e0, e1))
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        spp = pt.spp if pt is not None else 1
        key = f"{W}x{H}:{sname}:{mode}{suffix}"
        pmc = load_pmc(key)
        live_note = None
        if live is not None and live_args is not None and scale == 1:
            torch.cuda.synchronize()
            try:
                got = live(live_args() if callable(live_args) else live_args)
            except Exception as ex:        # (the mesh scene's file could not be written, ...: the committed pass stands, the timing is kept)
                got = {"dropped": f"{type(ex).__name__}: {str(ex)[:200]}"}
            pmc, live_note = merge_pmc(got, pmc)
            live_note = {k: v for k, v in live_note.items() if k != "passes"}           # (the headline's entry keeps the per-pass log)
        stale = "dropped" in pmc
        e = {"workload": f"{W}x{H}, {sname}, " + (f"path trace {spp} spp, 3 diffuse bounces" if pt is not None else
                                                   "primary-ray sphere trace + shadow march") + (", " + note if note else ""),
             "ms_per_step": round(sec * 1e3, 4), "value": round(W * H * spp / sec / 1e6, 2), "unit": "Mray/s",
             "steps": steps, "warmup": warmup, "frames_in_flight": nbuf, "kernel_ms": round(kernel_ms, 4),
             "pmc_key": key, "pmc_stale": stale,
             "hbm_frac": None, "valu_frac_of_spec": None, "traffic": None, "valu_insts_per_frame": None,
             "profile": None if stale else pmc.get("profile"), "kernel_source_sha": kernel_source_hash()}
        if live_note is not None:
            e["live_pmc"] = live_note
        if stale:
            e["pmc_dropped"] = pmc["dropped"]
        else:
            e["traffic"] = int(pmc["hbm_bytes_per_frame"])
            e["hbm_frac"] = round(e["traffic"] / sec / 1e9 / HBM_PEAK_GBS, 4)
            best_gbs = copy_gbs.get("best_gbs") if isinstance(copy_gbs, dict) else copy_gbs
            e["hbm_frac_of_measured"] = round(e["traffic"] / sec / 1e9 / best_gbs, 4) if best_gbs else None
            if pmc.get("valu_insts_per_frame"):
                e["valu_insts_per_frame"] = int(pmc["valu_insts_per_frame"])
                e["valu_frac_of_spec"] = round(pmc["valu_insts_per_frame"] / sec / 1e9 / VALU_PEAK_SPEC_GINSTR, 4)
            e["limiting"] = "hbm" if (e["hbm_frac"] or 0) >= (e["valu_frac_of_spec"] or 0) else "valu"
            e.update(compulsory_fields(e["traffic"], load_compulsory(key)))
        if sustained > 0 and pt is None:
            # the same frames without a stop, the GPU's clocks sampled (bench_sustained.py)
            try:
                r = sustained_leg(torch, lambda k, si: launch(k), streams, sustained, telemetry)
                clk = ((r.get("telemetry") or {}).get("sclk_mhz") or {}).get("mean")
                r["valu_frac_of_spec_2400mhz"] = (round(e["valu_insts_per_frame"] / (r["ms_per_step"] * 1e-3) / 1e9 / VALU_PEAK_SPEC_GINSTR, 4)
                                                  if e.get("valu_insts_per_frame") else None)
                r["valu_frac_at_observed_clock"] = at_observed_clock(e.get("valu_insts_per_frame"), r["ms_per_step"], clk)
                r["mray_per_s"] = round(W * H * spp / (r["ms_per_step"] * 1e-3) / 1e6, 1)
                e["sustained"] = r
            except Exception as ex:
                e["sustained"] = {"error": f"{type(ex).__name__}: {ex}"}
        out[name] = e
        del bufs, streams
        torch.cuda.empty_cache()

    W4, H4, W2, H2 = 3840 // scale, 2160 // scale, 1920 // scale, 1080 // scale

    def guarded(name, fn):                 # a configuration that fails is reported as such; the others are still measured
        try:
            fn()
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    guarded("cfg3_4k", lambda: measure("cfg3_4k", scene, scene_name, W4, H4, "default", suffix9,
            note="BASELINE cfg-3's frame with compaction OFF (the default kernel: faster than every form of compaction built)",
            sustained=sustained_4k_seconds, live_args=["--size", f"{W4}x{H4}"]))
    guarded("cfg3_4k_compact", lambda: measure("cfg3_4k_compact", scene, scene_name, W4, H4, "compact", suffix9, flags=sb.FLAG_COMPACT,
            note="BASELINE cfg-3 as named: wavefront ray compaction ON (SDFHIP_FLAG_COMPACT: the shadow rays of waves that hold fewer than 32 "
                 "compacted by ballot / prefix into a queue and marched 64 to a wave by a second kernel)", live_args=["--size", f"{W4}x{H4}", "--compact", "1"]))

    def cfg5():
        pt = sb.PathTrace(spp=16)
        sb._lib.check(sb._lib.lib.sdfhip_scene_prepare_path(scene._h))       # the bounce levels' grid, at load time
        measure("cfg5_4k_spp16", scene, scene_name, W4, H4, "spp16", grid_suffix(scene, pt), pt=pt, steps=9, warmup=3, nbuf=3,
                note="BASELINE cfg-5 on one GPU", live_args=["--size", f"{W4}x{H4}", "--spp", "16"])
    guarded("cfg5_4k_spp16", cfg5)

    def depth10():                         # cfg-2 at the reference application's default depth (Model.MaxDepth = 10, SdfBox/Model.cs:18)
        t0 = time.time()
        od10 = sb.dragon_standin(depth + 1, nthreads=max(1, min(32, os.cpu_count() or 1)))
        t_gen = time.time() - t0
        with sb.Scene(od10, device=scene.device) as sc10:
            measure("cfg2_depth10", sc10, f"dragon_standin_d{depth + 1}", W2, H2, "default", grid_suffix(sc10),
                    note=f"N={od10.Length} nodes, {od10.nbytes / 1e6:.0f} MB, built in {t_gen:.1f} s", live_args=["--depth", str(depth + 1)])
    guarded("cfg2_depth10", depth10)
    # ... and on a MESH-derived scene at that depth: the reference's import flow (Program.cs:613-650: .ply -> SdfGen(depth 10) -> upload)
    # on a 1 M-point cloud (a torus-knot tube: no mesh ships with the reference), the tree built by the GPU builder and never
    # leaving HBM (sdfhip_sdfgen_scene)
    def mesh():
        pts = sb.knot_point_cloud(1_000_000)
        sb.Scene.FromPoints(pts[:2000], 3).close()                       # (module load)
        t0 = time.time()
        scm, stg = sb.Scene.FromPoints(pts, depth + 1, device=scene.device, want_stats=True)
        t_build = time.time() - t0
        tmpd = []

        def mesh_file():                    # the children load the same tree from a file (the builder's bytes are the oracle's: tests/test_sdfgen.py);
            tmpd.append(tempfile.mkdtemp(prefix="sdfhip_live_mesh_", dir="/tmp"))      # written AFTER the timed frames: the GPU idles meanwhile
            asdf = os.path.join(tmpd[0], f"knot_d{depth + 1}.asdf")
            sb.OctData.SdfGen(pts, depth + 1).Save(asdf)
            return ["--asdf", asdf]
        try:
            with scm:
                measure("cfg2_mesh_knot_d10", scm, f"knot_d{depth + 1}.asdf", W2, H2, "default", grid_suffix(scm),
                        note=f"1 M-point cloud -> sdfhip_sdfgen_scene: N={scm.Length} nodes in {t_build * 1e3:.0f} ms wall ({stg.total_ms:.0f} ms in the library)",
                        live_args=mesh_file)
        finally:
            for d in tmpd:
                shutil.rmtree(d, ignore_errors=True)
    if scale == 1:
        guarded("cfg2_mesh_knot_d10", mesh)
    return out
