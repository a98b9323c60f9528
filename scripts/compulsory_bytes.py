"""The compulsory bytes of THIS design, per configuration: what a frame MUST move through HBM, to set
beside what the counters say it DID move (profiles/hbm_traffic.json).

SURVEY 8d's algorithmic bytes (8 B per node find() reads + 8 B per sample + the pixel) are the REFERENCE's loads
(Compute.hlsl:88-108); this design does not perform them -- find() is a lookup in a grid built at upload.  Its own compulsory
traffic is: every distinct 128-byte line of the grid that the frame's lookups touch, fetched once (the laboratory library's
counting kernels mark them in a bitmap per grid array and XCD: sdfhip_debug_touch_begin / _end), + the frame it stores; for the
path-traced mode, per kernel launch of its pipeline (camera segments, each bounce level), + the bytes of the hit queues and the
per-path results that one launch writes and the next reads.  Two denominators:

    compulsory_bytes           lines distinct over the whole chip x 128: one ideal cache in front of HBM
    compulsory_bytes_per_xcd   lines distinct per XCD, summed x 128: eight ideal L2s that share nothing -- what the L2s' fabric-side
                               request counters (FETCH_SIZE) could at best show, given which XCD renders which tile

Writes profiles/<round>_compulsory_bytes.json keyed like profiles/hbm_traffic.json (bench.py reads both and prints
roofline.compulsory_bytes / traffic_over_compulsory).  GPU:  gpurun -- python scripts/compulsory_bytes.py r06 [--only cfg2,...]; then here: python scripts/compulsory_bytes.py r06 --install
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def install(tag):
    """here, after the GPU run: gpurun_out/<tag>_compulsory_bytes.json -> profiles/<tag>_compulsory_bytes.json and the entries of
    profiles/compulsory_bytes.json (what bench.py reads: like hbm_traffic.json, every entry names the build it belongs to)"""
    got = json.load(open(os.path.join(REPO, "gpurun_out", f"{tag}_compulsory_bytes.json")))
    with open(os.path.join(REPO, "profiles", f"{tag}_compulsory_bytes.json"), "w") as f:
        json.dump(got, f, indent=1, sort_keys=True)
        f.write("\n")
    cur_path = os.path.join(REPO, "profiles", "compulsory_bytes.json")
    cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
    for k, e in got.items():
        cur[k] = dict(e, profile=f"profiles/{tag}_compulsory_bytes.json")
    with open(cur_path, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
        f.write("\n")
    print("installed", len(got), "entries of", tag)


if "--install" in sys.argv:
    install(sys.argv[1])
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
import sdfbox_amd.lab  # noqa: E402

sb = sdfbox_amd.lab.load()
LINE = 128
PT_RECORD_BYTES = 3 * 16            # raymarch_kernels.h: PT_RECORDS x 16 bytes per queue entry


def count(scene, W, H, pt=None, flags=0):
    """one counting render between touch_begin and touch_end -> (touch dict, Stats)"""
    cam = bench.bench_camera(sb, W, H)
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    st = sb.Stats()
    s = torch.cuda.current_stream().cuda_stream
    scene.touch_begin()
    if pt is not None:
        scene.DrawPathDevice(cam, W, H, buf.data_ptr(), pt=pt, flags=flags | sb.FLAG_COUNT, stream=s, stats=st)
    else:
        scene.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags | sb.FLAG_COUNT, stream=s, stats=st)
    torch.cuda.synchronize()
    t = scene.touch_end()
    del buf
    return t, st


def entry(key, scene, W, H, pt=None, flags=0, note=None):
    t0 = time.time()
    t, st = count(scene, W, H, pt, flags)
    frame_bytes = 16 * W * H
    lines = sum(p["coarse_lines"] + p["fine_lines"] for p in t["phases"])
    lines_xcd = sum(p["coarse_lines_xcd_sum"] + p["fine_lines_xcd_sum"] for p in t["phases"])
    stream_bytes, stream_is = 0, None
    if pt is not None:
        # the pipeline's own streams, each byte written by one launch and read by a later one (useful bytes, not lines):
        # E queue entries of 48 B (written + read); per path: the camera kernel stores its throughput and step count (8 B), a bounce
        # entry reads the step count and stores the light it received and the new count (12 B) and, when it escapes, the throughput
        # (counted for every entry: 4 B); the ordered sum reads count + throughput (8 B per path) and one light term per entry (4 B)
        E, npaths = st.n_hits, W * H * pt.spp
        if E == 0:
            raise SystemExit("the counting build reports no queue entries (sdfhip_stats.n_hits): a library older than round 6's")
        stream_bytes = 2 * PT_RECORD_BYTES * E + 8 * npaths + 16 * E + 8 * npaths + 4 * E
        stream_is = (f"{E} queue entries x (48 B written + 48 B read) + per-path results ({npaths} paths: 16 B each + 20 B per entry), "
                     "byte-granular (what is useful, not what a 128-byte line costs)")
    e = {"frame": f"{W}x{H}", "phases": t["phases"], "grid_array_bytes": t["array_bytes"],
         "distinct_lines": lines, "distinct_lines_summed_over_xcds": lines_xcd,
         "frame_store_bytes": frame_bytes, "pipeline_stream_bytes": stream_bytes, "pipeline_stream_bytes_is": stream_is,
         "compulsory_bytes": lines * LINE + frame_bytes + stream_bytes,
         "compulsory_bytes_per_xcd": lines_xcd * LINE + frame_bytes + stream_bytes,
         "lookups": st.n_loads, "lookups_per_distinct_line": round(st.n_loads / max(1, lines), 2),
         "kernel_source_sha": bench.kernel_source_hash(), "seconds": round(time.time() - t0, 2)}
    if note:
        e["note"] = note
    print(f"{key}: {lines} lines chip-wide ({lines * LINE / 1e6:.1f} MB), {lines_xcd} summed over XCDs ({lines_xcd * LINE / 1e6:.1f} MB), "
          f"frame {frame_bytes / 1e6:.1f} MB, streams {stream_bytes / 1e6:.1f} MB -> compulsory {e['compulsory_bytes'] / 1e6:.1f} MB "
          f"({e['compulsory_bytes_per_xcd'] / 1e6:.1f} MB per-XCD flavour); {st.n_loads} lookups", flush=True)
    for i, p in enumerate(t["phases"]):
        print(f"    phase {i} ({p['grid']} grid): coarse {p['coarse_lines']} fine {p['fine_lines']} lines; summed over XCDs {p['coarse_lines_xcd_sum']} / {p['fine_lines_xcd_sum']}",
              flush=True)
    return e


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r06"
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only"):
            only = set(a.split("=", 1)[1].split(",")) if "=" in a else set(sys.argv[sys.argv.index(a) + 1].split(","))
    want = lambda n: only is None or n in only      # noqa: E731
    out = {}
    od9 = sb.dragon_standin(9, nthreads=min(32, os.cpu_count() or 1))
    with sb.Scene(od9, device=0) as sc:
        suffix = bench.grid_suffix(sc)
        if want("cfg2"):
            out[f"1920x1080:dragon_standin_d9:default{suffix}"] = entry("cfg2", sc, 1920, 1080)
        if want("cfg3"):
            out[f"3840x2160:dragon_standin_d9:default{suffix}"] = entry("cfg3", sc, 3840, 2160)
            out[f"3840x2160:dragon_standin_d9:compact{suffix}"] = entry("cfg3 compact", sc, 3840, 2160, flags=sb.FLAG_COMPACT)
        if want("cfg5"):
            pt = sb.PathTrace(spp=16)
            sb._lib.check(sb._lib.lib.sdfhip_scene_prepare_path(sc._h))
            out[f"3840x2160:dragon_standin_d9:spp16{bench.grid_suffix(sc, pt)}"] = entry("cfg5", sc, 3840, 2160, pt=pt)
    del od9
    if want("d10"):
        od10 = sb.dragon_standin(10, nthreads=min(32, os.cpu_count() or 1))
        with sb.Scene(od10, device=0) as sc:
            out[f"1920x1080:dragon_standin_d10:default{bench.grid_suffix(sc)}"] = entry("d10", sc, 1920, 1080)
        del od10
    if want("mesh"):
        pts = sb.knot_point_cloud(1_000_000)
        scm = sb.Scene.FromPoints(pts, 10, device=0)
        with scm:
            out[f"1920x1080:knot_d10.asdf:default{bench.grid_suffix(scm)}"] = entry("mesh", scm, 1920, 1080)
    path = os.path.join(REPO, "profiles", f"{tag}_compulsory_bytes.json")
    prev = {}
    if only and os.path.exists(path):
        prev = json.load(open(path))
    prev.update(out)
    with open(path, "w") as f:
        json.dump(prev, f, indent=1, sort_keys=True)
        f.write("\n")
    # gpurun merges gpurun_out/ back: a copy there travels home
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", f"{tag}_compulsory_bytes.json"), "w") as f:
        json.dump(prev, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
