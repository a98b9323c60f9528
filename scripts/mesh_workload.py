"""The reference's mesh import flow (Program.cs:613-650 -> dllmain.cpp:295-319) at dragon scale, end to end on the GPU box:
a 1 M-point .ply -> sdfhip_load_ply -> sdfhip_sdfgen (depth 9 and Model.MaxDepth = 10) -> .asdf -> bench.py --asdf.
Writes the bench lines to gpurun_out/mesh/ (committed copies: profiles/bench_lines/r05_mesh_knot_d*.json).
    python scripts/mesh_workload.py [points]"""
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
import sdfbox_amd as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = "gpurun_out/mesh"; os.makedirs(out, exist_ok=True)
ply = "/tmp/knot.ply"
sb.write_ply(ply, sb.knot_point_cloud(n))
sb.OctData.SdfGen(sb.knot_point_cloud(1000), 3)          # warm the builder up (module load)
for depth in (9, 10):
    t0 = time.perf_counter(); pts = sb.OctData.LoadPly(ply); t1 = time.perf_counter()
    od, st = sb.OctData.SdfGen(pts, depth, want_stats=True); t2 = time.perf_counter()
    asdf = f"/tmp/knot_d{depth}.asdf"
    od.Save(asdf); t3 = time.perf_counter()
    info = {"points": n, "depth": depth, "nodes": od.Length, "scene_mb": round(od.nbytes / 1e6, 1), "load_ply_ms": round((t1 - t0) * 1e3, 1),
            "sdfgen_ms_in_library": round(st.total_ms, 1), "sdfgen_ms_wall": round((t2 - t1) * 1e3, 1), "candidate_entries": int(st.candidate_entries),
            "save_asdf_ms": round((t3 - t2) * 1e3, 1)}
    print(info, flush=True)
    for extra, tag in (([], ""), (["--size", "3840x2160"], "_4k")):
        r = subprocess.run([sys.executable, "bench.py", "--asdf", asdf, "--no-cpu-baseline"] + extra, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(r.stderr[-2000:]); raise SystemExit(1)
        j = json.loads(line[-1]); j["mesh_import"] = info
        json.dump(j, open(f"{out}/r05_mesh_knot_d{depth}{tag}.json", "w"))
        print(f"depth {depth}{tag}: {j['ms_per_step']} ms/frame, {j['value']} Mray/s, latency {j['latency_ms']} ms", flush=True)
