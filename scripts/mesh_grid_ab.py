"""Dev script (GPU): the mesh-derived depth-10 scene (1 M-point knot) through other lookup grids -- the coarse level the upload
picks by default is capped by the size of the tree's own records (level 7 + 8^3-cell blocks here); SDFHIP_TOP_GRID_SPLIT forces one."""
import json, os, subprocess, sys
sys.path.insert(0, ".")
import sdfbox_amd as sb
asdf = "/tmp/knot_d10.asdf"
if not os.path.exists(asdf):
    sb.OctData.SdfGen(sb.knot_point_cloud(1000), 3)
    sb.OctData.SdfGen(sb.knot_point_cloud(1_000_000), 10).Save(asdf)
for split in ("", "6", "7", "8"):
    for size in ("1920x1080", "3840x2160"):
        env = dict(os.environ)
        if split: env["SDFHIP_TOP_GRID_SPLIT"] = split
        r = subprocess.run([sys.executable, "bench.py", "--asdf", asdf, "--no-cpu-baseline", "--configs", "none", "--size", size], capture_output=True, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(r.stderr[-1500:]); raise SystemExit(1)
        j = json.loads(line[-1])
        print(f"split {split or 'default'} {size}: {j['ms_per_step']} ms/frame, {j['value']} Mray/s, latency {j['latency_ms']}, {j['config'].get('kernel','')[:90]}", flush=True)
