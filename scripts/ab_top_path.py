"""Dev script: the path-traced mode for several top-grid levels (SDFHIP_TOP_GRID_LEVEL)."""
import os, subprocess, sys, json
for lv in sys.argv[1:] or ["0", "5", "6", "7", "8", "9"]:
    env = dict(os.environ, SDFHIP_TOP_GRID_LEVEL=lv)
    res = []
    for args in (["--size", "1920x1080", "--spp", "4", "--steps", "10", "--warmup", "2"], ["--size", "3840x2160", "--spp", "16", "--steps", "4", "--warmup", "1"]):
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, env=env, capture_output=True, text=True).stdout
        j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        res.append((args[1], args[3], j["ms_per_step"]))
    print("top grid level", lv, res, flush=True)
