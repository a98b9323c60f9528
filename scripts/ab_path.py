"""Dev script: the path-traced mode (config 5 and a 1080p 4 spp frame) for several builds of the library."""
import os, subprocess, sys, json
for lib in sys.argv[1:]:
    env = dict(os.environ, SDFHIP_LIB=lib)
    res = []
    for args in (["--size", "1920x1080", "--spp", "4", "--steps", "10", "--warmup", "2"], ["--size", "3840x2160", "--spp", "16", "--steps", "4", "--warmup", "1"]):
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, env=env, capture_output=True, text=True).stdout
        j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        res.append((args[1], args[3], j["ms_per_step"], j["value"]))
    print(lib, res, flush=True)
