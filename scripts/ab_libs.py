"""Dev script: run scripts/sweep.py against several builds of the library (SDFHIP_LIB)."""
import os, subprocess, sys
libs = sys.argv[1:]
for lib in libs:
    env = dict(os.environ, SDFHIP_LIB=lib)
    out = subprocess.run([sys.executable, "scripts/sweep.py", "--rounds", "6", "--sizes", "1920x1080,3840x2160"], env=env, capture_output=True, text=True).stdout
    print(lib, [l.strip()[17:] for l in out.splitlines() if l.strip().startswith("stack  ") and ("median" in l or "in-flight" in l)], flush=True)
