"""Dev script: scripts/sweep.py for several top-grid levels (SDFHIP_TOP_GRID_LEVEL; 0 = no grid)."""
import os, subprocess, sys
levels = sys.argv[1:] or ["0", "3", "5", "6", "7", "8", "9"]
for lv in levels:
    env = dict(os.environ, SDFHIP_TOP_GRID_LEVEL=lv)
    out = subprocess.run([sys.executable, "scripts/sweep.py", "--rounds", "6", "--sizes", "1920x1080,3840x2160"], env=env, capture_output=True, text=True).stdout
    print("top grid level", lv, [l.strip()[17:] for l in out.splitlines() if l.strip().startswith("stack  ") and ("median" in l or "in-flight" in l)], flush=True)
