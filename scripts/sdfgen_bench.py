"""N1 measurement: the HIP SdfGen builder against the CPU oracle (the restatement of
SdfGen/dllmain.cpp) on the sphere cloud of BASELINE.md section 2 (200 k points, r = 0.5)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sdfbox_amd as sb
import oracle
from test_sdfgen import fib_sphere

v = fib_sphere(200000)
sb.OctData.SdfGen(fib_sphere(1000), 3)          # warm-up (module load)
for depth in (4, 6, 7, 8, 9):
    t0 = time.perf_counter(); od, st = sb.OctData.SdfGen(v, depth, want_stats=True); tg = time.perf_counter() - t0
    line = f"depth {depth}: N={od.Length} nodes, {st.candidate_entries/1e6:.1f} M candidate entries, GPU {tg*1e3:.0f} ms wall ({st.total_ms:.0f} ms in the library)"
    if depth <= 8:
        t0 = time.perf_counter(); o = oracle.sdfgen(v, depth); tc = time.perf_counter() - t0
        same = (od.Structs == o["structs"]).all() and (od.Values == o["values"]).all()
        line += f"; CPU oracle (1 thread) {tc:.2f} s -> {tc/tg:.0f}x; identical bytes: {same}"
    print(line, flush=True)
