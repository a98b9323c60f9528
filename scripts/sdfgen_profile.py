"""Dev script: one GPU SdfGen build of the 200 k-point sphere at depth 8, for rocprofv3 (profiles/r01_sdfgen_d8_kernel_stats.csv)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sdfbox_amd as sb
from test_sdfgen import fib_sphere
v = fib_sphere(200000)
sb.OctData.SdfGen(fib_sphere(1000), 3)
od, st = sb.OctData.SdfGen(v, 8, want_stats=True)
print(od.Length, st.candidate_entries, st.total_ms)
