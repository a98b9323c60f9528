#!/bin/bash
# PMC + stats passes of the mesh-derived workloads (scripts/mesh_workload.py's scenes): builds the .asdf files on the box, then
# scripts/profile.sh on `bench.py --asdf`.  Afterwards, here: python scripts/summarise_all.py <round> (it knows the keys), then
# scripts/mesh_workload.py again so that its bench lines find their counters.     bash scripts/profile_mesh.sh r02
R=${1:-r02}
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import sdfbox_amd as sb
ply = "/tmp/knot.ply"
sb.write_ply(ply, sb.knot_point_cloud(1_000_000))
for depth in (9, 10):
    od = sb.OctData.SdfGen(sb.OctData.LoadPly(ply), depth)
    od.Save(f"/tmp/knot_d{depth}.asdf")
    print("built", depth, od.Length, flush=True)
PY
bash scripts/profile.sh ${R}_mesh_d9 --asdf /tmp/knot_d9.asdf && \
bash scripts/profile.sh ${R}_mesh_d9_4k --asdf /tmp/knot_d9.asdf --size 3840x2160 && \
bash scripts/profile.sh ${R}_mesh_d10 --asdf /tmp/knot_d10.asdf && \
bash scripts/profile.sh ${R}_mesh_d10_4k --asdf /tmp/knot_d10.asdf --size 3840x2160
