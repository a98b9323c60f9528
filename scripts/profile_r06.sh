#!/bin/bash
# Round 6's evidence set: only what bench.py's line reports -- the headline, the same command with a camera that moves every
# frame, and the `configs` rows -- instead of round 3-5's seventeen configurations (the A/B forms are history).
#   bash scripts/profile_r06.sh r06 1|2|3     (a gpurun call is at most 20 minutes: three chunks)
# then here: python scripts/summarise_all.py r06 && python scripts/compulsory_bytes.py r06 --install
R=${1:-r06}; C=${2:-0}
run() { bash scripts/profile.sh "$@" || exit 1; }
if [ "$C" = 0 ] || [ "$C" = 1 ]; then
run ${R}_1080p
run ${R}_1080p_orbit --orbit 90
run ${R}_4k --size 3840x2160
run ${R}_4k_compact --size 3840x2160 --compact 1
fi
if [ "$C" = 0 ] || [ "$C" = 2 ]; then
[ "$C" = 2 ] && run ${R}_1080p          # (again: chunk 1's first run of it still had the sustained legs' frames in its passes)
run ${R}_1080p_d10 --depth 10
run ${R}_cfg5 --size 3840x2160 --spp 16
fi
if [ "$C" = 0 ] || [ "$C" = 3 ]; then
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import sdfbox_amd as sb
ply = "/tmp/knot.ply"
sb.write_ply(ply, sb.knot_point_cloud(1_000_000))
od = sb.OctData.SdfGen(sb.OctData.LoadPly(ply), 10)
od.Save("/tmp/knot_d10.asdf")
print("built", od.Length, flush=True)
PY
run ${R}_mesh_d10 --asdf /tmp/knot_d10.asdf
python3 scripts/compulsory_bytes.py $R > gpurun_out/${R}_compulsory.log 2>&1 || echo "compulsory_bytes failed"
tail -3 gpurun_out/${R}_compulsory.log
fi
