#!/bin/bash
# Every configuration whose PMC figures bench.py reports (profiles/hbm_traffic.json): run on the GPU box via gpurun,
# then `python scripts/summarise_all.py <round>` here.   bash scripts/profile_all.sh r05 [chunk]
# chunk 1 / 2 / 3: a third of the list each (a gpurun call is at most 20 minutes); no chunk: everything
R=${1:-r05}; C=${2:-0}
run() { bash scripts/profile.sh "$@" || exit 1; }
if [ "$C" = 0 ] || [ "$C" = 1 ]; then
run ${R}_1080p
run ${R}_4k --size 3840x2160
run ${R}_4k_compact --size 3840x2160 --compact 1
run ${R}_1080p_display --display
run ${R}_1080p_d10 --depth 10
fi
if [ "$C" = 0 ] || [ "$C" = 2 ]; then
run ${R}_cfg5 --size 3840x2160 --spp 16
run ${R}_1080p_onekernel --lab --one-kernel
run ${R}_1080p_queue --lab --shadow-queue
run ${R}_4k_queue --size 3840x2160 --lab --shadow-queue
fi
if [ "$C" = 0 ] || [ "$C" = 3 ]; then
run ${R}_1080p_dense --top-grid-level 9
run ${R}_4k_dense --size 3840x2160 --top-grid-level 9
run ${R}_1080p_split7 --top-grid-split 7
run ${R}_1080p_split6 --top-grid-split 6
fi
