#!/bin/bash
# Every configuration whose PMC figures bench.py reports (profiles/hbm_traffic.json): run on the GPU box via gpurun,
# then `python scripts/summarise_all.py <round>` here.   bash scripts/profile_all.sh r02
R=${1:-r04}
bash scripts/profile.sh ${R}_1080p                                  && \
bash scripts/profile.sh ${R}_1080p_onekernel --lab --one-kernel           && \
bash scripts/profile.sh ${R}_1080p_queue --lab --shadow-queue             && \
bash scripts/profile.sh ${R}_4k_queue --size 3840x2160 --lab --shadow-queue && \
bash scripts/profile.sh ${R}_4k --size 3840x2160                    && \
bash scripts/profile.sh ${R}_4k_compact --size 3840x2160 --compact 1 && \
bash scripts/profile.sh ${R}_1080p_display --display                && \
bash scripts/profile.sh ${R}_1080p_d10 --depth 10                   && \
bash scripts/profile.sh ${R}_cfg5 --size 3840x2160 --spp 16           && \
bash scripts/profile.sh ${R}_1080p_dense --top-grid-level 9             && \
bash scripts/profile.sh ${R}_4k_dense --size 3840x2160 --top-grid-level 9 && \
bash scripts/profile.sh ${R}_1080p_split7 --top-grid-split 7            && \
bash scripts/profile.sh ${R}_1080p_split6 --top-grid-split 6
