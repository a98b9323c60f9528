#!/bin/bash
# profiles/r05_rank_emulation.txt: one rank's whole job of an N-rank run, emulated on one GPU (scripts/rank_emulation.py) --
# 1080p and 4K, 2 / 4 / 8 ranks, the shares in the default order and in tile order, the cost deal and rounds 2-4's credit deal.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/rank_emulation; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
for size in 1920x1080 3840x2160; do
  for v in "" "--order" "--deal weight" "--deal weight --order"; do
    echo "== python scripts/rank_emulation.py $size $v" >> $OUT/all.txt
    python3 scripts/rank_emulation.py $size $v 2>/dev/null >> $OUT/all.txt
  done
done
echo "== the (G, groups in flight) sweep at 8 ranks, 1080p, tile order" >> $OUT/all.txt
python3 scripts/rank_emulation.py 1920x1080 --worlds 8 --sweep --order 2>/dev/null >> $OUT/all.txt
cat $OUT/all.txt
