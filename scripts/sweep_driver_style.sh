#!/bin/bash
mkdir -p gpurun_out/sweep20
for rep in 1 2 3; do
for fif in 2 3 4 6 8 12; do
for to in "" "--tile-order"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs none --frames-in-flight $fif $to 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fif=$fif to=\"$to\" rep=$rep', d['ms_per_step'], d['value'])" >> gpurun_out/sweep20/out.txt || exit 1
done; done; done
for to in "" "--tile-order"; do
  python bench.py --no-cpu-baseline --configs none $to 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('400 steps to=\"$to\"', d['ms_per_step'], d['value'], d.get('latency'))" >> gpurun_out/sweep20/out.txt || exit 1
done
cat gpurun_out/sweep20/out.txt
