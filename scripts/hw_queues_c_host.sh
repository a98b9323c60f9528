#!/bin/bash
# A C host with four frames in flight, with and without GPU_MAX_HW_QUEUES=8 in ITS environment (INTEGRATION.md
# section 3).  GPU box: bash scripts/hw_queues_c_host.sh  -> gpurun_out/r06_hw_queues_c_host.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06_hw_queues_c_host.txt
gcc -std=gnu11 -O2 -Wall -I include -I /opt/rocm/include tests/c_frames_in_flight.c -o /tmp/c_frames_in_flight -L sdfbox_amd -lsdfhip \
    -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$ROOT/sdfbox_amd -Wl,-rpath,/opt/rocm/lib || exit 1
{
  echo "# tests/c_frames_in_flight.c: plain C + four HIP streams + the C ABI, cfg-2's frame, 2000 frames per pass (best of passes 1-3)"
  for rep in 1; do
    for Q in unset 8 unset 8; do
      if [ "$Q" = unset ]; then ( unset GPU_MAX_HW_QUEUES; /tmp/c_frames_in_flight 2000 4 | tail -1 ); else ( export GPU_MAX_HW_QUEUES=$Q; /tmp/c_frames_in_flight 2000 4 | tail -1 ); fi
    done
  done
  echo "# the variable exported by the program itself at the top of main(), before its first call that touches the GPU"
  ( unset GPU_MAX_HW_QUEUES; /tmp/c_frames_in_flight 2000 4 9 setenv | tail -1 )
  echo "# six streams on the runtime's four queues, and on eight"
  ( unset GPU_MAX_HW_QUEUES; /tmp/c_frames_in_flight 2000 6 | tail -1 )
  ( export GPU_MAX_HW_QUEUES=8; /tmp/c_frames_in_flight 2000 6 | tail -1 )
} > $OUT 2>&1
cat $OUT
