#!/bin/bash
# A plain C host with four frames in flight on four HIP streams (tests/c_frames_in_flight.c), by what decides the process's hardware
# queues.  GPU box: bash scripts/hw_queues_c_host.sh -> gpurun_out/r06_hw_queues_c_host.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06_hw_queues_c_host.txt
gcc -std=gnu11 -O2 -Wall -I include -I /opt/rocm/include tests/c_frames_in_flight.c -o /tmp/c_frames_in_flight -L sdfbox_amd -lsdfhip \
    -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$ROOT/sdfbox_amd -Wl,-rpath,/opt/rocm/lib || exit 1
run() { echo "## $1"; shift; ( unset GPU_MAX_HW_QUEUES SDFHIP_KEEP_ENV; for kv in "$@"; do case "$kv" in *=*) export "$kv";; esac; done; /tmp/c_frames_in_flight 2000 $STREAMS 9 $MODE | tail -1 ); }
{
  echo "# tests/c_frames_in_flight.c: plain C + HIP streams + the C ABI, cfg-2's frame (1920x1080, 28 M nodes), 2000 frames per pass, best of passes 1-3"
  for rep in 1 2; do
    STREAMS=4 MODE=""
    run "the runtime's default: nothing in the environment, the library told to leave it alone (SDFHIP_KEEP_ENV=1) -> 4 hardware queues" SDFHIP_KEEP_ENV=1
    run "nothing in the environment: libsdfhip.so exports GPU_MAX_HW_QUEUES=8 when it is loaded"
    run "the host exports GPU_MAX_HW_QUEUES=8 itself" GPU_MAX_HW_QUEUES=8
    run "the host exports GPU_MAX_HW_QUEUES=4: its word stands" GPU_MAX_HW_QUEUES=4
  done
  STREAMS=4 MODE="setenv"
  run "SDFHIP_KEEP_ENV=1 and the program exports the variable at the top of main(), before its first call that touches the GPU (the runtime reads it at its first call)" SDFHIP_KEEP_ENV=1
  STREAMS=6 MODE=""
  run "six streams on the runtime's four queues (SDFHIP_KEEP_ENV=1)" SDFHIP_KEEP_ENV=1
  run "six streams, eight queues"
} > $OUT 2>&1
cat $OUT
