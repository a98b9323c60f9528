#!/bin/bash
# Runs on the GPU box (via gpurun): what a kernel's loads WAIT for -- separate rocprofv3 --pmc passes (never beside a trace) of a bench
# command with the counters of the memory path: L1 -> L2 read latency, L2 -> fabric read latency and credit stalls, the L1's
# pending-miss stalls, address translation (UTCL1), the texture addresser, the L2's tag stalls, the SQ's own VMEM latency.
#   [FROM=n] bash scripts/pmc_latency_study.sh <tag> [bench args...]    ->  gpurun_out/lat_<tag>/p<i>/...; then scripts/pmc_latency_summary.py <tag>
set -u
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
TAG=${1:-lat}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=$ROOT/gpurun_out/lat_$TAG; mkdir -p "$OUT"
i=0
for pass in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum" \
            "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" \
            "TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_LATENCY_FIFO_FULL_sum" \
            "TCC_HIT_sum TCC_MISS_sum" \
            "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  i=$((i+1))
  [ $i -lt ${FROM:-1} ] && continue
  # (a request the hardware cannot collect -- three TA counters in one pass -- makes rocprofv3 abort and then wait for ever: every pass is bounded)
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --configs none --sustained off --only-timed --live-pmc off "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err" || echo "pass $i failed: $pass"
  echo "pass $i done: $pass"
done
