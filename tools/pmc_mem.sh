#!/bin/bash
# Memory-system counter passes for one bench workload (development aid).  usage: tools/pmc_mem.sh <workload> [bench args]
# At most two counters of one hardware block per pass (more and rocprofv3 aborts with "exceeds the capabilities").
W=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcmem_$W/p$i -- python3 $R/bench.py --workload $W --inflight 1 --steps 4 --warmup 1 --cpu-seconds 0 --no-others "$@" > $R/gpurun_out/pmcmem_$W/log$i.txt 2>&1 || { echo "pass $i failed"; grep -m2 -iE "error|exceed" $R/gpurun_out/pmcmem_$W/log$i.txt; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcmem_$W/p$i
done
