#!/bin/bash
# SQ counter passes (stall / latency picture) of one bench workload, one stream (development aid).
# usage: tools/sq_passes.sh <workload> [bench args]   -> gpurun_out/sqp_<workload>/summary.txt
#        SQP_SETS=mem tools/sq_passes.sh ...             -> the L1 / TLB / L2 / fabric picture instead of the SQ one
W=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sqp_$W
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
if [ "$SQP_SETS" = mem ]; then
  SETS=("TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum"
        "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum"
        "TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum"
        "TA_BUSY_avr")
  # round 2 had TA_BUSY_avr in ONE pass with three TA_*_sum counters and rocprofv3 aborted with signal 6 (profiles/r02_cbvh_leaf_sq_stalls.txt).
  # Round 3 located it: with the derived metric in a pass of its own (it works: 88 781) the THREE _sum counters together still abort, and the
  # kept log says why - "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect", a fatal
  # check of the profiler tool at the first dispatch (inside rtcCommitScene's upload): every TA_*_sum sums one hardware counter per TA instance and
  # the TA block has too few counter slots for three of them.  One TA counter per pass.  The names offered by the box are recorded first
  # (rocprofv3 --list-avail), and a pass that fails is reported with the end of its log and the sweep goes on.
  (rocprofv3 --list-avail 2>/dev/null | grep -oE "TA_(BUSY|TOTAL_WAVEFRONTS|ADDR_STALLED_BY_TC_CYCLES|DATA_STALLED_BY_TC_CYCLES)[A-Za-z_]*" | sort -u | tr "\n" " "; echo) > $O/ta_counters_available.txt
  echo "# TA counters offered by rocprofv3 --list-avail on this box: $(cat $O/ta_counters_available.txt)" >> $O/summary.txt
else
  SETS=("SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
        "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
        "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
        "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU")
fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  echo "== pass $i: $set" >> $O/summary.txt
  if timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/bench.py --workload $W "$@" --inflight 1 --steps 6 --warmup 2 --repeats 1 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none > $O/log_$i.txt 2>&1; then
    python3 $R/tools/sq_summary.py $O/p$i >> $O/summary.txt
  else
    echo "pass failed (rc $?); end of its log (kept whole as log_$i.txt next to this file):" >> $O/summary.txt
    grep -B1 -A3 "failed with error code" $O/log_$i.txt | cut -c1-300 >> $O/summary.txt
    tail -n 4 $O/log_$i.txt | cut -c1-300 >> $O/summary.txt
  fi
  rm -rf $O/p$i
done
cat $O/summary.txt
