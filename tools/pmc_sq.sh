#!/bin/bash
# SQ counter passes for one bench workload (development aid; summary via tools/pmc_summary.py).
W=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq_$W/p$i -- python3 $R/bench.py --workload $W --inflight 1 --steps 4 --warmup 1 --cpu-seconds 0 --no-others "$@" > $R/gpurun_out/pmcsq_$W/log$i.txt 2>&1 || { echo "pass $i failed"; grep -m2 -iE "error|exceed" $R/gpurun_out/pmcsq_$W/log$i.txt; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcsq_$W/p$i
done
