#!/bin/bash
# SQ counters + kernel durations of one workload under two settings of an environment knob (development aid).
# usage: tools/sq_ab.sh <workload> <VAR> "<values>"     -> gpurun_out/sqab_<workload>/
W=$1; VAR=$2; VALS=$3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sqab_$W
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    echo "== $VAR=$v pass $i: $set"
    timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${v}_p$i -- python3 $R/bench.py --workload $W --inflight 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-others > $O/log_${v}_$i.txt 2>&1 || { echo "pass failed"; tail -n 3 $O/log_${v}_$i.txt; exit 1; }
    python3 $R/tools/sq_summary.py $O/${v}_p$i
  done
done
