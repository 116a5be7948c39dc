#!/bin/bash
# Sweep of the launch knobs on the three workloads (development aid).  INFLIGHT=n selects the pipelined pass.
for kv in "RTAMD_CHUNK=128" "RTAMD_BLOCKS_PER_CU=2" "RTAMD_CHUNK=64" "RTAMD_CHUNK=256" "RTAMD_REFILL_BATCH=16" "RTAMD_REFILL_BATCH=32" "RTAMD_LEAF_BATCH=16" "RTAMD_LEAF_BATCH=32" "RTAMD_LEAF_BATCH=48"; do
  echo "== $kv"; env $kv bash tools/quick_bench.sh
done
