#!/bin/bash
# Sweep of the launch knobs on the three workloads (development aid).
for kv in "RTAMD_BLOCKS_PER_CU=1" "RTAMD_BLOCKS_PER_CU=2" "RTAMD_BLOCKS_PER_CU=3" "RTAMD_CHUNK=32" "RTAMD_CHUNK=64" "RTAMD_CHUNK=256" "RTAMD_REFILL_BATCH=1" "RTAMD_REFILL_BATCH=16" "RTAMD_REFILL_BATCH=32" "RTAMD_LEAF_BATCH=8" "RTAMD_LEAF_BATCH=40"; do
  echo "== $kv"; env $kv bash tools/quick_bench.sh
done
