#!/bin/bash
# Combined knob settings, pipelined pass (value) and one-stream pass (kernel ms) side by side (development aid).
for kv in "RTAMD_CHUNK=256 RTAMD_LEAF_BATCH=32" "RTAMD_CHUNK=256 RTAMD_BLOCKS_PER_CU=2" "RTAMD_CHUNK=256 RTAMD_BLOCKS_PER_CU=2 RTAMD_LEAF_BATCH=32" "RTAMD_CHUNK=192" "RTAMD_CHUNK=512" "RTAMD_CHUNK=384 RTAMD_BLOCKS_PER_CU=2"; do
  echo "== $kv"; env $kv INFLIGHT=4 bash tools/quick_bench.sh
done
