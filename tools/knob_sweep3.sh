#!/bin/bash
# Workgroups per CU x batches in flight (development aid).
for kv in "RTAMD_BLOCKS_PER_CU=1 4" "RTAMD_BLOCKS_PER_CU=1 6" "RTAMD_BLOCKS_PER_CU=1 8" "RTAMD_BLOCKS_PER_CU=2 4" "RTAMD_BLOCKS_PER_CU=3 3"; do
  set -- $kv
  env $1 python bench.py --inflight $2 --steps 40 --warmup 4 --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 inflight $2: %.0f Mrays/s, alone %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
done
