#!/bin/bash
# Sweep of the octet node-step threshold (RTAMD_OCT_MAX) on one workload: kernel alone (one stream) and four batches in flight.
w=${1:-cbvh.leaf}
for o in ${OCTS:-0 8 16 32}; do
  RTAMD_OCT_MAX=$o python bench.py --workload $w --steps 40 --warmup 4 --cpu-seconds 0 --no-others --inflight 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w oct $o: in flight %.0f Mrays/s, alone %.4f ms (frac %.3f)' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
