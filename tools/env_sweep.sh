#!/bin/bash
# Generic knob sweep: tools/env_sweep.sh <workload> "VAR=val [VAR2=val]" ... ; prints rate with four batches in flight and kernel time alone.
w=${1:-cbvh.leaf}; shift
for cfg in "$@"; do
  env $cfg python bench.py --workload $w --steps 40 --warmup 4 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none --inflight 4 ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w [$cfg]: in flight %.0f Mrays/s, alone %.4f ms (frac %.3f)' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
