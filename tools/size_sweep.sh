#!/bin/bash
# Batch-size sweep (development aid): how much of the kernel time is ramp/drain.
for w in cbvh.leaf tri; do for n in 250000 1000000 4000000 16000000; do
  python bench.py --workload $w --rays $n --steps 6 --warmup 2 --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s rays %9d %9.1f Mrays/s  kernel %.4f ms frac %.3f' % ('$w', $n, d['value'], r['kernel_ms'], r['frac']))"
done; done
