#!/bin/bash
# Short GPU timing of the three main workloads (development aid): prints workload, Mrays/s, kernel ms.
for w in cbvh.leaf eager tri; do
  python bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 0 --no-others --inflight ${INFLIGHT:-1} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s %9.1f Mrays/s  kernel %.4f ms  nodes/ray %.2f leaves/ray %.3f inner/ray %.2f' % ('$w', d['value'], r['kernel_ms'], r['nodes_per_ray'], r['leaf_visits_per_ray'], r['inner_steps_per_ray']))"
done
