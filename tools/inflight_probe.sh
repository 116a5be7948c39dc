#!/bin/bash
# Which of (steps, workload, in-flight count) changes the pipelined rate (development aid).
for args in "--workload cbvh.leaf --steps 5 --warmup 2" "--workload cbvh.leaf --steps 20 --warmup 3" "--workload tri --steps 20 --warmup 3" "--workload cbvh.leaf --steps 20 --warmup 8" "--workload cbvh.leaf --steps 20 --warmup 3 --inflight 2"; do
  python bench.py $args --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$args', '->', round(d['value']), 'Mrays/s', d['config']['in_flight_note'][-40:])"
done
