#!/bin/bash
# A/B of one environment knob (development aid): kernel alone and four batches in flight, per workload and value.
# usage: tools/ab_env.sh RTAMD_CULL "0 1" [steps] [extra bench.py args]
VAR=$1; VALS=$2; STEPS=${3:-40}; shift 3
for wl in cbvh.leaf tri eager; do
  for d in $VALS; do
    out=$(env $VAR=$d python3 bench.py --workload $wl --steps $STEPS --cpu-seconds 0 --no-others "$@" 2>/dev/null)
    echo "$wl $VAR=$d $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone %.4f ms/step frac %.3f | B/ray %.1f" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["bytes_per_ray"]))')"
  done
done
