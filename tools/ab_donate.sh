#!/bin/bash
# Drain consolidation A/B (development aid): kernel alone and four batches in flight, per workload and RTAMD_DONATE value.
# usage: tools/ab_donate.sh "0 8 16 24 31" [steps]
for wl in cbvh.leaf tri eager; do
  for d in $1; do
    out=$(RTAMD_DONATE=$d python3 bench.py --workload $wl --steps ${2:-40} --cpu-seconds 0 --no-others 2>/dev/null)
    echo "$wl donate=$d $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone kernel %.4f ms frac %.3f" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))')"
  done
done
