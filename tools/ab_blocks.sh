#!/bin/bash
# Grid sweep for the octet-only leaf kernels (development aid): workgroups per CU alone (RTAMD_ALONE_BLOCKS) / in flight (RTAMD_BUSY_BLOCKS).
for wl in ${1:-cbvh.leaf}; do
  for a in 2 3 4; do for b in 1 2; do
    out=$(RTAMD_ALONE_BLOCKS=$a RTAMD_BUSY_BLOCKS=$b python3 bench.py --workload $wl --steps 40 --cpu-seconds 0 --no-others 2>/dev/null)
    echo "$wl alone_blocks=$a busy_blocks=$b $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone %.4f ms/step frac %.3f" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))')"
  done; done
done
