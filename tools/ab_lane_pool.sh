#!/bin/bash
# lane-per-ray skeleton against the ray-pool skeleton (RTAMD_KERNEL=lane|pool) over batch sizes: tools/ab_lane_pool.sh [workload] [sizes...]
W=${1:-cbvh.leaf}; shift
SIZES=${*:-1000000 2000000 3000000 4000000 8000000}
B="--workload $W --cpu-seconds 0 --no-others --no-pcie --scaled-levels none --steps 10 --warmup 2"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("in flight %8.0f Mrays/s [%.0f..%.0f] | one stream %8.0f | alone %.4f ms" % (d["value"], d["value_min_max"][0], d["value_min_max"][1], d["one_stream"]["value"], r["kernel_ms"]))'
for n in $SIZES; do for k in lane pool; do
  echo "$W rays $n RTAMD_KERNEL=$k: $(RTAMD_KERNEL=$k timeout -k 10 120 python3 bench.py --rays $n $B 2>/dev/null | python3 -c "$fmt")"
done; done
