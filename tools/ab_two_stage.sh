#!/bin/bash
# A/B of two builds of the library over all cBVH modes, levels and ray kinds (development aid): embree-compressed_amd/lib_wbase (a copy of lib/ made
# before a change: cp -r embree-compressed_amd/lib embree-compressed_amd/lib_wbase) against embree-compressed_amd/lib.  Run on the GPU box from the repo root.
R=$PWD
B="--cpu-seconds 0 --no-others --no-pcie --scaled-levels none --steps 20 --warmup 3"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("in flight %8.0f Mrays/s [%.0f..%.0f] | one stream %8.0f | alone %.4f ms" % (d["value"], d["value_min_max"][0], d["value_min_max"][1], d["one_stream"]["value"], r["kernel_ms"]))'
for args in "--workload cbvh.leaf" "--workload cbvh.box" "--workload cbvh.full" "--workload cbvh.grid" "--workload cbvh.leaf --levels 8,3" "--workload cbvh.leaf --levels 6,2" "--workload cbvh.leaf --levels 6,4" "--workload cbvh.leaf --rays-kind primary" "--workload cbvh.leaf --rays-kind secondary" "--workload cbvh.leaf --rays-kind secondary --query occluded"; do
  for lib in lib_wbase lib; do
    echo "$lib $args: $(RTAMD_LIB=$R/embree-compressed_amd/$lib/libembree3.so timeout -k 10 120 python3 bench.py $args $B 2>/dev/null | python3 -c "$fmt")"
  done
done
