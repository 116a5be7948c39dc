#!/bin/bash
# Compare library variants (lib, lib_w1, lib_w2) under the same knobs (development aid).
for v in lib lib_w1 lib_w2; do
  echo "== $v"; RTAMD_LIB=$PWD/embree-compressed_amd/$v/libembree3.so bash tools/quick_bench.sh; RTAMD_LIB=$PWD/embree-compressed_amd/$v/libembree3.so INFLIGHT=4 bash tools/quick_bench.sh
done
