#!/bin/bash
# A/B/C... of several builds of the library on one bench workload: tools/lib_sweep.sh "<libdir> <libdir> ..." [bench args...]
# prints the rate with four batches in flight and the kernel time alone, two repetitions, builds interleaved.
libs=$1; shift
for rep in 1 2; do
  for lib in $libs; do
    out=$(RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so python3 bench.py "$@" --steps 40 --warmup 4 --cpu-seconds 0 --no-others 2>/dev/null)
    echo "$lib $*: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone %.4f ms" % (d["value"], d["roofline"]["kernel_ms"]))')"
  done
done
