#!/bin/bash
# A/B/C... of several builds of the library on one bench workload: tools/lib_sweep.sh "<libdir> <libdir> ..." [bench args...]
# prints the rate with four batches in flight, the one-stream rate and the kernel time alone (medians of bench.py's repeats), two
# repetitions, builds interleaved.
libs=$1; shift
for rep in 1 2; do
  for lib in $libs; do
    out=$(RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so python3 bench.py "$@" --steps 40 --warmup 4 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none 2>/dev/null)
    echo "$lib $*: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s [%.0f..%.0f] | one stream %.0f | alone %.4f ms" % (d["value"], d["value_min_max"][0], d["value_min_max"][1], d["one_stream"]["value"], d["roofline"]["kernel_ms"]))')"
  done
done
