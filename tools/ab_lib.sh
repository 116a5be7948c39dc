#!/bin/bash
# A/B of two builds of the library on the bench workloads (development aid): tools/ab_lib.sh <libdir A> <libdir B> [bench args...]
# e.g. make -C embree-compressed_amd OUT=lib_cw0 EXTRA=-DTRACE_CBVH_PREFETCH=0 && tools/ab_lib.sh lib lib_cw0 --workload cbvh.leaf
A=$1; B=$2; shift 2
run() { lib=$1; shift; out=$(RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so python3 bench.py "$@" --steps 40 --warmup 4 --cpu-seconds 0 --no-others 2>/dev/null); echo "$lib $*: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone %.4f ms" % (d["value"], d["roofline"]["kernel_ms"]))')"; }
for rep in 1 2; do
  run $A "$@"
  run $B "$@"
done
