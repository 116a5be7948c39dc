#!/bin/bash
# The one parameterised A/B / sweep script (development aid; replaces knob_sweep*.sh, ab_*.sh, lib_sweep.sh, env_sweep.sh, size_sweep.sh,
# oct_sweep.sh, inflight_sweep.sh, quick_bench.sh, variant_bench.sh, pool_variants.sh of rounds 1-2).  Every line is one bench.py run:
# rate with four batches in flight (median of the repeated timed regions, min..max), the one-stream rate, the kernel time alone.
#   tools/sweep.sh env   <workload> "VAR=val [VAR2=val]" ...          knob sweep (env RTAMD_* / bench flags through BENCH_ARGS)
#   tools/sweep.sh libs  "<libdir> <libdir> ..." [bench args]          builds of the library (make OUT=lib_wX EXTRA=-D...), interleaved, 2 repetitions
#   tools/sweep.sh sizes "<libdir> ..." [workload]                      batch-size sweep 4 k .. 4 M rays
#   tools/sweep.sh workloads                                            cbvh.leaf / eager / tri on the default build
# BENCH_ARGS adds bench.py arguments to every run (e.g. BENCH_ARGS="--levels 8,3" or "--inflight 2").
mode=$1; shift
COMMON="--cpu-seconds 0 --no-others --no-pcie --scaled-levels none ${BENCH_ARGS}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("in flight %8.0f Mrays/s [%.0f..%.0f] | one stream %8.0f | alone %.4f ms (frac %.3f) | nodes/ray %.2f leaves/ray %.3f" % (d["value"], d["value_min_max"][0], d["value_min_max"][1], d["one_stream"]["value"], r["kernel_ms"], r["frac"], r["nodes_per_ray"], r["leaf_visits_per_ray"]))'
run() { python3 bench.py "$@" $COMMON 2>/dev/null | python3 -c "$fmt"; }
case $mode in
  env)
    w=$1; shift
    for cfg in "$@"; do echo "$w [$cfg]: $(env $cfg python3 bench.py --workload $w --steps 40 --warmup 4 $COMMON 2>/dev/null | python3 -c "$fmt")"; done ;;
  libs)
    libs=$1; shift
    for rep in 1 2; do for lib in $libs; do
      echo "$lib $*: $(RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so run "$@" --steps 40 --warmup 4)"
    done; done ;;
  sizes)
    libs=${1:-lib}; w=${2:-cbvh.leaf}
    for n in 4096 16384 65536 131072 250000 500000 1000000 4000000; do for lib in $libs; do
      echo "$lib $w rays $n: $(RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so run --workload $w --rays $n --steps 10 --warmup 2)"
    done; done ;;
  workloads)
    for w in cbvh.leaf eager tri; do echo "$w: $(run --workload $w --steps 10 --warmup 2)"; done ;;
  *) echo "usage: tools/sweep.sh env|libs|sizes|workloads ... (see the header of this file)"; exit 2 ;;
esac
