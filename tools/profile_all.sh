#!/bin/bash
# The committed profile recipe (run on the GPU box from the repo root):
#   tools/profile_all.sh r01            -> gpurun_out/profiles_out/r01_{cbvh_leaf,tri}_{kernel_stats.csv,pmc.json,bench.json}
# One rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass with the trace domains), the
# program itself right after `--`.  bench.py runs with --inflight 1: one stream, strictly back-to-back launches, so that
# the per-launch duration of the kernel trace is the duration roofline.kernel_ms reports.
# Environment: PROF_WORKLOADS (default "cbvh.leaf tri"), PROF_ARGS (extra bench.py arguments, e.g. "--levels 8,3"), PROF_SUFFIX (appended to
# the workload in the output names, e.g. "_L8": profiles/r03_cbvh_leaf_L8_pmc.json is what bench.py's roofline_scaled looks for),
# PROF_INFLIGHT=0 skips the four-batches-in-flight kernel trace.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for w in ${PROF_WORKLOADS:-cbvh.leaf tri}; do
  O=$R/gpurun_out/prof_$w${PROF_SUFFIX}
  rm -rf $O && mkdir -p $O
  A="--workload $w --inflight 1 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none --repeats 1 --steps 20 --warmup 3 $PROF_ARGS"
  echo "== $w: kernel trace"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $A > $O/bench_trace.json 2> $O/trace.err
  echo "== $w: FETCH_SIZE"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A > $O/bench_fetch.json 2> $O/fetch.err
  echo "== $w: WRITE_SIZE"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A > $O/bench_write.json 2> $O/write.err
  mkdir -p $R/gpurun_out/profiles_out
  P=$R/gpurun_out/profiles_out/${TAG}_$(echo $w | tr . _)${PROF_SUFFIX}   # only gpurun_out/ travels back: copy these into profiles/ afterwards
  python3 $R/tools/summarize_prof.py $O $P
  cp $O/bench_trace.json ${P}_bench.json
  if [ $w = cbvh.leaf ] && [ "${PROF_INFLIGHT:-1}" = 1 ]; then
    # the mode `value` is measured in: four batches in flight on four streams (kernel trace only: counter collection would
    # serialise the dispatches)
    echo "== $w: kernel trace, 4 batches in flight"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_if4 -- python3 $R/bench.py --workload $w --inflight 4 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none --repeats 1 --steps 40 --warmup 4 $PROF_ARGS > $O/bench_if4.json 2> $O/if4.err
    python3 $R/tools/summarize_inflight.py $O/trace_if4 ${P}_inflight.json 40
    cp $O/bench_if4.json ${P}_inflight_bench.json
  fi
done
ls -la $R/gpurun_out/profiles_out
