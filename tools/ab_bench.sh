#!/bin/bash
# A/B comparison of two library builds on the headline workload, alternating runs (development aid).  INFL = batches in flight.
for rep in 1 2; do for v in lib lib_w2; do
  RTAMD_LIB=$PWD/embree-compressed_amd/$v/libembree3.so python bench.py --inflight ${INFL:-4} --steps 40 --warmup 4 --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: in flight %.0f Mrays/s, alone %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
done; done
