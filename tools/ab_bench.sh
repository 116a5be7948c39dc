#!/bin/bash
# A/B comparison of two library builds on the headline workload, alternating runs (development aid).
for rep in 1 2; do for v in lib_w2 lib_w1 lib_w3; do
  RTAMD_LIB=$PWD/embree-compressed_amd/$v/libembree3.so python bench.py --steps 40 --warmup 4 --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: in flight %.0f Mrays/s, alone %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
done; done
