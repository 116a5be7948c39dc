#!/bin/bash
# Pool-kernel geometry variants (development aid): lib = 128 slots / 8 stack entries, lib_w1 = 96 / 6, lib_w2 = 112 / 6.
export RTAMD_KERNEL=pool
for v in lib lib_w1 lib_w2; do
  for n in 1000000 4000000; do
    RTAMD_LIB=$PWD/embree-compressed_amd/$v/libembree3.so python bench.py --workload cbvh.leaf --rays $n --steps 8 --warmup 2 --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v rays $n: in flight %.2f Grays/s, alone %.4f ms (frac %.3f)' % (d['value']/1e3, r['kernel_ms'], r['frac']))"
  done
done
