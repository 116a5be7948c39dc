#!/bin/bash
# Batch-size sweep (development aid): kernel time alone and rate in flight against the batch size, optionally for several library
# builds: tools/size_sweep.sh "lib lib_wX" [workload]
libs=${1:-lib}; w=${2:-cbvh.leaf}
for n in 4096 16384 65536 131072 250000 500000 1000000 4000000; do for lib in $libs; do
  RTAMD_LIB=$PWD/embree-compressed_amd/$lib/libembree3.so python3 bench.py --workload $w --rays $n --steps 10 --warmup 2 --cpu-seconds 0 --no-others --no-pcie --scaled-levels none 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s %-10s rays %9d  in flight %9.1f Mrays/s  one stream %9.1f  kernel %.4f ms frac %.3f' % ('$lib', '$w', $n, d['value'], d['one_stream']['value'], r['kernel_ms'], r['frac']))"
done; done
