#!/bin/bash
# In-flight tuning sweep for the quad-form cBVH kernel (development aid).
run() { out=$(env "$@" python3 bench.py --workload cbvh.leaf --steps 40 --cpu-seconds 0 --no-others --inflight ${INFL:-4} 2>/dev/null); echo "$* inflight=${INFL:-4}: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("in-flight %.0f Mrays/s | alone %.4f ms" % (d["value"], d["roofline"]["kernel_ms"]))')"; }
for l in 8 16 24 30; do run RTAMD_OCT_LEAF=$l; done
for i in 3 6 8; do INFL=$i run RTAMD_BUSY_BLOCKS=1; INFL=$i run RTAMD_BUSY_BLOCKS=2; done
for c in 128 512; do run RTAMD_CHUNK=$c; done
run RTAMD_OCT_MAX=8; run RTAMD_OCT_MAX=24
run RTAMD_REFILL_BATCH=16
