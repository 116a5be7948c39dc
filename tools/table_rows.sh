#!/bin/bash
# Re-measures the extra rows of DESIGN.md section 3 (camera rays, the scaled L8 scenes).  Development aid.
run() {
  python bench.py "$@" --cpu-seconds 0 --no-others --no-pcie --scaled-levels none --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$*', '|', c['rays_per_step_per_gpu'], 'rays | in flight %.2f Grays/s | one stream %.2f Grays/s, kernel %.4f ms | n_node %.2f n_leaf %.3f B/ray %.0f frac %.3f aggregate %.3f | accel %.0f MB' % (d['value']/1e3, d['one_stream']['value']/1e3, r['kernel_ms'], r['nodes_per_ray'], r['leaf_visits_per_ray'], r['bytes_per_ray'], r['frac'], r['aggregate_frac_in_flight'], c['accel_bytes']/1e6))"
}
run --workload cbvh.leaf
run --workload eager
run --workload tri
run --workload cbvh.box
run --workload cbvh.grid
run --workload cbvh.full
run --workload cbvh.leaf --rays-kind primary
run --workload eager --rays-kind primary
run --workload tri --rays-kind primary
run --workload cbvh.leaf --levels 8,3
run --workload cbvh.leaf --levels 8,3 --rays-kind primary
run --workload eager --levels 8,3 --rays-kind primary
# config 5: recorded bounce rays (closest hit) and shadow rays (any hit)
run --workload cbvh.leaf --rays-kind secondary
run --workload eager --rays-kind secondary
run --workload tri --rays-kind secondary
run --workload eager --rays-kind secondary --query occluded
run --workload tri --rays-kind secondary --query occluded
run --workload tri --query occluded
