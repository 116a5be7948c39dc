#!/bin/bash
# Pipelined rate against the number of batches in flight (development aid).
for n in 2 3 4 6 8; do
  python bench.py --inflight $n --cpu-seconds 0 --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('in flight', $n, round(d['value']), 'Mrays/s, aggregate frac %.3f, isolated frac %.3f' % (d['roofline']['aggregate_frac_in_flight'], d['roofline']['frac']))"
done
