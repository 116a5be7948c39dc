#!/usr/bin/env python3
"""Average every collected PMC counter over the dispatches of the (non-counted, intersect) traversal kernel.
usage: pmc_summary.py <rocprofv3 output dir>..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if ('trace_kernel' in k or 'trace_pool_kernel' in k) and 'false, false, true>' in k:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, x in sorted(agg.items()):
            print('%-40s %16.1f  (n=%d)' % (c, sum(x) / len(x), len(x)))
