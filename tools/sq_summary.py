#!/usr/bin/env python3
"""Summarise an SQ counter pass (rocprofv3 --pmc SQ_* --kernel-trace csv) for the traversal kernels."""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if ('trace_kernel' in k and 'false, false, true>' in k) or 'cull_kernel' in k:
        agg[k[:100]][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k[:100]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, 'avg ns', sum(dur[k]) / len(dur[k]))
    print(' ', {c: round(x) for c, x in m.items()})
    if 'SQ_ACTIVE_INST_VALU' in m and 'SQ_THREAD_CYCLES_VALU' in m:
        print('  VALU lane utilisation %.3f' % (m['SQ_THREAD_CYCLES_VALU'] / (m['SQ_ACTIVE_INST_VALU'] * 64)))
    if 'SQ_WAVE_CYCLES' in m:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY'):
            if c in m: print('  %s / WAVE_CYCLES = %.3f' % (c, m[c] / m['SQ_WAVE_CYCLES']))
    if 'SQ_WAVES' in m and 'SQ_INSTS_VALU' in m:
        print('  VALU insts per wave %.0f, waves %d' % (m['SQ_INSTS_VALU'] / m['SQ_WAVES'], m['SQ_WAVES']))
