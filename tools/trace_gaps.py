#!/usr/bin/env python3
"""Timeline of the traversal kernels from a rocprofv3 --kernel-trace csv: per-kernel duration, how many run concurrently,
idle time of the device between the first start and the last end of the timed steps.  usage: trace_gaps.py <dir> [skip]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = [r for r in csv.DictReader(open(f)) if 'trace_kernel' in r['Kernel_Name'] or 'trace_pool_kernel' in r['Kernel_Name']]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)[skip:]
t0, t1 = ev[0][0], max(e for _, e in ev)
dur = [e - s for s, e in ev]
print('%d kernels, span %.1f us, mean duration %.1f us, min %.1f, max %.1f' % (len(ev), (t1 - t0) / 1e3, sum(dur) / len(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3))
pts = sorted([(s, 1) for s, _ in ev] + [(e, -1) for _, e in ev])
level, last, hist = 0, t0, {}
for t, d in pts:
    hist[level] = hist.get(level, 0) + (t - last)
    level += d
    last = t
print('time with k kernels running:', {k: '%.1f us' % (v / 1e3) for k, v in sorted(hist.items())})
starts = [s for s, _ in ev]
print('start-to-start gaps (us):', ' '.join('%.0f' % ((b - a) / 1e3) for a, b in zip(starts, starts[1:]))[:600])
