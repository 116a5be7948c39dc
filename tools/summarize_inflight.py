#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace of `bench.py --inflight N` into the summary committed as profiles/rNN_<workload>_inflight.json:
durations of the traversal kernels of the timed pass with N batches in flight and, from their start / end stamps, for how long
k of them ran concurrently (the overlap that `value` rests on).

    python tools/summarize_inflight.py <rocprof output dir> <out.json> <launches of the in-flight pass (steps)>

(PMC counters cannot back this mode: counter collection serialises dispatches, which removes the overlap being measured.)"""
import csv
import glob
import json
import os
import sys


def main():
    src, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"] or "trace_pool_kernel" in r["Kernel_Name"]]
    name = max(set(r["Kernel_Name"] for r in rows), key=lambda n: sum(1 for r in rows if r["Kernel_Name"] == n))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"] == name)
    # the timed in-flight steps = the `steps` consecutive launches with the smallest span (bench.py also launches this kernel in
    # its one-stream pass, in warm-ups and in the host-pointer sample, all of which are spread out much further)
    best = min(range(len(ev) - steps + 1), key=lambda i: max(e for _, e in ev[i:i + steps]) - ev[i][0])
    ev = ev[best:best + steps]
    t0, t1 = ev[0][0], max(e for _, e in ev)
    dur = sorted(e - s for s, e in ev)
    pts = sorted([(s, 1) for s, _ in ev] + [(e, -1) for _, e in ev])
    level, last, hist = 0, t0, {}
    for t, d in pts:
        hist[level] = hist.get(level, 0) + (t - last)
        level += d
        last = t
    span = t1 - t0
    try:
        bh = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "embree-compressed_amd", "lib", "KERNEL_HASH")).read().strip()
    except OSError:
        bh = None
    out = {"build_hash": bh, "kernel": name[:140], "launches": len(ev), "span_us": span / 1e3, "us_per_launch_of_span": span / 1e3 / len(ev),
           "kernel_us": {"mean": sum(dur) / len(dur) / 1e3, "min": dur[0] / 1e3, "median": dur[len(dur) // 2] / 1e3, "max": dur[-1] / 1e3},
           "time_share_with_k_kernels_running": {str(k): v / span for k, v in sorted(hist.items())},
           "mean_kernels_running": sum(k * v for k, v in hist.items()) / span}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
