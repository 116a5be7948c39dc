#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/...) into the small summaries committed under profiles/.

    python tools/summarize_prof.py gpurun_out/prof profiles/r01_tri

Writes <prefix>_kernel_stats.csv (the --kernel-trace --stats table, our kernels + top others) and
<prefix>_pmc.json (FETCH_SIZE / WRITE_SIZE per launch of the traversal kernel, with the gfx950 correction of
MI355X_MICROARCH.md section HBM applied: FETCH_SIZE counts 64 B per 128-B request on wide reads -> x2).
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:12]:
                r = list(r)
                r[0] = r[0][:160]
                w.writerow(r)
    try:
        bh = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "embree-compressed_amd", "lib", "KERNEL_HASH")).read().strip()
    except OSError:
        bh = None
    out = {"build_hash": bh,
           "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch as reported by rocprofv3; *_bytes are per launch",
           "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> read_bytes = 2 * FETCH_SIZE * 1024"}
    for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == key and "trace_" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:120]].append(float(r["Counter_Value"]))
        out[key] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)} for k, v in agg.items()}
    for k, v in out.get("FETCH_SIZE", {}).items():
        v["read_bytes"] = 2 * v["mean_KiB"] * 1024
    for k, v in out.get("WRITE_SIZE", {}).items():
        v["write_bytes"] = v["mean_KiB"] * 1024
    json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
    print(open(prefix + "_pmc.json").read())


if __name__ == "__main__":
    main()
