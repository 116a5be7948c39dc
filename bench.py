#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of incoherent random rays through the MI355X traversal path.

    python bench.py --gpus N --steps K --warmup W            (N=1 directly; N>1 under torch.distributed.run)

One "step" = one rtcIntersect1M call over one batch of `--rays` device-resident RTCRayHit records (default
1 M, BASELINE.json configs[1]: random incoherent rays vs the BVH8 of the bomberman triangles).  Every step
traces a DIFFERENT, freshly generated batch (rays are modified in place by a trace, so re-tracing a batch
would shrink the work).  Rank r owns its own batches (weak scaling, no data-path collective: SURVEY.md 8e);
the BVH is replicated per GPU.  Timing: barrier + synchronize on both sides of exactly K steps, MAX over
ranks; `value` = all rays of all ranks / that time.

The JSON line also carries
  roofline      algorithmic bytes per launch / mean kernel time (HIP events on the launch stream) vs 8 TB/s
  cpu_baseline  the oracle (a scalar C port of the reference's AVX2 path) on the host cores, bounded sample
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec peak
IO_BYTES_PER_RAY = 84  # 48 B RTCRay read + 36 B (tfar, Ng, u, v, primID, geomID, instID) written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=1_000_000, help="rays per step per GPU")
    ap.add_argument("--accel", default="bvh8.triangle4v", help="tri_accel: bvh8.triangle4v (robust/Pluecker) or bvh8.triangle4")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline sample (0 = skip)")
    return ap.parse_args()


def pmc_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN_*_pmc.json, produced by tools/summarize_prof.py; FETCH_SIZE already doubled per the gfx950
    correction).  bench.py cannot collect PMC counters itself; None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    for path in reversed(files):
        try:
            d = json.load(open(path))
            rd = [v["read_bytes"] for k, v in d.get("FETCH_SIZE", {}).items() if kernel_tag in k]
            wr = [v["write_bytes"] for k, v in d.get("WRITE_SIZE", {}).items() if kernel_tag in k]
            if rd and wr:
                return rd[0] + wr[0], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def cpu_baseline(verts, tris, lo, hi, m, budget_s, mode):
    """Oracle timed on the host cores (kind "port"): the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    orc = po.TriangleScene(verts, tris, mode)
    src = po.make_random_rays(m, lo, hi, seed=12345)
    total, spent, reps = 0, 0.0, 0
    while spent < budget_s and reps < 4096:
        work = src.copy()
        t0 = time.perf_counter()
        orc.intersect1M(work, nthreads=cores)
        spent += time.perf_counter() - t0
        total += m
        reps += 1
    orc.free()
    return {"value": total / spent / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {m} rays of the same generator (seed 12345), oracle/liboracle.so, {cores} pthreads, blocks of 1024"}


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("embree-compressed_amd")
    rtc = pkg.rtc
    raygen = importlib.import_module("embree-compressed_amd.raygen")

    d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
    verts = d["verts"]
    tris = rtc.fan_triangulate(d["face_sizes"], d["face_index"])
    lo, hi = verts.min(0), verts.max(0)

    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    dev = rtc.Device(f"gpu={local_rank},tri_accel={args.accel}")
    dev.set_stream(stream.cuda_stream)
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, tris)
    sc.commit()
    st = sc.stats()

    m, K, W = args.rays, args.steps, args.warmup
    # distinct batches per step and per rank, generated on the host, resident in HBM before timing starts
    bufs = []
    for s in range(K + W + 1):
        host = raygen.make_random_rays(m, lo, hi, seed=rank * 100003 + s)
        bufs.append(torch.from_numpy(host).to("cuda", non_blocking=False))
    torch.cuda.synchronize()

    # work counters on the extra batch -> algorithmic bytes per ray (implementation's own visits x record sizes)
    cnt = sc.intersect1M_counted(bufs[K + W])
    n_node = cnt["nodeVisits"] / max(cnt["rays"], 1)
    n_prim = cnt["primTests"] / max(cnt["rays"], 1)
    bytes_per_ray = IO_BYTES_PER_RAY + n_node * st["nodeBytes"] + n_prim * st["primBytes"]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for s in range(W):
        sc.intersect1M(bufs[s], check=False)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for s in range(W, W + K):
        sc.intersect1M(bufs[s], check=False)
    ev1.record(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    dev.check("timed region")
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / K  # HIP events on the launch stream; one kernel per step

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    hits = int((bufs[W].view(torch.int32)[:, 18] != -1).sum().item())

    if rank == 0:
        total_rays = m * K * world
        achieved = bytes_per_ray * m / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, None)
        if m == 1_000_000:  # the committed PMC passes were taken on this exact workload
            tag = "trace_tri_kernel<true, false, false, true>" if args.accel.endswith("4v") else "trace_tri_kernel<false, false, false, true>"
            traffic, traffic_src = pmc_traffic(tag)
        out = {
            "metric": "Mrays/s (incoherent) on bomberman, device-resident ray batches",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 1M random incoherent rays (drand48 LCG, bbox-uniform endpoints) vs quantized BVH8 of bomberman.obj fan triangles (1454), rtcIntersect1M on device-resident RTCRayHit[80B]",
                       "rays_per_step_per_gpu": m, "accel": args.accel, "bvh_nodes": st["nodeCount"], "bvh_bytes": st["totalBytes"],
                       "hits_first_timed_batch": hits, "sharding": f"replicated BVH, {world} independent ray shards, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_ms,
                         "bytes_per_ray": bytes_per_ray, "nodes_per_ray": n_node, "prims_per_ray": n_prim,
                         "node_bytes": st["nodeBytes"], "prim_bytes": st["primBytes"]},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(verts, tris, lo, hi, m, args.cpu_seconds, 0 if args.accel.endswith("4v") else 1)
        print(json.dumps(out), flush=True)

    sc.release()
    dev.release()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
