#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of incoherent random rays through the MI355X traversal path.

    python bench.py --gpus N --steps K --warmup W            (N=1 directly; N>1 under torch.distributed.run)

Workload (default `cbvh.leaf`): the scene BASELINE.json's metric is quoted on — bomberman as a Catmull-Clark
subdivision surface at the reference's bomberman.ecs settings (subdivision level 6, compression level 3,
`--compress.leaf` = subdiv_accel=bvh4.compressed.leaf: 46 528 cBVH blobs under a quantized BVH8) — traced with
1 M random incoherent rays per step (the viewer's makeRandomRay generator, drand48 LCG).  `--workload tri`
selects BASELINE configs[1] (same rays vs the BVH8 of bomberman's 1454 fan triangles), `eager` the default
GridSOA-style subdiv accel, `cbvh.box` / `cbvh.grid` the other fork modes.  The JSON line of the default run also
carries the tri and eager rates under config.other_workloads (short runs of the same loop).

One "step" = one rtcIntersect1M call over one batch of device-resident RTCRayHit records.  Every step traces a
DIFFERENT, freshly generated batch (a trace modifies rays in place, so re-tracing a batch would shrink the work).
Rank r owns its own batches (weak scaling, no data-path collective: SURVEY.md 8e); the accel is replicated per
GPU.  Timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks; `value` = all rays of all
ranks / that time.

The JSON line also carries
  roofline      algorithmic bytes per launch / mean kernel time (HIP events on the launch stream) vs 8 TB/s
  cpu_baseline  the oracle (a scalar C port of the reference's AVX2 path) on the host cores, bounded sample
"""
import argparse
import glob
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

# name -> (device config, geometry kind, oracle mode, dominant-kernel tag in rocprof output, description)
WORKLOADS = {
    "cbvh.leaf": ("subdiv_accel=bvh4.compressed.leaf", "subdiv", 4, "CbvhLeaf<1, 3,",
                  "bomberman.ecs: Catmull-Clark subdiv L6, cBVH C3 pizza-box leaves (bvh4.compressed.leaf), 46528 blobs"),
    "cbvh.box": ("subdiv_accel=bvh4.compressed.box", "subdiv", 3, "CbvhLeaf<0, 3,", "bomberman subdiv L6/C3, cBVH voxel leaves"),
    "cbvh.grid": ("subdiv_accel=bvh4.compressed.grid", "subdiv", 5, "CbvhLeaf<2, 3,", "bomberman subdiv L6/C3, cBVH + float vertex grid"),
    "cbvh.full": ("subdiv_accel=bvh4.compressed.full", "subdiv", 6, "CbvhLeaf<3, 3,", "bomberman subdiv L6/C3, cBVH with float (uncompressed) quadtree nodes, voxel hits"),
    "eager": ("subdiv_accel=default", "subdiv", 2, "GridCellLeaf", "bomberman subdiv L6, eager 3x3-vertex grid cells (GridSOA semantics)"),
    "tri": ("tri_accel=bvh8.triangle4v", "tri", 0, "TriLeaf<true>", "configs[1]: bomberman.obj as 1454 fan triangles, BVH8 + Triangle4v/Pluecker"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--rays", type=int, default=1_000_000, help="rays per step per GPU")
    ap.add_argument("--workload", default="cbvh.leaf", choices=list(WORKLOADS))
    ap.add_argument("--levels", default="6,3", help="subdivision level, compression level")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-others", action="store_true", help="skip the short tri / eager side runs")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-record (PCIe-inclusive) side measurement: under a profiler its chunk launches would mix into the kernel statistics")
    ap.add_argument("--inflight", type=int, default=4,
                    help="ray batches in flight: step s is enqueued on HIP stream s %% inflight (1 = one stream, strictly back-to-back)")
    ap.add_argument("--query", default="intersect", choices=["intersect", "occluded"],
                    help="closest hit (rtcIntersect1M on RTCRayHit[80B]) or any hit (rtcOccluded1M on RTCRay[48B]); occluded is meant "
                         "for --rays-kind secondary (shadow rays)")
    ap.add_argument("--rays-kind", default="random", choices=["random", "primary", "secondary"],
                    help="random: the metric's incoherent bbox rays; primary: BASELINE config 4, 1920x1080 camera rays of bomberman.ecs")
    return ap.parse_args()


def build_hash():
    """sha256 over the library's DEVICE sources (*.hip, *.hip.h, accel.h, trace.h) as recorded by embree-compressed_amd/Makefile at
    build time (lib/KERNEL_HASH): a change of host code does not make a kernel profile stale, a change of kernel code does."""
    try:
        return open(os.path.join(ROOT, "embree-compressed_amd", "lib", "KERNEL_HASH")).read().strip()
    except OSError:
        return None


def pmc_traffic(kernel_tag, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN_<workload>_pmc.json by tools/summarize_prof.py; FETCH_SIZE already doubled per the gfx950
    correction).  bench.py cannot collect PMC counters itself.  A summary is used only if it was taken on a library built
    from the same sources as the one being measured (its "build_hash"); otherwise traffic is null rather than stale."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload.replace('.', '_')}_pmc.json")))
    for path in reversed(files):
        try:
            d = json.load(open(path))
            if d.get("build_hash") != build_hash():
                continue
            rd = [v["read_bytes"] for k, v in d.get("FETCH_SIZE", {}).items() if kernel_tag in k and "false, false, true>" in k]
            wr = [v["write_bytes"] for k, v in d.get("WRITE_SIZE", {}).items() if kernel_tag in k and "false, false, true>" in k]
            if rd and wr:
                return rd[0] + wr[0], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def rocprof_kernel_us(kernel_tag, traffic_source):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary that belongs to the PMC
    summary `traffic_source` (same tools/profile_all.sh run, same build hash): what the live HIP-event figure must agree with."""
    if not traffic_source:
        return None
    path = os.path.join(ROOT, traffic_source.replace("_pmc.json", "_kernel_stats.csv"))
    try:
        for line in open(path):
            if kernel_tag in line and "false, false, true>" in line:
                cols = line.rsplit('",', 1)[1].split(",")  # Calls, TotalDurationNs, AverageNs, ...
                return float(cols[2]) / 1e3
    except (OSError, ValueError, IndexError):
        pass
    return None


def build_scene(rtc, local_rank, workload, mesh, levels):
    cfg, kind, _, _, _ = WORKLOADS[workload]
    dev = rtc.Device(f"gpu={local_rank},{cfg}")
    sc = rtc.Scene(dev)
    verts, fs, fi = mesh
    if kind == "tri":
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(*levels)
    sc.commit()
    return dev, sc


def cpu_baseline(sc, rtc, workload, mesh, levels, lo, hi, m, budget_s):
    """Oracle timed on the host cores (kind "port"): the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    _, kind, mode, _, _ = WORKLOADS[workload]
    if kind == "tri":
        orc = po.TriangleScene(mesh[0], rtc.fan_triangulate(mesh[1], mesh[2]), mode)
        what = "oracle BVH8/Triangle4v restatement"
    else:
        st = sc.stats()
        orc = po.SubdivScene(sc.accel_data(2), st["primBytes"], mode, levels[1], qnodes=sc.accel_data(0), root=sc.accel_root())
        what = "oracle restatement of CompressedBVHIntersector1 / GridSOAIntersector1 over the same leaf records"
    src = po.make_random_rays(m, lo, hi, seed=12345)
    total, spent, reps = 0, 0.0, 0
    while spent < budget_s and reps < 4096:
        work = src.copy()
        t0 = time.perf_counter()
        orc.intersect1M(work, nthreads=cores)
        spent += time.perf_counter() - t0
        total += m
        reps += 1
    one = src[: min(m, 200_000)].copy()
    t0 = time.perf_counter()
    orc.intersect1M(one, nthreads=1)
    per_thread = one.shape[0] / (time.perf_counter() - t0) / 1e6
    orc.free()
    # true-reference rates measured by the survey (the reference library built there, AVX2, 1 thread,
    # Xeon 2.1 GHz, same scene and ray generator)
    ref_1t = {"cbvh.leaf": 5.6, "cbvh.box": 7.0, "cbvh.grid": 7.5, "cbvh.full": 6.1, "eager": 10.1, "tri": 30.4}.get(workload)
    return {"value": total / spent / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {m} rays of the same generator (seed 12345), {what} (oracle/liboracle.so), {cores} pthreads, blocks of 1024",
            "per_thread_Mrays": per_thread,
            "ref_ratio_note": (f"kind 'port' = scalar C restatement of the reference's AVX2 path, not the reference library (it cannot be built or "
                               f"shipped here).  This host, 1 thread: {per_thread:.2f} Mrays/s; true reference per BASELINE.md section 2 (survey build, "
                               f"1 thread of a 2.1 GHz Xeon, same scene and rays): {ref_1t} Mrays/s -> restatement / reference = "
                               f"{per_thread / ref_1t:.2f} across the two machines") if ref_1t else None}


def inflight_profile(workload):
    """Summary of the committed rocprofv3 --kernel-trace of `bench.py --inflight 4` (profiles/rNN_<workload>_inflight.json by
    tools/summarize_inflight.py): per-kernel durations and how many traversal kernels ran concurrently, from the start / end
    stamps of the trace.  Same build-hash rule as pmc_traffic."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload.replace('.', '_')}_inflight.json")))
    for path in reversed(files):
        try:
            d = json.load(open(path))
            if d.get("build_hash") == build_hash():
                d["source"] = os.path.relpath(path, ROOT)
                return d
        except (OSError, ValueError):
            continue
    return None


def pcie_inclusive(sc, raygen, D, m, lo, hi, rank, reps=3):
    """The same workload through HOST pointers: one rtcIntersect1M call over m records in pageable host memory, i.e. staging
    copy + H2D + traversal + D2H + scatter of tfar / hit, wall clock around the call.  Reported next to `value`, never as it."""
    best = None
    for r in range(reps):
        rays = np.ascontiguousarray(raygen.make_random_rays(m, lo, hi, seed=D.batch_seed(rank, 900 + r)))
        t0 = time.perf_counter()
        sc.intersect1M(rays)  # uint8 [m, 80]: host pointer, 80-byte stride
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return m / best / 1e6


def run_loop(torch, dist, sc, dev, streams, bufs, K, W, world, occluded=False):
    """W untimed + K timed steps; step s goes to streams[s % len(streams)] (the library keeps per-launch scratch, so
    batches on different streams overlap: the drain of one batch runs under the start of the next).  Returns the
    host-clock time of the timed region and, for ONE stream, the HIP-event time per launch on that stream."""
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    trace = sc.occluded1M if occluded else sc.intersect1M
    ctx = getattr(sc, "bench_ctx", None)  # primary rays carry RTC_INTERSECT_CONTEXT_FLAG_COHERENT like viewer_stream_device.cpp:305

    def step(s):
        st = streams[s % len(streams)]
        dev.set_stream(st.cuda_stream)
        trace(bufs[s], ctx=ctx, check=False)

    # every stream (and every per-launch context of the library) is exercised before the clock starts: first use of a
    # HIP stream costs milliseconds.  The extra untimed steps re-trace warm-up batches, never timed ones.
    for s in range(W):
        step(s)
    for s in range(W, 2 * len(streams) if len(streams) > 1 else 0):
        st = streams[s % len(streams)]
        dev.set_stream(st.cuda_stream)
        trace(bufs[s % max(W, 1)], ctx=ctx, check=False)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(streams[0])
    for s in range(W, W + K):
        step(s)
    ev1.record(streams[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    dev.check("timed region")
    dev.set_stream(streams[0].cuda_stream)
    return t1 - t0, (ev0.elapsed_time(ev1) / K if len(streams) == 1 else None)


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # BENCH_REHEARSE_GLOO=1: rehearsal of the N>1 launch on a ONE-GPU box (every rank on cuda:0, gloo for the barrier and
    # the MAX over ranks).  The numbers of such a run mean nothing; it exists to exercise rank handling end to end.
    rehearse = os.environ.get("BENCH_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("embree-compressed_amd")
    rtc = pkg.rtc
    raygen = importlib.import_module("embree-compressed_amd.raygen")
    D = importlib.import_module("embree-compressed_amd.dist")

    d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
    mesh = (d["verts"], d["face_sizes"], d["face_index"])
    lo, hi = mesh[0].min(0), mesh[0].max(0)  # bbox of the control vertices, like prepareRandomRays (viewer_device.cpp:394-429)
    levels = tuple(int(x) for x in args.levels.split(","))

    nfl = max(1, args.inflight)
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    m, K, W = args.rays, args.steps, args.warmup
    primary = None
    if args.rays_kind == "primary":
        primary = raygen.make_primary_rays()
        m = primary.shape[0]

    occluded = args.query == "occluded"

    def measure(workload, K, W):
        nonlocal m
        dev, sc = build_scene(rtc, local_rank, workload, mesh, levels)
        dev.set_stream(stream.cuda_stream)
        sc.bench_ctx = rtc.make_context(coherent=True) if args.rays_kind == "primary" else None
        # distinct batches per step and per rank, generated on the host, resident in HBM before timing starts
        nb = 2 * (K + W) + 1 if nfl > 1 else K + W + 1
        if args.rays_kind == "secondary":
            # config 5, wavefront style: trace the camera frame once on THIS scene, record one bounce ray and one shadow ray
            # per hit (raygen.make_secondary_rays); every step traces a fresh copy of the recorded batch
            first = torch.from_numpy(raygen.make_primary_rays()).to("cuda")
            sc.intersect1M(first)
            dev.synchronize()
            bounce, shadow = raygen.make_secondary_rays(first.cpu().numpy(), seed=11 + rank)
            tmpl = shadow if occluded else bounce
            m = tmpl.shape[0]
            bufs = [torch.from_numpy(tmpl).to("cuda") for s in range(nb)]
        elif primary is not None:  # the same camera frame every step, but a fresh copy (a trace modifies rays in place)
            bufs = [torch.from_numpy(primary[:, :48].copy() if occluded else primary).to("cuda") for s in range(nb)]
        else:
            bufs = [torch.from_numpy(np.ascontiguousarray(raygen.make_random_rays(m, lo, hi, seed=D.batch_seed(rank, s))[:, :48 if occluded else 80])).to("cuda")
                    for s in range(nb)]
        torch.cuda.synchronize()
        cnt = (sc.occluded1M_counted if occluded else sc.intersect1M_counted)(bufs[nb - 1], ctx=sc.bench_ctx)  # extra batch: work counters -> algorithmic bytes per ray
        # pass 1, one stream: the kernel alone, HIP-event time per launch (the roofline figure; agrees with rocprofv3
        # --kernel-trace of `bench.py --inflight 1`)
        elapsed1, kernel_ms = run_loop(torch, dist, sc, dev, streams[:1], bufs, K, W, world, occluded)
        if occluded:
            hits = int(torch.isneginf(bufs[W].view(torch.float32)[:, 8]).sum().item())
        else:
            hits = int((bufs[W].view(torch.int32)[:, 18] != -1).sum().item())
        # pass 2 (the reported value), fresh batches: `inflight` batches in flight on as many streams
        elapsed = elapsed1
        if nfl > 1:
            elapsed, _ = run_loop(torch, dist, sc, dev, streams, bufs[K + W:], K, W, world, occluded)
        return dev, sc, cnt, elapsed, kernel_ms, hits, elapsed1

    dev, sc, cnt, elapsed, kernel_ms, hits, elapsed1 = measure(args.workload, K, W)
    st = sc.stats()
    rate, worst = D.whole_job_rate(m * K, elapsed, world, device="cpu" if rehearse else "cuda")
    n_node = cnt["nodeVisits"] / max(cnt["rays"], 1)
    n_prim = cnt["primTests"] / max(cnt["rays"], 1)
    n_inner = cnt["innerVisits"] / max(cnt["rays"], 1)
    # implementation's own visits x record sizes; a subdiv leaf visit is priced at its whole record
    io_bytes = 48 + 4 if occluded else IO_BYTES_PER_RAY  # any hit: RTCRay read, tfar written
    bytes_per_ray = io_bytes + n_node * st["nodeBytes"] + n_prim * st["primBytes"]

    if rank == 0:
        _, _, _, tag, desc = WORKLOADS[args.workload]
        achieved = bytes_per_ray * m / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, None)
        if m == 1_000_000 and levels == (6, 3) and args.rays_kind == "random" and not occluded:  # the committed PMC passes were taken on this exact workload
            traffic, traffic_src = pmc_traffic(tag, args.workload)
        out = {
            "metric": "Mrays/s (incoherent) on bomberman displaced-subdiv scene, device-resident ray batches",
            "value": rate / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": worst / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; {m} "
                                   + {"random": "random incoherent rays per step (drand48 LCG, bbox-uniform endpoints), ",
                                      "primary": "coherent primary rays per step (config 4: 1920x1080, bomberman.ecs camera, 8x8 tile order), ",
                                      "secondary": "incoherent rays recorded from the config-4 camera frame (config 5: one bounce ray / one shadow ray "
                                                   "per primary hit, tnear 0.001), "}[args.rays_kind]
                                   + ("rtcOccluded1M on device-resident RTCRay[48B]" if occluded else "rtcIntersect1M on device-resident RTCRayHit[80B]"),
                       "rays_per_step_per_gpu": m, "accel_kind": st["accelKind"], "bvh_nodes": st["nodeCount"], "leaf_records": st["primCount"],
                       "leaf_record_bytes": st["primBytes"], "accel_bytes": st["totalBytes"], "hits_first_timed_batch": hits,
                       "batches_in_flight": nfl,
                       "in_flight_note": (f"step s is enqueued on HIP stream s % {nfl}; value = K steps / wall time of the region. "
                                          "One stream, strictly back-to-back (the roofline pass of this same run): "
                                          f"{m * K / elapsed1 / 1e6:.1f} Mrays/s on this rank") if nfl > 1 else "one stream, strictly back-to-back",
                       "sharding": f"replicated accel, {world} independent ray shards, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_ms, "kernel": tag,
                         "kernel_us_rocprofv3": rocprof_kernel_us(tag, traffic_src) if (m == 1_000_000 and levels == (6, 3) and args.rays_kind == "random" and not occluded) else None,
                         "aggregate_frac_in_flight": bytes_per_ray * m * K / elapsed / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_ray": bytes_per_ray, "nodes_per_ray": n_node, "leaf_visits_per_ray": n_prim,
                         "inner_steps_per_ray": n_inner, "node_bytes": st["nodeBytes"], "leaf_bytes": st["primBytes"],
                         "frac_note": "achieved / frac price ALGORITHMIC bytes (SURVEY.md 8d: every node / leaf record a ray visits, L1 / L2 hits "
                                      "included) against the HBM peak; `traffic` is what the PMC counters saw reach HBM per launch",
                         "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "in_flight_profile": inflight_profile(args.workload) if (m == 1_000_000 and levels == (6, 3) and args.rays_kind == "random" and not occluded) else None},
        }
        if world == 1 and args.rays_kind == "random" and not occluded and not args.no_pcie:
            out["config"]["pcie_inclusive_Mrays"] = pcie_inclusive(sc, raygen, D, m, lo, hi, rank)
            out["config"]["pcie_inclusive_note"] = ("one rtcIntersect1M call on pageable HOST records, best of 3: chunked pipeline of gather into pinned memory "
                                                    "(host thread pool) / H2D / traversal / D2H on two alternating streams / scatter of tfar + hit "
                                                    "(rt_trace.cpp trace_host_pipelined); `value` is the device-resident rate")
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(sc, rtc, args.workload, mesh, levels, lo, hi, m, args.cpu_seconds)
    sc.release()
    dev.release()

    if world == 1 and not args.no_others and args.workload == "cbvh.leaf":
        others = {}
        for w2 in ("tri", "eager"):
            d2, s2, c2, e2, k2, h2, _ = measure(w2, 5, 2)
            others[w2] = {"Mrays_per_s": m * 5 / e2 / 1e6, "kernel_ms": k2, "hits": h2, "steps": 5}
            s2.release()
            d2.release()
        out["config"]["other_workloads"] = others

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
