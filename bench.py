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

The timed region of K steps is REPEATED (`--repeats`, default: as many as make the timed regions add up to >= 60 ms, 5..40): every
repeat restores the K batches from pristine device copies (untimed), is bracketed by barrier + synchronize like the single region
of the contract, and `value` / `ms_per_step` are those of the MEDIAN repeat (min / max next to them): a 2-4 ms region alone was
what round 2's headline hung on.

The JSON line also carries
  one_stream       the same K steps strictly back-to-back on ONE stream (what a caller sees that does not pipeline batches)
  roofline         algorithmic bytes per launch / mean kernel time (HIP events on the launch stream) vs 8 TB/s; `bytes_per_ray_addressed`
                   prices a blob visit at the fields it actually reads instead of the whole record
  roofline_scaled  the same measurement on the scaled scene (--scaled-levels, default 8,3: 381 MB of blobs, larger than L2 + Infinity
                   Cache), with the PMC traffic of the committed profile of that scene
  cpu_baseline     the oracle (a scalar C port of the reference's AVX2 path) on the host cores, bounded sample
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
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps each (0 = auto: >= 60 ms of timed regions in total, 5..40)")
    ap.add_argument("--scaled-levels", default="8,3", help="levels of the scaled scene measured for roofline_scaled on the default workload ('none' = skip)")
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


def pmc_traffic(kernel_tag, workload, suffix=""):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN_<workload>_pmc.json by tools/summarize_prof.py; FETCH_SIZE already doubled per the gfx950
    correction).  bench.py cannot collect PMC counters itself.  A summary is used only if it was taken on a library built
    from the same sources as the one being measured (its "build_hash"); otherwise traffic is null rather than stale."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload.replace('.', '_')}{suffix}_pmc.json")))
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


def run_loop(torch, dist, sc, dev, streams, bufs, src, K, W, world, occluded=False, repeats=0):
    """W untimed + R x K timed steps; step s goes to streams[s % len(streams)] (the library keeps per-launch scratch, so
    batches on different streams overlap: the drain of one batch runs under the start of the next).  Every repeat first
    restores bufs[W .. W+K) from the pristine copies src[...] (untimed; a trace modifies rays in place), then times EXACTLY K
    steps bracketed by barrier + synchronize on both sides.  Returns the host-clock times of the repeats and, for ONE stream,
    the HIP-event times per launch on that stream."""
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
    times, ev_ms = [], []
    target_s, done = 0.060, 0
    while True:
        if done:  # fresh records for the next repeat
            with torch.cuda.stream(streams[0]):
                for s in range(W, W + K):
                    bufs[s].copy_(src[s])
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
        times.append(t1 - t0)
        if len(streams) == 1:
            ev_ms.append(ev0.elapsed_time(ev1) / K)
        done += 1
        if repeats == 0:  # auto: from the first region's length (same on every rank: MAX over ranks)
            first = times[0]
            if world > 1:
                t = torch.tensor([first], dtype=torch.float64, device="cpu" if os.environ.get("BENCH_REHEARSE_GLOO") == "1" else "cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                first = float(t.item())
            repeats = int(min(40, max(5, -(-target_s // max(first, 1e-6)))))
        if done >= repeats:
            break
    dev.check("timed region")
    dev.set_stream(streams[0].cuda_stream)
    return times, ev_ms


def median(v):
    w = sorted(v)
    return w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])


def leaf_record_bytes(st, levels):
    """Bytes a leaf visit is priced at in the algorithmic figure: the record's CONTENT.  Triangle records and grid cells: their size.
    A cBVH blob: line 0 + ids / uv + node words + cells (or grid) + tail, WITHOUT the padding that rounds the round-3 stride up to a
    multiple of 128 bytes (accel.h; C = 3 leaf mode: 448 of the 512-byte stride - the figure round 2 used for its 448-byte record)."""
    mode = {3: "box", 4: "leaf", 5: "grid", 7: "full"}.get(st["accelKind"])
    if mode is None:
        return st["primBytes"]
    C = levels[1]
    payload = (160 + (4 ** C - 1) // 3 * (96 if mode == "full" else 4) + 15) // 16 * 16
    extra = 2 * 4 ** C if mode == "leaf" else (12 * (2 ** C + 1) ** 2 if mode == "grid" else 0)
    return (payload + extra + 15) // 16 * 16 + 64


def addressed_bytes_per_ray(cnt, st, occluded, levels):
    """Second bytes figure (VERDICT r2 #2d): a leaf visit priced at the fields it READS instead of the whole padded record.
    cBVH blob (accel.h round-3 layout; counters of the instrumented twin: leafVisits = blob visits, primTests = visits that pass
    the frustum test, innerVisits = quadtree nodes entered + cells tested, hits):
      every visit        space 36 + frustum box 40 + root word 4                         = 80 B   (all in line 0 of the blob)
      a walk (passed)    + proj 36 + rcp_edges / extent 8                                = 44 B
      per node entered   <= 16 B (four child words; 8 B for the four 2-byte cells of a last-level node)
      per hit committed  uv window 16 + ids 8                                           = 24 B
    Other leaves: the record sizes of the first figure (a triangle record / a grid cell is read whole)."""
    rays = max(cnt["rays"], 1)
    io = 48 + 4 if occluded else IO_BYTES_PER_RAY
    n_node = cnt["nodeVisits"] / rays
    if st["accelKind"] in (3, 4, 5, 7):
        visits, walks, inner, hits = cnt["leafVisits"] / rays, cnt["primTests"] / rays, cnt["innerVisits"] / rays, cnt["hits"] / rays
        per_node = 96 if st["accelKind"] == 7 else 16
        return io + n_node * st["nodeBytes"] + visits * 80 + walks * 44 + inner * per_node + hits * 24
    return io + n_node * st["nodeBytes"] + (cnt["primTests"] / rays) * st["primBytes"]


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
    red_dev = "cpu" if rehearse else "cuda"

    def measure(workload, K, W, lv=levels, repeats=args.repeats):
        """One scene, two timed passes.  Returns a dict of everything the JSON line needs about it."""
        nonlocal m
        t_build = time.perf_counter()
        dev, sc = build_scene(rtc, local_rank, workload, mesh, lv)
        t_build = time.perf_counter() - t_build
        dev.set_stream(stream.cuda_stream)
        sc.bench_ctx = rtc.make_context(coherent=True) if args.rays_kind == "primary" else None
        # distinct batches per step and per rank, generated on the host, resident in HBM before timing starts: pristine copies
        # `src` and the working records `bufs` the traces modify in place
        nb = K + W + 1
        if args.rays_kind == "secondary":
            # config 5, wavefront style: trace the camera frame once on THIS scene, record one bounce ray and one shadow ray
            # per hit (raygen.make_secondary_rays); every step traces a fresh copy of the recorded batch
            first = torch.from_numpy(raygen.make_primary_rays()).to("cuda")
            sc.intersect1M(first)
            dev.synchronize()
            bounce, shadow = raygen.make_secondary_rays(first.cpu().numpy(), seed=11 + rank)
            tmpl = shadow if occluded else bounce
            m = tmpl.shape[0]
            one = torch.from_numpy(tmpl).to("cuda")
            src = [one for s in range(nb)]
        elif primary is not None:  # the same camera frame every step, but a fresh copy (a trace modifies rays in place)
            one = torch.from_numpy(primary[:, :48].copy() if occluded else primary).to("cuda")
            src = [one for s in range(nb)]
        else:
            src = [torch.from_numpy(np.ascontiguousarray(raygen.make_random_rays(m, lo, hi, seed=D.batch_seed(rank, s))[:, :48 if occluded else 80])).to("cuda")
                   for s in range(nb)]
        bufs = [t.clone() for t in src]
        torch.cuda.synchronize()
        cnt = (sc.occluded1M_counted if occluded else sc.intersect1M_counted)(bufs[nb - 1], ctx=sc.bench_ctx)  # extra batch: work counters -> algorithmic bytes per ray
        # pass 1, one stream: the kernel alone, HIP-event time per launch (the roofline figure; agrees with rocprofv3
        # --kernel-trace of `bench.py --inflight 1`)
        t1, ev1 = run_loop(torch, dist, sc, dev, streams[:1], bufs, src, K, W, world, occluded, repeats)
        if occluded:
            hits = int(torch.isneginf(bufs[W].view(torch.float32)[:, 8]).sum().item())
        else:
            hits = int((bufs[W].view(torch.int32)[:, 18] != -1).sum().item())
        # pass 2 (the reported value), fresh records: `inflight` batches in flight on as many streams
        tN = t1
        if nfl > 1:
            for s in range(nb):
                bufs[s].copy_(src[s])
            torch.cuda.synchronize()
            tN, _ = run_loop(torch, dist, sc, dev, streams, bufs, src, K, W, world, occluded, repeats)
        # per repeat: MAX over ranks of the region, then the median / min / max of the repeats
        worstN = [D.max_over_ranks(t, world, red_dev) for t in tN]
        worst1 = [D.max_over_ranks(t, world, red_dev) for t in t1]
        total_rays = D.sum_over_ranks(m * K, world, red_dev)
        st = sc.stats()
        rays = max(cnt["rays"], 1)
        n_node = cnt["nodeVisits"] / rays
        # leaf records visited per ray: triangle records tested, or subdivision leaves entered (grid cells / cBVH blobs)
        n_leaf = (cnt["primTests"] if st["accelKind"] in (1, 2) else cnt["leafVisits"]) / rays
        io_bytes = 48 + 4 if occluded else IO_BYTES_PER_RAY  # any hit: RTCRay read, tfar written
        return {"dev": dev, "sc": sc, "cnt": cnt, "st": st, "hits": hits, "m": m, "K": K, "build_s": t_build,
                "regionN": worstN, "region1": worst1, "kernel_ms": median(ev1), "kernel_ms_minmax": (min(ev1), max(ev1)),
                "total_rays": total_rays, "n_node": n_node, "n_leaf": n_leaf, "n_inner": cnt["innerVisits"] / rays,
                # implementation's own visits x record sizes; a subdiv leaf visit is priced at its whole record
                "leaf_bytes": leaf_record_bytes(st, lv),
                "bytes_per_ray": io_bytes + n_node * st["nodeBytes"] + n_leaf * leaf_record_bytes(st, lv),
                "bytes_per_ray_addressed": addressed_bytes_per_ray(cnt, st, occluded, lv)}

    def roofline_of(R, tag, traffic, traffic_src, with_profiles):
        kernel_ms = R["kernel_ms"]
        achieved = R["bytes_per_ray"] * R["m"] / (kernel_ms * 1e-3) / 1e9
        med = median(R["regionN"])
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_ms, "kernel_ms_min_max": list(R["kernel_ms_minmax"]), "kernel": tag,
                "kernel_us_rocprofv3": rocprof_kernel_us(tag, traffic_src) if with_profiles else None,
                "aggregate_frac_in_flight": R["bytes_per_ray"] * R["m"] * R["K"] / med / 1e9 / HBM_PEAK_GBS,
                "bytes_per_ray": R["bytes_per_ray"], "nodes_per_ray": R["n_node"], "leaf_visits_per_ray": R["n_leaf"],
                "inner_steps_per_ray": R["n_inner"], "node_bytes": R["st"]["nodeBytes"], "leaf_bytes": R["leaf_bytes"], "leaf_stride_bytes": R["st"]["primBytes"],
                "bytes_per_ray_addressed": R["bytes_per_ray_addressed"],
                "frac_addressed": R["bytes_per_ray_addressed"] * R["m"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_note": "achieved / frac price ALGORITHMIC bytes (SURVEY.md 8d: every node / leaf record a ray visits, L1 / L2 hits "
                             "included; a blob visit = the whole padded record, an upper bound - 28 % of the visits end at the frustum test, a walk reads a fraction of the record) against the HBM "
                             "peak; bytes_per_ray_addressed / frac_addressed price a blob visit at the fields it reads (bench.py addressed_bytes_per_ray); "
                             "`traffic` is what the PMC counters saw reach HBM per launch",
                "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}

    R = measure(args.workload, K, W)
    dev, sc, st, m = R["dev"], R["sc"], R["st"], R["m"]
    medN, med1 = median(R["regionN"]), median(R["region1"])
    rate = R["total_rays"] / medN
    metric_cfg = m == 1_000_000 and levels == (6, 3) and args.rays_kind == "random" and not occluded  # the committed profiles were taken on this exact workload

    if rank == 0:
        _, _, _, tag, desc = WORKLOADS[args.workload]
        traffic, traffic_src = pmc_traffic(tag, args.workload) if metric_cfg else (None, None)
        out = {
            "metric": "Mrays/s (incoherent) on bomberman displaced-subdiv scene, device-resident ray batches",
            "value": rate / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": medN / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "repeats": len(R["regionN"]),
            "timed_ms_total": sum(R["regionN"]) * 1e3,
            "value_min_max": [R["total_rays"] / max(R["regionN"]) / 1e6, R["total_rays"] / min(R["regionN"]) / 1e6],
            "value_note": f"median of {len(R['regionN'])} timed regions of exactly {K} steps each (barrier + synchronize on both sides of every region, MAX over ranks per region)",
            "one_stream": {"value": R["total_rays"] / med1 / 1e6, "unit": "Mrays/s", "ms_per_step": med1 / K * 1e3, "repeats": len(R["region1"]),
                           "value_min_max": [R["total_rays"] / max(R["region1"]) / 1e6, R["total_rays"] / min(R["region1"]) / 1e6],
                           "note": "the same K steps strictly back-to-back on ONE stream: what a caller gets that does not keep several batches in flight"},
            "config": {"workload": f"{args.workload}: {desc}; {m} "
                                   + {"random": "random incoherent rays per step (drand48 LCG, bbox-uniform endpoints), ",
                                      "primary": "coherent primary rays per step (config 4: 1920x1080, bomberman.ecs camera, 8x8 tile order), ",
                                      "secondary": "incoherent rays recorded from the config-4 camera frame (config 5: one bounce ray / one shadow ray "
                                                   "per primary hit, tnear 0.001), "}[args.rays_kind]
                                   + ("rtcOccluded1M on device-resident RTCRay[48B]" if occluded else "rtcIntersect1M on device-resident RTCRayHit[80B]"),
                       "levels": list(levels), "rays_per_step_per_gpu": m, "accel_kind": st["accelKind"], "bvh_nodes": st["nodeCount"], "leaf_records": st["primCount"],
                       "leaf_record_bytes": st["primBytes"], "accel_bytes": st["totalBytes"], "commit_seconds": R["build_s"], "hits_first_timed_batch": R["hits"],
                       "batches_in_flight": nfl,
                       "in_flight_note": (f"step s is enqueued on HIP stream s % {nfl}; value = K steps / wall time of the region; "
                                          "see one_stream for the strictly back-to-back rate") if nfl > 1 else "one stream, strictly back-to-back",
                       "sharding": f"replicated accel, {world} independent ray shards, no collective"
                                   + ("; device-resident, process-per-GPU batches are the only path that can scale near-linearly: ONE host-pointer call "
                                      "through rtcNewDevice(\"gpus=...\") is bound by the caller's gather / scatter (345 Mrays/s per process, DESIGN.md)" if world > 1 else "")},
            "roofline": roofline_of(R, tag, traffic, traffic_src, metric_cfg),
        }
        out["roofline"]["in_flight_profile"] = inflight_profile(args.workload) if metric_cfg else None
        if world == 1 and args.rays_kind == "random" and not occluded and not args.no_pcie:
            out["config"]["pcie_inclusive_Mrays"] = pcie_inclusive(sc, raygen, D, m, lo, hi, rank)
            out["config"]["pcie_inclusive_note"] = ("one rtcIntersect1M call on pageable HOST records, best of 3: chunked pipeline of gather into pinned memory "
                                                    "(host thread pool) / H2D / traversal / D2H on two alternating streams / scatter of tfar + hit "
                                                    "(rt_trace.cpp trace_host_pipelined); `value` is the device-resident rate")
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(sc, rtc, args.workload, mesh, levels, lo, hi, m, args.cpu_seconds)
    sc.release()
    dev.release()
    del R

    default_line = world == 1 and args.workload == "cbvh.leaf" and metric_cfg
    if default_line and not args.no_others:
        others = {}
        for w2 in ("tri", "eager"):
            R2 = measure(w2, 20, 3, repeats=5)  # (regions of 5 steps read 20-25 % low: four batches in flight need a few steps to overlap)
            others[w2] = {"Mrays_per_s": R2["total_rays"] / median(R2["regionN"]) / 1e6, "kernel_ms": R2["kernel_ms"], "hits": R2["hits"], "steps": 20, "repeats": 5}
            R2["sc"].release()
            R2["dev"].release()
            del R2
        out["config"]["other_workloads"] = others

    # the scaled scene ("report bomberman as-is and one scaled --subdLvl variant"): same rays, same kernel, an accel larger
    # than the 4 MiB L2s and the 256 MiB Infinity Cache, so that the blobs really come from HBM
    if default_line and args.scaled_levels != "none":
        lv2 = tuple(int(x) for x in args.scaled_levels.split(","))
        Ks = min(K, 20)  # (regions of 10 steps read the in-flight rate ~15 % low)
        R3 = measure(args.workload, Ks, 2, lv=lv2, repeats=5)
        _, _, _, tag, desc = WORKLOADS[args.workload]
        sfx = f"_L{lv2[0]}"
        tr3, src3 = pmc_traffic(tag, args.workload, suffix=sfx)
        rs = roofline_of(R3, tag, tr3, src3, True)
        rs.update({"levels": list(lv2), "accel_bytes": R3["st"]["totalBytes"], "leaf_records": R3["st"]["primCount"], "commit_seconds": R3["build_s"],
                   "steps": Ks, "repeats": len(R3["regionN"]), "hits_first_timed_batch": R3["hits"],
                   "value_in_flight": R3["total_rays"] / median(R3["regionN"]) / 1e6, "value_one_stream": R3["total_rays"] / median(R3["region1"]) / 1e6,
                   "note": f"bomberman at subdivision level {lv2[0]} / compression {lv2[1]}: {R3['st']['totalBytes'] / 1e6:.0f} MB accel (> 8 x 4 MiB L2 + 256 MiB "
                           "Infinity Cache); same ray generator, same kernel; traffic = PMC counters of the committed profile of this scene"})
        out["roofline_scaled"] = rs
        R3["sc"].release()
        R3["dev"].release()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
