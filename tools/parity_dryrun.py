#!/usr/bin/env python3
"""CPU dry run of the fork-parity comparison of the GPU suite (tests/helpers.py: check_fork_parity, PARITY leg).

The HIP kernels are byte-identical to the oracle run in PRODUCT arithmetic (the regression leg of every GPU parity test), so
oracle(product arithmetic) vs oracle(reference arithmetic) on the CPU reproduces, without a GPU, the classified figures the GPU
tests will see: hit/miss flips, ID flips, records beyond 1e-4 and their classes (subcell / neighbour / far).  Used to measure the
floors the tests bound (FORK_BEYOND_FLOOR, per-test `beyond_floor`) and to check a change of the classification before a GPU run.
Test infrastructure: uses oracle/ and a host-only device (`gpu=none`: builders only, no traversal).

    python tools/parity_dryrun.py [--quick]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po  # noqa: E402
from helpers import FORK_ORACLE_MODE, ORDERED_FORK, assert_fork_classes  # noqa: E402

rtc = importlib.import_module("embree-compressed_amd").rtc
rg = importlib.import_module("embree-compressed_amd.raygen")
d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
V, FS, FI = d["verts"], d["face_sizes"], d["face_index"]
QUICK = "--quick" in sys.argv
MODES = ("bvh4.compressed.box", "bvh4.compressed.leaf", "bvh4.compressed.full", "bvh4.compressed.grid")


def scene(accel, verts, fs, fi, L, C, extra=None, displacement=None, flags=0, level=None):
    dev = rtc.Device(f"gpu=none,subdiv_accel={accel}")
    sc = rtc.Scene(dev, flags)
    kw = {} if level is None else {"level": level}
    if extra is not None and extra[0] == "first":
        sc.add_triangles(*extra[1:])
    g = sc.add_subdiv(verts, fs, fi, displacement=displacement, **kw)
    if extra is not None and extra[0] == "last":
        sc.add_triangles(*extra[1:])
    sc.set_levels(L, C)
    sc.commit()
    same = accel in ORDERED_FORK
    orc = po.SubdivScene(sc.accel_data(2), sc.stats()["primBytes"], FORK_ORACLE_MODE[accel], C, qnodes=sc.accel_data(0) if same else None,
                         root=sc.accel_root() if same else None)
    return dev, sc, orc, g


def both(orcs, src, nthreads=8):
    """records of the oracle(s) in product arithmetic (= the kernels') and in reference arithmetic"""
    a, b = src.copy(), src.copy()
    with po.fork_arith(1):
        for o in orcs:
            o.intersect1M(a, nthreads=nthreads)
    for o in orcs:
        o.intersect1M(b, nthreads=nthreads)
    return a, b


def report(what, accel, a, b, cell, **kw):
    try:
        assert_fork_classes(a, b, accel, what=what, cell=cell, **kw)
    except AssertionError as e:
        print("   ^^^ WOULD FAIL:", str(e)[:200])


def bomberman_random(L, C, n, seed):
    for accel in MODES:
        dev, sc, orc, _ = scene(accel, V, FS, FI, L, C)
        src = po.make_random_rays(n, V.min(0), V.max(0), seed=seed, double_eval=True)
        a, b = both([orc], src)
        report(f"{accel} L{L} C{C}", accel, a, b, 2.0 ** -L)
        orc.free(); sc.release(); dev.release()


def primary(L, C, w, h):
    for accel in ("bvh4.compressed.leaf", "bvh4.compressed.full"):
        dev, sc, orc, _ = scene(accel, V, FS, FI, L, C)
        raw = rg.make_primary_rays(w, h)
        src = rtc.aligned_rayhits(raw.shape[0])
        src[:] = raw.reshape(-1).view(rtc.RAYHIT_DTYPE)
        a, b = both([orc], src)
        report(f"primary {w}x{h} {accel}", accel, a, b, 2.0 ** -L)
        orc.free(); sc.release(); dev.release()


def secondary():
    import test_gpu_secondary as ts
    for accel in MODES:
        dev, sc, orc, _ = scene(accel, V, FS, FI, 6, 3)
        raw = rg.make_primary_rays(640, 360)
        prim = rtc.aligned_rayhits(raw.shape[0])
        prim[:] = raw.reshape(-1).view(rtc.RAYHIT_DTYPE)
        with po.fork_arith(1):
            orc.intersect1M(prim, nthreads=8)
        src, _ = ts._bounce(rtc, prim, seed=11)
        a, b = both([orc], src)
        report(f"secondary {accel}", accel, a, b, 2.0 ** -6, hitmiss_max=8, beyond_floor=0.004)
        orc.free(); sc.release(); dev.release()


def degenerate():
    import test_gpu_degenerate_rays as td
    for accel in MODES:
        dev, sc, orc, _ = scene(accel, V, FS, FI, 5, 3)
        src = td._degenerate_rays(po, rtc, V.min(0), V.max(0))
        a, b = both([orc], src)
        report(f"{accel} degenerate rays", accel, a, b, 2.0 ** -5, beyond_floor=0.01)
        orc.free(); sc.release(); dev.release()


def prism():
    Vp = np.array([[0, 0, 0], [2, 0, 0], [1, 1.7, 0], [0, 0, 3], [2, 0, 3], [1, 1.7, 3]], np.float32)
    F = [(0, 2, 1), (3, 4, 5), (0, 1, 4, 3), (1, 2, 5, 4), (2, 0, 3, 5)]
    fs = np.array([len(f) for f in F], np.uint32)
    fi = np.concatenate([np.array(f, np.uint32) for f in F])
    for accel in MODES:
        dev, sc, orc, _ = scene(accel, Vp, fs, fi, 4, 2)
        src = po.make_random_rays(200_000, Vp.min(0) - 0.5, Vp.max(0) + 0.5, seed=8)
        a, b = both([orc], src)
        report(f"prism {accel}", accel, a, b, 2.0 ** -3)
        orc.free(); sc.release(); dev.release()


def cubes():
    import test_gpu_subdiv as tg
    from helpers import fill_rays, random_rays_np
    v, fs, fi = tg._cube()
    gv = np.array([[-10, -2, -10], [-10, -2, 10], [10, -2, -10], [10, -2, 10]], np.float32)
    gt = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    cb = tg.DISPL_CB(tg._displace)
    for accel in ("bvh4.compressed.leaf", "bvh4.compressed.box", "bvh4.compressed.grid"):
        dev, sc, orc, g_sub = scene(accel, v, fs, fi, 5, 4, extra=("last", gv, gt), displacement=cb, flags=rtc.RTC_SCENE_FLAG_ROBUST)
        orc_t = po.TriangleScene(gv, gt, 0, np.full(2, 1, np.uint32), np.arange(2, dtype=np.uint32))
        org, dd = random_rays_np(200_000, np.array([-4, -3, -4], np.float32), np.array([4, 4, 4], np.float32), 77)
        src = rtc.aligned_rayhits(200_000)
        fill_rays(src, org, dd)
        a, b = both([orc_t, orc], src)
        report(f"displaced cube {accel}", accel, a, b, 2.0 ** -5)
        orc.free(); orc_t.free(); sc.release(); dev.release()
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_tutorial.so")
    if not os.path.exists(lib):
        return
    import ctypes as C
    shader = C.CDLL(lib).ref_tutorial_displacementFunction
    v = np.array([[-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]], np.float32)
    fi = np.array([0, 4, 5, 1, 1, 5, 6, 2, 2, 6, 7, 3, 0, 3, 7, 4, 4, 7, 6, 5, 0, 1, 2, 3], np.uint32)
    for accel in MODES:
        dev, sc, orc, g_sub = scene(accel, v, fs, fi, 6, 4, extra=("first", gv, gt), displacement=shader, flags=rtc.RTC_SCENE_FLAG_ROBUST, level=256.0)
        orc_t = po.TriangleScene(gv, gt, 0, np.full(2, 0, np.uint32), np.arange(2, dtype=np.uint32))
        cam = rg.make_primary_rays(512, 512, frm=(1.5, 1.5, -1.5), to=(0, 0, 0), fov=90.0)
        rnd = po.make_random_rays(1_000_000, np.array([-10, -2, -10], np.float32), np.array([10, 2.5, 10], np.float32), seed=2)
        src = rtc.aligned_rayhits(cam.shape[0] + rnd.shape[0])
        src[: cam.shape[0]] = cam.reshape(-1).view(rtc.RAYHIT_DTYPE)
        src[cam.shape[0]:] = rnd
        a, b = both([orc_t, orc], src)
        report(f"config 3 {accel}", accel, a, b, 2.0 ** -6)
        orc.free(); orc_t.free(); sc.release(); dev.release()


if __name__ == "__main__":
    bomberman_random(3, 2, 100_000, 0)
    bomberman_random(6, 3, 1_000_000, 0)
    if not QUICK:
        for L, C in ((4, 1), (5, 4), (6, 5)):
            bomberman_random(L, C, 200_000, 3)
        primary(5, 3, 480, 270)
        primary(6, 3, 1920, 1080)
        prism()
        cubes()
        secondary()
        degenerate()
