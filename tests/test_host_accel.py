"""CPU tests of the host side of the product: C-ABI surface, error conventions, BVH8 builder + node codec.

These run on a `gpu=none` device (host-only object model: scenes can be built and inspected, every trace
call raises RTC_ERROR_INVALID_OPERATION).  No compute on a GPU is attempted here.
"""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NODE_DT = np.dtype([("origin", "<f4", 3), ("exp", "u1", 3), ("pad", "u1"), ("child", "<u4", 8), ("q", "u1", (6, 8))])
TRI_DT = np.dtype([("a", "<f4", 3), ("geomID", "<u4"), ("b", "<f4", 3), ("primID", "<u4"), ("c", "<f4", 3), ("pad", "<u4")])
LEAF, EMPTY = 0x80000000, 0xFFFFFFFF


def test_abi_exports_every_declared_symbol(rtc):
    """The shared library exports every RTC_API function include/embree3/*.h declares."""
    names = set()
    for h in ("rtcore.h", "rtcore_amd.h"):
        txt = open(os.path.join(ROOT, "include", "embree3", h)).read()
        names |= set(re.findall(r"RTC_API\s+[^;(]*?\b(rtc\w+)\s*\(", txt))
    assert len(names) > 90
    lib = rtc.lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts(rtc):
    assert rtc.RAYHIT_DTYPE.itemsize == 80 and rtc.RAY_DTYPE.itemsize == 48
    assert rtc.RAYHIT_DTYPE.fields["tfar"][1] == 32 and rtc.RAYHIT_DTYPE.fields["Ng_x"][1] == 48
    assert rtc.RAYHIT_DTYPE.fields["geomID"][1] == 72 and rtc.RAYHIT_DTYPE.fields["instID"][1] == 76
    assert NODE_DT.itemsize == 96 and TRI_DT.itemsize == 48


def test_device_creation_fails_loudly_without_gpu(rtc):
    """No HIP device -> rtcNewDevice returns NULL and the thread error is set; there is no CPU fallback."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    lib = rtc.lib()
    h = lib.rtcNewDevice(b"")
    assert not h
    assert lib.rtcGetDeviceError(None) != rtc.RTC_ERROR_NONE
    assert lib.rtcGetDeviceError(None) == rtc.RTC_ERROR_NONE  # returns-and-clears (device.cpp:250-256)


def test_error_conventions(rtc):
    dev = rtc.Device("gpu=none")
    lib = dev.lib
    sc = rtc.Scene(dev)
    rays = rtc.aligned_rayhits(1)
    # trace before commit -> INVALID_OPERATION (scene.cpp:25,54)
    sc.intersect1M(rays, check=False)
    assert dev.error() == rtc.RTC_ERROR_INVALID_OPERATION
    assert dev.error() == rtc.RTC_ERROR_NONE
    sc.commit()
    # host-only device: tracing is refused, never emulated
    sc.intersect1M(rays, check=False)
    assert dev.error() == rtc.RTC_ERROR_INVALID_OPERATION
    # packet entry points gather their active rays into a single-ray batch: null packet -> INVALID_ARGUMENT
    lib.rtcIntersect4(None, sc.handle, None, None)
    assert dev.error() == rtc.RTC_ERROR_INVALID_ARGUMENT
    # unsupported geometry types
    assert not lib.rtcNewGeometry(dev.handle, rtc.RTC_GEOMETRY_TYPE_QUAD)
    assert dev.error() == rtc.RTC_ERROR_INVALID_OPERATION
    # wrong buffer format
    g = lib.rtcNewGeometry(dev.handle, rtc.RTC_GEOMETRY_TYPE_TRIANGLE)
    buf = np.zeros(16, np.float32)
    lib.rtcSetSharedGeometryBuffer(g, rtc.RTC_BUFFER_TYPE_VERTEX, 0, rtc.RTC_FORMAT_UINT3, buf.ctypes.data, 0, 12, 4)
    assert dev.error() == rtc.RTC_ERROR_INVALID_OPERATION
    lib.rtcReleaseGeometry(g)
    # error callback receives code and message
    seen = []
    CB = rtc.C.CFUNCTYPE(None, rtc.C.c_void_p, rtc.C.c_int, rtc.C.c_char_p)
    cb = CB(lambda user, code, msg: seen.append((code, msg)))
    lib.rtcSetDeviceErrorFunction(dev.handle, rtc.C.cast(cb, rtc.C.c_void_p), None)
    lib.rtcGetGeometry(sc.handle, 1234)
    assert seen and seen[0][0] == rtc.RTC_ERROR_INVALID_ARGUMENT
    lib.rtcSetDeviceErrorFunction(dev.handle, None, None)
    dev.error()
    sc.release()
    # unknown accel name -> INVALID_ARGUMENT at commit (scene.cpp:209)
    dev2 = rtc.Device("gpu=none,tri_accel=bvh7.nonsense")
    sc2 = rtc.Scene(dev2)
    sc2.add_triangles(np.eye(3, dtype=np.float32), np.array([[0, 1, 2]], np.uint32))
    lib.rtcCommitScene(sc2.handle)
    assert dev2.error() == rtc.RTC_ERROR_INVALID_ARGUMENT
    sc2.release()
    dev2.release()
    dev.release()


def _decode_child(node, i):
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for a in range(3):
        s = np.array([int(node["exp"][a]) << 23], np.uint32).view(np.float32)[0]
        o = node["origin"][a]
        # fmaf(q, s, o): q*s is exact (8-bit integer times a power of two), so one rounding like the kernel's fma
        lo[a] = np.float32(np.float64(node["q"][2 * a][i]) * np.float64(s) + np.float64(o))
        hi[a] = np.float32(np.float64(node["q"][2 * a + 1][i]) * np.float64(s) + np.float64(o))
    return lo, hi


def _check_tree(nodes, tris, root, mode_pluecker):
    """Every decoded child box contains all triangles below it; every triangle is referenced exactly once."""
    seen = np.zeros(len(tris), np.int32)

    def tri_bounds(first, count):
        t = tris[first:first + count]
        a = t["a"]
        if mode_pluecker:
            pts = np.concatenate([a, t["b"], t["c"]])
        else:  # v1 = v0 - e1, v2 = v0 + e2
            pts = np.concatenate([a, a - t["b"], a + t["c"]])
        return pts.min(0), pts.max(0)

    def walk(ref, depth):
        if ref & LEAF:
            first, count = ref & ((1 << 26) - 1), (ref >> 26) & 31
            assert 1 <= count <= 28
            seen[first:first + count] += 1
            return tri_bounds(first, count) + (depth,)
        n = nodes[ref]
        lo, hi, md = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32), depth
        nchild = 0
        for i in range(8):
            c = int(n["child"][i])
            if c == EMPTY:
                assert np.all(n["q"][0::2, i] == 255) and np.all(n["q"][1::2, i] == 0)
                continue
            nchild += 1
            clo, chi, d = walk(c, depth + 1)
            blo, bhi = _decode_child(n, i)
            tol = 1e-6 * np.maximum(np.abs(clo), np.abs(chi)) if not mode_pluecker else 0.0
            assert np.all(blo <= clo + tol) and np.all(bhi >= chi - tol), (ref, i, blo, clo, bhi, chi)
            lo, hi, md = np.minimum(lo, clo), np.maximum(hi, chi), max(md, d)
        assert nchild >= 2
        return lo, hi, md

    _, _, maxdepth = walk(root, 0)
    assert np.all(seen == 1)
    return maxdepth


@pytest.mark.parametrize("cfg,pl", [("tri_accel=bvh8.triangle4v", True), ("tri_accel=bvh8.triangle4", False)])
def test_bvh8_builder_bomberman(rtc, bomberman_tris, cfg, pl):
    verts, tris = bomberman_tris
    dev = rtc.Device("gpu=none," + cfg)
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, tris)
    sc.commit()
    st = sc.stats()
    assert st["accelKind"] == (1 if pl else 2) and st["primCount"] == 1454 and st["nodeBytes"] == 96 and st["primBytes"] == 48
    nodes = sc.accel_data(0).view(NODE_DT)
    recs = sc.accel_data(1).view(TRI_DT)
    assert len(nodes) == st["nodeCount"] and len(recs) == 1454
    depth = _check_tree(nodes, recs, sc.accel_root(), pl)
    assert depth == st["maxDepth"]
    # ids: one geometry (geomID 0), primIDs are a permutation of 0..1453
    assert np.all(recs["geomID"] == 0) and np.array_equal(np.sort(recs["primID"]), np.arange(1454))
    lo, hi = sc.bounds()
    assert np.allclose(lo, verts.min(0)) and np.allclose(hi, verts.max(0))
    sc.release()
    dev.release()


@pytest.mark.parametrize("n,seed", [(1, 0), (4, 1), (5, 2), (29, 3), (3000, 4)])
def test_bvh8_builder_random(rtc, n, seed):
    from helpers import random_soup
    verts, tris = random_soup(n, seed, extent=100.0, size=3.0)
    # degenerate inputs: a flat axis and duplicated triangles
    verts[: min(n, 8) * 3, 2] = 5.0
    dev = rtc.Device("gpu=none,tri_accel=bvh8.triangle4v")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, np.concatenate([tris, tris[:1]]))
    sc.commit()
    nodes = sc.accel_data(0).view(NODE_DT)
    recs = sc.accel_data(1).view(TRI_DT)
    assert len(recs) == n + 1
    _check_tree(nodes, recs, sc.accel_root(), True)
    sc.release()
    dev.release()


def test_invalid_triangles_are_skipped(rtc):
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [np.nan, 0, 0]], np.float32)
    tris = np.array([[0, 1, 2], [0, 1, 3], [0, 1, 9]], np.uint32)  # NaN vertex, out-of-range index
    dev = rtc.Device("gpu=none")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, tris)
    sc.commit()
    recs = sc.accel_data(1).view(TRI_DT)
    assert len(recs) == 1 and recs["primID"][0] == 0
    sc.release()
    dev.release()


def test_host_pool_of_the_staging_pipeline(rtc):
    """The pool of staging threads behind the chunked pipeline of large host-pointer batches (rt_device.cpp HostPool: helpers that
    poll a ticket word while a batch runs, the caller works too) through its test hook on a gpu=none device: every part of every job
    runs exactly once with its own index, over many begin / end cycles and with more helpers than this container has cores."""
    dev = rtc.Device("gpu=none")
    L = rtc.lib()
    for threads, cycles, jobs, parts in ((1, 3, 50, 7), (4, 40, 60, 9), (12, 25, 80, 64), (4, 10, 200, 1)):
        assert L.rtcamdDebugHostPoolSelfTest(dev.handle, threads, cycles, jobs, parts) == cycles * jobs * parts
        assert dev.error() == 0
    dev.release()
