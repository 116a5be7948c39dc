"""Packet, packet-stream and SoA-stream entry points (rtcIntersect4/8/16, rtcIntersectNM, rtcIntersectNp and the occluded
twins; rtcore.cpp:306-401,450-539) give, lane by lane, what rtcIntersect1M gives for the same rays; masked lanes are not
touched."""
import ctypes as C

import numpy as np
import pytest

from helpers import INVALID

pytestmark = pytest.mark.gpu

RAYF = ["org_x", "org_y", "org_z", "tnear", "dir_x", "dir_y", "dir_z", "time", "tfar", "mask", "id", "flags"]
HITF = ["Ng_x", "Ng_y", "Ng_z", "u", "v", "primID", "geomID", "instID"]


def _soa(aos, n, with_hit):
    """RTCRayHitN / RTCRayN packet of width n from AoS records: field-major float32/uint32 words"""
    fields = RAYF + (HITF if with_hit else [])
    out = np.zeros((len(fields), n), np.uint32)
    for k, f in enumerate(fields):
        out[k] = aos[f][:n].view(np.uint32)
    return out


@pytest.mark.parametrize("W", [4, 8, 16])
def test_packets_match_single_rays(rtc, po, bomberman_tris, W):
    verts, tris = bomberman_tris
    dev = rtc.Device("tri_accel=bvh8.triangle4v")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, tris)
    sc.commit()
    L = sc.lib
    src = po.make_random_rays(4096, verts.min(0), verts.max(0), seed=5)
    want = rtc.aligned_rayhits(4096)
    want[:] = src
    sc.intersect1M(want)
    wocc = rtc.aligned_rays(4096)
    for f in wocc.dtype.names:
        wocc[f] = src[f]
    sc.occluded1M(wocc)
    ctx = rtc.make_context()
    fn_i, fn_o = getattr(L, f"rtcIntersect{W}"), getattr(L, f"rtcOccluded{W}")
    for fn in (fn_i, fn_o):
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.RandomState(W)
    nh = 0
    for p in range(0, 512, W):
        block = src[p:p + W]
        valid = np.where(rng.rand(W) < 0.8, -1, 0).astype(np.int32)
        pk = _soa(block, W, True)
        before = pk.copy()
        fn_i(valid.ctypes.data, sc.handle, C.addressof(ctx), pk.ctypes.data)
        dev.check("packet")
        for l in range(W):
            if valid[l] == 0:
                assert np.array_equal(pk[:, l], before[:, l])
                continue
            w = want[p + l]
            assert pk[18, l] == w["geomID"] and pk[17, l] == w["primID"]
            assert pk[8, l] == w["tfar"].view(np.uint32) and pk[15, l] == w["u"].view(np.uint32)
            nh += int(w["geomID"] != INVALID)
        po_ = _soa(block, W, False)
        fn_o(valid.ctypes.data, sc.handle, C.addressof(ctx), po_.ctypes.data)
        dev.check("packet occluded")
        for l in range(W):
            exp = wocc["tfar"][p + l] if valid[l] else block["tfar"][l]
            assert po_[8, l] == np.float32(exp).view(np.uint32)
    assert nh > 0
    # stream of packets with a stride larger than the packet, and the SoA pointer stream
    N, M = W, 16
    words = 20 * N + 8
    buf = np.zeros((M, words), np.uint32)
    for m in range(M):
        buf[m, :20 * N] = _soa(src[1024 + m * N: 1024 + (m + 1) * N], N, True).ravel()
    L.rtcIntersectNM.restype = None
    L.rtcIntersectNM.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t]
    L.rtcIntersectNM(sc.handle, C.addressof(ctx), buf.ctypes.data, N, M, words * 4)
    dev.check("rtcIntersectNM")
    for m in range(M):
        pk = buf[m, :20 * N].reshape(20, N)
        w = want[1024 + m * N: 1024 + (m + 1) * N]
        assert np.array_equal(pk[18], w["geomID"]) and np.array_equal(pk[17], w["primID"]) and np.array_equal(pk[8], w["tfar"].view(np.uint32))
        assert not buf[m, 20 * N:].any()
    n = 300
    cols = {f: np.ascontiguousarray(src[f][2048:2048 + n]) for f in RAYF + HITF[:-1]}
    inst = np.full(n, INVALID, np.uint32)

    class Np(C.Structure):
        _fields_ = [(f, C.c_void_p) for f in RAYF + HITF]

    a = Np(*[cols[f].ctypes.data for f in RAYF + HITF[:-1]], inst.ctypes.data)
    L.rtcIntersectNp.restype = None
    L.rtcIntersectNp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L.rtcIntersectNp(sc.handle, C.addressof(ctx), C.addressof(a), n)
    dev.check("rtcIntersectNp")
    w = want[2048:2048 + n]
    for f in ("tfar", "geomID", "primID", "u", "v", "Ng_x"):
        assert np.array_equal(cols[f].view(np.uint32), w[f].view(np.uint32)), f
    sc.release()
    dev.release()
