"""Multi-GPU behind the C API (north_star config 4/5): rtcNewDevice("gpus=...") replicates the accel on
every listed GPU at commit and splits one host-pointer rtcIntersect1M / rtcOccluded1M call into contiguous ray ranges, one per
GPU, each with its own stream, staging buffers and a disjoint D2H into the caller's records.  No collective.

On a one-GPU box the same ordinal is listed twice ("gpus=0:0"): two logical shards with their own replicas, streams and
staging, i.e. the whole sharded code path except the second physical device.  Results must be byte-identical to one shard.
(8 physical GPUs: unmeasured until the driver's SCALE run; bench.py shards one process per GPU on top of this.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rtc, cfg, bomberman, kind):
    verts, fs, fi = bomberman
    dev = rtc.Device(cfg + (",tri_accel=bvh8.triangle4v" if kind == "tri" else ",subdiv_accel=bvh4.compressed.leaf"))
    sc = rtc.Scene(dev)
    if kind == "tri":
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 2)
    sc.commit()
    return dev, sc


@pytest.mark.parametrize("kind", ["tri", "cbvh.leaf"])
@pytest.mark.parametrize("gpus", ["gpus=0:0", "gpus=0:0:0"])
def test_sharded_host_batch_matches_one_shard(rtc, po, bomberman, kind, gpus):
    verts = bomberman[0]
    lo, hi = verts.min(0), verts.max(0)
    d1, s1 = _scene(rtc, "gpu=0", bomberman, kind)
    dn, sn = _scene(rtc, gpus, bomberman, kind)
    for n in (300_001, 5, 1):  # uneven split; fewer rays than 2 per shard -> one shard
        a = po.make_random_rays(n, lo, hi, seed=21)
        b = a.copy()
        s1.intersect1M(a)
        sn.intersect1M(b)
        assert a.tobytes() == b.tobytes()
        if n > 100:
            assert int((a["geomID"] != 0xFFFFFFFF).sum()) > 0.1 * n
        # any hit, RTCRay records, strided (96-byte pitch)
        src = po.make_random_rays(n, lo, hi, seed=22)
        raw1 = np.zeros((n, 96), np.uint8)
        raw1[:, :48] = src.view(np.uint8).reshape(n, 80)[:, :48]
        raw2 = raw1.copy()
        r1 = np.ndarray((n,), dtype=rtc.RAY_DTYPE, buffer=raw1.data, strides=(96,))
        r2 = np.ndarray((n,), dtype=rtc.RAY_DTYPE, buffer=raw2.data, strides=(96,))
        s1.occluded1M(r1)
        sn.occluded1M(r2)
        assert raw1.tobytes() == raw2.tobytes()
    for s, d in ((s1, d1), (sn, dn)):
        s.release()
        d.release()


def test_device_resident_batch_and_counters_on_a_sharded_device(rtc, po, bomberman):
    import torch

    verts = bomberman[0]
    lo, hi = verts.min(0), verts.max(0)
    d1, s1 = _scene(rtc, "gpu=0", bomberman, "cbvh.leaf")
    dn, sn = _scene(rtc, "gpus=0:0", bomberman, "cbvh.leaf")
    n = 200_000
    src = po.make_random_rays(n, lo, hi, seed=5).view(np.uint8).reshape(n, 80)
    a = torch.from_numpy(src.copy()).cuda()
    b = torch.from_numpy(src.copy()).cuda()
    s1.intersect1M(a)
    sn.intersect1M(b)  # device pointer: traced in place on the shard that owns the GPU
    d1.synchronize()
    dn.synchronize()
    assert torch.equal(a, b)
    c1 = s1.intersect1M_counted(po.make_random_rays(n, lo, hi, seed=5))
    cn = sn.intersect1M_counted(po.make_random_rays(n, lo, hi, seed=5))
    assert c1["hits"] == cn["hits"] == int((a.view(torch.int32)[:, 18] != -1).sum().item()) and cn["stackSpills"] == 0
    assert c1["nodeVisits"] == cn["nodeVisits"]
    for s, d in ((s1, d1), (sn, dn)):
        s.release()
        d.release()


def test_bad_gpu_lists(rtc):
    for cfg in ("gpus=0:99", "gpus=7-9"):
        with pytest.raises(rtc.RTCError):
            rtc.Device(cfg)


def test_concurrent_device_resident_calls(rtc, po, bomberman):
    """rtcIntersect1M on device-resident batches from several host threads at once (re-entrant in the reference,
    rtcore.cpp:403-432): {launch context, queue heads, launch, event} is taken as one unit per launch, so concurrent callers
    never share a context.  Each thread's batches must equal the serial results."""
    import torch

    verts = bomberman[0]
    lo, hi = verts.min(0), verts.max(0)
    dev, sc = _scene(rtc, "gpu=0", bomberman, "tri")
    T, R, n = 6, 12, 60_000
    src = [[po.make_random_rays(n, lo, hi, seed=100 + 16 * t + r).view(np.uint8).reshape(n, 80) for r in range(R)] for t in range(T)]
    serial = [[torch.from_numpy(x.copy()).cuda() for x in row] for row in src]
    for row in serial:
        for b in row:
            sc.intersect1M(b)
    dev.synchronize()
    conc = [[torch.from_numpy(x.copy()).cuda() for x in row] for row in src]
    torch.cuda.synchronize()
    errs = []

    def work(t):
        try:
            for b in conc[t]:
                sc.intersect1M(b, check=False)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dev.synchronize()
    assert not errs
    for t in range(T):
        for a, b in zip(serial[t], conc[t]):
            assert torch.equal(a, b)
    sc.release()
    dev.release()
