"""Several ray batches in flight on different HIP streams (rtcamdSetDeviceStream between calls) give the same bytes as
the same batches traced one after the other: the per-launch scratch of the library (work-queue heads, stack overflow
area) is not shared between concurrent kernels.  No reference counterpart: embree traces synchronously on the caller."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["tri", "cbvh.leaf", "cbvh.leaf.coherent"])
def test_batches_on_several_streams_match_serial(rtc, po, bomberman, kind):
    """(cbvh.leaf.coherent: the batches carry RTC_INTERSECT_CONTEXT_FLAG_COHERENT, for which the library picks the blob walk form - four lanes
    per ray or one ray per lane - by the number of batches it finds running, so the pipelined batches mix both forms: same bytes.)"""
    import torch

    verts, fs, fi = bomberman
    ctx = rtc.make_context(coherent=True) if kind.endswith(".coherent") else None
    if kind == "tri":
        dev = rtc.Device("tri_accel=bvh8.triangle4v")
        sc = rtc.Scene(dev)
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        dev = rtc.Device("subdiv_accel=bvh4.compressed.leaf")
        sc = rtc.Scene(dev)
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 2)
    sc.commit()
    lo, hi = verts.min(0), verts.max(0)
    nb, n = 12, 200_000  # more batches than the library has launch contexts (8)
    src = [po.make_random_rays(n, lo, hi, seed=40 + b).view(np.uint8).reshape(n, 80) for b in range(nb)]
    serial = [torch.from_numpy(s.copy()).cuda() for s in src]
    for b in serial:
        sc.intersect1M(b, ctx=ctx)
    dev.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    piped = [torch.from_numpy(s.copy()).cuda() for s in src]
    torch.cuda.synchronize()
    for i, b in enumerate(piped):
        dev.set_stream(streams[i % 3].cuda_stream)
        sc.intersect1M(b, ctx=ctx, check=False)
    torch.cuda.synchronize()
    dev.check("pipelined batches")
    nh = 0
    for a, b in zip(serial, piped):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
        nh += int((a.view(torch.int32)[:, 18] != -1).sum().item())
    assert nh > 0
    sc.release()
    dev.release()
