"""Large HOST-pointer batches (what the callers of the drop-in API pass, viewer_stream_device.cpp:288-341) go through a chunked
pipeline (rt_trace.cpp trace_host_pipelined): gather into pinned memory by a host thread pool, upload / traversal / download of the
chunks on two alternating internal streams, scatter of tfar + hit.  A stream is M independent rays, so the records must be byte for
byte those of the unpipelined staging path and of a device-resident batch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rtc, cfg, bomberman, kind, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)  # the knobs are read when the device is created
    try:
        dev = rtc.Device(cfg + (",tri_accel=bvh8.triangle4v" if kind == "tri" else ",subdiv_accel=bvh4.compressed.leaf"))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    verts, fs, fi = bomberman
    sc = rtc.Scene(dev)
    if kind == "tri":
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 2)
    sc.commit()
    return dev, sc


UNPIPELINED = {"RTAMD_PIPE_MIN": "2000000000"}


@pytest.mark.parametrize("kind", ["tri", "cbvh.leaf"])
@pytest.mark.parametrize("cfg,env", [("gpu=0", {"RTAMD_PIPE_MIN": "1", "RTAMD_PIPE_CHUNK": "4096"}),          # many small chunks
                                     ("gpu=0", {"RTAMD_PIPE_MIN": "1", "RTAMD_PIPE_CHUNK": "70000", "RTAMD_HOST_THREADS": "1"}),  # ragged last chunk, no helpers
                                     ("gpus=0:0:0", {"RTAMD_PIPE_MIN": "1", "RTAMD_PIPE_CHUNK": "30000", "RTAMD_HOST_THREADS": "5"}),  # three shards
                                     ("gpu=0", {})])                                                            # defaults
def test_pipelined_host_batches_equal_the_unpipelined_path(rtc, po, bomberman, kind, cfg, env):
    import torch

    verts = bomberman[0]
    lo, hi = verts.min(0), verts.max(0)
    d0, s0 = _scene(rtc, "gpu=0", bomberman, kind, UNPIPELINED)
    d1, s1 = _scene(rtc, cfg, bomberman, kind, env)
    for n in (400_003, 4097, 3):
        a = po.make_random_rays(n, lo, hi, seed=31)
        a["tnear"][::7] = 5.0
        a["tfar"][::7] = 1.0  # skipped rays (tnear > tfar) stay untouched
        b = a.copy()
        g = torch.from_numpy(a.view(np.uint8).reshape(n, 80).copy()).cuda()
        s0.intersect1M(a)
        s1.intersect1M(b)
        s0.intersect1M(g)
        d0.synchronize()
        assert a.tobytes() == b.tobytes() == g.cpu().numpy().tobytes()
        if n > 100:
            assert int((a["geomID"] != 0xFFFFFFFF).sum()) > 0.08 * n
        # any hit, RTCRay records at a 96-byte pitch: the bytes between the records must not be touched
        src = po.make_random_rays(n, lo, hi, seed=32)
        raw0 = np.full((n, 96), 0xA5, np.uint8)
        raw0[:, :48] = src.view(np.uint8).reshape(n, 80)[:, :48]
        raw1 = raw0.copy()
        r0 = np.ndarray((n,), dtype=rtc.RAY_DTYPE, buffer=raw0.data, strides=(96,))
        r1 = np.ndarray((n,), dtype=rtc.RAY_DTYPE, buffer=raw1.data, strides=(96,))
        s0.occluded1M(r0)
        s1.occluded1M(r1)
        assert raw0.tobytes() == raw1.tobytes()
        assert (raw1[:, 48:] == 0xA5).all()
    for s, d in ((s0, d0), (s1, d1)):
        s.release()
        d.release()
