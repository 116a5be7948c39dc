"""Size-independent properties at the full sizes of BASELINE.json (the oracle needs minutes for these batches, the
properties need no oracle): on the metric scene (bomberman subdiv L6 / C3) and on the triangle scene, with 4 M random rays
on device-resident batches:
  * split invariance   one 4 M batch == four 1 M batches (a stream is M independent single-ray calls, rtcore.cpp:403-432)
  * stride invariance  the same rays embedded in 96-byte records give the same records
  * idempotence        tracing the traced batch again changes nothing (the hit is found again at t == tfar and accepted by
                       t <= tfar; misses stay untouched)
  * any-hit vs closest an occluded ray (tfar = -inf) is exactly a ray for which the closest-hit query found something
                       (triangles and the eager path; the fork's any-hit is a stub in all compressed modes, compressed.h:754-756)
  * determinism        two runs of the same batch give the same bytes although the work distribution is dynamic
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M = 4_000_000


def _scene(rtc, bomberman, kind):
    verts, fs, fi = bomberman
    if kind == "tri":
        dev = rtc.Device("tri_accel=bvh8.triangle4v")
        sc = rtc.Scene(dev)
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        dev = rtc.Device("subdiv_accel=" + kind)
        sc = rtc.Scene(dev)
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(6, 3)
    sc.commit()
    return dev, sc


@pytest.mark.parametrize("kind", ["tri", "bvh4.compressed.leaf", "default", "bvh4.compressed.grid"])
def test_full_size_properties(rtc, bomberman, kind):
    import torch

    rg = importlib.import_module("embree-compressed_amd.raygen")
    verts = bomberman[0]
    dev, sc = _scene(rtc, bomberman, kind)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)  # clones (torch's stream) and traces are then stream-ordered
    rays = torch.from_numpy(rg.make_random_rays(M, verts.min(0), verts.max(0), seed=2024)).cuda()  # [M, 80] bytes
    whole = rays.clone()
    sc.intersect1M(whole)
    dev.synchronize()
    hits = int((whole.view(torch.int32)[:, 18] != -1).sum().item())
    assert 0.1 * M < hits < 0.5 * M
    # split invariance
    parts = rays.clone()
    for k in range(4):
        sc.intersect1M(parts[k * (M // 4):(k + 1) * (M // 4)])
    dev.synchronize()
    assert torch.equal(parts, whole)
    # determinism under dynamic scheduling
    again = rays.clone()
    sc.intersect1M(again)
    dev.synchronize()
    assert torch.equal(again, whole)
    # stride invariance (96-byte records, payload in the first 80)
    wide = torch.zeros((M // 4, 96), dtype=torch.uint8, device="cuda")
    wide[:, :80] = rays[: M // 4]
    sc.intersect1M(wide)
    dev.synchronize()
    assert torch.equal(wide[:, :80], whole[: M // 4]) and int(wide[:, 80:].sum().item()) == 0
    # idempotence
    twice = whole.clone()
    sc.intersect1M(twice)
    dev.synchronize()
    if kind in ("tri", "default", "bvh4.compressed.grid"):
        assert torch.equal(twice, whole)
    else:  # fork leaf mode: a blob's local ray depends on the tfar it sees; IDs must still agree
        same = (twice.view(torch.int32)[:, 17:19] == whole.view(torch.int32)[:, 17:19]).all(1).float().mean().item()
        assert same > 0.999
    # any-hit vs closest hit
    if kind in ("tri", "default"):
        occ = rays[:, :48].contiguous().clone()
        sc.occluded1M(occ)
        dev.synchronize()
        occluded = torch.isneginf(occ.view(torch.float32)[:, 8])
        assert torch.equal(occluded, whole.view(torch.int32)[:, 18] != -1)
        untouched = ~occluded
        assert torch.equal(occ[untouched], rays[:, :48][untouched])
    sc.release()
    dev.release()


@pytest.mark.parametrize("kind", ["tri", "tri.moeller", "bvh4.compressed.leaf", "bvh4.compressed.full", "default"])
def test_node_step_variants_and_counted_twin(rtc, bomberman, kind, monkeypatch):
    """The child-parallel (octet) node and leaf steps (trace_loop.hip.h: 8 lanes per ray, taken when few lanes of a wave
    have node work / from a few rays waiting at a triangle or grid-cell leaf on) against the lane-per-ray steps, and the
    instrumented kernel twin against the plain one: identical ray records, byte for byte, closest hit and any hit, over
    repeated runs (the ray-to-wave assignment is dynamic, so every run mixes the lanes differently).  The twin's hit counter
    must equal the number of hit records."""
    import torch

    rg = importlib.import_module("embree-compressed_amd.raygen")
    verts = bomberman[0]
    n = 1_000_000
    rays = torch.from_numpy(rg.make_random_rays(n, verts.min(0), verts.max(0), seed=7)).cuda()

    def scene():
        if kind == "tri.moeller":
            dev = rtc.Device("tri_accel=bvh8.triangle4")
            sc = rtc.Scene(dev)
            sc.add_triangles(verts, rtc.fan_triangulate(bomberman[1], bomberman[2]))
            sc.commit()
        else:
            dev, sc = _scene(rtc, bomberman, kind)
        dev.set_stream(torch.cuda.current_stream().cuda_stream)  # clones (torch's stream) and traces are then stream-ordered
        return dev, sc

    # reference = lane-per-ray steps only.  Triangle leaves and grid cells have no lane-per-ray leaf code in the lane kernel
    # any more (they are always tested 8 lanes per ray): their reference comes from the ray-pool kernel, which runs
    # TriLeaf::intersect / GridCellLeaf::intersect, one lane per ray
    monkeypatch.setenv("RTAMD_KERNEL", "lane" if kind.startswith("bvh4.compressed") else "pool")
    monkeypatch.setenv("RTAMD_OCT_MAX", "0")  # knobs are read when the device is created
    monkeypatch.setenv("RTAMD_OCT_LEAF", "0")
    monkeypatch.setenv("RTAMD_CULL", "0")  # no root cull pre-pass: every ray is fetched by the traversal kernel itself
    monkeypatch.setenv("RTAMD_CBVH_FORM", "lane")  # cBVH blobs walked one ray per lane (the form the oracle-pinned round-1 kernels had)
    dev0, sc0 = scene()
    monkeypatch.setenv("RTAMD_KERNEL", "lane")
    ref = rays.clone()
    sc0.intersect1M(ref)
    occ_ref = rays[:, :48].contiguous().clone()
    sc0.occluded1M(occ_ref)
    dev0.synchronize()
    hits = int((ref.view(torch.int32)[:, 18] != -1).sum().item())
    # (octet node threshold, octet / quad leaf threshold, root cull pre-pass, cBVH blob walk: one ray per lane / four lanes per ray)
    # + rays parked after the frustum test before a walk pass runs (two-stage blob visits of the quad form; None = the library's default, 16)
    for octmax, octleaf, cull, form, walk in (("0", "0", "0", "lane", None), ("8", "1", "1", "quad", "1"), ("16", "8", "0", "quad", "40"),
                                              ("32", "32", "1", "lane", None), ("16", None, None, None, None)):
        monkeypatch.setenv("RTAMD_OCT_MAX", octmax)
        if walk is None:
            monkeypatch.delenv("RTAMD_WALK_BATCH", raising=False)
        else:
            monkeypatch.setenv("RTAMD_WALK_BATCH", walk)
        if form is None:
            monkeypatch.delenv("RTAMD_CBVH_FORM")  # the library's own default: by the context's coherent flag (here: quad form)
        else:
            monkeypatch.setenv("RTAMD_CBVH_FORM", form)
        if cull is None:
            monkeypatch.delenv("RTAMD_CULL")  # the library's own default
        else:
            monkeypatch.setenv("RTAMD_CULL", cull)
        if octleaf is None:
            monkeypatch.delenv("RTAMD_OCT_LEAF")  # the library's own default
        else:
            monkeypatch.setenv("RTAMD_OCT_LEAF", octleaf)
        dev, sc = scene()
        what = f"{kind}: octet thresholds node {octmax} leaf {octleaf}, cull {cull}, cBVH form {form}, walk batch {walk}"
        for rep in range(2):
            got = rays.clone()
            sc.intersect1M(got)
            dev.synchronize()
            assert torch.equal(got, ref), f"{what}, plain kernel, run {rep}"
            got = rays.clone()
            cnt = sc.intersect1M_counted(got)
            dev.synchronize()
            assert torch.equal(got, ref), f"{what}, counted twin, run {rep}"
            assert cnt["hits"] == hits and cnt["rays"] == n
            assert (cnt["reserved"] > 0) == (cull == "1")  # survivors of the pre-pass, when it ran (off by default)
            occ = rays[:, :48].contiguous().clone()
            sc.occluded1M(occ)
            dev.synchronize()
            assert torch.equal(occ, occ_ref), f"{what}, any hit, run {rep}"
        sc.release()
        dev.release()
    sc0.release()
    dev0.release()
