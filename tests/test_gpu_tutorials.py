"""The reference's UNCHANGED tutorial device code, running on the MI355X through this library (BASELINE configs[0] and [2]).

oracle/_ref/tut_<name> = tutorials/<name>/<name>_device.cpp + tutorials/common/tutorial/tutorial_device.cpp compiled from the
reference tree against this repository's include/embree3 and linked against libembree3.so in the build container (oracle/Makefile);
oracle/tut_harness.cpp stands in for the windowing framework.  The tutorial traces one rtcIntersect1 + one rtcOccluded1 per pixel
from all host threads (the call combiner batches them).  Its framebuffer is compared with the same frame shaded in numpy from
rtcIntersect1M / rtcOccluded1M batches through the Python binding: same scene, same camera (camera.h:74-91), same shading
(triangle_geometry_device.cpp:136-170, displacement_geometry_device.cpp:208-256)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
INVALID = 0xFFFFFFFF


def _run(name, cfg, w, h, out, extra=()):
    exe = os.path.join(REFDIR, "tut_" + name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs the reference tree in the build container)")
    r = subprocess.run([exe, cfg, str(w), str(h), out, "1.5", "1.5", "-1.5", "0", "0", "0", "90"] + list(extra), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(out, dtype=np.uint32).reshape(h, w), r.stdout


def _shade(rtc, sc, w, h, diffuse_of):
    rg = importlib.import_module("embree-compressed_amd.raygen")
    raw = rg.make_primary_rays(w, h, frm=(1.5, 1.5, -1.5), to=(0, 0, 0), fov=90.0, tile=1)  # row-major pixels
    rays = rtc.aligned_rayhits(w * h)
    rays[:] = raw.reshape(-1).view(rtc.RAYHIT_DTYPE)
    sc.intersect1M(rays)
    f32 = np.float32
    hit = rays["geomID"] != INVALID
    color = np.zeros((w * h, 3), f32)
    diffuse = diffuse_of(rays)
    color[hit] = diffuse[hit] * f32(0.5)
    light = (np.array([-1, -1, -1], f32) / np.sqrt(f32(3))).astype(f32)
    org = np.stack([rays["org_x"], rays["org_y"], rays["org_z"]], 1)
    d = np.stack([rays["dir_x"], rays["dir_y"], rays["dir_z"]], 1)
    sh = rtc.aligned_rays(int(hit.sum()))
    p = (org[hit] + rays["tfar"][hit, None] * d[hit]).astype(f32)
    sh["org_x"], sh["org_y"], sh["org_z"] = p[:, 0], p[:, 1], p[:, 2]
    sh["dir_x"], sh["dir_y"], sh["dir_z"] = -light[0], -light[1], -light[2]
    sh["tnear"], sh["tfar"], sh["time"], sh["mask"] = 0.001, np.inf, 0, 0xFFFFFFFF
    sc.occluded1M(sh)
    lit = sh["tfar"] >= 0
    ng = np.stack([rays["Ng_x"], rays["Ng_y"], rays["Ng_z"]], 1)[hit]
    ng = (ng / np.sqrt((ng * ng).sum(1, dtype=f32), dtype=f32)[:, None]).astype(f32)
    c = np.clip(-(ng @ light), 0, 1).astype(f32)
    idx = np.nonzero(hit)[0][lit]
    color[idx] += diffuse[idx] * c[lit, None]
    b = (f32(255.0) * np.clip(color, 0, 1)).astype(np.uint32)
    return ((b[:, 2] << 16) + (b[:, 1] << 8) + b[:, 0]).reshape(h, w), int(hit.sum())


def _compare(img, want, what):
    ch = lambda a, k: ((a >> (8 * k)) & 0xFF).astype(np.int32)
    diff = np.maximum.reduce([np.abs(ch(img, k) - ch(want, k)) for k in range(3)])
    frac = float((diff > 2).mean())
    print(f"[tutorial] {what}: {frac * 100:.3f} % of the pixels differ by more than 2/255")
    assert frac < 0.004, (what, frac)  # silhouette / shadow-edge pixels only


def test_triangle_geometry_tutorial_frame(rtc, tmp_path):
    w = h = 256
    img, log = _run("triangle_geometry", "", w, h, str(tmp_path / "tri.raw"))
    assert f"{w}x{h}" in log
    # the tutorial's scene (triangle_geometry_device.cpp:26-106), default scene flags
    v = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1], [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    t = np.array([[0, 1, 2], [1, 3, 2], [4, 6, 5], [5, 6, 7], [0, 4, 1], [1, 4, 5], [2, 3, 6], [3, 7, 6], [0, 2, 4], [2, 6, 4], [1, 5, 3], [3, 5, 7]], np.uint32)
    face = np.array([[1, 0, 0]] * 2 + [[0, 1, 0]] * 2 + [[.5, .5, .5]] * 2 + [[1, 1, 1]] * 2 + [[0, 0, 1]] * 2 + [[1, 1, 0]] * 2, np.float32)
    gv = np.array([[-10, -2, -10], [-10, -2, 10], [10, -2, -10], [10, -2, 10]], np.float32)
    gt = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    dev = rtc.Device("")
    sc = rtc.Scene(dev)
    assert sc.add_triangles(v, t) == 0 and sc.add_triangles(gv, gt) == 1
    sc.commit()
    want, nh = _shade(rtc, sc, w, h, lambda r: face[np.minimum(r["primID"], 11)])  # the tutorial indexes face_colors by primID only
    assert nh > 0.5 * w * h
    _compare(img, want, "triangle_geometry 256x256")
    sc.release()
    dev.release()


def test_displacement_geometry_tutorial_frame(rtc, tmp_path):
    """BASELINE configs[2] as the tutorial defines it: subdivision cube, rtcSetSceneLevels(6, 4), Perlin-noise displacement
    callback, ROBUST scene flag, ground plane; default subdiv accel and the fork's compressed.leaf."""
    lib = os.path.join(REFDIR, "libref_tutorial.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref/libref_tutorial.so not built")
    shader = C.CDLL(lib).ref_tutorial_displacementFunction
    w = h = 192
    v = np.array([[-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]], np.float32)
    fi = np.array([0, 4, 5, 1, 1, 5, 6, 2, 2, 6, 7, 3, 0, 3, 7, 4, 4, 7, 6, 5, 0, 1, 2, 3], np.uint32)
    fs = np.full(6, 4, np.uint32)
    gv = np.array([[-10, -2, -10], [-10, -2, 10], [10, -2, -10], [10, -2, 10]], np.float32)
    gt = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    for accel in ("default", "bvh4.compressed.leaf"):
        cfg = "subdiv_accel=" + accel
        img, _ = _run("displacement_geometry", cfg, w, h, str(tmp_path / "displ.raw"))
        dev = rtc.Device(cfg)
        sc = rtc.Scene(dev, rtc.RTC_SCENE_FLAG_ROBUST)
        assert sc.add_triangles(gv, gt) == 0
        assert sc.add_subdiv(v, fs, fi, level=256.0, displacement=shader) == 1
        sc.set_levels(6, 4)
        sc.commit()
        dif = np.array([[0.8, 0.0, 0.0], [0.9, 0.6, 0.5]], np.float32)
        want, nh = _shade(rtc, sc, w, h, lambda r: dif[np.minimum(r["geomID"], 1)])
        assert nh > 0.5 * w * h
        _compare(img, want, f"displacement_geometry {accel} {w}x{h}")
        sc.release()
        dev.release()
