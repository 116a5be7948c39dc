"""BASELINE configs 4 and 5 through the reference's UNCHANGED tutorial device code.

oracle/_ref/tut_viewer_stream and oracle/_ref/tut_pathtracer = tutorials/<name>/<name>_device.cpp + the reference's own scene framework
(tutorials/common/{scenegraph,image,lights,texture}, tutorial/scene.cpp, scene_device.cpp, tutorial_device.cpp, common/lexers, common/tasking,
common/sys), compiled where they lie in the build container against this repository's include/embree3 and linked against libembree3.so
(oracle/Makefile scene_tutorials); oracle/tut_scene_harness.cpp stands in for the windowing framework and loads the scene with the reference's
own OBJ loader in subdivision mode.  The .obj is written here from the committed data asset assets/bomberman.mesh.npz.

 * viewer_stream (config 4): the frame is traced by the tutorial as one rtcIntersect1M(M = 64, RTC_INTERSECT_CONTEXT_FLAG_COHERENT) call per
   8x8 tile from all host threads (viewer_stream_device.cpp:288-341) - 32 400 stream calls at 1920x1080 - and compared with the same frame
   shaded in numpy from ONE rtcIntersect1M batch through the Python binding (same camera, same eyelight shading with the default OBJ material).
 * pathtracer (config 5): one frame at 4 samples per pixel - rtcIntersect1 per bounce and rtcOccluded1 per light from all host threads
   (pathtracer_device.cpp:1442-1535), i.e. millions of single-ray calls through the call combiner.  The path tracer's materials and sampling are
   not restated; the frame must be the SAME whether the tutorial runs on one host thread (every call alone) or on all of them (calls combined
   into shared launches), and its primary visibility must agree with rtcIntersect1M."""
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
INVALID = 0xFFFFFFFF
CAM = dict(frm=(18.21240425, 20.05745888, 15.46878433), to=(0.0, 0.0, 0.0), fov=90.0)  # build/bomberman.ecs:2-6


def _obj(bomberman, path):
    verts, fs, fi = bomberman
    with open(path, "w") as f:
        for p in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        k = 0
        for n in fs:
            f.write("f " + " ".join(str(int(i) + 1) for i in fi[k:k + n]) + "\n")
            k += n


def _run(name, cfg, w, h, out, obj, L, Cl, extra=(), timeout=900):
    exe = os.path.join(REFDIR, "tut_" + name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs the reference tree in the build container)")
    cmd = [exe, cfg, str(w), str(h), out, obj] + [repr(float(x)) for x in CAM["frm"] + CAM["to"]] + [repr(CAM["fov"]), str(L), str(Cl)] + [str(x) for x in extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(out, dtype=np.uint32).reshape(h, w), r.stdout


def _channels(img):
    return np.stack([(img >> (8 * k)) & 0xFF for k in range(3)], -1).astype(np.int32)


@pytest.mark.parametrize("accel,w,h,L,Cl", [("bvh4.compressed.leaf", 1920, 1080, 6, 3), ("default", 640, 360, 5, 3)])
def test_viewer_stream_tutorial_frame(rtc, bomberman, tmp_path, accel, w, h, L, Cl):
    obj = str(tmp_path / "bomberman.obj")
    _obj(bomberman, obj)
    img, log = _run("viewer_stream", f"subdiv_accel={accel}", w, h, str(tmp_path / "vs.raw"), obj, L, Cl)
    print("[tutorial] " + log.strip())
    assert f"{w}x{h}" in log and f"levels {L}/{Cl}" in log
    # the same frame from one big batch: camera.h:74-91 / viewer_stream_device.cpp:297-325 rays, eyelight shading :344-372 with Kd = 1 (default OBJ material)
    rg = importlib.import_module("embree-compressed_amd.raygen")
    verts, fs, fi = bomberman
    dev = rtc.Device(f"subdiv_accel={accel}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(L, Cl)
    sc.commit()
    raw = rg.make_primary_rays(w, h, frm=CAM["frm"], to=CAM["to"], fov=CAM["fov"], tile=1)  # row-major pixels
    rays = rtc.aligned_rayhits(w * h)
    rays[:] = raw.reshape(-1).view(rtc.RAYHIT_DTYPE)
    sc.intersect1M(rays)
    f32 = np.float32
    hit = rays["geomID"] != INVALID
    d = np.stack([rays["dir_x"], rays["dir_y"], rays["dir_z"]], 1)
    ng = np.stack([rays["Ng_x"], rays["Ng_y"], rays["Ng_z"]], 1)
    ngn = (ng / np.maximum(np.sqrt((ng * ng).sum(1, dtype=f32), dtype=f32), f32(1e-30))[:, None]).astype(f32)
    c = np.abs((d * ngn).sum(1, dtype=f32)).astype(f32)  # dot(neg(dir), face_forward(dir, Ns))
    c[~hit] = 0
    b = (f32(255.0) * np.clip(c, 0, 1)).astype(np.uint32)
    want = ((b << 16) + (b << 8) + b).reshape(h, w)
    diff = np.abs(_channels(img) - _channels(want)).max(-1)
    frac = float((diff > 2).mean())
    hit_t = (img & 0xFFFFFF) != 0
    print(f"[tutorial] viewer_stream {accel} {w}x{h} L{L}/C{Cl}: {frac * 100:.3f} % of the pixels differ by more than 2/255; "
          f"hits {int(hit.sum())} (batch) vs non-black pixels {int(hit_t.sum())} (tutorial, {w * h // 64} stream calls of 64 rays)")
    assert hit.mean() > 0.7
    assert frac < 0.004, (accel, frac)  # silhouette pixels only
    sc.release()
    dev.release()


@pytest.mark.parametrize("accel", ["default", "bvh4.compressed.leaf"])
def test_pathtracer_tutorial_frame(rtc, bomberman, tmp_path, accel):
    obj = str(tmp_path / "bomberman.obj")
    _obj(bomberman, obj)
    w, h, L, Cl, spp = 192, 108, 5, 3, 4
    img_mt, log = _run("pathtracer", f"subdiv_accel={accel}", w, h, str(tmp_path / "pt_mt.raw"), obj, L, Cl, extra=(spp,))
    print("[tutorial] " + log.strip())
    img_1t, log1 = _run("pathtracer", f"subdiv_accel={accel}", w, h, str(tmp_path / "pt_1t.raw"), obj, L, Cl, extra=(spp, 1))
    # single-ray calls combined into shared launches (all host threads) give the frame of calls that were each alone (one host thread)
    assert np.array_equal(img_mt, img_1t), int((img_mt != img_1t).sum())
    ch = _channels(img_mt)
    lum = ch.sum(-1)
    assert (lum > 0).mean() > 0.9 and len(np.unique(lum)) > 30  # lit by the ambient + directional light of the harness, not flat
    # primary visibility: pixels that see geometry (rtcIntersect1M on the pixel centres) are darker than the ambient background on average
    rg = importlib.import_module("embree-compressed_amd.raygen")
    verts, fs, fi = bomberman
    dev = rtc.Device(f"subdiv_accel={accel}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(L, Cl)
    sc.commit()
    raw = rg.make_primary_rays(w, h, frm=CAM["frm"], to=CAM["to"], fov=CAM["fov"], tile=1)
    rays = rtc.aligned_rayhits(w * h)
    rays[:] = raw.reshape(-1).view(rtc.RAYHIT_DTYPE)
    sc.intersect1M(rays)
    hit = (rays["geomID"] != INVALID).reshape(h, w)
    print(f"[tutorial] pathtracer {accel} {w}x{h} {spp} spp: identical on 1 and on all host threads; mean luminance on hits {lum[hit].mean():.1f}, on misses {lum[~hit].mean() if (~hit).any() else float('nan'):.1f}")
    assert hit.mean() > 0.7
    sc.release()
    dev.release()
