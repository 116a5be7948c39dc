"""Shared helpers for the parity tests."""
import numpy as np

INVALID = 0xFFFFFFFF


def unit_triangle_rays(rtc_or_none, n=256, seed=1):
    """TriangleHitTest of the reference (tutorials/verify/verify.cpp:2118-2205): rays from (0,0,-1) to
    uniformly sampled interior points of the triangle (0,0,0),(1,0,0),(0,1,0).  Returns (rays, u, v)."""
    rng = np.random.RandomState(seed)
    u = rng.rand(n).astype(np.float32)
    v = rng.rand(n).astype(np.float32)
    su = np.sqrt(u).astype(np.float32)
    v = (v * su).astype(np.float32)
    u = (np.float32(1.0) - su).astype(np.float32)
    bad = (u < 0.001) | (v < 0.001) | ((u + v) > 0.999)
    u[bad] = 0.333
    v[bad] = 0.333
    a = np.array([1, 0, 0], np.float32)  # vertices[1]
    b = np.array([0, 1, 0], np.float32)  # vertices[2]
    c = np.array([0, 0, 0], np.float32)  # vertices[0]
    to = c[None, :] + u[:, None] * (a - c)[None, :] + v[:, None] * (b - c)[None, :]
    frm = np.array([0, 0, -1], np.float32)
    return frm, (to - frm[None, :]).astype(np.float32), u, v


def fill_rays(rays, org, dirs, tnear=0.0, tfar=np.inf):
    rays["org_x"], rays["org_y"], rays["org_z"] = org[..., 0], org[..., 1], org[..., 2]
    rays["dir_x"], rays["dir_y"], rays["dir_z"] = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    rays["tnear"] = tnear
    rays["tfar"] = tfar
    rays["time"] = 0
    rays["mask"] = 0xFFFFFFFF
    rays["id"] = np.arange(rays.shape[0], dtype=np.uint32)
    rays["flags"] = 0
    if "geomID" in rays.dtype.names:
        rays["geomID"] = INVALID
        rays["primID"] = INVALID
        rays["instID"] = INVALID
        for f in ("Ng_x", "Ng_y", "Ng_z", "u", "v"):
            rays[f] = 0


def compare_hits(got, want, rtol=1e-4, what=""):
    """IDs bit-exact, t/u/v within rtol (north_star: 1e-4 relative), Ng within rtol of its norm."""
    gh = got["geomID"] != INVALID
    wh = want["geomID"] != INVALID
    assert int((gh != wh).sum()) == 0, f"{what}: hit/miss differs for {(gh != wh).sum()} rays"
    assert np.array_equal(got["geomID"], want["geomID"]), f"{what}: geomID mismatch"
    bad = got["primID"][gh] != want["primID"][gh]
    assert int(bad.sum()) == 0, f"{what}: primID mismatch on {bad.sum()} rays"
    assert np.array_equal(got["instID"][gh], want["instID"][gh]), f"{what}: instID mismatch"
    t, tw = got["tfar"][gh].astype(np.float64), want["tfar"][gh].astype(np.float64)
    assert np.all(np.abs(t - tw) <= rtol * np.abs(tw) + 1e-30), f"{what}: t mismatch {np.abs(t - tw).max()}"
    # misses keep their tfar untouched
    assert np.array_equal(got["tfar"][~gh], want["tfar"][~gh]), f"{what}: tfar of misses changed"
    for f in ("u", "v"):
        a, b = got[f][gh].astype(np.float64), want[f][gh].astype(np.float64)
        assert np.all(np.abs(a - b) <= rtol * np.maximum(np.abs(b), 1e-3)), f"{what}: {f} mismatch {np.abs(a - b).max()}"
    ng = np.stack([got["Ng_x"][gh], got["Ng_y"][gh], got["Ng_z"][gh]], 1).astype(np.float64)
    nw = np.stack([want["Ng_x"][gh], want["Ng_y"][gh], want["Ng_z"][gh]], 1).astype(np.float64)
    if ng.size:
        scale = np.linalg.norm(nw, axis=1, keepdims=True) + 1e-30
        assert np.all(np.abs(ng - nw) <= rtol * scale), f"{what}: Ng mismatch"
    return int(gh.sum())


def random_soup(n_tris, seed, extent=10.0, size=1.0):
    rng = np.random.RandomState(seed)
    c = (rng.rand(n_tris, 1, 3) * extent).astype(np.float32)
    v = (c + (rng.rand(n_tris, 3, 3).astype(np.float32) - 0.5) * size).astype(np.float32)
    verts = v.reshape(-1, 3)
    tris = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    return verts, tris


def random_rays_np(m, lo, hi, seed):
    rng = np.random.RandomState(seed)
    p1 = (lo + rng.rand(m, 3) * (hi - lo)).astype(np.float32)
    p2 = (lo + rng.rand(m, 3) * (hi - lo)).astype(np.float32)
    d = p2 - p1
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return p1, d
