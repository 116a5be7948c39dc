"""Degenerate rays through the hot path, against the oracle: directions along the coordinate axes and with one or two
components exactly zero (the zero_fix / rcp_safe paths: vec3fa.h:163-165, compressed_help.h:111; a zero component turns a slab
test into 0 * inf), positive and huge tnear, short tfar, rays starting on the bounding box, unnormalised and tiny directions.
Orthographic cameras and straight-down shadow rays are exactly such rays.  Same bar as everywhere: IDs exact, t / u / v within
1e-4 (triangles, eager grid cells), byte for byte for the fork's modes in the product's arithmetic.

Finding (the fork's, reproduced not fixed): the compressed modes lose most of such rays - 1 330 hits where the eager accel finds
over 12 000 on the same rays.  A ray along a blob's local z axis has a zero projected direction, `intersect_line` then divides
0 by 0 and `intersect_frustum` rejects the blob through its NaN check (compressed_help.h:93-133): every straight-down ray misses
the ground plane.  Oracle and kernels agree on every one of them, in both arithmetic modes."""
import numpy as np
import pytest

from helpers import FORK_ORACLE_MODE, INVALID, ORDERED_FORK, check_fork_parity, compare_hits, fill_rays

pytestmark = pytest.mark.gpu


def _degenerate_rays(po, rtc, lo, hi, n_grid=70):
    """origins on planes outside / inside the model's box, directions along +-x, +-y, +-z, and with zero components"""
    rng = np.random.RandomState(4)
    ext = hi - lo
    org, dirs, tnear, tfar = [], [], [], []
    g = (np.arange(n_grid) + 0.37) / n_grid
    for axis in range(3):
        a, b = [k for k in range(3) if k != axis]
        for sign in (1.0, -1.0):
            o = np.zeros((n_grid * n_grid, 3), np.float32)
            o[:, a] = (lo[a] + np.repeat(g, n_grid) * ext[a]).astype(np.float32)
            o[:, b] = (lo[b] + np.tile(g, n_grid) * ext[b]).astype(np.float32)
            o[:, axis] = (lo[axis] - 3.0) if sign > 0 else (hi[axis] + 3.0)
            d = np.zeros_like(o)
            d[:, axis] = sign
            org.append(o); dirs.append(d)
            tnear.append(np.zeros(len(o), np.float32)); tfar.append(np.full(len(o), np.inf, np.float32))
    # one zero component, the other two random; two zero components with an unnormalised / tiny third one
    m = 6000
    o = (lo + rng.rand(m, 3) * ext).astype(np.float32)
    d = (rng.rand(m, 3).astype(np.float32) - 0.5)
    d[np.arange(m), rng.randint(0, 3, m)] = 0.0
    d[::5] *= 1e-3   # unnormalised directions scale t
    d[1::5] *= 250.0
    org.append(o); dirs.append(d)
    tn = np.zeros(m, np.float32)
    tn[::3] = 7.5                          # hits in front of tnear do not count (a negative tnear is outside the API: the ray segment must lie in
                                           # [0, inf], and whether a hit at t < 0 is found then depends on which boxes straddle the origin, i.e. on the
                                           # BVH builder - product, oracle and reference each have their own)
    tn[1::7] = 1e30                        # tnear > tfar for finite tfar -> skipped, else nothing in range
    tf = np.full(m, np.inf, np.float32)
    tf[::4] = (rng.rand(len(tf[::4])) * 40.0).astype(np.float32)  # short rays
    tnear.append(tn); tfar.append(tf)
    # rays that start exactly on the faces of the bounding box, pointing inwards along an axis
    k = 3000
    o = (lo + rng.rand(k, 3) * ext).astype(np.float32)
    ax = rng.randint(0, 3, k)
    side = rng.randint(0, 2, k)
    o[np.arange(k), ax] = np.where(side == 0, lo[ax], hi[ax]).astype(np.float32)
    d = np.zeros((k, 3), np.float32)
    d[np.arange(k), ax] = np.where(side == 0, 1.0, -1.0)
    org.append(o); dirs.append(d)
    tnear.append(np.zeros(k, np.float32)); tfar.append(np.full(k, np.inf, np.float32))
    org, dirs = np.concatenate(org), np.concatenate(dirs)
    rays = rtc.aligned_rayhits(len(org))
    fill_rays(rays, org, dirs, tnear=np.concatenate(tnear), tfar=np.concatenate(tfar))
    return rays


@pytest.mark.parametrize("accel", ["bvh8.triangle4v", "bvh8.triangle4"])
def test_degenerate_rays_on_triangles(rtc, po, bomberman, accel):
    verts, fs, fi = bomberman
    tris = rtc.fan_triangulate(fs, fi)
    dev = rtc.Device(f"tri_accel={accel}")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, tris)
    sc.commit()
    got = _degenerate_rays(po, rtc, verts.min(0), verts.max(0))
    want = got.copy()
    sc.intersect1M(got)
    orc = po.TriangleScene(verts, tris, 0 if accel.endswith("4v") else 1)
    orc.intersect1M(want)
    nh = compare_hits(got, want, what=accel)
    assert nh > 0.2 * len(got)
    occ = rtc.aligned_rays(len(got))
    src = _degenerate_rays(po, rtc, verts.min(0), verts.max(0))
    for f in occ.dtype.names:
        occ[f] = src[f]
    wocc = occ.copy()
    sc.occluded1M(occ)
    orc.occluded1M(wocc)
    assert np.array_equal(occ["tfar"], wocc["tfar"])
    orc.free()
    sc.release()
    dev.release()


@pytest.mark.parametrize("accel", list(FORK_ORACLE_MODE))
def test_degenerate_rays_on_subdivision_accels(rtc, po, bomberman, accel):
    verts, fs, fi = bomberman
    L, Cl = 5, 3
    dev = rtc.Device(f"subdiv_accel={accel}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(L, Cl)
    sc.commit()
    st = sc.stats()
    if accel in ORDERED_FORK:
        orc = po.SubdivScene(sc.accel_data(2), st["primBytes"], FORK_ORACLE_MODE[accel], Cl, qnodes=sc.accel_data(0), root=sc.accel_root())
    else:
        orc = po.SubdivScene(sc.accel_data(2), st["primBytes"], FORK_ORACLE_MODE[accel], Cl)
    lo, hi = verts.min(0), verts.max(0)
    got = _degenerate_rays(po, rtc, lo, hi)
    src = got.copy()
    sc.intersect1M(got)

    def trace_oracle():
        w = src.copy()
        orc.intersect1M(w, nthreads=8)
        return w

    if accel == "default":
        nh = compare_hits(got, trace_oracle(), what=accel)
    else:
        # (reference arithmetic: axis-parallel rays run along cell borders, and the sample of hits is small - 10 of 1 250 hits are beyond 1e-4
        # with the other reciprocal, 0.8 %, 8 of them in the same cell and 2 in the neighbouring one)
        check_fork_parity(po, got, trace_oracle, accel, what=f"{accel} degenerate rays", cell=2.0 ** -L, beyond_floor=0.008)
        nh = int((got["geomID"] != INVALID).sum())
    assert nh > (0.2 if accel == "default" else 0.02) * len(got)  # see the module docstring for the fork's modes
    assert sc.intersect1M_counted(src.copy())["stackSpills"] == 0
    orc.free()
    sc.release()
    dev.release()


@pytest.mark.parametrize("cfg", ["tri_accel=bvh8.triangle4v", "subdiv_accel=default", "subdiv_accel=bvh4.compressed.leaf"])
def test_invalid_rays_terminate_and_leave_the_others_alone(rtc, po, bomberman, cfg):
    """NaN / infinite origins, directions and intervals are outside the API (a renderer still produces one now and then): such a ray
    must come back - it may visit the whole tree when every comparison of its slab tests is decided by the NaN rules - and must not
    change the result of any other ray of the batch.  (What it reports itself is not checked: x86 max/min and IEEE maxNum treat
    NaN differently, the reference defines nothing here.)"""
    verts, fs, fi = bomberman
    dev = rtc.Device(cfg)
    sc = rtc.Scene(dev)
    if cfg.startswith("tri"):
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 2)
    sc.commit()
    lo, hi = verts.min(0), verts.max(0)
    n = 40_000
    clean = po.make_random_rays(n, lo, hi, seed=91)
    want = clean.copy()
    sc.intersect1M(want)
    dirty = clean.copy()
    bad = np.arange(100, n, 4001)  # ten rays
    fields = ["org_x", "dir_y", "tnear", "tfar", "dir_z", "org_z"]
    values = [np.nan, np.nan, np.nan, np.nan, np.inf, -np.inf]
    for k, i in enumerate(bad):
        dirty[fields[k % 6]][i] = values[k % 6]
    dirty["dir_x"][bad[-1]] = dirty["dir_y"][bad[-1]] = dirty["dir_z"][bad[-1]] = 0.0  # null direction
    sc.intersect1M(dirty)
    keep = np.ones(n, bool)
    keep[bad] = False
    assert dirty[keep].tobytes() == want[keep].tobytes()
    occ = rtc.aligned_rays(n)
    for f in occ.dtype.names:
        occ[f] = clean[f]
    wocc = occ.copy()
    sc.occluded1M(wocc)
    for k, i in enumerate(bad):
        occ[fields[k % 6]][i] = values[k % 6]
    sc.occluded1M(occ)
    assert occ[keep].tobytes() == wocc[keep].tobytes()
    assert dev.error() == 0
    sc.release()
    dev.release()
