"""Edge and vertex creases of subdivision geometry (rtcore_geometry.h crease buffers; rules of
CatmullClark1RingT::subdivide, kernels/subdiv/catmullclark_ring.h:213-315) on a `gpu=none` device, against closed forms
and an independent numpy implementation of the same published rules."""
import numpy as np
import pytest

CUBE_V = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1], [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
CUBE_F = [(0, 4, 5, 1), (1, 5, 7, 3), (3, 7, 6, 2), (2, 6, 4, 0), (4, 6, 7, 5), (0, 1, 3, 2)]
CUBE_E = sorted({(min(f[k], f[(k + 1) % 4]), max(f[k], f[(k + 1) % 4])) for f in CUBE_F for k in range(4)})


def _grids(sc, L):
    raw = sc.accel_data(4)
    w = 2 ** L + 1
    per = 12 + 12 * w * w
    return [raw[p * per + 12: (p + 1) * per].view(np.float32).reshape(3, w, w) for p in range(len(raw) // per)]


def _scene(rtc, L, edge_creases=None, vertex_creases=None):
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    sc.add_subdiv(CUBE_V, np.full(6, 4, np.uint32), np.array(CUBE_F, np.uint32).ravel(), edge_creases=edge_creases, vertex_creases=vertex_creases)
    sc.set_levels(L, 2)
    sc.commit()
    return dev, sc


def _cc_creases(V, faces, ecrease, vcrease):
    """one Catmull-Clark step with semi-sharp creases on a closed mesh, per catmullclark_ring.h:213-315 (numpy, independent)"""
    V = np.asarray(V, np.float64)
    nV = len(V)
    fp = np.array([V[list(f)].mean(0) for f in faces])
    edges, vfaces, vnbr = {}, [[] for _ in range(nV)], [set() for _ in range(nV)]
    for fi, f in enumerate(faces):
        for k, a in enumerate(f):
            b = f[(k + 1) % len(f)]
            edges.setdefault((min(a, b), max(a, b)), []).append(fi)
            vfaces[a].append(fi)
            vnbr[a].add(b)
            vnbr[b].add(a)
    eidx = {e: i for i, e in enumerate(edges)}
    w_of = lambda a, b: ecrease.get((min(a, b), max(a, b)), 0.0)
    ep = np.zeros((len(edges), 3))
    for (a, b), fs in edges.items():
        smooth = (V[a] + V[b] + fp[fs[0]] + fp[fs[1]]) / 4
        mid = (V[a] + V[b]) / 2
        w = w_of(a, b)
        ep[eidx[(a, b)]] = smooth if w <= 0 else (mid if w >= 1 else (1 - w) * smooth + w * mid)
    newV = np.zeros_like(V)
    new_e, new_v = {}, {}
    for v in range(nV):
        n = len(vnbr[v])
        Q = fp[vfaces[v]].mean(0)
        R = np.mean([(V[v] + V[x]) / 2 for x in vnbr[v]], 0)
        smooth = (Q + 2 * R + (n - 3) * V[v]) / n
        cr = [(x, w_of(v, x)) for x in sorted(vnbr[v]) if w_of(v, x) > 0]
        vw = vcrease.get(v, 0.0)
        new_v[v] = max(vw - 1, 0.0)
        child = {x: max(w - 1, 0.0) for x, w in cr}
        if vw > 0:
            newV[v] = V[v] if vw >= 1 else (1 - vw) * smooth + vw * V[v]
        elif len(cr) <= 1:
            newV[v] = smooth
        elif len(cr) == 2:
            (a, wa), (b, wb) = cr
            sharp = (V[a] + 6 * V[v] + V[b]) / 8
            blend = 0.5 * (wa + wb)
            newV[v] = sharp if blend >= 1 else (1 - blend) * smooth + blend * sharp
            child[a] = max(0.25 * (3 * wa + wb) - 1, 0.0)
            child[b] = max(0.25 * (3 * wb + wa) - 1, 0.0)
        else:
            newV[v] = V[v]
        for x, w in child.items():
            if w > 0:
                new_e[(v, nV + eidx[(min(v, x), max(v, x))])] = w
    allV = np.concatenate([newV, ep, fp])
    quads = []
    for fi, f in enumerate(faces):
        N = len(f)
        for k in range(N):
            quads.append((f[k], nV + eidx[(min(f[k], f[(k + 1) % N]), max(f[k], f[(k + 1) % N]))], nV + len(ep) + fi,
                          nV + eidx[(min(f[k - 1], f[k]), max(f[k - 1], f[k]))]))
    return allV, quads, {(min(a, b), max(a, b)): w for (a, b), w in new_e.items()}, {v: w for v, w in new_v.items() if w > 0}


def _smooth_limits(V, quads):
    out = np.zeros_like(V)
    ring = [(set(), set()) for _ in range(len(V))]
    for q in quads:
        for k in range(4):
            ring[q[k]][0].update([q[(k + 1) % 4], q[(k + 3) % 4]])
            ring[q[k]][1].add(q[(k + 2) % 4])
    for v, (E, F) in enumerate(ring):
        n = len(E)
        out[v] = (n * n * V[v] + 4 * V[list(E)].sum(0) + V[list(F)].sum(0)) / (n * (n + 5))
    return out


def test_fully_creased_cube_is_the_cube(rtc):
    L = 3
    dev, sc = _scene(rtc, L, edge_creases=(np.array(CUBE_E, np.uint32), np.full(12, 1e9, np.float32)))
    for f, g in zip(CUBE_F, _grids(sc, L)):
        c = CUBE_V[list(f)]
        axis = int(np.argmax((c == c[0]).all(0)))  # the coordinate that is constant on this face
        assert np.array_equal(g[axis], np.full_like(g[axis], c[0, axis]))
        t = np.linspace(0, 1, 9, dtype=np.float64)
        want = ((1 - t)[None, :, None] * (1 - t)[:, None, None] * c[0] + t[None, :, None] * (1 - t)[:, None, None] * c[1]
                + t[None, :, None] * t[:, None, None] * c[2] + (1 - t)[None, :, None] * t[:, None, None] * c[3])  # [j][i][xyz]
        assert np.abs(np.moveaxis(g, 0, -1) - want).max() < 1e-6  # straight creases, flat faces: the bilinear quad itself
    sc.release()
    dev.release()


def test_zero_weights_change_nothing_and_fractions_interpolate(rtc):
    L = 3
    d0, s0 = _scene(rtc, L)
    d1, s1 = _scene(rtc, L, edge_creases=(np.array(CUBE_E, np.uint32), np.zeros(12, np.float32)), vertex_creases=(np.arange(8, dtype=np.uint32), np.zeros(8, np.float32)))
    for a, b in zip(_grids(s0, L), _grids(s1, L)):
        assert np.array_equal(a, b)
    # a pinned corner (vertex crease >= level) is interpolated by the surface
    d2, s2 = _scene(rtc, L, vertex_creases=(np.array([7], np.uint32), np.array([50.0], np.float32)))
    g = _grids(s2, L)
    f = next(i for i, q in enumerate(CUBE_F) if q[2] == 7)  # face 1 = (1,5,7,3): corner index 2 is uv (1,1)
    assert np.array_equal(g[f][:, -1, -1], CUBE_V[7])
    smooth_corner = _grids(s0, L)[f][:, -1, -1]
    assert np.linalg.norm(smooth_corner) < 0.8 * np.linalg.norm(CUBE_V[7])  # the smooth cube pulls its corners in
    for s, d in ((s0, d0), (s1, d1), (s2, d2)):
        s.release()
        d.release()


@pytest.mark.parametrize("case", ["top2", "mixed"])
def test_semi_sharp_creases_match_an_independent_implementation(rtc, case):
    """weights that decay to zero within the tessellation level: after L numpy steps with the crease rules every weight is
    0 and the smooth limit stencil is exact; the tessellator's grid points must be exactly those limit points"""
    L = 3
    if case == "top2":  # the four edges of the face x = +1 with weight 2 (Chaikin keeps them equal)
        e = [(4, 6), (6, 7), (5, 7), (4, 5)]
        ew = [2.0, 2.0, 2.0, 2.0]
        vc = {}
    else:  # unequal weights meeting at vertices (Chaikin mixes them), a fractional one, a dart and a vertex crease
        e = [(4, 6), (6, 7), (5, 7), (4, 5), (0, 1), (2, 3)]
        ew = [3.0, 1.0, 2.5, 0.5, 1.5, 0.75]
        vc = {3: 1.5, 0: 0.5}
    dev, sc = _scene(rtc, L, edge_creases=(np.array(e, np.uint32), np.array(ew, np.float32)),
                     vertex_creases=(np.array(list(vc), np.uint32), np.array(list(vc.values()), np.float32)) if vc else None)
    V, F = CUBE_V.astype(np.float64), [tuple(f) for f in CUBE_F]
    ec = {(min(a, b), max(a, b)): w for (a, b), w in zip(e, ew)}
    for _ in range(L):
        V, F, ec, vc = _cc_creases(V, F, ec, vc)
    assert not ec and not vc  # all creases have decayed: the smooth limit applies
    lim = _smooth_limits(V, F)
    pts = np.concatenate([np.moveaxis(g, 0, -1).reshape(-1, 3) for g in _grids(sc, L)]).astype(np.float64)
    d = np.sqrt(((pts[:, None, :] - lim[None, :, :]) ** 2).sum(-1))
    assert d.min(1).max() < 2e-6   # every grid point is one of the independent limit points ...
    assert d.min(0).max() < 2e-6   # ... and every limit point is hit
    sc.release()
    dev.release()
