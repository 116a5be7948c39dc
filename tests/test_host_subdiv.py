"""CPU tests of the subdivision host side (rows a11, a13, a14, a17, f1 of SURVEY.md section 8): tessellator,
eager grid cells, fork cBVH encoder + node/leaf codec, on a `gpu=none` device.  The traversal itself is not run here
by the product (there is no CPU path); where hits are checked they come from the oracle on the exported records.
"""
import numpy as np
import pytest

LEAF, EMPTY = 0x80000000, 0xFFFFFFFF
CELL_DT = np.dtype([("px", "<f4", 9), ("py", "<f4", 9), ("pz", "<f4", 9), ("uv", "<u4", 9), ("geomID", "<u4"), ("primID", "<u4"), ("pad", "<u4", 2)])
from helpers import CBVH_HDR_DT, CBVH_NODES, CBVH_TAIL_DT, cbvh_header, cbvh_layout  # noqa: E402
T1 = np.array([0.0, 0.005, 0.01, 0.05, 0.1, 0.2, 0.4, 0.6], np.float32)  # compressed_node.h:31-38
T2 = np.array([0.0, 0.4, 0.48, 0.49, 0.5, 0.51, 0.52, 0.6], np.float32)  # :22-29
T3 = np.array([0.0, 0.25, 0.5, 0.75], np.float32)


def _cube():
    v = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1], [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    fi = np.array([0, 4, 5, 1, 1, 5, 7, 3, 3, 7, 6, 2, 2, 6, 4, 0, 4, 6, 7, 5, 0, 1, 3, 2], np.uint32)
    return v, np.full(6, 4, np.uint32), fi


def _grids(sc, L):
    raw = sc.accel_data(4)
    w = 2 ** L + 1
    per = 12 + 12 * w * w
    out = {}
    for p in range(len(raw) // per):
        h = raw[p * per: p * per + 12].view(np.uint32)
        out[int(h[1])] = raw[p * per + 12: (p + 1) * per].view(np.float32).reshape(3, w, w)
    return out


def test_struct_sizes():
    assert CELL_DT.itemsize == 160 and CBVH_HDR_DT.itemsize == CBVH_NODES == 160 and CBVH_TAIL_DT.itemsize == 64
    assert cbvh_layout(3, "leaf") == (256, 384, 512)  # line 0 | ids + nodes | cells | tail


def test_tessellator_plane_is_reproduced_exactly(rtc):
    """A flat 5x5-quad regular grid: interior faces are bicubic B-spline patches of coplanar, uniformly spaced control
    points, so the limit surface is the plane itself and the grid points are the affine images of (i/n, j/n)."""
    n = 6
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32))
    verts = np.stack([xs.ravel() * 2 + 1, ys.ravel() * 3 - 2, 0.5 * xs.ravel() + 0.25 * ys.ravel()], 1).astype(np.float32)
    faces = [(j * n + i, j * n + i + 1, (j + 1) * n + i + 1, (j + 1) * n + i) for j in range(n - 1) for i in range(n - 1)]
    fi = np.array(faces, np.uint32).ravel()
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, np.full(len(faces), 4, np.uint32), fi)
    sc.set_levels(3, 2)
    sc.commit()
    g = _grids(sc, 3)
    f = 2 * (n - 1) + 2  # an interior face: (i,j) = (2,2)
    t = np.arange(9) / 8.0
    want_x = (2 + t)[None, :] * 2 + 1 + 0 * t[:, None]
    want_y = (2 + t)[:, None] * 3 - 2 + 0 * t[None, :]
    want_z = 0.5 * (2 + t)[None, :] + 0.25 * (2 + t)[:, None]
    assert np.allclose(g[f][0], want_x, atol=1e-5) and np.allclose(g[f][1], want_y, atol=1e-5) and np.allclose(g[f][2], want_z, atol=1e-5)
    sc.release()
    dev.release()


def test_tessellator_cube_limit_surface(rtc):
    """Cube (8 extraordinary vertices of valence 3): symmetry, shared borders between faces, and the closed-form limit
    position of a valence-3 corner, (n^2 V + 4 sum(E) + sum(F)) / (n (n+5)) = V * (9 + 4*1 - 1) / 24 ... evaluated numerically."""
    v, fs, fi = _cube()
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi)
    sc.set_levels(4, 2)
    sc.commit()
    g = _grids(sc, 4)
    assert len(g) == 6
    pts = np.concatenate([np.stack([g[f][0].ravel(), g[f][1].ravel(), g[f][2].ravel()], 1) for f in range(6)])
    # octahedral symmetry of the limit surface: the point set is invariant under coordinate permutations and sign flips
    key = lambda P: np.unique(np.round(P * 1e4).astype(np.int64), axis=0)
    base = key(pts)
    assert np.array_equal(base, key(pts[:, [1, 2, 0]])) and np.array_equal(base, key(pts * np.array([-1, 1, 1])))
    # corner limit position: level-0 stencil with edge neighbours E_i and face-diagonal neighbours F_i of corner V=(1,1,1)
    V = np.array([1, 1, 1.0])
    E = [np.array([-1, 1, 1.0]), np.array([1, -1, 1.0]), np.array([1, 1, -1.0])]
    F = [np.array([-1, -1, 1.0]), np.array([1, -1, -1.0]), np.array([-1, 1, -1.0])]
    lim = (9 * V + 4 * sum(E) + sum(F)) / (3 * 8)
    corner = pts[np.argmax(pts.sum(1))]
    assert np.allclose(corner, lim, atol=1e-5), (corner, lim)
    # every border vertex is shared bit-exactly by the two faces that meet there (one evaluation per vertex)
    rounded = np.round(pts * 1e6).astype(np.int64)
    uniq = np.unique(rounded, axis=0)
    assert len(uniq) == 6 * 16 * 16 + 2  # V - E + F = 2 on the refined closed quad mesh: 6*16*16 faces -> F + 2 vertices
    sc.release()
    dev.release()


def test_eager_cells_cover_the_grid(rtc, bomberman):
    verts, fs, fi = bomberman
    L = 3
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(L, 2)
    sc.commit()
    st = sc.stats()
    assert st["accelKind"] == 6 and st["primBytes"] == 160 and st["primCount"] == 727 * 16
    cells = sc.accel_data(2).view(CELL_DT)
    g = _grids(sc, L)
    n = 2 ** L
    seen = {}
    for c in cells:
        f = int(c["primID"])
        u = (c["uv"] & 0xFFFF).astype(np.float64) / 8192.0  # decodeUV, grid_soa.h:248-257
        v = (c["uv"] >> 16).astype(np.float64) / 8192.0
        i0, j0 = int(round(u[0] * n)), int(round(v[0] * n))
        assert i0 % 2 == 0 and j0 % 2 == 0
        assert np.allclose(u.reshape(3, 3), (i0 + np.arange(3))[None, :] / n) and np.allclose(v.reshape(3, 3), (j0 + np.arange(3))[:, None] / n)
        assert np.array_equal(c["px"].reshape(3, 3), g[f][0][j0:j0 + 3, i0:i0 + 3])
        assert np.array_equal(c["pz"].reshape(3, 3), g[f][2][j0:j0 + 3, i0:i0 + 3])
        seen[(f, i0, j0)] = seen.get((f, i0, j0), 0) + 1
    assert len(seen) == 727 * 16 and all(x == 1 for x in seen.values())
    sc.release()
    dev.release()


def _decode_child(word, P, loc):
    xz, x, yz, y = word & 0xFF, (word >> 8) & 0xFF, (word >> 16) & 0xFF, (word >> 24) & 0xFF
    dim = (P[1] - P[0]).astype(np.float32)
    mn = np.array([T2[(xz >> 2) & 7] if loc & 1 else T1[xz >> 5], T2[(yz >> 2) & 7] if loc & 2 else T1[yz >> 5], T3[xz & 3]], np.float32)
    mx = np.array([1 - T1[(x >> 2) & 7] if loc & 1 else 1 - T2[x >> 5], 1 - T1[(y >> 2) & 7] if loc & 2 else 1 - T2[y >> 5], 1 - T3[yz & 3]], np.float32)
    return (mn * dim + P[0]).astype(np.float32), (mx * dim + P[0]).astype(np.float32)


def _morton(i):
    x = y = 0
    for b in range(8):
        x |= ((i >> (2 * b)) & 1) << b
        y |= ((i >> (2 * b + 1)) & 1) << b
    return x, y


def test_full_precision_nodes_hold_the_merged_cell_boxes(rtc, bomberman):
    """subdiv_accel=bvh4.compressed.full (compressed.h:40: Node<flavor::ref,...>, float boxes[4][6]; here plane-major, 96 B):
    the header is the box mode's byte for byte except for `box` / world bounds (made from the stored instead of the re-decoded
    cell boxes), a last-level child box is the bounding box of its cell's four projected vertices, an inner child box the union
    of its four children (build_box_hierarchy, compressed.h:381-405), and every float box lies inside the box the 4-byte code
    of the box mode decodes to for the same child (the quantizer only ever widens)."""
    verts, fs, fi = bomberman
    L, Cl = 4, 2
    blobs = {}
    for mode in ("bvh4.compressed.full", "bvh4.compressed.box"):
        dev = rtc.Device(f"gpu=none,keep_grids=1,subdiv_accel={mode}")
        sc = rtc.Scene(dev)
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(L, Cl)
        sc.commit()
        st = sc.stats()
        blobs[mode] = sc.accel_data(2).reshape(-1, st["primBytes"]).copy()
        if mode.endswith("full"):
            g = _grids(sc, L)
            assert st["accelKind"] == 7
        sc.release()
        dev.release()
    elems = (4 ** Cl - 1) // 3
    F, B = blobs["bvh4.compressed.full"], blobs["bvh4.compressed.box"]
    assert F.shape[1] == cbvh_layout(Cl, "full")[2] and B.shape[1] == cbvh_layout(Cl, "box")[2] and len(F) == len(B)
    n, sub = 2 ** L, 2 ** Cl
    rng = np.random.RandomState(1)
    for k in rng.choice(len(F), 200, replace=False):
        H, HB = cbvh_header(F[k], Cl, "full"), cbvh_header(B[k], Cl, "box")
        for f in ("geomID", "primID", "uv0", "uv1", "rcp_edges", "elems", "grid_width", "levels", "space", "proj", "iproj"):
            assert np.array_equal(H[f], HB[f]), f
        nodes = F[k][CBVH_NODES:CBVH_NODES + 96 * elems].view(np.float32).reshape(elems, 6, 4)  # [node][lx ux ly uy lz uz][child]
        words = B[k][CBVH_NODES:CBVH_NODES + 4 * elems].view(np.uint32)
        assert HB["rootWord"] == words[0]  # line 0 carries a copy of the root's word
        x0, y0 = int(round(H["uv0"][0] * n)), int(round(H["uv0"][1] * n))
        G = g[int(H["primID"])]
        P = np.stack([G[0][y0:y0 + sub + 1, x0:x0 + sub + 1], G[1][y0:y0 + sub + 1, x0:x0 + sub + 1], G[2][y0:y0 + sub + 1, x0:x0 + sub + 1]], -1).astype(np.float64)
        loc = P @ H["space"].reshape(3, 3).astype(np.float64).T
        hom = np.concatenate([loc[..., :2], np.ones(loc.shape[:2] + (1,))], -1) @ H["proj"].reshape(3, 3).astype(np.float64).T
        pr = np.concatenate([hom[..., :2] / hom[..., 2:3], loc[..., 2:3]], -1)

        def child_box(idx, r):
            return nodes[idx, 0::2, r], nodes[idx, 1::2, r]

        rootq = (np.array([-1, -1, HB["box"][0]], np.float32), np.array([1, 1, HB["box"][1]], np.float32))
        stack = [(0, rootq)]
        zlo, zhi = np.inf, -np.inf
        while stack:
            idx, qbox = stack.pop()
            for r in range(4):
                lo, hi = child_box(idx, r)
                qlo, qhi = _decode_child(int(words[idx]), qbox, r)
                # quantized box contains the float box - up to rounding of the decode, and up to the root's nominal [-1,1]^2
                # (compressed.h:517-519), which the rescaled projection meets only to a few 1e-4
                eps = 1e-3 * np.abs(qhi - qlo) + 1e-6 * np.maximum(np.maximum(np.abs(qlo), np.abs(qhi)), 1.0)
                assert np.all(lo >= qlo - eps) and np.all(hi <= qhi + eps)
                child = 4 * idx + 1 + r
                if child < elems:
                    clo = np.min(nodes[child, 0::2, :], axis=1)
                    chi = np.max(nodes[child, 1::2, :], axis=1)
                    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)  # union of the four children, exactly
                    stack.append((child, (qlo, qhi)))
                else:
                    cx, cy = _morton(child - elems)
                    q = pr[cy:cy + 2, cx:cx + 2].reshape(4, 3)
                    # (the encoder projects in fp32 from world coordinates of a few hundred: ~1e-4 of the [-1,1] window)
                    tol = np.array([5e-4, 5e-4, 2e-5 * max(1.0, float(np.abs(q[:, 2]).max()))])
                    assert np.all(np.abs(lo - q.min(0)) <= tol) and np.all(np.abs(hi - q.max(0)) <= tol)  # tight around the cell
                    zlo, zhi = min(zlo, lo[2]), max(zhi, hi[2])
        assert H["box"][0] == np.float32(zlo) and H["box"][1] == np.float32(zhi)  # frustum z range = the stored cells' (compressed.h:277-292)


@pytest.mark.parametrize("mode,Cl", [("bvh4.compressed.leaf", 3), ("bvh4.compressed.box", 2), ("bvh4.compressed.grid", 3)])
def test_cbvh_encoder_is_conservative(rtc, bomberman, mode, Cl):
    """Decode every blob with the tables of compressed_node.h: each cell's decoded box (in the blob's projected
    frame) must contain the four projected grid vertices of that cell up to the quantizer's documented slack, and
    the header must be self-consistent (iproj = proj^-1, uv window, elems, world bounds contain the vertices)."""
    verts, fs, fi = bomberman
    L = 4
    dev = rtc.Device(f"gpu=none,keep_grids=1,subdiv_accel={mode}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(L, Cl)
    sc.commit()
    st = sc.stats()
    stride = st["primBytes"]
    blobs = sc.accel_data(2).reshape(-1, stride)
    g = _grids(sc, L)
    n, sub = 2 ** L, 2 ** Cl
    assert len(blobs) == 727 * (n // sub) ** 2 == st["primCount"]
    elems = (4 ** Cl - 1) // 3
    rng = np.random.RandomState(0)
    worst = 0.0
    for b in blobs[rng.choice(len(blobs), 300, replace=False)]:
        H = cbvh_header(b, Cl, mode.split(".")[-1])
        assert H["elems"] == elems and H["grid_width"] == sub + 1 and H["levels"] == Cl
        assert np.allclose(H["proj"].reshape(3, 3) @ H["iproj"].reshape(3, 3), np.eye(3), atol=2e-3)
        x0, y0 = int(round(H["uv0"][0] * n)), int(round(H["uv0"][1] * n))
        assert np.allclose(H["uv1"], sub / n) and x0 % sub == 0 and y0 % sub == 0
        G = g[int(H["primID"])]
        P = np.stack([G[0][y0:y0 + sub + 1, x0:x0 + sub + 1], G[1][y0:y0 + sub + 1, x0:x0 + sub + 1], G[2][y0:y0 + sub + 1, x0:x0 + sub + 1]], -1).astype(np.float64)
        # the reference's bounds_o is NOT strictly conservative: its corner list omits (lx,ly,uz) (compressed.h:260-267,
        # copied as is), so vertices may stick out by a fraction of the blob's extent
        wslack = 0.25 * (H["whi"] - H["wlo"]) + 1e-4
        assert np.all(P >= H["wlo"] - wslack) and np.all(P <= H["whi"] + wslack)
        S = H["space"].reshape(3, 3).astype(np.float64)
        loc = P @ S.T
        M = H["proj"].reshape(3, 3).astype(np.float64)
        hom = np.concatenate([loc[..., :2], np.ones(loc.shape[:2] + (1,))], -1) @ M.T
        pr = np.concatenate([hom[..., :2] / hom[..., 2:3], loc[..., 2:3]], -1)
        assert np.all(np.abs(pr[..., :2]) <= 1.0 + 1e-3)  # rescaled to the projected bounding box
        words = b[CBVH_NODES:CBVH_NODES + 4 * elems].view(np.uint32)
        assert H["rootWord"] == words[0]
        # walk the implicit quadtree; the traversal root box is [-1,1]^2 x [box0,box1] (compressed.h:517-519)
        root = (np.array([-1, -1, H["box"][0]], np.float32), np.array([1, 1, H["box"][1]], np.float32))
        stack = [(0, root)]
        while stack:
            idx, box = stack.pop()
            for r in range(4):
                cb = _decode_child(int(words[idx]), box, r)
                child = 4 * idx + 1 + r
                if child < elems:
                    stack.append((child, cb))
                else:
                    cx, cy = _morton(child - elems)
                    q = pr[cy:cy + 2, cx:cx + 2].reshape(4, 3)
                    ext = np.maximum(cb[1] - cb[0], 1e-6)
                    # table steps are coarse (largest entry <= value): the decoded box may cut in by less than one step
                    slack = np.array([0.2, 0.2, 0.25]) * np.maximum(box[1] - box[0], 1e-6) + 1e-4
                    worst = max(worst, float(np.max(np.maximum(cb[0] - q, q - cb[1]) / ext)))
                    assert np.all(q >= cb[0] - slack) and np.all(q <= cb[1] + slack)
    sc.release()
    dev.release()


def test_oracle_hits_match_across_subdiv_modes(rtc, po, bomberman):
    """compressed.grid keeps the exact vertex grid, so its hit SET equals the eager path's (the reference reports
    162 467 for both); here at L=4 on 100 k rays through the oracle."""
    verts, fs, fi = bomberman
    res = {}
    for accel, mode in (("default", 2), ("bvh4.compressed.grid", 5)):
        dev = rtc.Device(f"gpu=none,subdiv_accel={accel}")
        sc = rtc.Scene(dev)
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 2)
        sc.commit()
        st = sc.stats()
        orc = po.SubdivScene(sc.accel_data(2), st["primBytes"], mode, 2)
        rays = po.make_random_rays(100_000, verts.min(0), verts.max(0), seed=1)
        orc.intersect1M(rays, nthreads=4)
        res[accel] = rays
        orc.free()
        sc.release()
        dev.release()
    a, b = res["default"], res["bvh4.compressed.grid"]
    ha, hb = a["geomID"] != 0xFFFFFFFF, b["geomID"] != 0xFFFFFFFF
    assert int((ha != hb).sum()) <= 2  # different edge rules (Pluecker vs Moeller-style) may flip a grazing ray
    both = ha & hb
    assert np.mean(a["primID"][both] == b["primID"][both]) > 0.999
    # the cBVH's quantized boxes are not strictly conservative (fork approximation): a few rays see a farther triangle
    close = np.isclose(a["tfar"][both], b["tfar"][both], rtol=1e-3, atol=1e-3)
    assert close.mean() > 0.995


def test_leaf_quantiser_matches_the_forks_header(rtc, po):
    """quantTris<4>::setZ / estimateExtent / getZ / getDelta of kernels/geometry/compressed_leaf.h, compiled where it lies
    (oracle/_ref/libref_fork.so, -ffp-contract=off), against the encoder's restatement (test hook of the product library)
    and against the decode formula the kernel and the oracle use.  Bit for bit."""
    import ctypes as C
    R = po.ref_fork()
    if R is None:
        pytest.skip("oracle/_ref/libref_fork.so not built (no reference tree)")
    L = rtc.lib()
    rng = np.random.RandomState(21)
    checked = 0
    for trial in range(4000):
        lo = (rng.randn(3) * 2).astype(np.float32)
        hi = (lo + np.abs(rng.randn(3)).astype(np.float32) * (0.01 if trial % 3 == 0 else 1.0) + np.float32(1e-3)).astype(np.float32)
        if trial % 97 == 0:
            hi[2] = lo[2]  # flat box: setZero branch
        box = np.concatenate([lo, hi]).astype(np.float32)
        # four corner vertices roughly over the box corners, heights scattered inside and outside the z range
        xy = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [lo[0], hi[1]], [hi[0], hi[1]]], np.float32)
        xy = (xy + rng.randn(4, 2).astype(np.float32) * 0.05 * (hi[:2] - lo[:2])).astype(np.float32)
        z = (lo[2] + (hi[2] - lo[2]) * rng.uniform(-0.4, 1.4, 4)).astype(np.float32)
        v = np.ascontiguousarray(np.concatenate([xy, z[:, None]], 1).astype(np.float32))
        ext_ref = R.ref_fork_estimate_extent(box.ctypes.data, v.ctypes.data)
        got_b, want_b = np.zeros(2, np.uint8), np.zeros(2, np.uint8)
        ext = C.c_float(0)
        extent_in = float(np.float32(ext_ref if trial % 2 else rng.uniform(0, 0.5)))
        L.rtcamdDebugCbvhLeafCodec(box.ctypes.data, v.ctypes.data, extent_in, got_b.ctypes.data, C.byref(ext))
        R.ref_fork_leaf_setZ(box.ctypes.data, v.ctypes.data, extent_in, want_b.ctypes.data)
        if np.isfinite(ext_ref):
            assert np.float32(ext.value) == np.float32(ext_ref), (trial, ext.value, ext_ref)
            assert np.array_equal(got_b, want_b), (trial, got_b, want_b)
            checked += 1
    assert checked > 3500
    # getDelta() = rcp(16.f) is rcpss + one Newton step in the reference (common/math/math.h:60-75): 0x3D7FFFFF on Intel
    # CPUs, exactly 0.0625 where rcpss is exact for powers of two.  Device code and oracle use the exact 0.0625
    # (rcp is a correctly rounded division on the GPU).
    delta = np.float32(R.ref_fork_leaf_delta())
    assert abs(float(delta) - 0.0625) <= 2.0 ** -28
    out = np.zeros(4, np.float32)
    for z12, z34 in ((0x00, 0xff), (0x1e, 0x73), (0xa5, 0x5a), (0xff, 0x00)):
        for rng_, off in ((np.float32(0.37), np.float32(-1.25)), (np.float32(1e-3), np.float32(17.5)), (np.float32(251.0), np.float32(0.0))):
            R.ref_fork_leaf_getZ(z12, z34, float(rng_), float(off), out.ctypes.data)
            rcpF = delta * rng_  # same operation order as the device / oracle decode: (delta*range) * nibble + offset
            want = [off + rcpF * np.float32(z12 >> 4), off + rcpF * np.float32(z12 & 15), off + rcpF * np.float32(z34 >> 4), off + rcpF * np.float32(z34 & 15)]
            assert np.array_equal(out, np.array(want, np.float32))


@pytest.mark.parametrize("accel,mode", [("default", 2), ("bvh4.compressed.box", 3), ("bvh4.compressed.leaf", 4), ("bvh4.compressed.grid", 5), ("bvh4.compressed.full", 6)])
def test_subdiv_golden_fixture_is_reproduced(rtc, po, bomberman, accel, mode):
    """tests/golden/bomberman_subdiv_hits.npz (tests/golden/make_golden_subdiv.py): host pipeline (tessellator, encoders,
    outer BVH) + oracle reproduce the committed hits; guards all of them against silent changes.  REGRESSION vectors of this
    implementation (product tessellator + product encoder + oracle in product arithmetic), not a parity claim."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bomberman_subdiv_hits.npz"))
    verts, fs, fi = bomberman
    dev = rtc.Device(f"gpu=none,subdiv_accel={accel}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(int(g["level"]), int(g["compression"]))
    sc.commit()
    ordered = mode in (3, 4, 6)
    orc = po.SubdivScene(sc.accel_data(2), sc.stats()["primBytes"], mode, int(g["compression"]),
                         qnodes=sc.accel_data(0) if ordered else None, root=sc.accel_root() if ordered else None)
    rays = po.make_random_rays(int(g["count"]), verts.min(0), verts.max(0), seed=int(g["seed"]))
    with po.fork_arith(1):  # the fixture holds vectors of THIS implementation: oracle in the product's arithmetic
        orc.intersect1M(rays, nthreads=4)
    key = accel.split(".")[-1]
    assert np.array_equal(rays["geomID"], g[f"{key}_geomID"]) and np.array_equal(rays["primID"], g[f"{key}_primID"])
    hit = rays["geomID"] != 0xFFFFFFFF
    for f in ("tfar", "u", "v"):
        a, b = rays[f][hit].astype(np.float64), g[f"{key}_{f}"][hit].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-3)), f  # rcpps differs between CPU vendors
    if mode != 2:
        # the same rays in the REFERENCE's arithmetic (rcp / rsqrt with Newton steps): same IDs; t/u/v of the box / leaf
        # approximations within 1e-4 except for a few rays per thousand that land in the neighbouring cell
        from helpers import fork_parity_stats
        ref = po.make_random_rays(int(g["count"]), verts.min(0), verts.max(0), seed=int(g["seed"]))
        orc.intersect1M(ref, nthreads=4)
        st = fork_parity_stats(rays, ref)
        assert st["hitmiss_flips"] == 0 and st["id_flips"] <= 1 and st["beyond_frac"] <= (0.0 if mode == 5 else 0.006), st
    orc.free()
    sc.release()
    dev.release()


@pytest.mark.parametrize("accel", ["default", "bvh4.compressed.leaf"])
def test_parallel_commit_is_deterministic(rtc, bomberman, accel):
    """`threads=N` only changes who does the work: nodes and leaf records are byte-identical to the single-threaded build
    (the upper levels of the BVH are built in forked subtrees and spliced, the cBVH blobs are encoded in parallel)."""
    verts, fs, fi = bomberman
    out = []
    for cfg in ("threads=1", "threads=8"):
        dev = rtc.Device(f"gpu=none,subdiv_accel={accel},{cfg}")
        sc = rtc.Scene(dev)
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(5, 3)
        sc.commit()
        out.append((sc.accel_data(0).tobytes(), sc.accel_data(2).tobytes(), sc.accel_root()))
        sc.release()
        dev.release()
    assert out[0] == out[1]
