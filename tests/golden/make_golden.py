#!/usr/bin/env python3
"""Generates tests/golden/bomberman_tri_hits.npz: closest hits of the first 20 000 rays of the seed-0
BASELINE generator against bomberman's fan triangles, computed by the oracle (oracle/liboracle.so) AFTER the
oracle was checked against the reference's TriangleHitTest and the reference outputs of SURVEY.md section 8d
(227 188 hits / sum primID 10 389 122 on the full 1 M-ray set).  IDs are identical for the Pluecker (mode 0)
and Moeller (mode 1) paths; t/u/v are stored per mode.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
v, fs, fi = d["verts"], d["face_sizes"], d["face_index"]
tris, p = [], 0
for n in fs:
    for k in range(2, int(n)):
        tris.append((fi[p], fi[p + k - 1], fi[p + k]))
    p += int(n)
tris = np.asarray(tris, np.uint32)
M, SEED = 20000, 0
out = {"count": M, "seed": SEED}
for mode in (0, 1):
    rays = po.make_random_rays(M, v.min(0), v.max(0), seed=SEED)
    sc = po.TriangleScene(v, tris, mode)
    sc.intersect1M(rays)
    if mode == 0:
        out["geomID"], out["primID"] = rays["geomID"].copy(), rays["primID"].copy()
    else:
        assert np.array_equal(out["geomID"], rays["geomID"]) and np.array_equal(out["primID"], rays["primID"])
    for f in ("tfar", "u", "v"):
        out[f"{f}_{mode}"] = rays[f].copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bomberman_tri_hits.npz"), **out)
print("wrote bomberman_tri_hits.npz:", int((out["geomID"] != 0xFFFFFFFF).sum()), "hits of", M)
