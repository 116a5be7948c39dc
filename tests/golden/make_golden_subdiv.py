#!/usr/bin/env python3
"""Generates tests/golden/bomberman_subdiv_hits.npz: closest hits of the first 20 000 rays of the seed-0 BASELINE generator
against bomberman as Catmull-Clark subdivision surface at level 4 / compression 2, for the eager GridSOA path and the
fork's four compressed modes.  The leaf records are built by the product on a host-only device (`gpu=none`: tessellator
+ encoders, no traversal) and traversed by the ORACLE (oracle/liboracle.so), after the oracle and the pipeline were pinned
(tests/test_oracle.py; reference hit counts of SURVEY.md section 6 at level 6: eager and compressed.grid 162 467).
For the order-dependent box / leaf modes the oracle walks the product's outer BVH8.

These are regression vectors of THIS implementation (the reference library cannot be built under this round's rules): they
freeze tessellator, encoders, outer BVH and oracle together, and let the GPU tests compare against committed numbers.

    python tests/golden/make_golden_subdiv.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

rtc = importlib.import_module("embree-compressed_amd").rtc
d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
v, fs, fi = d["verts"], d["face_sizes"], d["face_index"]
M, SEED, L, C = 20000, 0, 4, 2
MODES = {"default": 2, "bvh4.compressed.box": 3, "bvh4.compressed.leaf": 4, "bvh4.compressed.grid": 5, "bvh4.compressed.full": 6}
PATH = os.path.join(ROOT, "tests", "golden", "bomberman_subdiv_hits.npz")
out = {"count": M, "seed": SEED, "level": L, "compression": C}
if os.path.exists(PATH) and "--all" not in sys.argv:  # default: keep the committed vectors, add the modes that are missing (round 2: full)
    old = np.load(PATH)
    assert all(int(old[k]) == out[k] for k in out)
    out.update({k: old[k] for k in old.files})
for name, mode in MODES.items():
    if name.split(".")[-1] + "_geomID" in out:
        continue
    dev = rtc.Device(f"gpu=none,subdiv_accel={name}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi)
    sc.set_levels(L, C)
    sc.commit()
    st = sc.stats()
    ordered = mode in (3, 4, 6)
    orc = po.SubdivScene(sc.accel_data(2), st["primBytes"], mode, C, qnodes=sc.accel_data(0) if ordered else None, root=sc.accel_root() if ordered else None)
    rays = po.make_random_rays(M, v.min(0), v.max(0), seed=SEED)
    with po.fork_arith(1):  # product arithmetic: what the tests compare the vectors in
        orc.intersect1M(rays, nthreads=8)
    key = name.split(".")[-1]
    for f in ("geomID", "primID", "tfar", "u", "v"):
        out[f"{key}_{f}"] = rays[f].copy()
    print(name, int((rays["geomID"] != 0xFFFFFFFF).sum()), "hits of", M)
    orc.free()
    sc.release()
    dev.release()
np.savez_compressed(PATH, **out)
