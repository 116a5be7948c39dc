#!/usr/bin/env python3
"""Generates tests/golden/bomberman_fork_refarith.npz: closest hits of the first 100 000 rays of the seed-0 BASELINE generator
against bomberman at level 6 / compression 3 (the metric's scene) for the fork's order-dependent modes compressed.leaf / box / full,
traced by the ORACLE in the REFERENCE's arithmetic (rcp / rsqrt = the CPU's rcpss / rsqrtss estimate + one Newton step, dpps dot:
oracle/subdiv_oracle.inc, each building block pinned to the reference's own headers by tests/test_oracle.py) ON THE CPU OF THE
BUILD CONTAINER (recorded in the file; round 3: Intel Xeon).  The leaf records and the outer BVH8 come from the product's host
builders (`gpu=none`), the oracle walks the product's outer tree (the modes depend on the order blobs are reached in).

Why: rcpss / rsqrtss tables differ between CPU vendors, so the reference's own results on these discontinuous modes differ between an
Intel and an AMD host.  With this file the GPU box (AMD EPYC host) can report oracle(EPYC) vs fixture(Intel) next to GPU vs
fixture(Intel) (tests/test_gpu_fork_fixture.py, profiles/r03_parity_report.txt): the first is the reference-arithmetic spread between
CPU vendors, the second the product's distance from a reference-arithmetic run - both in the same classes.

Vectors of the ORACLE, not of the reference library (which cannot be built here): the rows stay "parity unpinned".
Only the hits are stored (ray index, IDs, t, u, v); a ray that is not listed missed.

    python tests/golden/make_golden_fork_ref.py
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
M, SEED, L, C = 100_000, 0, 6, 3
MODES = {"bvh4.compressed.box": 3, "bvh4.compressed.leaf": 4, "bvh4.compressed.full": 6}
cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
out = {"count": M, "seed": SEED, "level": L, "compression": C, "cpu": cpu, "rcp_probe": np.array([po.lib().orc_rcp(x) for x in (3.0, 7.0, 1.1, 1e-3)], np.float32)}
for name, mode in MODES.items():
    dev = rtc.Device(f"gpu=none,subdiv_accel={name}")
    sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi)
    sc.set_levels(L, C)
    sc.commit()
    orc = po.SubdivScene(sc.accel_data(2), sc.stats()["primBytes"], mode, C, qnodes=sc.accel_data(0), root=sc.accel_root())
    rays = po.make_random_rays(M, v.min(0), v.max(0), seed=SEED, double_eval=True)
    assert po.lib().orc_get_fork_arith() == 0  # reference arithmetic
    orc.intersect1M(rays, nthreads=8)
    hit = np.nonzero(rays["geomID"] != 0xFFFFFFFF)[0].astype(np.uint32)
    key = name.split(".")[-1]
    out[f"{key}_index"] = hit
    for f in ("geomID", "primID", "tfar", "u", "v"):
        out[f"{key}_{f}"] = rays[f][hit].copy()
    print(name, len(hit), "hits of", M, "on", cpu)
    orc.free()
    sc.release()
    dev.release()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bomberman_fork_refarith.npz"), **out)
