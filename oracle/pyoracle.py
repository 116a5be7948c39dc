"""ORACLE — TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (and oracle/_ref/libref_prims.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref_prims.so")

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.orc_scene_new_triangles.restype = vp
        L.orc_scene_new_triangles.argtypes = [vp, sz, vp, vp, vp, sz, C.c_int]
        L.orc_scene_free.argtypes = [vp]
        L.orc_intersect1.argtypes = [vp, vp, u32]
        L.orc_occluded1.argtypes = [vp, vp]
        L.orc_intersect1M.argtypes = [vp, vp, u32, sz, u32, C.c_int]
        L.orc_occluded1M.argtypes = [vp, vp, u32, sz, C.c_int]
        L.orc_get_counters.argtypes = [vp]
        L.orc_rcp.restype = C.c_float
        L.orc_rcp.argtypes = [C.c_float]
        for n in ("orc_rsqrt", "orc_dot3fa", "orc_length3"):
            getattr(L, n).restype = C.c_float
        L.orc_rsqrt.argtypes = [C.c_float]
        L.orc_dot3fa.argtypes = [vp, vp]
        L.orc_length3.argtypes = [vp]
        L.orc_normalize3.argtypes = [vp, vp]
        L.orc_set_fork_arith.argtypes = [C.c_int]
        L.orc_get_fork_arith.restype = C.c_int
        L.orc_dot.restype = C.c_float
        L.orc_dot.argtypes = [vp, vp]
        L.orc_cross.argtypes = [vp, vp, vp]
        L.orc_stable_triangle_normal.argtypes = [vp, vp, vp, vp]
        L.orc_pluecker_block.restype = C.c_int
        L.orc_pluecker_block.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_float, vp]
        L.orc_moeller_block.restype = C.c_int
        L.orc_moeller_block.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_float, vp]
        L.orc_make_random_rays.argtypes = [vp, u32, sz, vp, vp, C.c_uint64, C.c_int]
        _lib = L
    return _lib


def ref():
    """The reference's own common/ primitives (oracle/_ref), or None if the prebuilt library is absent."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        R = C.CDLL(REF_PATH)
        vp = C.c_void_p
        R.ref_rcp.restype = C.c_float
        R.ref_rcp.argtypes = [C.c_float]
        for n in ("ref_rcp4", "ref_rcp_safe3", "ref_zero_fix3", "ref_normalize3", "ref_rcp_safe3f"):
            getattr(R, n).argtypes = [vp, vp]
        for n in ("ref_rsqrt", "ref_dot3fa", "ref_length3"):
            getattr(R, n).restype = C.c_float
        R.ref_rsqrt.argtypes = [C.c_float]
        R.ref_dot3fa.argtypes = [vp, vp]
        R.ref_length3.argtypes = [vp]
        for n in ("ref_dot4", "ref_cross4"):
            getattr(R, n).argtypes = [vp, vp, vp]
        R.ref_stable_triangle_normal4.argtypes = [vp, vp, vp, vp]
        R.ref_select_min4.restype = C.c_int
        R.ref_select_min4.argtypes = [C.c_uint, vp]
        R.ref_max4of.argtypes = [vp] * 5
        R.ref_min4of.argtypes = [vp] * 5
        _ref = R
    return _ref


_ref_fork = None


def ref_fork():
    """The fork's leaf codec header compiled in place (oracle/_ref/libref_fork.so), or None if absent."""
    global _ref_fork
    path = os.path.join(os.path.dirname(REF_PATH), "libref_fork.so")
    if _ref_fork is None and os.path.exists(path):
        R = C.CDLL(path)
        vp = C.c_void_p
        R.ref_fork_leaf_setZ.argtypes = [vp, vp, C.c_float, vp]
        R.ref_fork_estimate_extent.restype = C.c_float
        R.ref_fork_estimate_extent.argtypes = [vp, vp]
        R.ref_fork_leaf_getZ.argtypes = [C.c_ubyte, C.c_ubyte, C.c_float, C.c_float, vp]
        R.ref_fork_leaf_delta.restype = C.c_float
        _ref_fork = R
    return _ref_fork


class TriangleScene:
    """mode 0: BVH8/Triangle4v/robust/Pluecker; mode 1: BVH8/Triangle4/fast/Moeller."""

    def __init__(self, verts, tris, mode=0, geom_ids=None, prim_ids=None):
        self.L = lib()
        self.v = np.ascontiguousarray(verts, dtype=np.float32)
        self.t = np.ascontiguousarray(tris, dtype=np.uint32)
        self.g = None if geom_ids is None else np.ascontiguousarray(geom_ids, dtype=np.uint32)
        self.p = None if prim_ids is None else np.ascontiguousarray(prim_ids, dtype=np.uint32)
        self.handle = self.L.orc_scene_new_triangles(
            self.v.ctypes.data, self.v.shape[0], self.t.ctypes.data,
            None if self.g is None else self.g.ctypes.data, None if self.p is None else self.p.ctypes.data,
            self.t.shape[0], mode)

    def intersect1M(self, rayhits, inst_id=0xFFFFFFFF, nthreads=1):
        self.L.orc_intersect1M(self.handle, rayhits.ctypes.data, rayhits.shape[0], rayhits.strides[0], inst_id, nthreads)

    def occluded1M(self, rays, nthreads=1):
        self.L.orc_occluded1M(self.handle, rays.ctypes.data, rays.shape[0], rays.strides[0], nthreads)

    def counters(self):
        out = (C.c_ulonglong * 3)()
        self.L.orc_get_counters(out)
        return {"nodes": out[0], "leaves": out[1], "blocks": out[2]}

    def free(self):
        if self.handle:
            self.L.orc_scene_free(self.handle)
            self.handle = None


class SubdivScene(TriangleScene):
    """Scene over leaf records exported by the product (rtcamdGetAccelData kind 2).
    mode 2: eager grid cells (stride 160); 3/4/5/6: fork cBVH blobs box/leaf/grid/full."""

    def __init__(self, blobs, stride, mode, levels=3, qnodes=None, root=None):
        """qnodes/root given: traverse the product's outer BVH8 (same visiting order as the device kernels)."""
        self.L = lib()
        self.L.orc_scene_new_subdiv.restype = C.c_void_p
        self.L.orc_scene_new_subdiv.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
        self.L.orc_scene_new_subdiv_qbvh.restype = C.c_void_p
        self.L.orc_scene_new_subdiv_qbvh.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.c_uint32]
        self.blobs = np.ascontiguousarray(blobs, dtype=np.uint8)
        assert self.blobs.size % stride == 0
        if qnodes is None:
            self.handle = self.L.orc_scene_new_subdiv(self.blobs.ctypes.data, stride, self.blobs.size // stride, mode, levels)
        else:
            self.q = np.ascontiguousarray(qnodes, dtype=np.uint8)
            assert self.q.size % 96 == 0
            self.handle = self.L.orc_scene_new_subdiv_qbvh(self.blobs.ctypes.data, stride, self.blobs.size // stride, mode, levels,
                                                           self.q.ctypes.data, self.q.size // 96, root)


class fork_arith:
    """with fork_arith(1): ...  -> the cBVH helpers of the oracle use the PRODUCT's device arithmetic (bit-for-bit regression
    checks of the HIP kernels); default 0 = the reference's rcp / rsqrt / dpps arithmetic (embree_oracle.h)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib().orc_get_fork_arith()
        lib().orc_set_fork_arith(self.mode)

    def __exit__(self, *a):
        lib().orc_set_fork_arith(self.prev)


def make_random_rays(m, lo, hi, seed=0, double_eval=False, dtype=None):
    """RTCRayHit records from the drand48-compatible generator of BASELINE.md section 3."""
    from numpy import dtype as _dt  # noqa: F401
    rec = np.dtype([("org_x", "<f4"), ("org_y", "<f4"), ("org_z", "<f4"), ("tnear", "<f4"),
                    ("dir_x", "<f4"), ("dir_y", "<f4"), ("dir_z", "<f4"), ("time", "<f4"),
                    ("tfar", "<f4"), ("mask", "<u4"), ("id", "<u4"), ("flags", "<u4"),
                    ("Ng_x", "<f4"), ("Ng_y", "<f4"), ("Ng_z", "<f4"), ("u", "<f4"), ("v", "<f4"),
                    ("primID", "<u4"), ("geomID", "<u4"), ("instID", "<u4")])
    raw = np.zeros(m * 80 + 16, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    rays = raw[off: off + m * 80].view(rec)
    lo = np.ascontiguousarray(lo, dtype=np.float32)
    hi = np.ascontiguousarray(hi, dtype=np.float32)
    lib().orc_make_random_rays(rays.ctypes.data, m, 80, lo.ctypes.data, hi.ctypes.data, seed, 1 if double_eval else 0)
    return rays
