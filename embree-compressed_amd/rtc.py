"""ctypes binding of the C-ABI library (lib/libembree3.so) — the same entry points a C application or the
reference's tutorials call (include/embree3/rtcore.h), plus the rtcamd* extensions (rtcore_amd.h).

This module is plumbing for tests and bench.py.  There is no Python or CPU implementation of the hot path
behind it: if the HIP library is missing, importing fails; if no GPU is present, rtcNewDevice fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB", os.path.join(_HERE, "lib", "libembree3.so"))  # RTAMD_LIB: A/B builds of the same library

RTC_INVALID_GEOMETRY_ID = 0xFFFFFFFF

# enums (include/embree3/rtcore.h)
RTC_GEOMETRY_TYPE_TRIANGLE = 0
RTC_GEOMETRY_TYPE_QUAD = 1
RTC_GEOMETRY_TYPE_SUBDIVISION = 8
RTC_BUFFER_TYPE_INDEX = 0
RTC_BUFFER_TYPE_VERTEX = 1
RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE = 2
RTC_BUFFER_TYPE_FACE = 16
RTC_BUFFER_TYPE_LEVEL = 17
RTC_BUFFER_TYPE_EDGE_CREASE_INDEX = 18
RTC_BUFFER_TYPE_EDGE_CREASE_WEIGHT = 19
RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX = 20
RTC_BUFFER_TYPE_VERTEX_CREASE_WEIGHT = 21
RTC_FORMAT_UINT = 0x5001
RTC_FORMAT_UINT3 = 0x5003
RTC_FORMAT_FLOAT = 0x9001
RTC_FORMAT_FLOAT3 = 0x9003
RTC_SCENE_FLAG_NONE = 0
RTC_SCENE_FLAG_ROBUST = 4
RTC_ERROR_NONE, RTC_ERROR_UNKNOWN, RTC_ERROR_INVALID_ARGUMENT, RTC_ERROR_INVALID_OPERATION = 0, 1, 2, 3
RTC_ERROR_OUT_OF_MEMORY, RTC_ERROR_UNSUPPORTED_CPU, RTC_ERROR_CANCELLED = 4, 5, 6

# RTCRayHit / RTCRay as numpy record types (80 / 48 bytes)
RAY_FIELDS = [("org_x", "<f4"), ("org_y", "<f4"), ("org_z", "<f4"), ("tnear", "<f4"),
              ("dir_x", "<f4"), ("dir_y", "<f4"), ("dir_z", "<f4"), ("time", "<f4"),
              ("tfar", "<f4"), ("mask", "<u4"), ("id", "<u4"), ("flags", "<u4")]
HIT_FIELDS = [("Ng_x", "<f4"), ("Ng_y", "<f4"), ("Ng_z", "<f4"), ("u", "<f4"), ("v", "<f4"),
              ("primID", "<u4"), ("geomID", "<u4"), ("instID", "<u4")]
RAY_DTYPE = np.dtype(RAY_FIELDS)
RAYHIT_DTYPE = np.dtype(RAY_FIELDS + HIT_FIELDS)
assert RAY_DTYPE.itemsize == 48 and RAYHIT_DTYPE.itemsize == 80


class RTCIntersectContext(C.Structure):
    _fields_ = [("flags", C.c_int), ("filter", C.c_void_p), ("instID", C.c_uint * 1)]


class RTCBounds(C.Structure):
    _fields_ = [("lower_x", C.c_float), ("lower_y", C.c_float), ("lower_z", C.c_float), ("align0", C.c_float),
                ("upper_x", C.c_float), ("upper_y", C.c_float), ("upper_z", C.c_float), ("align1", C.c_float)]


class RTCAMDSceneStats(C.Structure):
    _fields_ = [("byteSize", C.c_size_t), ("accelKind", C.c_uint), ("branching", C.c_uint),
                ("nodeCount", C.c_size_t), ("nodeBytes", C.c_size_t), ("primCount", C.c_size_t),
                ("primBytes", C.c_size_t), ("leafCount", C.c_size_t), ("totalBytes", C.c_size_t),
                ("maxDepth", C.c_uint), ("reserved", C.c_uint)]


class RTCFilterFunctionNArguments(C.Structure):
    _fields_ = [("valid", C.POINTER(C.c_int)), ("geometryUserPtr", C.c_void_p), ("context", C.c_void_p), ("ray", C.c_void_p),
                ("hit", C.c_void_p), ("N", C.c_uint)]


FILTER_FUNC = C.CFUNCTYPE(None, C.POINTER(RTCFilterFunctionNArguments))


class RTCInterpolateArguments(C.Structure):
    _fields_ = [("geometry", C.c_void_p), ("primID", C.c_uint), ("u", C.c_float), ("v", C.c_float), ("bufferType", C.c_int),
                ("bufferSlot", C.c_uint), ("P", C.POINTER(C.c_float)), ("dPdu", C.POINTER(C.c_float)), ("dPdv", C.POINTER(C.c_float)),
                ("ddPdudu", C.POINTER(C.c_float)), ("ddPdvdv", C.POINTER(C.c_float)), ("ddPdudv", C.POINTER(C.c_float)),
                ("valueCount", C.c_uint)]


class RTCInterpolateNArguments(C.Structure):
    _fields_ = [("geometry", C.c_void_p), ("valid", C.c_void_p), ("primIDs", C.POINTER(C.c_uint)), ("u", C.POINTER(C.c_float)),
                ("v", C.POINTER(C.c_float)), ("N", C.c_uint), ("bufferType", C.c_int), ("bufferSlot", C.c_uint),
                ("P", C.POINTER(C.c_float)), ("dPdu", C.POINTER(C.c_float)), ("dPdv", C.POINTER(C.c_float)),
                ("ddPdudu", C.POINTER(C.c_float)), ("ddPdvdv", C.POINTER(C.c_float)), ("ddPdudv", C.POINTER(C.c_float)),
                ("valueCount", C.c_uint)]


# rtcore_amd.h: enum RTCAMDDeviceProperty
RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES = 240
RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS = 241
RTCAMD_DEVICE_PROPERTY_COMBINED_BATCHES = 242
RTCAMD_DEVICE_PROPERTY_SERVICE_CALLS = 243


class RTCAMDTraceCounters(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in
                ("rays", "nodeVisits", "leafVisits", "primTests", "innerVisits", "hits", "stackSpills", "reserved",
                 "cyclesFetch", "cyclesNode", "cyclesLeaf", "cyclesPop", "cyclesTotal", "iterations", "leafPhases", "waves",
                 "activeLaneIters", "startInv", "maxRaySteps", "drainTicksSum", "drainTicksMax")] + [("waveEndHist", C.c_ulonglong * 64), ("waveIterHist", C.c_ulonglong * 64)]


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C embree-compressed_amd` "
                          f"(or __graft_entry__.build()); there is no fallback implementation")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.  If this
    # library were loaded first it would bind the system copies and torch would then find "no HIP GPUs".  Importing
    # torch first makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped, so
    # torch tensors and this library share one runtime (device pointers, streams and events are interchangeable).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, u, sz = C.c_void_p, C.c_uint, C.c_size_t
    sig = {
        "rtcNewDevice": (vp, [C.c_char_p]),
        "rtcRetainDevice": (None, [vp]),
        "rtcReleaseDevice": (None, [vp]),
        "rtcGetDeviceProperty": (C.c_ssize_t, [vp, C.c_int]),
        "rtcGetDeviceError": (C.c_int, [vp]),
        "rtcSetDeviceErrorFunction": (None, [vp, vp, vp]),
        "rtcNewBuffer": (vp, [vp, sz]),
        "rtcNewSharedBuffer": (vp, [vp, vp, sz]),
        "rtcGetBufferData": (vp, [vp]),
        "rtcReleaseBuffer": (None, [vp]),
        "rtcNewGeometry": (vp, [vp, C.c_int]),
        "rtcRetainGeometry": (None, [vp]),
        "rtcReleaseGeometry": (None, [vp]),
        "rtcCommitGeometry": (None, [vp]),
        "rtcEnableGeometry": (None, [vp]),
        "rtcDisableGeometry": (None, [vp]),
        "rtcSetGeometryBuffer": (None, [vp, C.c_int, u, C.c_int, vp, sz, sz, sz]),
        "rtcSetSharedGeometryBuffer": (None, [vp, C.c_int, u, C.c_int, vp, sz, sz, sz]),
        "rtcSetNewGeometryBuffer": (vp, [vp, C.c_int, u, C.c_int, sz, sz]),
        "rtcGetGeometryBufferData": (vp, [vp, C.c_int, u]),
        "rtcSetGeometryDisplacementFunction": (None, [vp, vp]),
        "rtcSetGeometrySubdivisionMode": (None, [vp, u, C.c_int]),
        "rtcSetGeometryTessellationRate": (None, [vp, C.c_float]),
        "rtcSetGeometryUserData": (None, [vp, vp]),
        "rtcSetGeometryIntersectFilterFunction": (None, [vp, vp]),
        "rtcSetGeometryOccludedFilterFunction": (None, [vp, vp]),
        "rtcNewScene": (vp, [vp]),
        "rtcRetainScene": (None, [vp]),
        "rtcReleaseScene": (None, [vp]),
        "rtcAttachGeometry": (u, [vp, vp]),
        "rtcAttachGeometryByID": (None, [vp, vp, u]),
        "rtcDetachGeometry": (None, [vp, u]),
        "rtcGetGeometry": (vp, [vp, u]),
        "rtcSetGeometryVertexAttributeCount": (None, [vp, u]),
        "rtcSetGeometryTopologyCount": (None, [vp, u]),
        "rtcSetGeometrySubdivisionMode": (None, [vp, u, C.c_int]),
        "rtcSetGeometryVertexAttributeTopology": (None, [vp, u, u]),
        "rtcInterpolate": (None, [C.POINTER(RTCInterpolateArguments)]),
        "rtcInterpolateN": (None, [C.POINTER(RTCInterpolateNArguments)]),
        "rtcSetSceneLevels": (None, [vp, u, u]),
        "rtcSetSceneFlags": (None, [vp, C.c_int]),
        "rtcGetSceneFlags": (C.c_int, [vp]),
        "rtcSetSceneBuildQuality": (None, [vp, C.c_int]),
        "rtcCommitScene": (None, [vp]),
        "rtcJoinCommitScene": (None, [vp]),
        "rtcGetSceneBounds": (None, [vp, C.POINTER(RTCBounds)]),
        "rtcIntersect1": (None, [vp, C.POINTER(RTCIntersectContext), vp]),
        "rtcIntersect1M": (None, [vp, C.POINTER(RTCIntersectContext), vp, u, sz]),
        "rtcOccluded1": (None, [vp, C.POINTER(RTCIntersectContext), vp]),
        "rtcOccluded1M": (None, [vp, C.POINTER(RTCIntersectContext), vp, u, sz]),
        "rtcIntersect1Mp": (None, [vp, C.POINTER(RTCIntersectContext), vp, u]),
        "rtcOccluded1Mp": (None, [vp, C.POINTER(RTCIntersectContext), vp, u]),
        "rtcIntersect4": (None, [vp, vp, C.POINTER(RTCIntersectContext), vp]),
        "rtcamdGetDeviceStream": (vp, [vp]),
        "rtcamdSetDeviceStream": (None, [vp, vp]),
        "rtcamdSynchronizeDevice": (None, [vp]),
        "rtcamdGetDeviceOrdinal": (C.c_int, [vp]),
        "rtcamdGetSceneStats": (None, [vp, C.POINTER(RTCAMDSceneStats)]),
        "rtcamdIntersect1MCounted": (None, [vp, C.POINTER(RTCIntersectContext), vp, u, sz, C.POINTER(RTCAMDTraceCounters)]),
        "rtcamdOccluded1MCounted": (None, [vp, C.POINTER(RTCIntersectContext), vp, u, sz, C.POINTER(RTCAMDTraceCounters)]),
        "rtcamdGetAccelData": (vp, [vp, u, C.POINTER(sz)]),
        "rtcamdGetAccelRoot": (u, [vp]),
        "rtcamdDebugReadWaveLog": (C.c_size_t, [vp, vp, C.c_size_t]),
        "rtcamdDebugHoldCombiner": (None, [vp, C.c_int]),
        "rtcamdDebugCbvhLeafCodec": (None, [vp, vp, C.c_float, vp, C.POINTER(C.c_float)]),
        "rtcamdDebugHostPoolSelfTest": (C.c_ulonglong, [vp, u, u, u, u]),
    }
    for name, (res, args) in sig.items():
        if name.startswith("rtcamdDebug") and not hasattr(lib, name):
            continue  # development hooks: an older build of the library (RTAMD_LIB=... in an A/B sweep) may lack the newest one
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


class RTCError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"embree3 error {code} in {where}")
        self.code = code


RTC_INTERSECT_CONTEXT_FLAG_INCOHERENT = 0
RTC_INTERSECT_CONTEXT_FLAG_COHERENT = 1


def make_context(inst_id=RTC_INVALID_GEOMETRY_ID, coherent=False):
    """coherent=True: RTC_INTERSECT_CONTEXT_FLAG_COHERENT, the hint the reference's viewer_stream sets for primary rays; on
    subdiv_accel=bvh4.compressed.* it selects the one-ray-per-lane blob walk instead of the quad form."""
    ctx = RTCIntersectContext()
    ctx.flags = RTC_INTERSECT_CONTEXT_FLAG_COHERENT if coherent else RTC_INTERSECT_CONTEXT_FLAG_INCOHERENT
    ctx.filter = None
    ctx.instID[0] = inst_id
    return ctx


class Device:
    """RTCDevice wrapper.  cfg is the embree config string; 'gpu=N' selects the HIP device."""

    def __init__(self, cfg=""):
        self.lib = lib()
        self.handle = self.lib.rtcNewDevice(cfg.encode())
        if not self.handle:
            raise RTCError(self.lib.rtcGetDeviceError(None), f"rtcNewDevice({cfg!r})")

    def error(self):
        return self.lib.rtcGetDeviceError(self.handle)

    def check(self, where=""):
        e = self.error()
        if e != RTC_ERROR_NONE:
            raise RTCError(e, where)

    def stream(self):
        return self.lib.rtcamdGetDeviceStream(self.handle)

    def get_property(self, prop):
        v = self.lib.rtcGetDeviceProperty(self.handle, int(prop))
        self.check("rtcGetDeviceProperty")
        return int(v)

    def set_stream(self, hip_stream):
        self.lib.rtcamdSetDeviceStream(self.handle, hip_stream)
        self.check("rtcamdSetDeviceStream")

    def synchronize(self):
        self.lib.rtcamdSynchronizeDevice(self.handle)
        self.check("rtcamdSynchronizeDevice")

    def release(self):
        if self.handle:
            self.lib.rtcReleaseDevice(self.handle)
            self.handle = None


def _ptr_and_count(buf):
    """(address, record count, byte stride) of a numpy record array or a torch uint8/structured tensor."""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data, buf.shape[0], buf.strides[0]
    # torch tensor [M, recBytes] uint8 or [M, recBytes/4] float32/int32
    return buf.data_ptr(), buf.shape[0], buf.stride(0) * buf.element_size()


class Scene:
    def __init__(self, device, flags=RTC_SCENE_FLAG_NONE):
        self.device = device
        self.lib = device.lib
        self.handle = self.lib.rtcNewScene(device.handle)
        device.check("rtcNewScene")
        if flags:
            self.lib.rtcSetSceneFlags(self.handle, flags)
        self._keep = []  # shared buffers must outlive the scene

    def add_triangles(self, verts, tris, geom_id=None):
        """verts float32 [nv,3], tris uint32 [nt,3]; uses shared buffers like the reference's tutorials."""
        L = self.lib
        v = np.ascontiguousarray(verts, dtype=np.float32)
        # 16 bytes of readable padding after the last vertex (verify.cpp:2143-2148)
        vpad = np.zeros((v.shape[0] + 2, 3), dtype=np.float32)
        vpad[: v.shape[0]] = v
        t = np.ascontiguousarray(tris, dtype=np.uint32)
        g = L.rtcNewGeometry(self.device.handle, RTC_GEOMETRY_TYPE_TRIANGLE)
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, vpad.ctypes.data, 0, 12, v.shape[0])
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, t.ctypes.data, 0, 12, t.shape[0])
        L.rtcCommitGeometry(g)
        if geom_id is None:
            gid = L.rtcAttachGeometry(self.handle, g)
        else:
            L.rtcAttachGeometryByID(self.handle, g, geom_id)
            gid = geom_id
        L.rtcReleaseGeometry(g)
        self._keep += [vpad, t]
        self.device.check("add_triangles")
        return gid

    def add_subdiv(self, verts, face_sizes, face_index, level=1.0, geom_id=None, displacement=None, user_data=None,
                   edge_creases=None, vertex_creases=None):
        """edge_creases: (uint32 [k,2] vertex pairs, float32 [k] weights); vertex_creases: (uint32 [m], float32 [m])."""
        L = self.lib
        v = np.ascontiguousarray(verts, dtype=np.float32)
        vpad = np.zeros((v.shape[0] + 2, 3), dtype=np.float32)
        vpad[: v.shape[0]] = v
        fs = np.ascontiguousarray(face_sizes, dtype=np.uint32)
        fi = np.ascontiguousarray(face_index, dtype=np.uint32)
        lv = np.full(fi.shape[0], level, dtype=np.float32)
        g = L.rtcNewGeometry(self.device.handle, RTC_GEOMETRY_TYPE_SUBDIVISION)
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, vpad.ctypes.data, 0, 12, v.shape[0])
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_FACE, 0, RTC_FORMAT_UINT, fs.ctypes.data, 0, 4, fs.shape[0])
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT, fi.ctypes.data, 0, 4, fi.shape[0])
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_LEVEL, 0, RTC_FORMAT_FLOAT, lv.ctypes.data, 0, 4, lv.shape[0])
        if edge_creases is not None:
            ei = np.ascontiguousarray(edge_creases[0], dtype=np.uint32).reshape(-1, 2)
            ew = np.ascontiguousarray(edge_creases[1], dtype=np.float32)
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_EDGE_CREASE_INDEX, 0, RTC_FORMAT_UINT + 1, ei.ctypes.data, 0, 8, ei.shape[0])
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_EDGE_CREASE_WEIGHT, 0, RTC_FORMAT_FLOAT, ew.ctypes.data, 0, 4, ew.shape[0])
            self._keep += [ei, ew]
        if vertex_creases is not None:
            vi = np.ascontiguousarray(vertex_creases[0], dtype=np.uint32)
            vw = np.ascontiguousarray(vertex_creases[1], dtype=np.float32)
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX, 0, RTC_FORMAT_UINT, vi.ctypes.data, 0, 4, vi.shape[0])
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX_CREASE_WEIGHT, 0, RTC_FORMAT_FLOAT, vw.ctypes.data, 0, 4, vw.shape[0])
            self._keep += [vi, vw]
        if displacement is not None:
            L.rtcSetGeometryDisplacementFunction(g, C.cast(displacement, C.c_void_p))
            self._keep.append(displacement)
        if user_data is not None:
            L.rtcSetGeometryUserData(g, user_data)
        L.rtcCommitGeometry(g)
        if geom_id is None:
            gid = L.rtcAttachGeometry(self.handle, g)
        else:
            L.rtcAttachGeometryByID(self.handle, g, geom_id)
            gid = geom_id
        L.rtcReleaseGeometry(g)
        self._keep += [vpad, fs, fi, lv]
        self.device.check("add_subdiv")
        return gid

    def set_filters(self, geom_id, intersect=None, occluded=None):
        """intersect / occluded: FILTER_FUNC objects (kept alive by the scene wrapper) or None."""
        g = self.lib.rtcGetGeometry(self.handle, geom_id)
        self.lib.rtcSetGeometryIntersectFilterFunction(g, C.cast(intersect, C.c_void_p) if intersect is not None else None)
        self.lib.rtcSetGeometryOccludedFilterFunction(g, C.cast(occluded, C.c_void_p) if occluded is not None else None)
        self.lib.rtcCommitGeometry(g)
        self._keep += [intersect, occluded]
        self.device.check("set_filters")

    def set_vertex_attribute(self, geom_id, slot, values, topology_index=None, mode=None):
        """Bind a float32 [nv, k] (k <= 4) array as vertex attribute `slot` of an attached geometry and re-commit it.
        topology_index: uint32 face-vertex indices of a second topology for this attribute (face-varying data); mode: its
        RTCSubdivisionMode."""
        L = self.lib
        if topology_index is not None:
            g = L.rtcGetGeometry(self.handle, geom_id)
            ti = np.ascontiguousarray(topology_index, dtype=np.uint32)
            L.rtcSetGeometryTopologyCount(g, 2)
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 1, RTC_FORMAT_UINT, ti.ctypes.data, 0, 4, ti.shape[0])
            if mode is not None:
                L.rtcSetGeometrySubdivisionMode(g, 1, mode)
            self._keep.append(ti)
        a = np.ascontiguousarray(values, dtype=np.float32)
        pad = np.zeros((a.shape[0] + 2, a.shape[1]), dtype=np.float32)
        pad[: a.shape[0]] = a
        g = L.rtcGetGeometry(self.handle, geom_id)
        L.rtcSetGeometryVertexAttributeCount(g, slot + 1)
        fmt = {1: RTC_FORMAT_FLOAT, 2: RTC_FORMAT_FLOAT + 1, 3: RTC_FORMAT_FLOAT3, 4: RTC_FORMAT_FLOAT3 + 1}[a.shape[1]]
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot, fmt, pad.ctypes.data, 0, 4 * a.shape[1], a.shape[0])
        if topology_index is not None:
            L.rtcSetGeometryVertexAttributeTopology(g, slot, 1)
        L.rtcCommitGeometry(g)
        self._keep.append(pad)
        self.device.check("set_vertex_attribute")

    def interpolate(self, geom_id, prim_id, u, v, buffer_type=None, slot=0, count=3, derivs=2):
        """rtcInterpolate: returns (P, dPdu, dPdv, ddPdudu, ddPdvdv, ddPdudv) as float32 arrays (None beyond `derivs`)."""
        bt = RTC_BUFFER_TYPE_VERTEX if buffer_type is None else buffer_type
        out = [np.zeros(count, np.float32) for _ in range(6)]
        a = RTCInterpolateArguments()
        a.geometry = self.lib.rtcGetGeometry(self.handle, geom_id)
        a.primID, a.u, a.v, a.bufferType, a.bufferSlot, a.valueCount = prim_id, u, v, bt, slot, count
        fp = C.POINTER(C.c_float)
        a.P = out[0].ctypes.data_as(fp)
        if derivs >= 1:
            a.dPdu, a.dPdv = out[1].ctypes.data_as(fp), out[2].ctypes.data_as(fp)
        if derivs >= 2:
            a.ddPdudu, a.ddPdvdv, a.ddPdudv = out[3].ctypes.data_as(fp), out[4].ctypes.data_as(fp), out[5].ctypes.data_as(fp)
        self.lib.rtcInterpolate(C.byref(a))
        self.device.check("rtcInterpolate")
        return tuple(o if k == 0 or (k <= 2 and derivs >= 1) or derivs >= 2 else None for k, o in enumerate(out))

    def interpolateN(self, geom_id, prim_ids, u, v, buffer_type=None, slot=0, count=3, valid=None):
        """rtcInterpolateN: returns P [count, N] and dPdu, dPdv (SoA like the reference)."""
        bt = RTC_BUFFER_TYPE_VERTEX if buffer_type is None else buffer_type
        pid = np.ascontiguousarray(prim_ids, np.uint32)
        uu, vv = np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32)
        n = pid.shape[0]
        P, du, dv = (np.zeros((count, n), np.float32) for _ in range(3))
        a = RTCInterpolateNArguments()
        a.geometry = self.lib.rtcGetGeometry(self.handle, geom_id)
        fp = C.POINTER(C.c_float)
        a.primIDs, a.u, a.v, a.N = pid.ctypes.data_as(C.POINTER(C.c_uint)), uu.ctypes.data_as(fp), vv.ctypes.data_as(fp), n
        if valid is not None:
            vm = np.ascontiguousarray(valid, np.int32)
            a.valid = vm.ctypes.data
        a.bufferType, a.bufferSlot, a.valueCount = bt, slot, count
        a.P, a.dPdu, a.dPdv = P.ctypes.data_as(fp), du.ctypes.data_as(fp), dv.ctypes.data_as(fp)
        self.lib.rtcInterpolateN(C.byref(a))
        self.device.check("rtcInterpolateN")
        return P, du, dv

    def set_levels(self, subdivision_level, compression_level):
        self.lib.rtcSetSceneLevels(self.handle, subdivision_level, compression_level)

    def commit(self):
        self.lib.rtcCommitScene(self.handle)
        self.device.check("rtcCommitScene")

    def bounds(self):
        b = RTCBounds()
        self.lib.rtcGetSceneBounds(self.handle, C.byref(b))
        self.device.check("rtcGetSceneBounds")
        return (np.array([b.lower_x, b.lower_y, b.lower_z], np.float32), np.array([b.upper_x, b.upper_y, b.upper_z], np.float32))

    # ---- the hot path -------------------------------------------------------------------------------
    def intersect1M(self, rayhits, ctx=None, check=True):
        """rayhits: numpy RAYHIT_DTYPE array (host) or torch CUDA tensor [M, 80] uint8 / [M,20] (device)."""
        ptr, m, stride = _ptr_and_count(rayhits)
        ctx = ctx or make_context()
        self.lib.rtcIntersect1M(self.handle, C.byref(ctx), ptr, m, stride)
        if check:
            self.device.check("rtcIntersect1M")

    def occluded1M(self, rays, ctx=None, check=True):
        ptr, m, stride = _ptr_and_count(rays)
        ctx = ctx or make_context()
        self.lib.rtcOccluded1M(self.handle, C.byref(ctx), ptr, m, stride)
        if check:
            self.device.check("rtcOccluded1M")

    def intersect1(self, rayhit_record, ctx=None):
        """rayhit_record: 1-element RAYHIT_DTYPE array, 16-byte aligned."""
        ctx = ctx or make_context()
        self.lib.rtcIntersect1(self.handle, C.byref(ctx), rayhit_record.ctypes.data)
        self.device.check("rtcIntersect1")

    def occluded1(self, ray_record, ctx=None):
        ctx = ctx or make_context()
        self.lib.rtcOccluded1(self.handle, C.byref(ctx), ray_record.ctypes.data)
        self.device.check("rtcOccluded1")

    def intersect1M_counted(self, rayhits, ctx=None):
        ptr, m, stride = _ptr_and_count(rayhits)
        ctx = ctx or make_context()
        cnt = RTCAMDTraceCounters()
        self.lib.rtcamdIntersect1MCounted(self.handle, C.byref(ctx), ptr, m, stride, C.byref(cnt))
        self.device.check("rtcamdIntersect1MCounted")
        return {n: (list(getattr(cnt, n)) if n.endswith("Hist") else getattr(cnt, n)) for n, _ in RTCAMDTraceCounters._fields_}

    def occluded1M_counted(self, rays, ctx=None):
        ptr, m, stride = _ptr_and_count(rays)
        ctx = ctx or make_context()
        cnt = RTCAMDTraceCounters()
        self.lib.rtcamdOccluded1MCounted(self.handle, C.byref(ctx), ptr, m, stride, C.byref(cnt))
        self.device.check("rtcamdOccluded1MCounted")
        return {n: (list(getattr(cnt, n)) if n.endswith("Hist") else getattr(cnt, n)) for n, _ in RTCAMDTraceCounters._fields_}

    def stats(self):
        st = RTCAMDSceneStats()
        st.byteSize = C.sizeof(RTCAMDSceneStats)
        self.lib.rtcamdGetSceneStats(self.handle, C.byref(st))
        self.device.check("rtcamdGetSceneStats")
        return {n: getattr(st, n) for n, _ in RTCAMDSceneStats._fields_}

    def accel_data(self, kind):
        n = C.c_size_t(0)
        p = self.lib.rtcamdGetAccelData(self.handle, kind, C.byref(n))
        self.device.check("rtcamdGetAccelData")
        if not p or n.value == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    def accel_root(self):
        return self.lib.rtcamdGetAccelRoot(self.handle)

    def release(self):
        if self.handle:
            self.lib.rtcReleaseScene(self.handle)
            self.handle = None


def aligned_rayhits(m):
    """RAYHIT_DTYPE array of m records whose base address is 16-byte aligned (rtcIntersect1 contract)."""
    raw = np.zeros(m * 80 + 16, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off: off + m * 80].view(RAYHIT_DTYPE)


def aligned_rays(m):
    raw = np.zeros(m * 48 + 16, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off: off + m * 48].view(RAY_DTYPE)


def fan_triangulate(face_sizes, face_index):
    """Triangle fan per face: (0,1,2),(0,2,3),... like obj_loader.cpp:565-578."""
    tris = []
    p = 0
    for n in face_sizes:
        n = int(n)
        for k in range(2, n):
            tris.append((face_index[p], face_index[p + k - 1], face_index[p + k]))
        p += n
    return np.asarray(tris, dtype=np.uint32)
