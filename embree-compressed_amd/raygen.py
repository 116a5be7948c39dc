"""Synthetic ray batches of the BASELINE workload, generated without libc and without the oracle.

`make_random_rays` reproduces the reference viewer's random-ray benchmark generator (makeRandomRay,
tutorials/viewer/viewer_device.cpp:367-392): two points uniform in the scene's bounding box from a
drand48-compatible 48-bit LCG, org = p1, dir = normalize(p2 - p1), tnear = 0, tfar = inf.  The LCG is
evaluated with log-doubling jump-ahead in numpy so that multi-million-ray batches take milliseconds.
"""
import numpy as np

_A = np.uint64(0x5DEECE66D)
_C = np.uint64(0xB)
_MASK = np.uint64((1 << 48) - 1)

RAYHIT_BYTES = 80


def lcg48_sequence(n, seed):
    """x_1..x_n of drand48 after srand48(seed) as uint64 (48-bit states)."""
    x0 = np.uint64(((seed & 0xFFFFFFFF) << 16) | 0x330E)
    out = np.empty(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        out[0] = (_A * x0 + _C) & _MASK
        # (a_k, c_k): x_{i+k} = a_k x_i + c_k ; doubling: a_2k = a_k^2, c_2k = a_k c_k + c_k  (mod 2^48 via uint64 wrap)
        a, c, k = _A, _C, 1
        while k < n:
            m = min(k, n - k)
            out[k:k + m] = (a * out[:m] + c) & _MASK
            c = (a * c + c) & _MASK
            a = (a * a) & _MASK
            k *= 2
    return out


def make_random_rays(m, lo, hi, seed=0):
    """Returns a uint8 array [m, 80] of RTCRayHit records (BASELINE.md section 3, fp32 evaluation)."""
    lo = np.asarray(lo, dtype=np.float32)
    hi = np.asarray(hi, dtype=np.float32)
    u = (lcg48_sequence(6 * m, seed).astype(np.float64) * (1.0 / 281474976710656.0)).astype(np.float32).reshape(m, 2, 3)
    diam = (hi - lo).astype(np.float32)
    p = (u * diam[None, None, :]).astype(np.float32) + lo[None, None, :]
    d = (p[:, 1, :] - p[:, 0, :]).astype(np.float32)
    sq = (d * d).astype(np.float32)
    length = np.sqrt(((sq[:, 0] + sq[:, 1]).astype(np.float32) + sq[:, 2]).astype(np.float32)).astype(np.float32)
    d = (d / length[:, None]).astype(np.float32)
    rec = np.zeros((m, 20), dtype=np.float32)
    rec[:, 0:3] = p[:, 0, :]
    rec[:, 4:7] = d
    rec[:, 8] = np.inf
    w = rec.view(np.uint32)
    w[:, 9] = 0xFFFFFFFF
    w[:, 10] = np.arange(m, dtype=np.uint32)
    w[:, 17:20] = 0xFFFFFFFF
    return rec.view(np.uint8).reshape(m, RAYHIT_BYTES)


def make_primary_rays(width=1920, height=1080, frm=(18.21240425, 20.05745888, 15.46878433), to=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                      fov=90.0, tile=8):
    """Coherent primary rays of BASELINE config 4 (viewer_stream on build/bomberman.ecs): pinhole camera of
    Camera::getISPCCamera (tutorials/common/tutorial/camera.h:74-88, right-handed lookat), one ray per pixel,
    dir = normalize(x*vx + y*vy + vz), emitted tile by tile (8x8, row-major tiles, row-major inside a tile) like
    renderTileStandard (tutorials/viewer_stream/viewer_stream_device.cpp:288-325).  Returns uint8 [W*H, 80]."""
    f32 = np.float32
    frm, to, up = (np.asarray(a, dtype=f32) for a in (frm, to, up))

    def norm(a):
        return (a / np.sqrt(np.dot(a, a), dtype=f32)).astype(f32)

    Z = norm(to - frm)
    U = norm(np.cross(up, Z).astype(f32))
    V = norm(np.cross(Z, U).astype(f32))
    U = -U  # RIGHT_HANDED: local2world.l.vx = -vx
    fov_scale = f32(1.0) / np.tan(np.deg2rad(f32(0.5) * f32(fov)), dtype=f32)
    vx, vy = U, -V
    vz = (f32(-0.5 * width) * U + f32(0.5 * height) * V + f32(0.5 * height) * fov_scale * Z).astype(f32)
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    # tile order
    ty, tx = ys // tile, xs // tile
    order = np.lexsort((xs.ravel(), ys.ravel(), tx.ravel(), ty.ravel()))
    x = xs.ravel()[order].astype(f32)
    y = ys.ravel()[order].astype(f32)
    d = (x[:, None] * vx[None, :] + y[:, None] * vy[None, :] + vz[None, :]).astype(f32)
    d = (d / np.sqrt((d * d).sum(1, dtype=f32), dtype=f32)[:, None]).astype(f32)
    m = width * height
    rec = np.zeros((m, 20), dtype=f32)
    rec[:, 0:3] = frm[None, :]
    rec[:, 4:7] = d
    rec[:, 8] = np.inf
    w = rec.view(np.uint32)
    w[:, 9] = 0xFFFFFFFF
    w[:, 10] = np.arange(m, dtype=np.uint32)
    w[:, 17:20] = 0xFFFFFFFF
    return rec.view(np.uint8).reshape(m, RAYHIT_BYTES)


def make_secondary_rays(traced, seed=11, light=(50.0, 400.0, -120.0), tnear=0.001):
    """Incoherent rays of BASELINE config 5 (pathtracer), recorded wavefront-style from a traced batch (uint8 [M,80] records
    or the matching structured array): for every hit, one bounce ray from the hit point in a cosine-like lobe around the
    shading-side normal (tnear = 0.001, tfar = inf; pathtracer_device.cpp:1442-1535 traces such rays with rtcIntersect1)
    and one shadow ray towards a point light (tfar = distance; rtcOccluded1).  Returns (bounce uint8 [n,80], shadow uint8
    [n,48]); n = number of hits."""
    f = np.ascontiguousarray(traced).view(np.uint8).reshape(-1, RAYHIT_BYTES).view(np.float32)
    w = f.view(np.uint32)
    hit = w[:, 18] != 0xFFFFFFFF
    f, n = f[hit], int(hit.sum())
    o = (f[:, 0:3] + f[:, 8:9] * f[:, 4:7]).astype(np.float32)
    ng = f[:, 12:15].astype(np.float64)
    ng /= np.maximum(np.linalg.norm(ng, axis=1, keepdims=True), 1e-30)
    ng[(ng * f[:, 4:7]).sum(1) > 0] *= -1  # face the incoming ray
    rng = np.random.RandomState(seed)
    r = rng.normal(size=(n, 3))
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    d = ng + 0.999 * r
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rec = np.zeros((n, 20), dtype=np.float32)
    rec[:, 0:3], rec[:, 3], rec[:, 4:7], rec[:, 8] = o, tnear, d.astype(np.float32), np.inf
    rw = rec.view(np.uint32)
    rw[:, 9] = 0xFFFFFFFF
    rw[:, 10] = np.arange(n, dtype=np.uint32)
    rw[:, 17:20] = 0xFFFFFFFF
    ld = np.asarray(light, np.float64)[None, :] - o
    dist = np.linalg.norm(ld, axis=1)
    sh = np.zeros((n, 12), dtype=np.float32)
    sh[:, 0:3], sh[:, 3], sh[:, 4:7], sh[:, 8] = o, tnear, (ld / dist[:, None]).astype(np.float32), dist.astype(np.float32)
    sw = sh.view(np.uint32)
    sw[:, 9] = 0xFFFFFFFF
    sw[:, 10] = np.arange(n, dtype=np.uint32)
    return rec.view(np.uint8).reshape(n, RAYHIT_BYTES), sh.view(np.uint8).reshape(n, 48)


def shard_range(total, rank, world):
    """Contiguous ray-index range of `rank`: [rank*total/world, (rank+1)*total/world)."""
    return (rank * total) // world, ((rank + 1) * total) // world
