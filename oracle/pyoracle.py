"""ctypes binding of oracle/libmiro_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see oracle/miro_oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmiro_oracle.so")

RAY_DTYPE = np.dtype([("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("tmin", "<f4"),
                      ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<u4"), ("beta", "<f4"), ("gamma", "<f4")])
MISS = 0xFFFFFFFF


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, f32p, u32p, u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.orc_scene_new.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_add_obj.argtypes = [vp, C.c_char_p, f32p]
        L.orc_scene_add_triangle.argtypes = [vp, f32p, f32p]
        L.orc_scene_add_arrays.argtypes = [vp, C.c_int, f32p, C.c_int, f32p, C.c_int, u32p, u32p]
        L.orc_scene_counts.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for name, rt in (("orc_scene_vertices", f32p), ("orc_scene_normals", f32p),
                         ("orc_scene_vidx", u32p), ("orc_scene_nidx", u32p)):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = rt
        L.orc_scene_build.argtypes = [vp, C.c_int]
        L.orc_scene_tree_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_scene_export_tree.argtypes = [vp, f32p, C.POINTER(C.c_int32), u32p]
        L.orc_trace.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(Counters)]
        L.orc_trace_brute.argtypes = [vp, vp, C.c_uint64, vp]
        L.orc_trace_sse.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, C.POINTER(Counters)]
        L.orc_trace_sse.restype = C.c_int
        L.orc_eye_rays.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_uint32, vp]
        L.orc_shadow_rays.argtypes = [vp, vp, vp, C.c_uint64, f32p, vp, u64p, C.c_int]
        L.orc_shadow_rays.restype = C.c_uint64
        L.orc_hit_attrs.argtypes = [vp, vp, C.c_uint64, f32p, f32p]
        L.orc_hit_attrs_rays.argtypes = [vp, vp, vp, C.c_uint64, f32p, f32p]
        L.orc_scene_add_sphere.argtypes = [vp, f32p, C.c_float]
        L.orc_scene_add_plane.argtypes = [vp, f32p, f32p, C.c_uint32]
        L.orc_shade_direct.argtypes = [vp, vp, vp, C.c_uint64, vp, f32p, f32p, C.c_float, f32p, C.c_int, f32p]
        L.orc_tonemap.argtypes = [f32p, C.c_uint64, vp]
        L.orc_trace_scene.argtypes = [vp, f32p, u32p, vp, C.c_uint64, f32p, f32p, C.c_float, C.c_int, f32p]
        L.orc_trace_scene.restype = C.c_uint64
        L.orc_path_rays.argtypes = [vp, f32p, u32p, vp, vp, f32p, u32p, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, vp, f32p, u32p, u32p, u32p]
        L.orc_path_rays.restype = C.c_uint64
        L.orc_miro_math.argtypes = [f32p, f32p, C.c_uint64, f32p]
        L.orc_trace_scene_pt.argtypes = [vp, f32p, u32p, vp, C.c_uint64, f32p, f32p, C.c_float, C.c_int, C.c_uint32, C.c_uint32, f32p]
        L.orc_trace_scene_pt.restype = C.c_uint64
        L.orc_pmap_new.argtypes = [C.c_int]
        L.orc_pmap_new.restype = vp
        L.orc_pmap_free.argtypes = [vp]
        L.orc_pmap_count.argtypes = [vp]
        L.orc_pmap_store.argtypes = [vp, C.c_int, f32p, f32p, f32p]
        L.orc_pmap_scale.argtypes = [vp, C.c_float]
        L.orc_pmap_balance.argtypes = [vp]
        L.orc_pmap_irradiance.argtypes = [vp, C.c_uint64, f32p, f32p, C.c_float, C.c_int, f32p, C.POINTER(C.c_int), f32p]
        L.orc_pmap_irradiance_brute.argtypes = [vp, C.c_uint64, f32p, f32p, C.c_float, C.c_int, f32p, C.POINTER(C.c_int), f32p]
        L.orc_pmap_export.argtypes = [vp, f32p, C.POINTER(C.c_int), C.POINTER(C.c_ubyte), f32p]
        L.orc_hash.argtypes = [C.c_uint32]
        L.orc_hash.restype = C.c_uint32
        _lib = L
    return _lib


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def make_camera(eye, lookat, up, fov_deg):
    cam = Camera()
    cam.eye[:] = eye
    cam.lookat[:] = lookat
    cam.up[:] = up
    cam.fov_deg = fov_deg
    return cam


class Scene:
    """Reference-shaped scene: meshes appended in addObject order, then BVH::build."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.orc_scene_new()
        self.leaf_size = None

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def add_obj(self, path, ctm=None):
        m = None
        if ctm is not None:
            m = np.ascontiguousarray(ctm, dtype=np.float32).reshape(16)
        n = self.L.orc_scene_add_obj(self.h, os.fsencode(path), _f32p(m) if m is not None else None)
        if n < 0:
            raise FileNotFoundError(path)
        return n

    def add_triangle(self, verts, normals):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(9)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(9)
        return self.L.orc_scene_add_triangle(self.h, _f32p(v), _f32p(n))

    def add_arrays(self, v, n, vi, ni):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(n, dtype=np.float32).reshape(-1, 3)
        vi = np.ascontiguousarray(vi, dtype=np.uint32).reshape(-1, 3)
        ni = np.ascontiguousarray(ni, dtype=np.uint32).reshape(-1, 3)
        return self.L.orc_scene_add_arrays(self.h, len(v), _f32p(v), len(n), _f32p(n), len(vi), _u32p(vi), _u32p(ni))

    def add_sphere(self, center, radius):
        """Sphere as the next bounded object (Scene::addObject); returns its prim index."""
        c = np.ascontiguousarray(center, dtype=np.float32).reshape(3)
        return self.L.orc_scene_add_sphere(self.h, _f32p(c), float(radius))

    def add_plane(self, normal, origin, material=0):
        """Plane as the next unbounded object; hits carry prim = 0x80000000 | index."""
        n = np.ascontiguousarray(normal, dtype=np.float32).reshape(3)
        o = np.ascontiguousarray(origin, dtype=np.float32).reshape(3)
        return self.L.orc_scene_add_plane(self.h, _f32p(n), _f32p(o), int(material))

    def counts(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_scene_counts(self.h, a, b, c)
        return a.value, b.value, c.value

    def arrays(self):
        nv, nn, nt = self.counts()
        v = np.ctypeslib.as_array(self.L.orc_scene_vertices(self.h), shape=(nv, 3)).copy()
        n = np.ctypeslib.as_array(self.L.orc_scene_normals(self.h), shape=(nn, 3)).copy()
        vi = np.ctypeslib.as_array(self.L.orc_scene_vidx(self.h), shape=(nt, 3)).copy()
        ni = np.ctypeslib.as_array(self.L.orc_scene_nidx(self.h), shape=(nt, 3)).copy()
        return v, n, vi, ni

    def build(self, leaf_size=4):
        self.leaf_size = leaf_size
        return self.L.orc_scene_build(self.h, leaf_size)

    def tree_stats(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_scene_tree_stats(self.h, a, b, c)
        return a.value, b.value, c.value

    def export_tree(self):
        nodes, _, _ = self.tree_stats()
        _, _, nt = self.counts()
        corners = np.zeros((nodes, 6), np.float32)
        meta = np.zeros((nodes, 3), np.int32)
        prims = np.zeros(nt, np.uint32)
        self.L.orc_scene_export_tree(self.h, _f32p(corners), meta.ctypes.data_as(C.POINTER(C.c_int32)), _u32p(prims))
        return corners, meta, prims

    def trace(self, rays, counters=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.empty(len(rays), HIT_DTYPE)
        ctr = Counters(0, 0)
        self.L.orc_trace(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, C.byref(ctr))
        if counters:
            return hits, (ctr.box_tests, ctr.tri_tests)
        return hits

    def trace_brute(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.empty(len(rays), HIT_DTYPE)
        self.L.orc_trace_brute(self.h, rays.ctypes.data, len(rays), hits.ctypes.data)
        return hits

    def trace_sse(self, rays, threads=0, counters=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.empty(len(rays), HIT_DTYPE)
        ctr = Counters(0, 0)
        used = self.L.orc_trace_sse(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, threads, C.byref(ctr))
        if used < 0:
            raise RuntimeError("orc_trace_sse needs a scene built with leaf_size=8")
        if counters:
            return hits, used, (ctr.box_tests, ctr.tri_tests)
        return hits, used

    def shadow_rays(self, rays, hits, light, sse_order=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = np.empty(len(rays), RAY_DTYPE)
        src = np.empty(len(rays), np.uint64)
        l = np.ascontiguousarray(light, dtype=np.float32)
        k = self.L.orc_shadow_rays(self.h, rays.ctypes.data, hits.ctypes.data, len(rays), _f32p(l),
                                   out.ctypes.data, src.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if sse_order else 0)
        return out[:k].copy(), src[:k].copy()

    def shade_direct(self, rays, hits, occluded, light, wattage, spp=1, color=(1, 1, 1), diffuse=(1, 1, 1)):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        occ = np.ascontiguousarray(occluded, dtype=np.uint8)
        rgb = np.empty((len(rays) // spp, 3), np.float32)
        l, c, d = (np.ascontiguousarray(x, dtype=np.float32) for x in (light, color, diffuse))
        self.L.orc_shade_direct(self.h, rays.ctypes.data, hits.ctypes.data, len(rays), occ.ctypes.data,
                                _f32p(l), _f32p(c), wattage, _f32p(d), spp, _f32p(rgb))
        return rgb

    def trace_scene(self, materials, prim_mat, rays, light, wattage, depth=10, color=(1, 1, 1)):
        """Scene::traceScene per ray; materials [n,11] already clamped; returns (rgb [n,3], Scene::trace calls)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        m = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 11)
        pm = np.ascontiguousarray(prim_mat, dtype=np.uint32)
        rgb = np.empty((len(rays), 3), np.float32)
        l, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (light, color))
        calls = self.L.orc_trace_scene(self.h, _f32p(m), _u32p(pm), rays.ctypes.data, len(rays), _f32p(l), _f32p(c),
                                       wattage, depth, _f32p(rgb))
        return rgb, calls

    def trace_scene_pt(self, materials, prim_mat, rays, light, wattage, depth=10, seed=168, kinds=3, color=(1, 1, 1)):
        """Scene::traceScene as the PATH_TRACING build runs it (lobe-sampled reflect / refract; kinds & 4: + diffuse bounce)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        m = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 11)
        pm = np.ascontiguousarray(prim_mat, dtype=np.uint32)
        rgb = np.empty((len(rays), 3), np.float32)
        l, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (light, color))
        calls = self.L.orc_trace_scene_pt(self.h, _f32p(m), _u32p(pm), rays.ctypes.data, len(rays), _f32p(l), _f32p(c),
                                          wattage, depth, seed, kinds, _f32p(rgb))
        return rgb, calls

    def path_rays(self, materials, prim_mat, rays, hits, weights=None, pixels=None, ids=None, spp=1, seed=168, bounce=0, kinds=7):
        """The PATH_TRACING generators (miro_oracle_path.c): children of every hit in ray order.
        Returns (rays, weights [m,3], pixels, ids, kinds)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        m = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 11)
        pm = np.ascontiguousarray(prim_mat, dtype=np.uint32) if prim_mat is not None else None
        n = len(rays)
        w = np.ascontiguousarray(weights, dtype=np.float32) if weights is not None else None
        px = np.ascontiguousarray(pixels, dtype=np.uint32) if pixels is not None else None
        idv = np.ascontiguousarray(ids, dtype=np.uint32) if ids is not None else None
        out = np.empty(4 * n, RAY_DTYPE)
        ow = np.empty((4 * n, 3), np.float32)
        op, oi, ok = (np.empty(4 * n, np.uint32) for _ in range(3))
        cnt = self.L.orc_path_rays(self.h, _f32p(m), _u32p(pm) if pm is not None else None, rays.ctypes.data, hits.ctypes.data,
                                   _f32p(w) if w is not None else None, _u32p(px) if px is not None else None,
                                   _u32p(idv) if idv is not None else None, n, spp, seed, bounce, kinds, out.ctypes.data,
                                   _f32p(ow), _u32p(op), _u32p(oi), _u32p(ok))
        return out[:cnt].copy(), ow[:cnt].copy(), op[:cnt].copy(), oi[:cnt].copy(), ok[:cnt].copy()

    def hit_attrs(self, hits, rays=None):
        """HitInfo::P / ::N; `rays` is needed when the scene holds spheres or planes (P = o + t*d)."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        P = np.empty((len(hits), 3), np.float32)
        N = np.empty((len(hits), 3), np.float32)
        if rays is None:
            self.L.orc_hit_attrs(self.h, hits.ctypes.data, len(hits), _f32p(P), _f32p(N))
        else:
            rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
            self.L.orc_hit_attrs_rays(self.h, rays.ctypes.data, hits.ctypes.data, len(hits), _f32p(P), _f32p(N))
        return P, N


def miro_math(x, y):
    """include/miro_math.h on the host: columns sin, cos, asin01, acos01, pow01(x, y)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty((len(x), 5), np.float32)
    lib().orc_miro_math(_f32p(x), _f32p(y), len(x), _f32p(out))
    return out


class PhotonMap:
    """Photon_map (PhotonMap.h:42-105): store, scale_photon_power, balance, irradiance_estimate."""

    def __init__(self, max_photons):
        self.L = lib()
        self.h = self.L.orc_pmap_new(max_photons)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_pmap_free(self.h)
                self.h = None
        except Exception:
            pass

    def store(self, power, pos, direction):
        power, pos, direction = (np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (power, pos, direction))
        self.L.orc_pmap_store(self.h, len(pos), _f32p(power), _f32p(pos), _f32p(direction))

    def scale_photon_power(self, scale):
        self.L.orc_pmap_scale(self.h, scale)

    def balance(self):
        self.L.orc_pmap_balance(self.h)

    def count(self):
        return self.L.orc_pmap_count(self.h)

    def irradiance_estimate(self, pos, normal, max_dist=1e10, nphotons=500, brute=False):
        pos, normal = (np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (pos, normal))
        n = len(pos)
        irr = np.empty((n, 3), np.float32)
        found = np.empty(n, np.int32)
        r2 = np.empty(n, np.float32)
        fn = self.L.orc_pmap_irradiance_brute if brute else self.L.orc_pmap_irradiance
        fn(self.h, n, _f32p(pos), _f32p(normal), max_dist, nphotons, _f32p(irr), found.ctypes.data_as(C.POINTER(C.c_int)), _f32p(r2))
        return irr, found, r2

    def export(self):
        n = self.count()
        pos, power = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        plane = np.empty(n, np.int32)
        tp = np.empty((n, 2), np.uint8)
        self.L.orc_pmap_export(self.h, _f32p(pos), plane.ctypes.data_as(C.POINTER(C.c_int)), tp.ctypes.data_as(C.POINTER(C.c_ubyte)), _f32p(power))
        return pos, plane, tp, power


def tonemap(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(rgb.shape, np.uint8)
    lib().orc_tonemap(_f32p(rgb), rgb.size, out.ctypes.data)
    return out


def eye_rays(cam, W, H, spp=1, jitter=False, seed=168, y0=0, y1=None):
    if y1 is None:
        y1 = H
    rays = np.empty((y1 - y0) * W * spp, RAY_DTYPE)
    lib().orc_eye_rays(C.byref(cam), W, H, y0, y1, spp, 1 if jitter else 0, seed, rays.ctypes.data)
    return rays
