"""ctypes binding of libmiro_hip.so.  Names follow the reference: Scene.addObject-style assembly,
Scene.preCalc() -> BVH::build, Scene.trace() (batched)."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_PKG, "lib", "libmiro_hip.so")

RAY_DTYPE = np.dtype([("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("tmin", "<f4"),
                      ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<u4"), ("beta", "<f4"), ("gamma", "<f4")])
MISS = 0xFFFFFFFF

MR_TRACE_CLOSEST = 0
MR_PLANE_BIT = 0x80000000
MR_TRACE_ANY = 1 << 0
MR_RAYS_ON_DEVICE = 1 << 1
MR_HITS_ON_DEVICE = 1 << 2
MR_MATH_FAST = 1 << 3
MR_COUNT_STATS = 1 << 4
MR_TRACE_PERSISTENT = 1 << 5
MR_MATH_PRODUCT = 1 << 6
MR_TRACE_INCOHERENT = 1 << 7
MR_FRAME_NO_SHADOWS = 1 << 8
MR_ORDER_GIVEN = 0x80000000

MR_PATH_MIRROR, MR_PATH_REFRACT, MR_PATH_DIFFUSE = 1, 2, 4
MR_LEVEL_LAST, MR_LEVEL_SPECULAR, MR_LEVEL_PATH = 0, 1, 2
MR_LAYOUT_DFS, MR_LAYOUT_PAIRS, MR_LAYOUT_TREELETS, MR_LAYOUT_ALIGN_LEAVES = 0, 1, 2, 16

MR_OK, MR_ERR_INVALID, MR_ERR_IO, MR_ERR_NOMEM, MR_ERR_HIP, MR_ERR_STATE = 0, -1, -2, -3, -4, -5

# every symbol include/miro_hip.h declares (tests check the library exports each one)
EXPORTED_SYMBOLS = [
    "mr_scene_create", "mr_scene_destroy", "mr_scene_add_mesh", "mr_scene_add_obj", "mr_scene_add_triangle",
    "mr_scene_add_sphere", "mr_scene_add_plane",
    "mr_bvh_build", "mr_scene_get_info", "mr_scene_get_mesh", "mr_scene_export_tree",
    "mr_trace", "mr_host_alloc", "mr_host_free", "mr_trace_indirect", "mr_trace_grouped", "mr_order_by_octant", "mr_trace_get_stats", "mr_gen_eye_rays", "mr_gen_eye_rays_tiled", "mr_tile_pixel_map", "mr_untile_pixels", "mr_gen_shadow_rays", "mr_hit_attrs",
    "mr_shade_direct", "mr_render_direct", "mr_band_locate", "mr_band_rows_of", "mr_deinterleave_bands", "mr_gen_path_rays", "mr_trace_level", "mr_tonemap",
    "mr_scene_set_materials", "mr_shade_accumulate", "mr_gen_secondary_rays",
    "mr_photon_map_create", "mr_photon_map_destroy", "mr_photon_map_store", "mr_photon_map_scale",
    "mr_photon_map_balance", "mr_photon_map_count", "mr_photon_map_export", "mr_irradiance_estimate",
    "mr_photon_map_count_stats", "mr_photon_map_get_stats",
    "mr_final_gather",
    "mr_last_error", "mr_version",
]


class MiroError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("miro_hip status %d: %s" % (status, msg))
        self.status = status


class MeshDesc(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_vertices", C.c_uint32),
                ("normals", C.POINTER(C.c_float)), ("n_normals", C.c_uint32),
                ("vidx", C.POINTER(C.c_uint32)), ("nidx", C.POINTER(C.c_uint32)),
                ("n_triangles", C.c_uint32)]


class BuildOpts(C.Structure):
    _fields_ = [("leaf_size", C.c_uint32), ("builder", C.c_uint32), ("host_only", C.c_uint32),
                ("layout", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class SceneInfo(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_normals", C.c_uint32), ("n_triangles", C.c_uint32),
                ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("max_depth", C.c_uint32),
                ("leaf_size", C.c_uint32), ("built", C.c_uint32), ("device_bytes", C.c_uint64),
                ("device", C.c_int32), ("reserved", C.c_uint32 * 3)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3), ("wattage", C.c_float)]


class FrameDesc(C.Structure):
    """mr_frame_desc (miro_hip.h): one window of a direct-light frame for mr_render_direct"""
    _fields_ = [("camera", Camera), ("W", C.c_uint32), ("H", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32),
                ("band_rows", C.c_uint32), ("band_rank", C.c_uint32), ("band_world", C.c_uint32),
                ("spp", C.c_uint32), ("jitter", C.c_uint32), ("seed", C.c_uint32), ("tiled", C.c_uint32),
                ("flags", C.c_uint32), ("light", Light), ("diffuse", C.c_float * 3), ("reserved", C.c_uint32 * 4)]


class LevelDesc(C.Structure):
    """mr_level_desc (miro_hip.h): one level of traceScene's recursion for mr_trace_level"""
    _fields_ = [("light", Light), ("spp", C.c_uint32), ("flags", C.c_uint32), ("children", C.c_uint32),
                ("path_kinds", C.c_uint32), ("seed", C.c_uint32), ("bounce", C.c_uint32),
                ("out_capacity_lo", C.c_uint32), ("out_capacity_hi", C.c_uint32), ("reserved", C.c_uint32),
                ("d_out_octants", C.c_void_p), ("d_order", C.c_void_p)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("transmission", C.c_float * 3),
                ("shininess", C.c_float), ("refract_index", C.c_float)]


def lib_path():
    return _LIB


_lib = None


def build_library():
    """Compile libmiro_hip.so in-tree with hipcc (cse168-raytracer_amd/Makefile)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", _PKG, "-j4"])
    return _LIB


def load_library(path=None):
    """Load libmiro_hip.so, compiling it first if it is not there; raises if that fails (there is no fallback:
    without the HIP library nothing in this package computes anything)."""
    global _lib
    path = path or os.environ.get("MIRO_LIB") or _LIB       # MIRO_LIB: an A/B build of the library (Makefile VARIANT=...)
    if not os.path.exists(path) and path == _LIB:
        build_library()
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build it with `make -C cse168-raytracer_amd` or __graft_entry__.build()" % path)
    try:  # share torch's HIP runtime (same soname) when torch is present
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(path)
    vp, f32p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.mr_last_error.restype = C.c_char_p
    L.mr_version.restype = C.c_char_p
    L.mr_scene_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.mr_scene_destroy.argtypes = [vp]
    L.mr_scene_add_mesh.argtypes = [vp, C.POINTER(MeshDesc)]
    L.mr_scene_add_obj.argtypes = [vp, C.c_char_p, f32p, u32p]
    L.mr_scene_add_triangle.argtypes = [vp, f32p, f32p]
    L.mr_bvh_build.argtypes = [vp, C.POINTER(BuildOpts)]
    L.mr_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    L.mr_scene_get_mesh.argtypes = [vp, C.POINTER(MeshDesc)]
    L.mr_scene_export_tree.argtypes = [vp, f32p, C.POINTER(C.c_int32), u32p]
    L.mr_trace.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    L.mr_trace_indirect.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    L.mr_trace_grouped.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.mr_order_by_octant.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp]
    L.mr_host_alloc.argtypes = [C.POINTER(vp), C.c_uint64]
    L.mr_host_free.argtypes = [vp]
    L.mr_trace_get_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
    L.mr_gen_eye_rays.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.mr_gen_eye_rays_tiled.argtypes = L.mr_gen_eye_rays.argtypes
    L.mr_tile_pixel_map.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p]
    L.mr_untile_pixels.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.mr_gen_shadow_rays.argtypes = [vp, vp, vp, C.c_uint64, f32p, vp, vp, vp, vp]
    L.mr_hit_attrs.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, vp]
    L.mr_scene_add_sphere.argtypes = [vp, f32p, C.c_float, u32p]
    L.mr_scene_add_plane.argtypes = [vp, f32p, f32p, C.c_uint32, u32p]
    L.mr_shade_direct.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, vp, C.POINTER(Light), f32p, C.c_uint32, vp, vp]
    L.mr_band_locate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    L.mr_band_rows_of.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
    L.mr_deinterleave_bands.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.mr_render_direct.argtypes = [vp, C.POINTER(FrameDesc), vp, vp, vp, vp, vp]
    L.mr_tonemap.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.mr_scene_set_materials.argtypes = [vp, C.POINTER(Material), C.c_uint32, u32p]
    L.mr_shade_accumulate.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, C.POINTER(Light), C.c_uint32, vp, vp]
    L.mr_gen_secondary_rays.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.mr_gen_path_rays.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.mr_trace_level.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mr_final_gather.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_float, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.mr_photon_map_create.argtypes = [C.c_int32, C.c_uint32, C.POINTER(vp)]
    L.mr_photon_map_destroy.argtypes = [vp]
    L.mr_photon_map_store.argtypes = [vp, C.c_uint32, f32p, f32p, f32p]
    L.mr_photon_map_scale.argtypes = [vp, C.c_float]
    L.mr_photon_map_balance.argtypes = [vp, C.c_uint32]
    L.mr_photon_map_count.argtypes = [vp, u32p]
    L.mr_photon_map_export.argtypes = [vp, f32p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), f32p]
    L.mr_irradiance_estimate.argtypes = [vp, vp, vp, C.c_uint64, C.c_float, C.c_uint32, vp, vp, vp, vp]
    L.mr_photon_map_count_stats.argtypes = [vp, C.c_int32]
    L.mr_photon_map_get_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32]
    for name in EXPORTED_SYMBOLS:
        if hasattr(L, name) and getattr(L, name).restype is C.c_int:
            getattr(L, name).restype = C.c_int32
    _lib = L
    return L


def lib():
    return _lib if _lib is not None else load_library()


def _check(st):
    if st != MR_OK:
        raise MiroError(st, lib().mr_last_error().decode("utf-8", "replace"))


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class PinnedArray:
    """A numpy array in page-locked host memory (mr_host_alloc); keep the object alive while `.array` is in use."""

    def __init__(self, n, dtype):
        self.L = lib()
        dtype = np.dtype(dtype)
        self.ptr = C.c_void_p()
        _check(self.L.mr_host_alloc(C.byref(self.ptr), max(1, n) * dtype.itemsize))
        buf = (C.c_char * (n * dtype.itemsize)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=n)

    def close(self):
        if self.ptr is not None and self.ptr.value:
            self.array = None
            self.L.mr_host_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_locate(H, band_rows, world, y):
    """(rank, row inside that rank's shard) of image row y -- mr_band_locate"""
    r, l = C.c_uint32(), C.c_uint32()
    _check(lib().mr_band_locate(H, band_rows, world, y, C.byref(r), C.byref(l)))
    return r.value, l.value


def band_rows_of(H, band_rows, rank, world):
    n = C.c_uint32()
    _check(lib().mr_band_rows_of(H, band_rows, rank, world, C.byref(n)))
    return n.value


def tile_pixel_map(W, rows, spp):
    """pixel_of_slot[p] = local row-major pixel index of slot p of a tiled window (mr_tile_pixel_map), uint32 numpy."""
    out = np.empty(W * rows, np.uint32)
    _check(lib().mr_tile_pixel_map(W, rows, spp, _u32p(out)))
    return out


def make_camera(eye, lookat, up, fov_deg):
    cam = Camera()
    cam.eye[:] = eye
    cam.lookat[:] = lookat
    cam.up[:] = up
    cam.fov_deg = fov_deg
    return cam


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


class Scene:
    """Scene (Scene.h:14-73) on one device: addObject-style assembly, preCalc() = BVH::build, trace()."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _check(self.L.mr_scene_create(device, C.byref(self.h)))
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.mr_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- assembly (Scene::addObject over TriangleMesh triangles, assignment2.cpp:449-461)
    def add_obj(self, path, ctm=None):
        m = None
        if ctm is not None:
            m = np.ascontiguousarray(ctm, dtype=np.float32).reshape(16)
        n = C.c_uint32(0)
        _check(self.L.mr_scene_add_obj(self.h, os.fsencode(path), _f32p(m) if m is not None else None, C.byref(n)))
        return n.value

    def add_triangle(self, verts, normals):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(9)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(9)
        _check(self.L.mr_scene_add_triangle(self.h, _f32p(v), _f32p(n)))
        return 1

    def add_sphere(self, center, radius):
        """Sphere as the next bounded object (Scene::addObject); returns its prim index."""
        c = np.ascontiguousarray(center, dtype=np.float32).reshape(3)
        prim = C.c_uint32(0)
        _check(self.L.mr_scene_add_sphere(self.h, _f32p(c), float(radius), C.byref(prim)))
        return prim.value

    def add_plane(self, normal, origin, material=0):
        """Plane as the next unbounded object; hits carry prim = MR_PLANE_BIT | index."""
        n = np.ascontiguousarray(normal, dtype=np.float32).reshape(3)
        o = np.ascontiguousarray(origin, dtype=np.float32).reshape(3)
        idx = C.c_uint32(0)
        _check(self.L.mr_scene_add_plane(self.h, _f32p(n), _f32p(o), int(material), C.byref(idx)))
        return idx.value

    def add_arrays(self, v, n, vi, ni):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(n, dtype=np.float32).reshape(-1, 3)
        vi = np.ascontiguousarray(vi, dtype=np.uint32).reshape(-1, 3)
        ni = np.ascontiguousarray(ni, dtype=np.uint32).reshape(-1, 3)
        d = MeshDesc(_f32p(v), len(v), _f32p(n), len(n), _u32p(vi), _u32p(ni), len(vi))
        _check(self.L.mr_scene_add_mesh(self.h, C.byref(d)))
        return len(vi)

    # ---- Scene::preCalc -> BVH::build (Scene.cpp:72)
    def build(self, leaf_size=4, host_only=False, layout=None):
        """layout: MR_LAYOUT_* (storage order of the device records; default from MIRO_LAYOUT for A/B probes, else 0)"""
        o = BuildOpts()
        o.leaf_size = leaf_size
        o.builder = 0
        o.host_only = 1 if host_only else 0
        o.layout = int(os.environ.get("MIRO_LAYOUT", "0")) if layout is None else layout
        _check(self.L.mr_bvh_build(self.h, C.byref(o)))
        return self.info()

    preCalc = build

    def info(self):
        i = SceneInfo()
        _check(self.L.mr_scene_get_info(self.h, C.byref(i)))
        return i

    def arrays(self):
        d = MeshDesc()
        _check(self.L.mr_scene_get_mesh(self.h, C.byref(d)))
        def arr(p, n, dt):
            if n == 0:
                return np.zeros((0, 3), dt)
            return np.ctypeslib.as_array(p, shape=(n, 3)).copy()
        return (arr(d.vertices, d.n_vertices, np.float32), arr(d.normals, d.n_normals, np.float32),
                arr(d.vidx, d.n_triangles, np.uint32), arr(d.nidx, d.n_triangles, np.uint32))

    def export_tree(self):
        i = self.info()
        corners = np.zeros((i.n_nodes, 6), np.float32)
        meta = np.zeros((i.n_nodes, 3), np.int32)
        prims = np.zeros(max(i.n_triangles, 1), np.uint32)
        _check(self.L.mr_scene_export_tree(self.h, _f32p(corners), meta.ctypes.data_as(C.POINTER(C.c_int32)), _u32p(prims)))
        return corners, meta, prims[:i.n_triangles]

    # ---- Scene::trace, batched
    def trace(self, rays, flags=0, hits=None):
        """Host numpy rays (RAY_DTYPE) -> host numpy hits (HIT_DTYPE); `hits` may be a caller's (e.g. pinned) array."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        if hits is None:
            hits = np.empty(len(rays), HIT_DTYPE)
        assert hits.dtype == HIT_DTYPE and len(hits) == len(rays) and hits.flags["C_CONTIGUOUS"]
        flags &= ~(MR_RAYS_ON_DEVICE | MR_HITS_ON_DEVICE)
        _check(self.L.mr_trace(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, flags, None))
        return hits

    def trace_device(self, d_rays, n, d_hits, flags=0, stream=None):
        """Device buffers (torch tensors or raw pointers); only enqueues work."""
        rp = d_rays.data_ptr() if hasattr(d_rays, "data_ptr") else int(d_rays)
        hp = d_hits.data_ptr() if hasattr(d_hits, "data_ptr") else int(d_hits)
        _check(self.L.mr_trace(self.h, rp, n, hp, flags | MR_RAYS_ON_DEVICE | MR_HITS_ON_DEVICE, _stream_ptr(stream)))

    def trace_indirect(self, d_rays, d_count, max_rays, d_hits, flags=0, stream=None):
        """Batch size read on the device from d_count (uint64/int64 tensor written by gen_shadow_rays)."""
        _check(self.L.mr_trace_indirect(self.h, d_rays.data_ptr(), d_count.data_ptr(), max_rays, d_hits.data_ptr(),
                                        flags, _stream_ptr(stream)))

    def trace_grouped(self, d_rays, n, d_hits, d_order, flags=0, chunk_log2=0, stream=None, d_octants=None):
        """mr_trace_grouped: a bounce queue traced with its rays grouped by direction octant inside chunks of 2^chunk_log2
        (d_order: int32 / uint32 tensor of n entries, written by the call unless chunk_log2 holds MR_ORDER_GIVEN); the hit
        buffer is mr_trace's.  d_octants: the generator's octant bytes (uint8 per ray), cheaper to order from than the rays."""
        _check(self.L.mr_trace_grouped(self.h, d_rays.data_ptr(), d_octants.data_ptr() if d_octants is not None else None, n,
                                       d_hits.data_ptr(), d_order.data_ptr(), chunk_log2, flags, _stream_ptr(stream)))

    def order_by_octant(self, d_rays, n, d_order, chunk_log2=0, stream=None, d_octants=None):
        """mr_order_by_octant: the ray order alone (for trace_level(d_order=...) or trace_grouped(chunk_log2 | MR_ORDER_GIVEN))"""
        _check(self.L.mr_order_by_octant(self.h, d_rays.data_ptr() if d_rays is not None else None,
                                         d_octants.data_ptr() if d_octants is not None else None, n, chunk_log2,
                                         d_order.data_ptr(), _stream_ptr(stream)))

    def stats(self, reset=True):
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self.L.mr_trace_get_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    # ---- callers on the device
    def gen_eye_rays(self, cam, W, H, d_rays, y0=0, y1=None, spp=1, jitter=False, seed=168, stream=None, tiled=False):
        """Camera::eyeRay for rows [y0, y1); tiled=True: the tiled order of mr_gen_eye_rays_tiled (see tile_pixel_map)."""
        y1 = H if y1 is None else y1
        fn = self.L.mr_gen_eye_rays_tiled if tiled else self.L.mr_gen_eye_rays
        _check(fn(self.h, C.byref(cam), W, H, y0, y1, spp, 1 if jitter else 0, seed, d_rays.data_ptr(), _stream_ptr(stream)))
        return (y1 - y0) * W * spp

    def untile_pixels(self, d_slots, d_image, W, rows, spp, channels=3, stream=None):
        """Scatter a tiled window's pixel slots to image order on the device (mr_untile_pixels)."""
        _check(self.L.mr_untile_pixels(self.h, d_slots.data_ptr(), d_image.data_ptr(), W, rows, spp, channels,
                                       _stream_ptr(stream)))

    def gen_shadow_rays(self, d_rays, d_hits, n, light, d_out, d_src, d_count, stream=None):
        l = np.ascontiguousarray(light, dtype=np.float32)
        _check(self.L.mr_gen_shadow_rays(self.h, d_rays.data_ptr() if d_rays is not None else None, d_hits.data_ptr(), n,
                                         _f32p(l), d_out.data_ptr(), d_src.data_ptr() if d_src is not None else None,
                                         d_count.data_ptr(), _stream_ptr(stream)))

    def hit_attrs(self, d_hits, n, d_P, d_N, stream=None, d_rays=None):
        _check(self.L.mr_hit_attrs(self.h, d_rays.data_ptr() if d_rays is not None else None, d_hits.data_ptr(), n,
                                   d_P.data_ptr() if d_P is not None else None,
                                   d_N.data_ptr() if d_N is not None else None, _stream_ptr(stream)))

    def shade_direct(self, d_rays, d_hits, n, d_shadow_hits, d_shadow_src, d_shadow_count, light_pos, wattage,
                     d_rgb, spp=1, color=(1.0, 1.0, 1.0), diffuse=(1.0, 1.0, 1.0), stream=None):
        lt = Light()
        lt.position[:] = light_pos
        lt.color[:] = color
        lt.wattage = wattage
        df = np.ascontiguousarray(diffuse, dtype=np.float32)
        _check(self.L.mr_shade_direct(self.h, d_rays.data_ptr(), d_hits.data_ptr(), n, d_shadow_hits.data_ptr(),
                                      d_shadow_src.data_ptr(), d_shadow_count.data_ptr(), C.byref(lt), _f32p(df), spp,
                                      d_rgb.data_ptr(), _stream_ptr(stream)))

    def gen_path_rays(self, d_rays, d_hits, d_weights, d_pixels, d_ids, n, d_out_rays, d_out_weights, d_out_pixels, d_out_ids,
                      d_count, spp=1, seed=168, bounce=0, kinds=MR_PATH_MIRROR | MR_PATH_REFRACT | MR_PATH_DIFFUSE, stream=None,
                      out_capacity=None, d_out_octants=None):
        """mr_gen_path_rays: the PATH_TRACING generators (Ray.h:124-158,235-239): up to 4 children per hit.
        out_capacity: rays the output tensors hold (default: the shortest of them)"""
        def ptr(t):
            return t.data_ptr() if t is not None else None
        if out_capacity is None:
            out_capacity = min(t.shape[0] for t in (d_out_rays, d_out_weights, d_out_pixels, d_out_ids) if t is not None)
        _check(self.L.mr_gen_path_rays(self.h, d_rays.data_ptr(), d_hits.data_ptr(), ptr(d_weights), ptr(d_pixels), ptr(d_ids),
                                       n, spp, seed, bounce, kinds, d_out_rays.data_ptr(), d_out_weights.data_ptr(),
                                       d_out_pixels.data_ptr(), ptr(d_out_ids), d_count.data_ptr(), out_capacity, ptr(d_out_octants),
                                       _stream_ptr(stream)))

    def trace_level(self, d_rays, d_weights, d_pixels, d_ids, n, d_rgb, light_pos, wattage, children=MR_LEVEL_LAST, d_out_rays=None,
                    d_out_weights=None, d_out_pixels=None, d_out_ids=None, d_out_count=None, d_counts=None, spp=1, flags=0,
                    seed=168, bounce=0, kinds=MR_PATH_MIRROR | MR_PATH_REFRACT, color=(1.0, 1.0, 1.0), stream=None, out_capacity=None,
                    d_out_octants=None, d_order=None):
        """mr_trace_level: trace -> shadow ray -> trace -> Phong::shade x weight -> pixel, and the next level's queue, in
        one launch (Scene.cpp:270-346)"""
        def ptr(t):
            return t.data_ptr() if t is not None else None
        ld = LevelDesc()
        ld.light.position[:] = light_pos
        ld.light.color[:] = color
        ld.light.wattage = wattage
        ld.spp, ld.flags, ld.children, ld.path_kinds, ld.seed, ld.bounce = spp, flags, children, kinds, seed, bounce
        if out_capacity is None:
            outs = [t for t in (d_out_rays, d_out_weights, d_out_pixels, d_out_ids) if t is not None]
            out_capacity = min(t.shape[0] for t in outs) if outs else 0
        ld.out_capacity_lo, ld.out_capacity_hi = out_capacity & 0xFFFFFFFF, out_capacity >> 32
        ld.d_out_octants, ld.d_order = ptr(d_out_octants), ptr(d_order)
        _check(self.L.mr_trace_level(self.h, C.byref(ld), d_rays.data_ptr(), ptr(d_weights), ptr(d_pixels), ptr(d_ids), n,
                                     d_rgb.data_ptr(), ptr(d_out_rays), ptr(d_out_weights), ptr(d_out_pixels), ptr(d_out_ids),
                                     ptr(d_out_count), ptr(d_counts), _stream_ptr(stream)))

    def deinterleave_bands(self, d_recv, d_full, W, H, band_rows, world, shard_rows, floats_per_pixel=3, stream=None):
        _check(self.L.mr_deinterleave_bands(self.h, d_recv.data_ptr(), d_full.data_ptr(), W, H, band_rows, world, shard_rows,
                                            floats_per_pixel, _stream_ptr(stream)))

    def render_direct(self, cam, W, H, d_rgb, light_pos, wattage, y0=0, y1=None, bands=None, spp=1, jitter=False, seed=168,
                      tiled=False, flags=0, color=(1.0, 1.0, 1.0), diffuse=(1.0, 1.0, 1.0), d_hits=None, d_shadow_hits=None,
                      d_counts=None, stream=None):
        """mr_render_direct: eye rays -> trace -> shadow rays -> trace -> Phong shade -> pixel means in one launch.
        bands = (rows per band, rank, world) selects one rank's interleaved bands instead of the rows [y0,y1)."""
        fd = FrameDesc()
        fd.camera = cam
        fd.W, fd.H, fd.y0, fd.y1 = W, H, y0, H if y1 is None else y1
        fd.band_rows, fd.band_rank, fd.band_world = bands if bands is not None else (1, 0, 1)
        fd.spp, fd.jitter, fd.seed, fd.tiled, fd.flags = spp, 1 if jitter else 0, seed, 1 if tiled else 0, flags
        fd.light.position[:] = light_pos
        fd.light.color[:] = color
        fd.light.wattage = wattage
        fd.diffuse[:] = diffuse
        _check(self.L.mr_render_direct(self.h, C.byref(fd), d_rgb.data_ptr(),
                                       d_hits.data_ptr() if d_hits is not None else None,
                                       d_shadow_hits.data_ptr() if d_shadow_hits is not None else None,
                                       d_counts.data_ptr() if d_counts is not None else None, _stream_ptr(stream)))

    def final_gather(self, global_map, caustic_map, d_rays, d_hits, n, d_scratch, d_rgb, max_dist=1e10, nphotons=500,
                     spp=1, stream=None):
        """Scene.cpp:285-299: irradiance + caustic estimate at every diffuse hit, added to the pixels."""
        _check(self.L.mr_final_gather(self.h, global_map.h if global_map is not None else None,
                                      caustic_map.h if caustic_map is not None else None, d_rays.data_ptr(),
                                      d_hits.data_ptr(), n, max_dist, nphotons, spp, d_scratch.data_ptr(), d_rgb.data_ptr(),
                                      _stream_ptr(stream)))

    def set_materials(self, materials, prim_material=None):
        """materials: list of (diffuse, specular, transmission, shininess, refract_index) as Phong's constructor takes
        them (Phong.h:10-14); prim_material: material id per triangle (None: material 0 everywhere)."""
        arr = (Material * len(materials))()
        for i, (kd, ks, kt, sh, ri) in enumerate(materials):
            arr[i].diffuse[:] = kd
            arr[i].specular[:] = ks
            arr[i].transmission[:] = kt
            arr[i].shininess = sh
            arr[i].refract_index = ri
        pm = None
        if prim_material is not None:
            pm = np.ascontiguousarray(prim_material, dtype=np.uint32)
        _check(self.L.mr_scene_set_materials(self.h, arr, len(materials), _u32p(pm) if pm is not None else None))

    def shade_accumulate(self, d_rays, d_hits, d_weights, d_pixels, n, d_shadow_rays, d_shadow_hits, d_shadow_src,
                         d_shadow_count, light_pos, wattage, d_rgb, spp=1, color=(1.0, 1.0, 1.0), stream=None):
        lt = Light()
        lt.position[:] = light_pos
        lt.color[:] = color
        lt.wattage = wattage
        _check(self.L.mr_shade_accumulate(self.h, d_rays.data_ptr(), d_hits.data_ptr(),
                                          d_weights.data_ptr() if d_weights is not None else None,
                                          d_pixels.data_ptr() if d_pixels is not None else None, n,
                                          d_shadow_rays.data_ptr(), d_shadow_hits.data_ptr(), d_shadow_src.data_ptr(),
                                          d_shadow_count.data_ptr(), C.byref(lt), spp, d_rgb.data_ptr(), _stream_ptr(stream)))

    def gen_secondary_rays(self, d_rays, d_hits, d_weights, d_pixels, n, d_out_rays, d_out_weights, d_out_pixels, d_count,
                           spp=1, stream=None, out_capacity=None, d_out_octants=None):
        if out_capacity is None:
            out_capacity = min(t.shape[0] for t in (d_out_rays, d_out_weights, d_out_pixels))
        _check(self.L.mr_gen_secondary_rays(self.h, d_rays.data_ptr(), d_hits.data_ptr(),
                                            d_weights.data_ptr() if d_weights is not None else None,
                                            d_pixels.data_ptr() if d_pixels is not None else None, n, spp,
                                            d_out_rays.data_ptr(), d_out_weights.data_ptr(), d_out_pixels.data_ptr(),
                                            d_count.data_ptr(), out_capacity,
                                            d_out_octants.data_ptr() if d_out_octants is not None else None, _stream_ptr(stream)))

    def tonemap(self, d_rgb, n_values, d_out, stream=None):
        _check(self.L.mr_tonemap(self.h, d_rgb.data_ptr(), n_values, d_out.data_ptr(), _stream_ptr(stream)))


class PhotonMap:
    """Photon_map (PhotonMap.h:42-105) on one device: store, scale_photon_power, balance, irradiance_estimate."""

    def __init__(self, max_photons, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _check(self.L.mr_photon_map_create(device, max_photons, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.mr_photon_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store(self, power, pos, direction):
        power, pos, direction = (np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (power, pos, direction))
        _check(self.L.mr_photon_map_store(self.h, len(pos), _f32p(power), _f32p(pos), _f32p(direction)))

    def scale_photon_power(self, scale):
        _check(self.L.mr_photon_map_scale(self.h, scale))

    def balance(self, host_only=False):
        _check(self.L.mr_photon_map_balance(self.h, 1 if host_only else 0))

    def count(self):
        n = C.c_uint32(0)
        _check(self.L.mr_photon_map_count(self.h, C.byref(n)))
        return n.value

    def export(self):
        n = self.count()
        pos, power = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        plane = np.empty(n, np.int32)
        tp = np.empty((n, 2), np.uint8)
        _check(self.L.mr_photon_map_export(self.h, _f32p(pos), plane.ctypes.data_as(C.POINTER(C.c_int32)),
                                           tp.ctypes.data_as(C.POINTER(C.c_uint8)), _f32p(power)))
        return pos, plane, tp, power

    def irradiance_estimate(self, d_pos, d_normal, n, d_irrad, max_dist=1e10, nphotons=500, d_found=None, d_r2=None,
                            stream=None):
        """Device tensors: d_pos / d_normal [n,3] float32, d_irrad [n,3]; optional d_found int32 [n], d_r2 float32 [n]."""
        _check(self.L.mr_irradiance_estimate(self.h, d_pos.data_ptr(), d_normal.data_ptr(), n, max_dist, nphotons,
                                             d_irrad.data_ptr(), d_found.data_ptr() if d_found is not None else None,
                                             d_r2.data_ptr() if d_r2 is not None else None, _stream_ptr(stream)))

    STAT_NAMES = ("queries", "blocks", "records_searched", "tightenings", "records_prepass", "repeated_searches", "boxes_measured",
                  "candidates", "expansions", "unused9", "unguessed", "unused11")

    def count_stats(self, enable=True):
        """mr_photon_map_count_stats: the estimates on this map run the counting build of the kernel while enabled."""
        _check(self.L.mr_photon_map_count_stats(self.h, 1 if enable else 0))

    def stats(self, reset=True):
        """mr_photon_map_get_stats as a dict (synchronises the device)."""
        out = (C.c_uint64 * 12)()
        _check(self.L.mr_photon_map_get_stats(self.h, out, 1 if reset else 0))
        return dict(zip(self.STAT_NAMES, (int(v) for v in out)))
