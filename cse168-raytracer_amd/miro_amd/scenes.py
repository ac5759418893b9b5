"""Scene descriptions of the BASELINE configs: camera / light / model constants restated from the
reference's hard-coded scene functions (assignment2.cpp), plus the procedural stand-in for the
missing models/sponza.obj (.MISSING_LARGE_BLOBS:5).

A description is plain data; ``populate(scene, desc)`` feeds it to anything with the
add_obj / add_triangle interface (miro_amd.Scene, or the test oracle's Scene).
"""
import os

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODEL_DIR = os.path.join(_REPO, "tests", "golden", "models")


def _model(name):
    return os.path.join(MODEL_DIR, name)


UP = (0.0, 1.0, 0.0)

# name -> description.  floor = the single "floor triangle" each make*Scene adds after the mesh.
SCENES = {
    # BASELINE config 1: models/cornell_box.obj with makeCornellScene's camera (assignment2.cpp:384-387,397)
    "cornell": dict(models=[("cornell_box.obj", None)], floor=None,
                    eye=(2.5, 3.0, 3.0), lookat=(2.5, 2.5, 0.0), up=UP, fov=90.0, light=(2.5, 4.9, -1.0),
                    wattage=160.0),
    # BASELINE config 2: makeTeapotScene (assignment2.cpp:24-70)
    "teapot": dict(models=[("teapot.obj", None)],
                   floor=((-10, 0, -10), (0, 0, 10), (10, 0, -10)),
                   eye=(0.0, 3.0, 6.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(10.0, 10.0, 10.0),
                   wattage=700.0),
    # BASELINE config 3: makeBunny1Scene (assignment2.cpp:73-119)
    "bunny": dict(models=[("bunny.obj", None)],
                  floor=((-100, 0, -100), (0, 0, 100), (100, 0, -100)),
                  eye=(0.0, 5.0, 15.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(10.0, 20.0, 10.0),
                  wattage=1000.0),
    # BASELINE config 4: makeSponzaScene (assignment2.cpp:341-371); the model is absent from the
    # reference, "sponza" resolves to the real file when MIRO_SPONZA_OBJ points at it, else to the
    # procedural atrium below (every number then carries the label "sponza-standin").
    "sponza": dict(models=[("@sponza", None)], floor=None,
                   eye=(8.0, 1.5, 1.0), lookat=(0.0, 2.5, -1.0), up=UP, fov=55.0, light=(0.0, 10.0, 0.0),
                   wattage=200.0),
    # micro fixtures
    "testobj": dict(models=[("testobj.obj", None)], floor=None,
                    eye=(0.5, 0.5, 3.0), lookat=(0.5, 0.5, 0.0), up=UP, fov=60.0, light=(0.0, 5.0, 5.0), wattage=100.0),
    "sphere": dict(models=[("sphere.obj", None)], floor=None,
                   eye=(0.0, 0.0, 4.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(5.0, 5.0, 5.0), wattage=100.0),
}


def _spiral_objects():
    """makeSpiralScene (assignment1.cpp:31-72): 149 spheres on a spiral, the plane y = -2, one triangle.  Centres and
    radii are evaluated in double and rounded to fp32 once (the reference's own float/double mix depends on which
    cos/sin overload its headers select; both sides of every parity test are fed these same numbers)."""
    objs = []
    max_i, a = 150, np.float32(0.15)
    pi = np.float32(3.1415926535897932384626433832795028841972)
    for i in range(1, max_i):
        t = np.float32(i) / np.float32(max_i)
        theta = np.float32(np.float32(4) * pi * t)
        r = np.float32(a * theta)
        x, y = np.float32(r * np.cos(np.float64(theta))), np.float32(r * np.sin(np.float64(theta)))
        z = np.float32(np.float32(2) * (np.float32(2) * pi * a - r))
        objs.append(("sphere", (float(x), float(y), float(z)), float(np.float32(r / np.float32(10)))))
    objs.append(("plane", (0.0, 1.0, 0.0), (0.0, -2.0, 0.0)))
    n2 = np.asarray([0.1, 0.1, -1.0], np.float32)
    n3 = np.asarray([-0.1, -0.2, -1.0], np.float32)
    n2 = n2 * (np.float32(1) / np.sqrt((n2 * n2).sum(dtype=np.float32)))
    n3 = n3 * (np.float32(1) / np.sqrt((n3 * n3).sum(dtype=np.float32)))
    objs.append(("tri", (0, 0, 0, 0, 3, 0, 5, 5, 0), (0, 0, -1) + tuple(float(c) for c in n2) + tuple(float(c) for c in n3)))
    return objs


# scenes with spheres / planes (assignment1.cpp); `objects` are added in order after any models
SCENES["spiral"] = dict(models=[], floor=None, objects=_spiral_objects(),
                        eye=(0.0, 0.0, -5.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(-3.0, 15.0, -15.0),
                        wattage=1000.0)
# A1makeSphereScene (assignment1.cpp:383-433): the floor triangle, then a sphere whose centre is never set and
# therefore is Vector3() = (0, 1, 2) (Vector3.h:26-27)
SCENES["a1sphere"] = dict(models=[], floor=None,
                          objects=[("tri", (0, -1.5, 10, 10, -1.5, -10, -10, -1.5, -10), (0, 1, 0) * 3),
                                   ("sphere", (0.0, 1.0, 2.0), 1.5)],
                          eye=(-2.0, 1.0, 5.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(-3.0, 15.0, 3.0),
                          wattage=500.0)


# ---------------------------------------------------------------------------------------------
# makeBunny20Scene (assignment2.cpp:124-339): twenty transformed copies of bunny.obj.  The transforms are scene
# constants, composed here with the reference's own fp32 arithmetic -- Matrix4x4::operator*= (Matrix4x4.h:463-493:
# A = A * B, every element ((a*b + c*d) + e*f) + g*h in float) and translate / scale / rotate (assignment2.cpp:464-511:
# angle in degrees, rad = angle * (PI / 180.) evaluated in double and stored as float, the axis NOT normalised).
# The write-up publishes this scene's counters (writeup/A2/Readme.tex:97,101): 876 137 nodes, 438 069 leaves,
# 495 502 rays with shadows = 262 144 primary + 233 358 hits; tests/test_oracle_kat.py reproduces them exactly.
# ---------------------------------------------------------------------------------------------
_F = np.float32


def _mat_mul(A, B):
    R = np.empty((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            R[i, j] = _F(_F(_F(_F(A[i, 0] * B[0, j]) + _F(A[i, 1] * B[1, j])) + _F(A[i, 2] * B[2, j])) + _F(A[i, 3] * B[3, j]))
    return R


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[0, 3], m[1, 3], m[2, 3] = _F(x), _F(y), _F(z)
    return m


def _scale(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = _F(x), _F(y), _F(z)
    return m


def _rotate(angle, x, y, z):
    angle, x, y, z = _F(angle), _F(x), _F(y), _F(z)
    pi = _F(3.1415926535897932384626433832795028841972)
    rad = _F(float(angle) * (float(pi) / 180.0))
    c, s = _F(np.cos(np.float64(rad))), _F(np.sin(np.float64(rad)))
    x2, y2, z2 = _F(x * x), _F(y * y), _F(z * z)
    cinv = _F(_F(1) - c)
    xy, xz, yz, xs, ys, zs = _F(x * y), _F(x * z), _F(y * z), _F(x * s), _F(y * s), _F(z * s)
    xzc, xyc, yzc = _F(xz * cinv), _F(xy * cinv), _F(yz * cinv)
    m = np.eye(4, dtype=np.float32)
    m[0, :3] = [_F(x2 + _F(c * _F(_F(1) - x2))), _F(_F(xy * cinv) + zs), _F(xzc - ys)]
    m[1, :3] = [_F(xyc - zs), _F(y2 + _F(c * _F(_F(1) - y2))), _F(yzc + xs)]
    m[2, :3] = [_F(xzc + ys), _F(yzc - xs), _F(z2 + _F(c * _F(_F(1) - z2)))]
    return m


def _bunny20_models():
    T, S, R = _translate, _scale, _rotate
    sequences = [[S(0.3, 2.0, 0.7), T(-1, .4, .3), R(25, .3, .1, .6)],
                 [S(.6, 1.2, .9), T(7.6, .8, .6)],
                 [T(.7, 0, -2), R(120, 0, .6, 1)],
                 [T(3.6, 3, -1)],
                 [T(-2.4, 2, 3), S(1, .8, 2)],
                 [T(5.5, -.5, 1), S(1, 2, 1)],
                 [R(15, 0, 0, 1), T(-4, -.5, -6), S(1, 2, 1)],
                 [R(60, 0, 1, 0), T(5, .1, 3)],
                 [T(-3, .4, 6), R(-30, 0, 1, 0)],
                 [T(3, 0.5, -2), R(180, 0, 1, 0), S(1.5, 1.5, 1.5)]]
    xform2 = np.eye(4, dtype=np.float32)
    for m in (R(110, 0, 1, 0), S(.6, 1, 1.1)):                 # assignment2.cpp:150-152
        xform2 = _mat_mul(xform2, m)
    models = []
    for start in (np.eye(4, dtype=np.float32), xform2):        # bunnies 1-10 from the identity, 11-20 from xform2
        for seq in sequences:
            m = start.copy()
            for b in seq:
                m = _mat_mul(m, b)
            models.append(("bunny.obj", m))
    return models


SCENES["bunny20"] = dict(models=_bunny20_models(), floor=((-100, 0, -100), (0, 0, 100), (100, 0, -100)),
                         eye=(0.0, 5.0, 15.0), lookat=(0.0, 0.0, 0.0), up=UP, fov=45.0, light=(10.0, 20.0, 10.0),
                         wattage=1000.0)


def sponza_label():
    p = os.environ.get("MIRO_SPONZA_OBJ", "")
    return "sponza" if p and os.path.exists(p) else "sponza-standin"


def sponza_path(cache_dir=None):
    """Path of the OBJ used for the 'sponza' scene (real file if supplied, else the generated atrium)."""
    p = os.environ.get("MIRO_SPONZA_OBJ", "")
    if p and os.path.exists(p):
        return p
    cache_dir = cache_dir or os.environ.get("MIRO_CACHE_DIR") or os.path.join("/tmp", "miro_amd_cache_%d" % os.getuid())
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, "atrium_standin_v3.obj")
    if not os.path.exists(path):
        tmp = path + ".%d.tmp" % os.getpid()
        write_obj(tmp, *atrium_mesh())
        os.replace(tmp, path)
    return path


def populate(scene, desc, cache_dir=None):
    """Assemble `desc` into `scene` in the reference's order: meshes first, floor triangle last."""
    if isinstance(desc, str):
        desc = SCENES[desc]
    n = 0
    for name, ctm in desc["models"]:
        path = sponza_path(cache_dir) if name == "@sponza" else (name if os.path.isabs(name) else _model(name))
        n += scene.add_obj(path, ctm)
    if desc.get("floor") is not None:
        f = np.asarray(desc["floor"], np.float32).reshape(9)
        scene.add_triangle(f, np.asarray([0, 1, 0] * 3, np.float32))
        n += 1
    for obj in desc.get("objects", ()):
        if obj[0] == "sphere":
            scene.add_sphere(obj[1], obj[2])
        elif obj[0] == "plane":
            scene.add_plane(obj[1], obj[2])
        else:
            scene.add_triangle(np.asarray(obj[1], np.float32), np.asarray(obj[2], np.float32))
        n += 1
    return n


# ---------------------------------------------------------------------------------------------
# procedural stand-in for sponza.obj: a closed two-storey colonnaded atrium, ~77k triangles,
# sized so that makeSponzaScene's camera (8,1.5,1)->(0,2.5,-1) and light (0,10,0) sit inside it.
# Deterministic (no RNG beyond a fixed-seed relief), pure numpy.
# ---------------------------------------------------------------------------------------------
class _MeshBuilder:
    def __init__(self):
        self.v, self.f = [], []
        self.nv = 0

    def add(self, verts, faces):
        verts = np.asarray(verts, np.float64).reshape(-1, 3)
        faces = np.asarray(faces, np.int64).reshape(-1, 3)
        self.v.append(verts)
        self.f.append(faces + self.nv)
        self.nv += len(verts)

    def grid(self, origin, du, dv, nu, nv, height=None):
        """(nu x nv)-cell rectangle patch origin + s*du + t*dv, optional displacement height(s,t) along the normal."""
        origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
        s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
        p = origin + s[..., None] * du + t[..., None] * dv
        if height is not None:
            nrm = np.cross(du, dv)
            nrm /= np.linalg.norm(nrm)
            p = p + height(s, t)[..., None] * nrm
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
        self.add(p.reshape(-1, 3), faces)

    def tube(self, path, radius, nseg, closed_caps=True):
        """Swept circle of varying radius along a polyline (columns, balusters, arches)."""
        path = np.asarray(path, np.float64)
        radius = np.broadcast_to(np.asarray(radius, np.float64), (len(path),))
        tang = np.gradient(path, axis=0)
        tang /= np.linalg.norm(tang, axis=1, keepdims=True)
        ref = np.where(np.abs(tang[:, 1:2]) < 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
        e1 = np.cross(tang, ref)
        e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
        e2 = np.cross(tang, e1)
        ang = np.linspace(0, 2 * np.pi, nseg, endpoint=False)
        ring = (np.cos(ang)[None, :, None] * e1[:, None, :] + np.sin(ang)[None, :, None] * e2[:, None, :])
        p = path[:, None, :] + radius[:, None, None] * ring
        m = len(path)
        idx = np.arange(m * nseg).reshape(m, nseg)
        a, b = idx[:-1, :], idx[1:, :]
        a2, b2 = np.roll(a, -1, axis=1), np.roll(b, -1, axis=1)
        faces = [np.stack([a, b, b2], -1).reshape(-1, 3), np.stack([a, b2, a2], -1).reshape(-1, 3)]
        verts = [p.reshape(-1, 3)]
        if closed_caps:
            c0, c1 = m * nseg, m * nseg + 1
            verts.append(path[[0, -1]])
            r0, r1 = idx[0], idx[-1]
            faces.append(np.stack([np.full(nseg, c0), np.roll(r0, -1), r0], -1))
            faces.append(np.stack([np.full(nseg, c1), r1, np.roll(r1, -1)], -1))
        self.add(np.concatenate(verts), np.concatenate(faces))

    def box(self, lo, hi):
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                      [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
        f = [[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
             [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]]
        self.add(c, f)

    def finish(self):
        v = np.concatenate(self.v)
        f = np.concatenate(self.f)
        return v, f


def atrium_mesh():
    """Returns (vertices float64 [nv,3], faces int [nt,3]).

    Calibrated against what the reference's write-up publishes for the real sponza.obj under its own builder
    (writeup/A2/Readme.tex:95-102): 42 645 BVH nodes, 54.8 node visits + 9.93 triangle tests per primary ray, 51.2 + 10.33
    per ray with shadow rays.  This mesh: 76 552 triangles, 46 701 nodes, 51.9 + 10.50 per primary ray, 42.3 + 10.13 with
    shadows (1920x1080 frame, counted by the restated traversal) -- i.e. about the real scene's work per ray, where a
    plain tessellated box would need half of it.  What buys the visits is clutter that rays graze: slender posts on the
    nave floor, long banners hanging along the view direction, a wire lantern around the light."""
    rng = np.random.RandomState(168)
    relief = rng.rand(64, 64)

    def brick(s, t):
        i = np.minimum((s * 63).astype(int), 63)
        j = np.minimum((t * 63).astype(int), 63)
        return 0.03 * relief[i, j]

    mb = _MeshBuilder()
    X0, X1, Y0, Y1, Z0, Z1 = -14.0, 14.0, 0.0, 12.0, -6.0, 6.0
    # shell: floor, ceiling, four walls (inward-facing, relief on the walls)
    mb.grid((X0, Y0, Z0), (0, 0, Z1 - Z0), (X1 - X0, 0, 0), 20, 48)                 # floor
    mb.grid((X0, Y1, Z0), (X1 - X0, 0, 0), (0, 0, Z1 - Z0), 14, 6)                   # ceiling
    mb.grid((X0, Y0, Z0), (X1 - X0, 0, 0), (0, Y1 - Y0, 0), 36, 15, brick)           # wall z=Z0
    mb.grid((X0, Y0, Z1), (0, Y1 - Y0, 0), (X1 - X0, 0, 0), 15, 36, brick)           # wall z=Z1
    mb.grid((X0, Y0, Z0), (0, Y1 - Y0, 0), (0, 0, Z1 - Z0), 12, 12, brick)           # wall x=X0
    mb.grid((X1, Y0, Z0), (0, 0, Z1 - Z0), (0, Y1 - Y0, 0), 12, 12, brick)           # wall x=X1
    # two storeys of colonnades along both long sides
    col_x = np.arange(-12.0, 12.0 + 1e-9, 3.0)
    for zc in (-3.5, 3.5):
        for storey, (yb, yt, r) in enumerate(((0.0, 4.2, 0.32), (5.0, 9.0, 0.24))):
            for x in col_x:
                ys = np.linspace(yb, yt, 12)
                prof = r * (1.0 + 0.35 * np.exp(-((ys - yb) / 0.25) ** 2) + 0.45 * np.exp(-((ys - yt) / 0.3) ** 2)
                            - 0.08 * np.sin(np.pi * (ys - yb) / (yt - yb)))
                path = np.stack([np.full_like(ys, x), ys, np.full_like(ys, zc)], 1)
                mb.tube(path, prof, 16)
                mb.box((x - 0.5, yb, zc - 0.5), (x + 0.5, yb + 0.18, zc + 0.5))      # plinth
                mb.box((x - 0.5, yt, zc - 0.5), (x + 0.5, yt + 0.2, zc + 0.5))       # abacus
            # arches between neighbouring columns
            for xa, xb in zip(col_x[:-1], col_x[1:]):
                th = np.linspace(0, np.pi, 14)
                cx, rad = 0.5 * (xa + xb), 0.5 * (xb - xa)
                path = np.stack([cx - rad * np.cos(th), yt + 0.2 + 0.75 * rad * np.sin(th), np.full_like(th, zc)], 1)
                mb.tube(path, 0.16, 10, closed_caps=False)
        # gallery slab (with thickness) and balustrade on the upper floor
        zin, zout = (zc + 0.6, Z0) if zc < 0 else (zc - 0.6, Z1)
        zlo, zhi = min(zin, zout), max(zin, zout)
        mb.grid((X0, 4.95, zlo), (0, 0, zhi - zlo), (X1 - X0, 0, 0), 6, 56)          # gallery floor (top)
        mb.grid((X0, 4.55, zlo), (X1 - X0, 0, 0), (0, 0, zhi - zlo), 56, 6)          # gallery underside
        mb.grid((X0, 4.55, zin), (X1 - X0, 0, 0), (0, 0.4, 0), 56, 1)                # fascia
        for x in np.arange(-13.5, 13.5 + 1e-9, 1.0):
            ys = np.linspace(4.95, 5.85, 7)
            prof = 0.045 * (1.0 + 0.8 * np.sin(np.pi * (ys - 4.95) / 0.9) ** 2)
            mb.tube(np.stack([np.full_like(ys, x), ys, np.full_like(ys, zin)], 1), prof, 8)
        mb.box((X0, 5.85, zin - 0.07), (X1, 5.97, zin + 0.07))                       # hand rail
    # hanging drapes: wavy sheets across the nave
    for x0, amp in ((-7.0, 0.35), (3.5, 0.45), (10.5, 0.3)):
        def wave(s, t, amp=amp):
            return amp * np.sin(6 * np.pi * t) * (0.3 + 0.7 * s)
        mb.grid((x0, 10.5, -2.4), (0, -3.5, 0), (0, 0, 4.8), 20, 28, wave)
    # a few floor objects (vases) for small-scale geometry near the camera
    for x, z in ((5.0, -1.5), (5.0, 1.8), (2.0, 0.0), (-3.0, -1.0), (-8.0, 1.0), (9.5, -2.0)):
        ys = np.linspace(0.0, 1.2, 14)
        prof = 0.12 + 0.28 * np.sin(np.pi * ys / 1.2) ** 1.5 * (1.0 - 0.35 * ys / 1.2)
        mb.tube(np.stack([np.full_like(ys, x), ys, np.full_like(ys, z)], 1), prof, 24)
    # clutter that rays graze (what makes the real scene expensive): slender posts scattered over the nave floor ...
    prng = np.random.RandomState(5)
    for _ in range(400):
        x, z = prng.uniform(-13, 13), prng.uniform(-2.6, 2.6)
        h = prng.uniform(1.0, 4.0)
        ys = np.linspace(0.0, h, 6)
        mb.tube(np.stack([np.full_like(ys, x), ys, np.full_like(ys, z)], 1), 0.03 + 0.05 * np.sin(np.pi * ys / h) ** 2, 5,
                closed_caps=False)
    # ... long banners hanging along the nave axis, i.e. along the view direction ...
    for k in range(6):
        z = -2.2 + 4.4 * (k + 0.5) / 6

        def ripple(s, t, k=k):
            return 0.15 * np.sin(9 * np.pi * t + k)
        mb.grid((-13.0, 10.8, z), (0, -4.0 - (k % 2), 0), (26.0, 0, 0), 8, 60, ripple)
    # ... and a wire lantern (two shells of meridians and parallels) around the light at (0, 10, 0)
    c = np.array([0.0, 10.0, 0.0])
    for shell in range(2):
        rad = 1.6 * (1.0 + 0.45 * shell)
        for k in range(12):
            ph = 2 * np.pi * (k + 0.5 * shell) / 12
            th = np.linspace(0.12, np.pi - 0.12, 14)
            mb.tube(c + rad * np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], 1), 0.02, 5, closed_caps=False)
        for k in range(6):
            th = np.pi * (k + 1) / 7
            ph = np.linspace(0, 2 * np.pi, 28)
            mb.tube(c + rad * np.stack([np.sin(th) * np.cos(ph), np.full_like(ph, np.cos(th)), np.sin(th) * np.sin(ph)], 1),
                    0.02, 5, closed_caps=False)
    return mb.finish()


def write_obj(path, verts, faces):
    """Plain 'v' / 'f' records (no normals: the loader synthesises and smooths them, like bunny.obj)."""
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64) + 1
    with open(path, "w") as fh:
        fh.write("# procedural atrium stand-in for sponza.obj (miro_amd.scenes.atrium_mesh, seed 168)\n")
        np.savetxt(fh, verts, fmt="v %.6f %.6f %.6f")
        np.savetxt(fh, faces, fmt="f %d %d %d")


def synthetic_photons(v, vi, n, seed=168, power=1.0):
    """Config 5's synthetic photon set: n photons uniform on the scene's surfaces (area-weighted),
    incoming directions cosine-distributed about the inward side of the face normal, equal power.  Returns
    (power[n,3], pos[n,3], dir[n,3]) float32.  Deterministic (numpy RandomState(seed))."""
    rng = np.random.RandomState(seed)
    v = np.asarray(v, np.float64)
    vi = np.asarray(vi, np.int64)
    a, b, c = v[vi[:, 0]], v[vi[:, 1]], v[vi[:, 2]]
    fn = np.cross(b - a, c - a)
    area = 0.5 * np.linalg.norm(fn, axis=1)
    cdf = np.cumsum(area)
    tri = np.searchsorted(cdf, rng.rand(n) * cdf[-1])
    tri = np.minimum(tri, len(vi) - 1)
    r1, r2 = np.sqrt(rng.rand(n)), rng.rand(n)
    pos = (1 - r1)[:, None] * a[tri] + (r1 * (1 - r2))[:, None] * b[tri] + (r1 * r2)[:, None] * c[tri]
    nrm = fn[tri] / np.maximum(np.linalg.norm(fn[tri], axis=1, keepdims=True), 1e-30)
    # cosine-distributed direction about nrm, then reversed: the photon arrives against the normal
    u1, u2 = rng.rand(n), rng.rand(n)
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    t1 = np.cross(nrm, np.where(np.abs(nrm[:, :1]) < 0.9, [[1.0, 0, 0]], [[0, 1.0, 0]]))
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    d = rr[:, None] * np.cos(phi)[:, None] * t1 + rr[:, None] * np.sin(phi)[:, None] * t2 + np.sqrt(1 - u1)[:, None] * nrm
    pw = np.full((n, 3), power, np.float32)
    return pw, pos.astype(np.float32), (-d).astype(np.float32)
