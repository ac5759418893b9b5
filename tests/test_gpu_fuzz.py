"""Differential fuzzing of the traversal: seeded random scenes built to provoke ties and special values (vertices on a
coarse grid, duplicated and zero-area triangles, spheres, planes, every leaf size) and seeded ray sets built the same
way (grid origins, axis-parallel and signed-zero direction components, rays from vertex to vertex, tMax placed exactly
on / one ulp either side of a known hit).  The HIP path must return the oracle's hit records bit for bit in the default
mode and the reference's counters under MR_COUNT_STATS; the any-hit flag must agree on hit / miss.

MIRO_FUZZ_SEEDS sets the number of scenes and MIRO_FUZZ_BASE the first seed (default 8 from 1680 keeps the suite
short; the round-1 campaigns are in profiles/r01_fuzz.log)."""
import os

import numpy as np
import pytest

from helpers import assert_hits_bit_exact

pytestmark = pytest.mark.gpu

SEEDS = int(os.environ.get("MIRO_FUZZ_SEEDS", "8"))
BASE = int(os.environ.get("MIRO_FUZZ_BASE", "1680"))
RAYS = 24000


def make_scene(rng):
    """(list of construction steps, leaf size): the same list is replayed on the oracle and on the product."""
    steps = []
    grid = float(rng.choice([0.25, 0.5, 1.0, 0.1]))
    span = int(rng.integers(2, 12))
    for _ in range(int(rng.integers(1, 4))):
        nt = int(rng.choice([1, 2, 5, 40, 300, 2500]))
        kind = int(rng.integers(0, 3))
        if kind == 0:      # soup on a grid: shared planes, shared edges, exact ties
            v = rng.integers(-span, span + 1, (3 * nt, 3)).astype(np.float32) * np.float32(grid)
        elif kind == 1:    # small triangles scattered on a grid
            c = rng.integers(-span, span + 1, (nt, 1, 3)).astype(np.float32) * np.float32(grid)
            v = (c + rng.integers(-1, 2, (nt, 3, 3)).astype(np.float32) * np.float32(grid)).reshape(-1, 3)
        else:              # generic positions
            v = ((rng.random((3 * nt, 3)) - 0.5) * (2 * span * grid)).astype(np.float32)
        f = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
        dup = rng.integers(0, nt, max(1, nt // 10))            # duplicated triangles: equal t, first in order wins
        f = np.concatenate([f, f[dup]])
        if rng.random() < 0.5:                                 # zero-area ones
            z = f[rng.integers(0, len(f), max(1, nt // 20))].copy()
            z[:, 2] = z[:, 1]
            f = np.concatenate([f, z])
        n = rng.standard_normal((len(v), 3)).astype(np.float32)
        steps.append(("mesh", v, n, f))
        if rng.random() < 0.4:
            for _ in range(int(rng.integers(1, 30))):
                c = rng.integers(-span, span + 1, 3).astype(np.float32) * np.float32(grid)
                steps.append(("sphere", c, float(np.float32(rng.choice([grid, 0.5 * grid, 3 * grid, 0.0])))))
    if rng.random() < 0.4:
        for _ in range(int(rng.integers(1, 4))):
            nrm = rng.integers(-1, 2, 3).astype(np.float32)
            if not nrm.any():
                nrm[1] = 1.0
            org = rng.integers(-span, span + 1, 3).astype(np.float32) * np.float32(grid)
            steps.append(("plane", nrm, org))
    return steps, int(rng.choice([1, 2, 4, 8])), grid * span


def replay(s, steps, leaf):
    for st in steps:
        if st[0] == "mesh":
            s.add_arrays(st[1], st[2], st[3], st[3])
        elif st[0] == "sphere":
            s.add_sphere(st[1], st[2])
        else:
            s.add_plane(st[1], st[2])
    s.build(leaf)
    return s


def make_rays(rng, dtype, steps, extent, n=RAYS):
    verts = np.concatenate([st[1] for st in steps if st[0] == "mesh"])
    o = ((rng.random((n, 3)) - 0.5) * 3 * extent).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    q = n // 8
    # grid origins, axis-parallel directions with +0 / -0 in the other components
    o[:q] = np.round(o[:q] / np.float32(0.25)) * np.float32(0.25)
    ax = rng.integers(0, 3, q)
    d[:q] = np.where(rng.random((q, 3)) < 0.5, np.float32(0.0), np.float32(-0.0))
    d[np.arange(q), ax] = np.where(rng.random(q) < 0.5, 1.0, -1.0).astype(np.float32)
    # one zero component
    d[q:2 * q, 0] = 0.0
    d[2 * q:3 * q, rng.integers(0, 3)] = -0.0
    # from a vertex towards a vertex (through edges and corners, starting on surfaces)
    a, b = verts[rng.integers(0, len(verts), q)], verts[rng.integers(0, len(verts), q)]
    o[3 * q:4 * q] = a
    d[3 * q:4 * q] = b - a
    # towards a vertex from outside, un-normalised
    o[4 * q:5 * q] = (o[4 * q:5 * q] * 2).astype(np.float32)
    d[4 * q:5 * q] = verts[rng.integers(0, len(verts), q)] - o[4 * q:5 * q]
    norm = np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    keep = (np.arange(n) >= 4 * q) & (np.arange(n) < 5 * q)
    d = np.where(keep[:, None] | (norm == 0), d, d / np.where(norm == 0, 1, norm)).astype(np.float32)
    rays = np.zeros(n, dtype)
    rays["ox"], rays["oy"], rays["oz"] = o[:, 0], o[:, 1], o[:, 2]
    rays["dx"], rays["dy"], rays["dz"] = d[:, 0], d[:, 1], d[:, 2]
    rays["tmin"] = np.where(rng.random(n) < 0.2, rng.random(n) * extent, 0.0).astype(np.float32)
    rays["tmax"] = np.where(rng.random(n) < 0.3, rng.random(n) * 3 * extent, 1e12).astype(np.float32)
    return rays


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_scene_bit_exact(oracle, miro, seed):
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(BASE + seed)
    steps, leaf, extent = make_scene(rng)
    a = replay(oracle.Scene(), steps, leaf)
    b = replay(miro.Scene(), steps, leaf)
    rays = make_rays(rng, oracle.RAY_DTYPE, steps, extent)
    for rnd in range(2):
        want, ctr = a.trace(rays, counters=True)
        want = want.view(miro.HIT_DTYPE)
        r = rays.view(miro.RAY_DTYPE)
        assert_hits_bit_exact(b.trace(r), want)
        b.stats()
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_COUNT_STATS), want)
        assert b.stats() == ctr
        anyh = b.trace(r, flags=miro.MR_TRACE_ANY)
        assert np.array_equal(anyh["prim"] != 0xFFFFFFFF, want["prim"] != 0xFFFFFFFF)
        # the product form may only differ where a box comparison ties to within its 3 ulp; count, do not fail
        prod = b.trace(r, flags=miro.MR_MATH_PRODUCT)
        diff = int((prod.view(np.uint32).reshape(-1, 4) != want.view(np.uint32).reshape(-1, 4)).any(1).sum())
        assert diff <= len(rays) // 100, diff
        # the persistent kernel computes what the one-shot kernel computes, in either arithmetic (with spheres or
        # planes in the scene the flag is a no-op and the one-shot kernels answer)
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_TRACE_PERSISTENT), want)
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_TRACE_PERSISTENT | miro.MR_MATH_PRODUCT), prod)
        # round 2: the voting control flow (alone and with the persistent refill) is the same per-ray sequence of steps
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_TRACE_INCOHERENT), want)
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_TRACE_INCOHERENT | miro.MR_TRACE_PERSISTENT), want)
        assert_hits_bit_exact(b.trace(r, flags=miro.MR_TRACE_INCOHERENT | miro.MR_MATH_PRODUCT), prod)
        # round 2: the octant-specialised slab tests only run when a whole wave's rays point into one octant, which random
        # batches almost never do: trace the same rays grouped by octant (a permutation, answers are per ray) so that
        # they do, in both arithmetic modes
        octant = (rays["dx"] < 0).astype(np.int32) | ((rays["dy"] < 0).astype(np.int32) << 1) | ((rays["dz"] < 0).astype(np.int32) << 2)
        order = np.argsort(octant, kind="stable")
        grouped = np.ascontiguousarray(r[order])
        assert_hits_bit_exact(b.trace(grouped), want[order])
        assert_hits_bit_exact(b.trace(grouped, flags=miro.MR_MATH_PRODUCT), prod[order])
        if diff and os.environ.get("MIRO_FUZZ_VERBOSE"):
            print(f"seed {seed} round {rnd}: product form differs on {diff} of {len(rays)} rays")
        if rnd == 0:
            # second round: tMax on, just below and just above each known hit distance
            hit = want["prim"] != 0xFFFFFFFF
            t = want["t"].copy()
            k = rng.integers(0, 3, len(rays))
            t = np.where(k == 0, t, np.where(k == 1, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(np.inf))))
            rays = rays.copy()
            rays["tmax"] = np.where(hit, t, rays["tmax"]).astype(np.float32)
            rays["tmin"] = np.where(hit & (rng.random(len(rays)) < 0.1), want["t"], rays["tmin"]).astype(np.float32)


def _same_bits_or_nan(got, want):
    g, w = np.ascontiguousarray(got), np.ascontiguousarray(want)
    gn, wn = np.isnan(g), np.isnan(w)
    return np.array_equal(gn, wn) and np.array_equal(g.view(np.uint32)[~gn], w.view(np.uint32)[~wn])


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_scene_surfaces_and_shadow_rays(oracle, miro, seed):
    """The callers either side of the traversal on the same scenes: HitInfo::P / ::N for triangles, spheres and planes
    (Triangle.cpp:160-162, Sphere.cpp:58-66, Plane.cpp:42-45) and the Phong shadow rays (Phong.cpp:80-97), bit for bit
    (NaNs, which zero-length normals produce, compared by position)."""
    import torch
    rng = np.random.default_rng(BASE + seed)
    steps, leaf, extent = make_scene(rng)
    a = replay(oracle.Scene(), steps, leaf)
    b = replay(miro.Scene(), steps, leaf)
    rays = make_rays(rng, oracle.RAY_DTYPE, steps, extent)
    hits = a.trace(rays)
    n = len(rays)
    light = ((rng.random(3) - 0.5) * 4 * extent).astype(np.float32)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_hits = torch.from_numpy(hits.view(np.float32).reshape(-1, 4).copy()).cuda()
    P = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    N = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    b.hit_attrs(d_hits, n, P, N, d_rays=d_rays)
    Pw, Nw = a.hit_attrs(hits, rays)
    hit = hits["prim"] != 0xFFFFFFFF
    assert _same_bits_or_nan(P.cpu().numpy()[hit], Pw[hit])
    assert _same_bits_or_nan(N.cpu().numpy()[hit], Nw[hit])
    want, src_want = a.shadow_rays(rays, hits, light)
    d_out = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_src = torch.empty(n, dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    b.gen_shadow_rays(d_rays, d_hits, n, light, d_out, d_src, d_cnt)
    k = int(d_cnt.item())
    assert k == len(want)
    src = d_src[:k].cpu().numpy().astype(np.int64)
    order = np.argsort(src, kind="stable")
    assert np.array_equal(src[order], src_want.astype(np.int64))
    got = d_out[:k].cpu().numpy()[order]
    assert _same_bits_or_nan(got, want.view(np.float32).reshape(-1, 8))


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_cameras_bit_exact(oracle, miro, seed):
    """Camera::eyeRay (Camera.cpp:104-161) for random cameras -- grid and generic eye points, near-degenerate up
    vectors, fov from 1 to 179 degrees, odd image sizes, row windows, 1-64 samples per pixel, jitter seeds."""
    import torch
    from miro_amd import binding
    rng = np.random.default_rng(BASE + seed)
    s = miro.Scene()
    s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    s.build(4)
    for _ in range(6):
        eye = (rng.integers(-8, 9, 3) * 0.5 if rng.random() < 0.5 else (rng.random(3) - 0.5) * 40).astype(np.float32)
        look = (rng.integers(-8, 9, 3) * 0.5 if rng.random() < 0.5 else (rng.random(3) - 0.5) * 40).astype(np.float32)
        if np.array_equal(eye, look):
            look = look + np.float32(1.0)
        up = rng.standard_normal(3).astype(np.float32) if rng.random() < 0.5 else np.asarray([0, 1, 0], np.float32)
        if rng.random() < 0.2:                                   # nearly along the view direction
            up = ((look - eye) + rng.standard_normal(3) * 1e-3).astype(np.float32)
        fov = float(rng.choice([1.0, 20.0, 45.0, 55.0, 90.0, 120.0, 179.0]))
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        spp = int(rng.choice([1, 2, 3, 16, 64]))
        jitter = bool(rng.random() < 0.6)
        sd = int(rng.integers(0, 2 ** 31))
        y0 = int(rng.integers(0, H))
        y1 = int(rng.integers(y0, H + 1))
        want = oracle.eye_rays(oracle.make_camera(eye, look, up, fov), W, H, spp=spp, jitter=jitter, seed=sd, y0=y0, y1=y1)
        d = torch.empty((max(1, (y1 - y0) * W * spp), 8), dtype=torch.float32, device="cuda")
        n = s.gen_eye_rays(binding.make_camera(eye, look, up, fov), W, H, d, y0=y0, y1=y1, spp=spp, jitter=jitter, seed=sd)
        assert n == len(want)
        got = d[:n].cpu().numpy()
        assert _same_bits_or_nan(got, want.view(np.float32).reshape(-1, 8)), (eye, look, up, fov, W, H, spp, jitter, sd, y0, y1)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_scene_fused_frame_equals_batched_pipeline(miro, seed):
    """mr_render_direct (one launch, rays in registers) against the batched pipeline on the fuzz scenes -- grid soups with
    exact ties, spheres of radius 0, planes, leaf sizes 1-8 -- under random cameras, frame sizes, sample counts, sample
    orders, arithmetic modes and band splits: primary records, the shadow record of every sample, ray counts and the float
    framebuffer must be the same bits (NaN pixels, which zero-length normals produce, compared by position)."""
    import torch
    from miro_amd import frame as mframe
    rng = np.random.default_rng(BASE + seed + 7919)
    steps, leaf, extent = make_scene(rng)
    sc = replay(miro.Scene(), steps, leaf)
    for _ in range(3):
        eye = ((rng.random(3) - 0.5) * 4 * extent).astype(np.float32)
        desc = dict(eye=[float(x) for x in eye], lookat=[float(x) for x in (rng.random(3) - 0.5) * extent], up=[0.0, 1.0, 0.0],
                    fov=float(rng.choice([20.0, 45.0, 90.0, 140.0])), light=[float(x) for x in (rng.random(3) - 0.5) * 3 * extent],
                    wattage=float(rng.choice([10.0, 700.0])))
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        spp = int(rng.choice([1, 2, 4, 16, 64]))
        tiled = bool(rng.random() < 0.6)
        flags = int(rng.choice([0, miro.MR_MATH_PRODUCT, miro.MR_TRACE_INCOHERENT]))
        ref = mframe.FrameRenderer(sc, desc, W, H, spp=spp, flags=flags, tiled=tiled)
        ref.generate()
        ref.step()
        fu = mframe.FusedFrame(sc, desc, W, H, spp=spp, flags=flags, tiled=tiled, keep_hits=True)
        fu.step()
        torch.cuda.synchronize()
        n_p, n_s = ref.ray_counts()
        assert fu.ray_counts() == (n_p, n_s)
        assert torch.equal(fu.d_hits.view(torch.int32), ref.d_hits.view(torch.int32))
        src = ref.d_src[:n_s].to(torch.int64)
        assert torch.equal(fu.d_shadow_hits[src].view(torch.int32), ref.d_shadow_hits[:n_s].view(torch.int32))
        assert _same_bits_or_nan(fu.d_rgb.cpu().numpy(), ref.d_rgb.cpu().numpy())
        # the same frame as two ranks' interleaved bands
        band = int(rng.integers(1, 9))
        full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        for r in range(2):
            part = mframe.FusedFrame(sc, desc, W, H, spp=spp, flags=flags, tiled=tiled, band=band, rank=r, world=2)
            part.step()
            rows = torch.from_numpy(mframe.rows_of(mframe.band_rows(H, band, r, 2))).to("cuda")
            if len(rows):
                full[rows] = part.d_rgb.view(len(rows), W, 3)
        torch.cuda.synchronize()
        assert _same_bits_or_nan(full.view(-1, 3).cpu().numpy(), ref.d_rgb.cpu().numpy())


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_scene_level_equals_batched_calls(miro, seed):
    """mr_trace_level (one launch per level of traceScene's recursion) against the batched calls on the same random scenes
    with random materials -- mirrors, glass, glossy lobes, zero-area triangles (NaN normals), spheres, planes: two levels,
    either build of the generators, children compared as sets bit for bit, equal ray counts, pixel sums within the order of
    the float atomics."""
    import torch
    from miro_amd import binding
    from miro_amd import frame as mframe
    from test_level import batched_level, canon, fused_level
    rng = np.random.default_rng(BASE + 500000 + seed)
    steps, leaf, extent = make_scene(rng)
    sc = replay(miro.Scene(), steps, leaf)
    n_tri = sum(len(st[3]) for st in steps if st[0] == "mesh") + sum(1 for st in steps if st[0] == "sphere")
    n_mat = int(rng.integers(1, 6))
    mats = []
    for _ in range(n_mat):
        kd = tuple(float(x) for x in rng.random(3))
        ks = tuple(float(x) for x in rng.random(3) * rng.choice([0.0, 0.6])) 
        kt = tuple(float(x) for x in rng.random(3) * rng.choice([0.0, 0.9]))
        mats.append((kd, ks, kt, float(rng.choice([1.0, 5.0, 200.0, float("inf")])), float(rng.choice([1.0, 1.33, 1.5]))))
    sc.set_materials(mats, rng.integers(0, n_mat, n_tri).astype(np.uint32))
    eye = ((rng.random(3) - 0.5) * 4 * extent).astype(np.float32)
    desc = dict(eye=[float(x) for x in eye], lookat=[float(x) for x in (rng.random(3) - 0.5) * extent], up=[0.0, 1.0, 0.0],
                fov=float(rng.choice([45.0, 90.0])), light=[float(x) for x in (rng.random(3) - 0.5) * 3 * extent], wattage=700.0)
    W, H, spp = int(rng.integers(8, 70)), int(rng.integers(8, 60)), int(rng.choice([1, 2, 4]))
    fr = mframe.FrameRenderer(sc, desc, W, H, spp=spp, tiled=bool(rng.random() < 0.5))
    fr.generate()
    path = bool(rng.random() < 0.5)
    children = binding.MR_LEVEL_PATH if path else binding.MR_LEVEL_SPECULAR
    kinds = int(rng.integers(1, 8))
    flags0 = int(rng.choice([0, miro.MR_MATH_PRODUCT]))
    rays, weights, pixels, ids, n = fr.d_rays, None, None, None, fr.n
    L, Wt = desc["light"], desc["wattage"]
    # (tiled order: ray k belongs to pixel SLOT k // spp, which is all the level calls need)
    for level in range(2):
        fl = flags0 | (miro.MR_TRACE_INCOHERENT if level else 0)
        rgb_b, ns_b, out_b = batched_level(torch, sc, rays, weights, pixels, ids, n, L, Wt, spp, fl, children, level, 31, kinds)
        rgb_f, ns_f, out_f = fused_level(torch, sc, rays, weights, pixels, ids, n, L, Wt, spp, fl, children, level, 31, kinds, rgb_b.shape[0])
        assert ns_b == ns_f
        a, b = rgb_b.cpu().numpy(), rgb_f.cpu().numpy()
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.array_equal(np.isfinite(a), np.isfinite(b))
        scale = float(np.abs(a[fin]).max()) if fin.any() else 0.0
        assert np.allclose(a[fin], b[fin], rtol=1e-4, atol=1e-5 * scale)
        with_ids = children == binding.MR_LEVEL_PATH
        qa = canon(*[o.cpu().numpy() for o in (out_b if with_ids else out_b[:3])])
        qb = canon(*[o.cpu().numpy() for o in (out_f if with_ids else out_f[:3])])
        assert np.array_equal(qa, qb)
        n = len(out_b[0])
        if n == 0:
            break
        rays, weights, pixels = out_b[0].contiguous(), out_b[1].contiguous(), out_b[2].contiguous()
        ids = out_b[3].contiguous() if with_ids else None
