"""The PATH_TRACING build of the secondary-ray generators (Ray::random, Ray::reflect / ::refract under PATH_TRACING,
Ray.h:124-158,235-239): mr_gen_path_rays against the oracle's restatement (oracle/miro_oracle_path.c).

frand() is replaced on both sides by the counter-based generator of the eye-ray jitter, the transcendentals by
include/miro_math.h, so the generated ray sets have to be the same bits, and so do the hit records of tracing them.
The diffuse bounce (Ray::random as a child of every diffuse hit) is an EXTENSION of the reference's recursion:
Scene::traceScene at HEAD never calls Ray::random."""
import numpy as np
import pytest

from helpers import camera_of, oracle_scene, product_scene
from miro_amd import binding
from miro_amd import scenes


def test_miro_math_matches_correctly_rounded_double(oracle):
    """CPU: the shared transcendentals (include/miro_math.h) against numpy's double functions rounded to float -- the
    float result must be that value (what a correctly rounded libm float function returns) on every sampled argument."""
    rng = np.random.default_rng(168)
    u = (rng.integers(0, 1 << 24, 400000).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)
    u[:4] = [0.0, 1.0 - 2.0 ** -24, 0.5, 0.75]
    th = (np.float32(2.0) * np.float32(np.pi)) * u
    y = (np.float32(1.0) / (np.float32(1.0) + rng.integers(0, 200, len(u)).astype(np.float32))).astype(np.float32)
    got_th = oracle.miro_math(th, y)
    got_u = oracle.miro_math(u, y)
    d = np.float64
    assert np.array_equal(got_th[:, 0], np.sin(th.astype(d)).astype(np.float32))
    assert np.array_equal(got_th[:, 1], np.cos(th.astype(d)).astype(np.float32))
    assert np.array_equal(got_u[:, 2], np.arcsin(u.astype(d)).astype(np.float32))
    assert np.array_equal(got_u[:, 3], np.arccos(u.astype(d)).astype(np.float32))
    want_pow = np.power(u.astype(d), y.astype(d)).astype(np.float32)
    assert np.array_equal(got_u[:, 4], want_pow)
    # the corners the generators rely on: pow(x, 0) = 1 (infinite shininess -> phi = 0), pow(0, y) = 0
    edge = oracle.miro_math(np.array([0.3, 0.0, 1.0], np.float32), np.array([0.0, 0.5, 0.25], np.float32))
    assert edge[0, 4] == 1.0 and edge[1, 4] == 0.0 and edge[2, 4] == 1.0 and edge[2, 3] == 0.0


def _canon(rays, w, pix, ids):
    """order-free form of a child batch: sort by (id, ray bytes)"""
    key = np.lexsort((rays.view(np.uint32).reshape(len(rays), 8)[:, 4], ids))
    return rays[key], w[key], pix[key], ids[key]


def _device_children(miro, sc, d_rays, d_hits, n, spp, seed, bounce, kinds, d_w=None, d_pix=None, d_ids=None):
    import torch
    out = torch.empty((4 * n, 8), dtype=torch.float32, device="cuda")
    ow = torch.empty((4 * n, 3), dtype=torch.float32, device="cuda")
    op = torch.empty(4 * n, dtype=torch.int32, device="cuda")
    oi = torch.empty(4 * n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    sc.gen_path_rays(d_rays, d_hits, d_w, d_pix, d_ids, n, out, ow, op, oi, cnt, spp=spp, seed=seed, bounce=bounce, kinds=kinds)
    torch.cuda.synchronize()
    m = int(cnt.item())
    return (out[:m].cpu().numpy().view(miro.RAY_DTYPE).reshape(-1), ow[:m].cpu().numpy(), op[:m].cpu().numpy().view(np.uint32),
            oi[:m].cpu().numpy().view(np.uint32), out, ow, op, oi, m)


@pytest.mark.gpu
def test_specular_lobes_match_oracle(oracle, miro):
    """Glossy mirror + glass spheres (A1makeSphereScene-like materials on the spiral scene): mirror, Fresnel and refraction
    children with finite shininess, two bounces deep, ids propagated."""
    import torch
    name, W, H, spp = "spiral", 96, 64, 2
    a, b = oracle_scene(oracle, name), product_scene(miro, name)
    d = scenes.SCENES[name]
    nobj = b.info().n_triangles
    mats = [((0.6, 0.6, 0.6), (0, 0, 0), (0, 0, 0), 1.0, 1.0),                # Lambert
            ((0.1, 0.1, 0.1), (0.8, 0.8, 0.8), (0, 0, 0), 40.0, 1.0),         # glossy mirror
            ((0.0, 0.0, 0.0), (0.1, 0.1, 0.1), (0.9, 0.9, 0.9), 200.0, 1.5),  # rough glass
            ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0, 0, 0), float("inf"), 1.0)]  # perfect mirror: the lobe collapses
    pm = (np.arange(nobj) % 4).astype(np.uint32)
    b.set_materials(mats, pm)
    # the clamped table the kernels use (Phong's constructor, Phong.cpp:12-33)
    def clamp(m):
        kd, ks, kt, sh, ix = m
        ks = np.array(ks, np.float32); kt = np.minimum(np.array(kt, np.float32), 1 - ks).clip(0); kd = np.minimum(np.array(kd, np.float32), 1 - ks - kt).clip(0)
        return np.concatenate([kd, ks, kt, [sh, ix]]).astype(np.float32)
    table = np.stack([clamp(m) for m in mats])
    cam = camera_of(oracle, name)
    rays = oracle.eye_rays(cam, W, H, spp=spp, jitter=True, seed=168)
    hits = a.trace(rays)
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8).copy()).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.trace_device(d_rays, n, d_hits)
    assert d_hits.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1).tobytes() == hits.tobytes()
    # bounce 0
    want = a.path_rays(table, pm, rays, hits, spp=spp, seed=7, bounce=0, kinds=7)
    got = _device_children(miro, b, d_rays, d_hits, n, spp, 7, 0, 7)
    assert got[8] == len(want[0]) > n // 4
    wr, ww, wp, wi = _canon(*want[:4])
    gr, gw, gp, gi = _canon(*got[:4])
    assert wr.tobytes() == gr.tobytes() and np.array_equal(wi, gi) and np.array_equal(wp, gp)
    assert np.array_equal(ww.view(np.uint32), gw.view(np.uint32))
    assert set(np.unique(want[4])) == {0, 1, 2, 3}                      # every kind of child occurs
    # bounce 1: trace the children on both sides, generate theirs from the propagated weights / pixels / ids
    m = got[8]
    d_h1 = torch.empty((m, 4), dtype=torch.float32, device="cuda")
    b.trace_device(got[4], m, d_h1)
    h1 = a.trace(want[0])
    want2 = a.path_rays(table, pm, want[0], h1, weights=want[1], pixels=want[2], ids=want[3], spp=spp, seed=7, bounce=1, kinds=3)
    got2 = _device_children(miro, b, got[4], d_h1, m, spp, 7, 1, 3, d_w=got[5], d_pix=got[6], d_ids=got[7])
    assert got2[8] == len(want2[0]) > 0
    wr, ww, wp, wi = _canon(*want2[:4])
    gr, gw, gp, gi = _canon(*got2[:4])
    assert wr.tobytes() == gr.tobytes() and np.array_equal(wi, gi) and np.array_equal(wp, gp)
    assert np.array_equal(ww.view(np.uint32), gw.view(np.uint32))


@pytest.mark.gpu
def test_bunny_config3_with_one_diffuse_bounce(oracle, miro):
    """BASELINE config 3 read as a path trace (EXTENSION, see the module docstring): bunny 1024x1024 x 16 spp, primary rays
    -> hits -> one cosine-weighted bounce per diffuse hit (Ray::random) -> hits.  16.8 M primary + 12 M bounce rays run on
    the device; the oracle checks every ray and hit record of a 64-row band (rows 480..544, all 16 samples), bit for
    bit; the whole frame is covered by size-independent properties."""
    import torch
    name, W, H, spp, seed = "bunny", 1024, 1024, 16, 168
    a, b = oracle_scene(oracle, name), product_scene(miro, name)
    cam_o, dsc = camera_of(oracle, name), scenes.SCENES[name]
    cam_d = binding.make_camera(dsc["eye"], dsc["lookat"], dsc["up"], dsc["fov"])
    n = W * H * spp
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.gen_eye_rays(cam_d, W, H, d_rays, spp=spp, jitter=True, seed=seed)
    b.trace_device(d_rays, n, d_hits)
    out = torch.empty((n, 8), dtype=torch.float32, device="cuda")            # one child per ray at most (kinds = diffuse)
    ow = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    op = torch.empty(n, dtype=torch.int32, device="cuda")
    oi = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    b.gen_path_rays(d_rays, d_hits, None, None, None, n, out, ow, op, oi, cnt, spp=spp, seed=seed, bounce=0,
                    kinds=binding.MR_PATH_DIFFUSE)
    torch.cuda.synchronize()
    m = int(cnt.item())
    n_hit = int((d_hits[:, 1].view(torch.int32) != -1).sum().item())
    assert m == n_hit and 0.3 * n < m < n                                     # every hit is Lambert: one bounce ray each
    d_h1 = torch.empty((m, 4), dtype=torch.float32, device="cuda")
    b.trace_device(out, m, d_h1)
    torch.cuda.synchronize()
    # whole frame: unit directions, origins epsilon along them from the surface, the weight is the material's diffuse colour
    dirs = out[:m, 4:7]
    assert float((dirs.norm(dim=1) - 1).abs().max()) < 1e-6
    assert bool((ow[:m] == 1.0).all()) and bool((out[:m, 3] == 0).all()) and bool((out[:m, 7] == 1e12).all())
    pix = op[:m].to(torch.int64)
    assert int(pix.min()) >= 0 and int(pix.max()) < W * H
    # the band the oracle re-does: rows [y0, y1)
    y0, y1 = 480, 544
    rays_o = oracle.eye_rays(cam_o, W, H, spp=spp, jitter=True, seed=seed, y0=y0, y1=y1)
    lo, hi = y0 * W * spp, y1 * W * spp
    assert d_rays[lo:hi].cpu().numpy().view(miro.RAY_DTYPE).reshape(-1).tobytes() == rays_o.tobytes()
    hits_o = a.trace(rays_o)
    assert d_hits[lo:hi].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1).tobytes() == hits_o.tobytes()
    white = np.array([[1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1]], np.float32)
    ids_band = np.arange(lo, hi, dtype=np.uint32)                             # the frame's ray indices are the ids
    pix_band = (ids_band // spp).astype(np.uint32)
    cr, cw, cp, ci, ck = a.path_rays(white, None, rays_o, hits_o, pixels=pix_band, ids=ids_band, spp=spp, seed=seed, bounce=0, kinds=4)
    assert (ck == 3).all()
    band = ((pix >= y0 * W) & (pix < y1 * W)).nonzero().squeeze(1)
    g_r = out[band].cpu().numpy().view(miro.RAY_DTYPE).reshape(-1)
    g_i, g_p = oi[band].cpu().numpy().view(np.uint32), op[band].cpu().numpy().view(np.uint32)
    g_h = d_h1[band].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    assert len(g_r) == len(cr) > 100000
    ko, kg = np.argsort(ci, kind="stable"), np.argsort(g_i, kind="stable")
    assert len(np.unique(ci)) == len(ci)                                      # child ids are distinct
    assert np.array_equal(ci[ko], g_i[kg]) and np.array_equal(cp[ko], g_p[kg])
    assert cr[ko].tobytes() == g_r[kg].tobytes()                              # the bounce rays: same bits
    h_o = a.trace(cr)
    assert h_o[ko].tobytes() == g_h[kg].tobytes()                             # and their hit records
    assert 0.02 < (h_o["prim"] != oracle.MISS).mean() < 1.0                  # open scene: most bounce rays leave it


def _one_glass_triangle(oracle, miro):
    """a triangle in the plane z = 0 with vertex normals (0, 0, 1), material: rough glass"""
    verts = np.array([-4, -4, 0, 4, -4, 0, 0, 4, 0], np.float32)
    a, b = oracle.Scene(), miro.Scene()
    for s in (a, b):
        s.add_triangle(verts, [0, 0, 1] * 3)
    mats = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.9, 0.9, 0.9), 50.0, 1.5)]
    table = np.array([[0, 0, 0, 0, 0, 0, 0.9, 0.9, 0.9, 50.0, 1.5]], np.float32)
    a.build(4)
    b.set_materials(mats, np.zeros(1, np.uint32))
    b.build(4)
    return a, b, table


@pytest.mark.gpu
def test_incidence_cosine_above_one_gives_the_references_nan(oracle, miro):
    """Ray::getReflectionCoefficient takes acos(dot(-d, n)) (Ray.h:176-188); when that dot product exceeds 1 by
    rounding -- or, as here, because the caller's direction is a hair longer than 1 -- libm's acosf returns NaN, the Fresnel
    term Rs is NaN, `Rs > 0.01` is false (no Fresnel reflection child) and the refracted child carries the weight
    kt * (1 - NaN) = NaN into the pixel.  mm_acosf reproduces that on both sides (no clamp: the reference has none): the
    children of such a hit are the same on the device and in the oracle, NaN weights included; an in-range neighbour is
    finite."""
    import torch
    a, b, table = _one_glass_triangle(oracle, miro)
    up = np.float32(1.0) + np.float32(2.0 ** -23)
    rays = np.zeros(3, miro.RAY_DTYPE)
    rays["oz"], rays["tmax"] = 1.0, 1e12
    rays["dz"] = [-up, -1.0, -np.float32(0.8)]
    rays["dx"][2] = 0.6
    hits = a.trace(rays)
    assert (hits["prim"] == 0).all()
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8).copy()).cuda()
    d_hits = torch.from_numpy(hits.view(np.float32).reshape(n, 4).copy()).cuda()
    want = a.path_rays(table, np.zeros(1, np.uint32), rays, hits, spp=1, seed=5, bounce=0, kinds=7)
    got = _device_children(miro, b, d_rays, d_hits, n, 1, 5, 0, 7)
    assert got[8] == len(want[0])
    wr, ww, wp, wi = _canon(*want[:4])
    gr, gw, gp, gi = _canon(*got[:4])
    assert np.array_equal(wi, gi) and np.array_equal(wp, gp)
    assert np.array_equal(np.isnan(ww), np.isnan(gw)) and np.array_equal(ww[~np.isnan(ww)], gw[~np.isnan(gw)])
    kids = {int(p): ww[wp == p] for p in range(3)}
    assert len(kids[0]) == 1 and np.isnan(kids[0]).all()          # cos > 1: no Fresnel child, a NaN-weighted refraction
    assert len(kids[1]) >= 1 and np.isfinite(kids[1]).all()       # cos == 1 exactly: acos = 0, everything finite
    assert len(kids[2]) >= 1 and np.isfinite(kids[2]).all()
    # the directions of the finite children are the oracle's bits
    fin = ~np.isnan(ww).any(axis=1)
    assert wr[fin].tobytes() == gr[fin].tobytes()


@pytest.mark.gpu
def test_child_queues_are_bounded_by_out_capacity(miro):
    """the generators used to store at whatever slot the atomic counter handed out.  With out_capacity the
    children beyond the queue's room are counted, not stored: the call reports how many there are and writes nothing out of
    bounds -- mr_gen_path_rays, mr_gen_secondary_rays and mr_trace_level alike; a capacity of 0 is refused."""
    import torch
    from miro_amd import frame as mframe
    name, W, H, spp = "bunny", 128, 96, 2
    b = product_scene(miro, name)
    d = scenes.SCENES[name]
    n = W * H * spp
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.gen_eye_rays(binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), W, H, d_rays, spp=spp, jitter=True)
    b.trace_device(d_rays, n, d_hits)
    n_hit = int((d_hits[:, 1].view(torch.int32) != -1).sum().item())
    cap = 1000
    assert n_hit > 4 * cap
    sentinel = 12345.0

    def queues():
        return (torch.full((n, 8), sentinel, device="cuda"), torch.full((n, 3), sentinel, device="cuda"),
                torch.full((n,), 77, dtype=torch.int32, device="cuda"), torch.full((n,), 77, dtype=torch.int32, device="cuda"),
                torch.zeros(1, dtype=torch.int64, device="cuda"))

    def check(q):
        torch.cuda.synchronize()
        assert int(q[4].item()) == n_hit                                  # every child counted
        assert bool((q[0][cap:] == sentinel).all()) and bool((q[1][cap:] == sentinel).all())
        assert bool((q[2][cap:] == 77).all()) and bool((q[3][cap:] == 77).all())
        assert bool((q[0][:cap, 7] == 1e12).all())                       # ... and the first `cap` slots hold rays

    q = queues()
    b.gen_path_rays(d_rays, d_hits, None, None, None, n, q[0], q[1], q[2], q[3], q[4], spp=spp, kinds=binding.MR_PATH_DIFFUSE,
                    out_capacity=cap)
    check(q)
    q = queues()
    rgb = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
    b.trace_level(d_rays, None, None, None, n, rgb, d["light"], d["wattage"], children=binding.MR_LEVEL_PATH, d_out_rays=q[0],
                  d_out_weights=q[1], d_out_pixels=q[2], d_out_ids=q[3], d_out_count=q[4], spp=spp, kinds=binding.MR_PATH_DIFFUSE,
                  out_capacity=cap)
    check(q)
    with pytest.raises(miro.MiroError):
        b.gen_path_rays(d_rays, d_hits, None, None, None, n, q[0], q[1], q[2], q[3], q[4], spp=spp, out_capacity=0)
    with pytest.raises(miro.MiroError):
        b.trace_level(d_rays, None, None, None, n, rgb, d["light"], d["wattage"], children=binding.MR_LEVEL_SPECULAR, d_out_rays=q[0],
                      d_out_weights=q[1], d_out_pixels=q[2], d_out_count=q[4], spp=spp, out_capacity=0)
    with pytest.raises(miro.MiroError):
        b.gen_secondary_rays(d_rays, d_hits, None, None, n, q[0], q[1], q[2], q[4], spp=spp, out_capacity=0)


@pytest.mark.gpu
def test_octant_bytes_and_ordered_levels(miro):
    """The generators' d_out_octants are the sign bits of the children they wrote (batched generators and mr_trace_level
    alike); mr_order_by_octant makes the same index from those bytes as from the rays; mr_trace_grouped fed with the bytes
    returns mr_trace's hit buffer; and a level worked through that index (mr_level_desc.d_order) emits the same children --
    compared as sets by ray id -- and the same pixel sums up to the order of the float atomics."""
    import torch
    name, W, H, spp = "bunny", 160, 120, 4
    b = product_scene(miro, name)
    d = scenes.SCENES[name]
    n = W * H * spp
    i32 = dict(dtype=torch.int32, device="cuda")
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.gen_eye_rays(binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), W, H, d_rays, spp=spp, jitter=True)
    b.trace_device(d_rays, n, d_hits)

    def queue():
        return dict(rays=torch.zeros((n, 8), dtype=torch.float32, device="cuda"), w=torch.zeros((n, 3), dtype=torch.float32, device="cuda"),
                    pix=torch.zeros(n, **i32), ids=torch.zeros(n, **i32), oct=torch.full((n,), 255, dtype=torch.uint8, device="cuda"),
                    cnt=torch.zeros(1, dtype=torch.int64, device="cuda"))

    def signs(q, m):
        r = q["rays"][:m]
        return ((r[:, 4] < 0).to(torch.uint8) | ((r[:, 5] < 0).to(torch.uint8) << 1) | ((r[:, 6] < 0).to(torch.uint8) << 2))

    # ---- batched generator
    q = queue()
    b.gen_path_rays(d_rays, d_hits, None, None, None, n, q["rays"], q["w"], q["pix"], q["ids"], q["cnt"], spp=spp, seed=5,
                    kinds=binding.MR_PATH_DIFFUSE, d_out_octants=q["oct"])
    m = int(q["cnt"].item())
    assert m > 20000 and torch.equal(q["oct"][:m], signs(q, m)) and bool((q["oct"][m:] == 255).all())
    assert len(torch.unique(q["oct"][:m])) == 8
    # ---- the index from the bytes is the index from the rays
    o_rays, o_bytes = torch.full((m,), -1, **i32), torch.full((m,), -2, **i32)
    for lg in (8, 12, 0):
        b.order_by_octant(q["rays"], m, o_rays, chunk_log2=lg)
        b.order_by_octant(None, m, o_bytes, chunk_log2=lg, d_octants=q["oct"])
        assert torch.equal(o_rays, o_bytes)
    hits_a = torch.empty((m, 4), dtype=torch.float32, device="cuda")
    hits_b = torch.full((m, 4), 3.0, dtype=torch.float32, device="cuda")
    b.trace_device(q["rays"], m, hits_a)
    b.trace_grouped(q["rays"], m, hits_b, o_bytes, d_octants=q["oct"])
    assert torch.equal(hits_a.view(torch.int32), hits_b.view(torch.int32))
    # ---- one level of the bounce queue, as made and through the index
    res = []
    for order in (None, o_bytes):
        q2 = queue()
        rgb = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
        b.trace_level(q["rays"][:m], q["w"][:m], q["pix"][:m], q["ids"][:m], m, rgb, d["light"], d["wattage"],
                      children=binding.MR_LEVEL_PATH, d_out_rays=q2["rays"], d_out_weights=q2["w"], d_out_pixels=q2["pix"],
                      d_out_ids=q2["ids"], d_out_count=q2["cnt"], spp=spp, seed=5, bounce=1, kinds=binding.MR_PATH_DIFFUSE,
                      d_out_octants=q2["oct"], d_order=order)
        m2 = int(q2["cnt"].item())
        assert torch.equal(q2["oct"][:m2], signs(q2, m2))
        by_id = torch.argsort(((q2["ids"][:m2].to(torch.int64) & 0xFFFFFFFF) << 32) | (q2["rays"][:m2, 0].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF))
        res.append((m2, q2["ids"][:m2][by_id], q2["rays"][:m2][by_id], q2["w"][:m2][by_id], q2["pix"][:m2][by_id], rgb))
    assert res[0][0] == res[1][0] > 100
    for x, y in zip(res[0][1:5], res[1][1:5]):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    scale = float(res[0][5].abs().max())
    assert scale > 0 and float((res[0][5] - res[1][5]).abs().max()) <= 1e-5 * scale
