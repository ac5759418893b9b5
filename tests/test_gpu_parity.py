"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on identical
seeded ray sets.  Bar: the default ("exact") mode is bit-exact in t / prim / beta / gamma and reproduces the
reference's -DSTATS counters; MR_MATH_FAST is within the 1e-5 relative tolerance of BASELINE.json.
/root/reference is not needed at run time: scenes come from tests/golden/models and the seeded atrium."""
import json
import os

import numpy as np
import pytest

from helpers import (assert_hits_bit_exact, assert_hits_close, camera_of, oracle_scene, product_scene, random_rays)
from miro_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


_cache = {}


def both(oracle, miro, name, leaf=4):
    key = (name, leaf)
    if key not in _cache:
        _cache[key] = (oracle_scene(oracle, name, leaf), product_scene(miro, name, leaf))
    return _cache[key]


def scene_box(s):
    v = s.arrays()[0]
    return v.min(0), v.max(0)


# ----------------------------------------------------------------------------------------------- exact mode
@pytest.mark.parametrize("name,W,H", [("cornell", 256, 256), ("teapot", 512, 512), ("bunny", 256, 256),
                                      ("sponza", 256, 256), ("sphere", 64, 64), ("testobj", 64, 64)])
def test_primary_rays_bit_exact(oracle, miro, torch_cuda, name, W, H):
    """BASELINE configs 1-4 (reduced resolution where the oracle would take long): pixel-centre
    eye rays, closest hit."""
    a, b = both(oracle, miro, name)
    rays = oracle.eye_rays(camera_of(oracle, name), W, H)
    want = a.trace(rays)
    got = b.trace(rays.view(miro.RAY_DTYPE))
    assert_hits_bit_exact(got, want.view(miro.HIT_DTYPE))
    assert (want["prim"] != oracle.MISS).any()


@pytest.mark.parametrize("name", ["teapot", "bunny", "sponza"])
def test_quotient_and_product_slabs_bit_exact(oracle, miro, torch_cuda, name):
    """The default trace (slab distances as the reference's quotients, BVH.cpp:601-602, by the fma correction step) and
    MR_MATH_PRODUCT (products with the rounded 1/d): same hits as the oracle on camera, shadow and random rays,
    including closest/any agreement on hit or miss."""
    a, b = both(oracle, miro, name)
    rays = oracle.eye_rays(camera_of(oracle, name), 160, 120)
    hits = a.trace(rays)
    sh, _ = a.shadow_rays(rays, hits, scenes.SCENES[name]["light"])
    lo, hi = scene_box(a)
    rnd = random_rays(oracle.RAY_DTYPE, 20000, np.maximum(lo, -20), np.minimum(hi, 20), seed=21)
    allr = np.concatenate([rays, sh, rnd])
    want = a.trace(allr).view(miro.HIT_DTYPE)
    assert_hits_bit_exact(b.trace(allr.view(miro.RAY_DTYPE)), want)
    assert_hits_bit_exact(b.trace(allr.view(miro.RAY_DTYPE), flags=miro.MR_MATH_PRODUCT), want)
    for fl in (miro.MR_TRACE_ANY, miro.MR_MATH_PRODUCT | miro.MR_TRACE_ANY):
        anyh = b.trace(allr.view(miro.RAY_DTYPE), flags=fl)
        assert np.array_equal(anyh["prim"] == oracle.MISS, want["prim"] == oracle.MISS)


@pytest.mark.parametrize("name", ["teapot", "bunny", "sponza"])
def test_shadow_rays_bit_exact(oracle, miro, torch_cuda, name):
    """Phong shadow batch (finite tMax, origins on surfaces): closest hit as the reference traces them."""
    a, b = both(oracle, miro, name)
    rays = oracle.eye_rays(camera_of(oracle, name), 192, 192)
    hits = a.trace(rays)
    sh, _ = a.shadow_rays(rays, hits, scenes.SCENES[name]["light"])
    assert len(sh) > 1000
    assert_hits_bit_exact(b.trace(sh.view(miro.RAY_DTYPE)), a.trace(sh).view(miro.HIT_DTYPE))


@pytest.mark.parametrize("name", ["cornell", "teapot", "bunny", "sponza"])
def test_incoherent_rays_bit_exact(oracle, miro, torch_cuda, name):
    """Random origins and directions (divergent traversal, deep stacks, rays starting inside boxes)."""
    a, b = both(oracle, miro, name)
    lo, hi = scene_box(a)
    lo, hi = np.maximum(lo, -20), np.minimum(hi, 20)
    rays = random_rays(oracle.RAY_DTYPE, 20000, lo, hi, seed=11)
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), a.trace(rays).view(miro.HIT_DTYPE))


def test_t_interval_and_degenerate_rays(oracle, miro, torch_cuda):
    """tMin is inclusive at the triangle (Triangle.cpp:158); a hit at exactly t == tMax passes the triangle
    test but loses the strict-less comparison against minHit.t = tMax (BVH.cpp:477,500), so it is a miss;
    axis-parallel directions give
    +-inf / NaN slab distances (BVH.cpp:451-457); zero-length direction; rays behind the origin."""
    a, b = both(oracle, miro, "teapot")
    base = oracle.eye_rays(camera_of(oracle, "teapot"), 96, 96)
    ref = a.trace(base)
    hit = ref["prim"] != oracle.MISS
    cases = []
    r = base[hit].copy(); r["tmax"] = ref["t"][hit]; cases.append(r)                       # t == tMax: miss
    r = base[hit].copy(); r["tmax"] = np.nextafter(ref["t"][hit], np.float32(np.inf)); cases.append(r)   # one ulp more: hit
    r = base[hit].copy(); r["tmin"] = ref["t"][hit]; cases.append(r)                       # t == tMin accepted
    r = base[hit].copy(); r["tmin"] = np.nextafter(ref["t"][hit], np.float32(np.inf)); cases.append(r)
    r = base.copy(); r["tmax"] = 0.0; cases.append(r)                                      # empty interval
    r = base.copy(); r["dx"] *= -1; r["dy"] *= -1; r["dz"] *= -1; cases.append(r)          # pointing away
    ax = np.zeros(6 * 400, oracle.RAY_DTYPE)                                               # axis-parallel
    rng = np.random.RandomState(5)
    for k, d in enumerate(((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))):
        sl = slice(400 * k, 400 * (k + 1))
        o = (rng.rand(400, 3).astype(np.float32) - 0.5) * 8
        o[:, 1] = np.abs(o[:, 1])
        o[:50] = np.round(o[:50] * 2) / 2            # some origins on "nice" coordinates
        ax["ox"][sl], ax["oy"][sl], ax["oz"][sl] = (o[:, 0] - 6 * d[0]), (o[:, 1] - 6 * d[1]), (o[:, 2] - 6 * d[2])
        ax["dx"][sl], ax["dy"][sl], ax["dz"][sl] = d
        ax["tmax"][sl] = 1e12
    cases.append(ax)
    z = base[:64].copy(); z["dx"] = 0; z["dy"] = 0; z["dz"] = 0; cases.append(z)           # zero direction
    neg0 = ax.copy(); neg0["dx"] = np.where(neg0["dx"] == 0, np.float32(-0.0), neg0["dx"]); cases.append(neg0)
    for rays in cases:
        assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), a.trace(rays).view(miro.HIT_DTYPE))
        # (in the default trace, axis-parallel and zero directions are "irregular" rays -> the literal-division fallback)
        assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE), flags=miro.MR_MATH_PRODUCT), a.trace(rays).view(miro.HIT_DTYPE))
    # the bound cases really exercise both sides
    assert (a.trace(cases[0])["prim"] == oracle.MISS).all()
    assert (a.trace(cases[1])["prim"] != oracle.MISS).all()
    assert (a.trace(cases[2])["prim"] != oracle.MISS).mean() > 0.99
    assert (a.trace(cases[4])["prim"] == oracle.MISS).all()


def test_edge_and_vertex_hits(oracle, miro, torch_cuda):
    """Rays aimed exactly at shared edges and vertices of the mesh: the epsilon-slack barycentric test
    accepts both neighbours and the strict-less / first-found rule decides (BVH.cpp:500)."""
    a, b = both(oracle, miro, "sphere")
    v, _, vi, _ = a.arrays()
    eye = np.array([0.3, 0.2, 4.0], np.float32)
    targets = [v[vi[:, 0]], v[vi[:, 1]], 0.5 * (v[vi[:, 0]] + v[vi[:, 1]]), 0.5 * (v[vi[:, 1]] + v[vi[:, 2]]),
               (v[vi[:, 0]] + v[vi[:, 1]] + v[vi[:, 2]]) / 3]
    tg = np.concatenate(targets).astype(np.float32)
    d = tg - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(len(tg), oracle.RAY_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = eye
    rays["dx"], rays["dy"], rays["dz"] = d[:, 0], d[:, 1], d[:, 2]
    rays["tmax"] = 1e12
    want = a.trace(rays)
    assert (want["prim"] != oracle.MISS).mean() > 0.9
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), want.view(miro.HIT_DTYPE))


def test_empty_ragged_and_single_batches(oracle, miro, torch_cuda):
    a, b = both(oracle, miro, "teapot")
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), 70, 61)      # 4270 rays: not a multiple of 64 or 256
    want = a.trace(rays)
    assert len(b.trace(rays[:0].view(miro.RAY_DTYPE))) == 0
    for n in (1, 63, 64, 65, 255, 257, 4270):
        assert_hits_bit_exact(b.trace(rays[:n].view(miro.RAY_DTYPE)), want[:n].view(miro.HIT_DTYPE))


def test_degenerate_scenes(oracle, miro, torch_cuda):
    """Root is a leaf; empty scene; a depth-32 leaf holding 40 triangles (count escape of the leaf reference)."""
    rays = random_rays(oracle.RAY_DTYPE, 2000, (-1, -1, -1), (2, 2, 2), seed=3)
    one_v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    one_n = np.array([[0, 0, 1]] * 3, np.float32)
    f1 = np.array([[0, 1, 2]], np.uint32)
    many_v = np.tile(np.array([[0.25, 0.5, -1.0], [1.5, 0.125, 0.75], [-0.5, 2.0, 0.5]], np.float32), (40, 1))
    many_f = np.arange(120, dtype=np.uint32).reshape(40, 3)
    for v, n, f in ((None, None, None), (one_v, one_n, f1), (many_v, np.tile(one_n[:1], (120, 1)), many_f)):
        a, b = oracle.Scene(), miro.Scene()
        if v is not None:
            a.add_arrays(v, n, f, f)
            b.add_arrays(v, n, f, f)
        a.build(4)
        b.build(4)
        want = a.trace(rays)
        assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), want.view(miro.HIT_DTYPE))
        got, stats_want = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_COUNT_STATS), a.trace(rays, counters=True)[1]
        assert b.stats() == stats_want


@pytest.mark.parametrize("name", ["teapot", "sponza"])
def test_persistent_kernel_bit_exact(oracle, miro, torch_cuda, name):
    """MR_TRACE_PERSISTENT (resident waves, ballot/prefix re-arming of idle lanes) returns exactly what the
    one-ray-per-lane kernel returns: incoherent rays, a ragged count, closest and any-hit."""
    a, b = both(oracle, miro, name)
    lo, hi = scene_box(a)
    rays = random_rays(oracle.RAY_DTYPE, 50021, np.maximum(lo, -20), np.minimum(hi, 20), seed=31)
    want = a.trace(rays)
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE), miro.MR_TRACE_PERSISTENT), want.view(miro.HIT_DTYPE))
    for n in (1, 63, 65, 1023, 1025):
        assert_hits_bit_exact(b.trace(rays[:n].view(miro.RAY_DTYPE), miro.MR_TRACE_PERSISTENT), want[:n].view(miro.HIT_DTYPE))
    any_a = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_TRACE_ANY)
    any_b = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_TRACE_ANY | miro.MR_TRACE_PERSISTENT)
    assert_hits_bit_exact(any_b, any_a)


# ----------------------------------------------------------------------------------------------- counters
@pytest.mark.parametrize("name", ["teapot", "bunny"])
def test_stats_counters_match_reference(oracle, miro, torch_cuda, golden_dir, name):
    """-DSTATS on the device: the same node-visit and triangle-test counts as the reference's scalar
    build (BASELINE.md section 2 = 512x512 render + stride-8 probe grid, see kat_counters.json)."""
    k = json.load(open(os.path.join(golden_dir, "kat_counters.json")))["baseline"][name]
    a, b = both(oracle, miro, name)
    rays = oracle.eye_rays(camera_of(oracle, name), 512, 512)
    b.stats()
    hits = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_COUNT_STATS)
    primary = b.stats()
    assert int((hits["prim"] != miro.MISS).sum()) == k["primary_hits"]
    probe = rays.reshape(512, 512)[0::8, 0::8].reshape(-1).copy()
    b.trace(probe.view(miro.RAY_DTYPE), miro.MR_COUNT_STATS)
    pr = b.stats()
    assert (primary[0] + pr[0], primary[1] + pr[1]) == (k["no_shadows"]["box_tests"], k["no_shadows"]["tri_tests"])
    sh, _ = a.shadow_rays(rays, hits.view(oracle.HIT_DTYPE), scenes.SCENES[name]["light"])
    b.trace(sh.view(miro.RAY_DTYPE), miro.MR_COUNT_STATS)
    shc = b.stats()
    assert (primary[0] + pr[0] + shc[0], primary[1] + pr[1] + shc[1]) == (k["shadows"]["box_tests"], k["shadows"]["tri_tests"])


def test_published_sse_triangle_counts_belong_to_the_sse_traversal(oracle, miro, torch_cuda, golden_dir):
    """can the DEVICE reproduce the write-up's 892 848 / 1 817 141 ray-triangle tests (Readme.tex:95,99)?
    No, and this test pins why.  Those numbers come from the reference's SSE build: 8 triangles per leaf AND the packet
    traversal of BVH.cpp:513-584 (12-bit _mm_rcp_ps slab distances, strict comparisons, children culled against the
    running minHit.t, triangles counted per 4-wide packet).  The device implements the SCALAR traversal north_star names
    (BVH.cpp:587-651): on the very same 8-per-leaf tree it visits what the oracle's scalar traversal visits -- 987 237
    triangle tests for the teapot's primary rays, not 892 848 -- so the count is a property of the SSE control flow, which
    exists only as the timed CPU baseline (oracle/miro_oracle_sse.c; it reproduces both published numbers to the unit in
    tests/test_oracle_kat.py::test_writeup_table_teapot_sse).  What the device must and does reproduce is every counter of
    the scalar traversal, on this tree as on the 4-per-leaf one."""
    k = json.load(open(os.path.join(golden_dir, "kat_counters.json")))["writeup"]["teapot_sse"]
    name = "teapot"
    a = oracle_scene(oracle, name, 8)
    b = product_scene(miro, name, 8)
    assert (b.info().n_nodes, b.info().n_leaves) == (k["nodes"], k["leaves"])          # 199 / 100: the published tree
    rays = oracle.eye_rays(camera_of(oracle, name), 512, 512)
    want_hits, want = a.trace(rays, counters=True)
    b.stats()
    hits = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_COUNT_STATS)
    got = b.stats()
    assert hits.tobytes() == want_hits.tobytes()
    assert tuple(got) == tuple(want) == (1505697, 987237)
    # The SSE count is not even a property of the algorithm alone: _mm_rcp_ps is an approximation whose table differs between
    # CPU families -- this container's host reproduces the write-up's 892 848 to the unit (tests/test_oracle_kat.py, CPU
    # suite), the GPU box's EPYC counts 893 330 for the same rays.  Either way it is not the scalar traversal's count.
    _, _, sse = a.trace_sse(rays, threads=1, counters=True)
    assert k["no_shadows"]["tri_tests"] == 892848 and abs(sse[1] - 892848) < 2000 and sse[1] != got[1]


# ----------------------------------------------------------------------------------------------- fast mode
@pytest.mark.parametrize("name", ["cornell", "teapot", "bunny", "sponza"])
def test_fast_math_within_tolerance(oracle, miro, torch_cuda, name):
    a, b = both(oracle, miro, name)
    mesh = a.arrays()
    rays = oracle.eye_rays(camera_of(oracle, name), 256, 256)
    want = a.trace(rays)
    got = b.trace(rays.view(miro.RAY_DTYPE), miro.MR_MATH_FAST)
    def explain(got, want, flips):
        # every flipped ray must be explainable: the alternative is at (almost) the same distance (two
        # triangles sharing an edge both accept inside the epsilon slack), or the hit sat inside the
        # slack band of an edge / at the end of the interval and one side rejected it
        for i in flips:
            g, w = got[i], want[i]
            if g["prim"] != miro.MISS and w["prim"] != oracle.MISS:
                assert abs(g["t"] - w["t"]) <= 1e-4 * max(1.0, abs(w["t"]))
            else:
                h = w if w["prim"] != oracle.MISS else g
                assert min(h["beta"], h["gamma"], 1 - h["beta"] - h["gamma"]) < 1e-3

    explain(got, want, assert_hits_close(got, want.view(miro.HIT_DTYPE), mesh, rays))
    lo, hi = scene_box(a)
    rr = random_rays(oracle.RAY_DTYPE, 20000, np.maximum(lo, -20), np.minimum(hi, 20), seed=21)
    got, want = b.trace(rr.view(miro.RAY_DTYPE), miro.MR_MATH_FAST), a.trace(rr)
    explain(got, want, assert_hits_close(got, want.view(miro.HIT_DTYPE), mesh, rr))


# ----------------------------------------------------------------------------------------------- any-hit
def test_any_hit_agrees_on_occlusion(oracle, miro, torch_cuda):
    """MR_TRACE_ANY returns *a* valid hit inside [tMin,tMax] iff the closest-hit query hits."""
    a, b = both(oracle, miro, "bunny")
    rays = oracle.eye_rays(camera_of(oracle, "bunny"), 160, 160)
    hits = a.trace(rays)
    sh, _ = a.shadow_rays(rays, hits, scenes.SCENES["bunny"]["light"])
    want = a.trace(sh)
    got = b.trace(sh.view(miro.RAY_DTYPE), miro.MR_TRACE_ANY)
    assert np.array_equal(got["prim"] != miro.MISS, want["prim"] != oracle.MISS)
    occ = got["prim"] != miro.MISS
    assert occ.any() and (got["t"][occ] >= want["t"][occ]).all() and (got["t"][occ] <= sh["tmax"][occ]).all()
    # the reported hit is a genuine intersection of that primitive
    chk = np.nonzero(occ)[0][:200]
    for i in chk:
        one = sh[i:i + 1].copy()
        one["tmin"] = got["t"][i]
        one["tmax"] = np.nextafter(got["t"][i], np.float32(np.inf))     # t == tMax itself is a miss
        again = a.trace_brute(one)
        assert again["prim"][0] != oracle.MISS


# ----------------------------------------------------------------------------------------------- device-side callers
def test_device_buffers_and_streams(oracle, miro, torch_cuda):
    """Device-resident rays/hits on a side stream; result identical to the staged host path."""
    torch = torch_cuda
    a, b = both(oracle, miro, "teapot")
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), 200, 150)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        b.trace_device(d_rays, len(rays), d_hits, stream=st)
    st.synchronize()
    got = d_hits.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    assert_hits_bit_exact(got, a.trace(rays).view(miro.HIT_DTYPE))
    with pytest.raises(miro.MiroError):        # misaligned device pointer is refused, not faulted on
        b.trace_device(d_rays.data_ptr() + 4, 8, d_hits)


@pytest.mark.parametrize("name,spp,jitter", [("teapot", 1, False), ("bunny", 4, True), ("sponza", 2, True)])
def test_eye_ray_generator_bit_exact(oracle, miro, torch_cuda, name, spp, jitter):
    """Camera::eyeRay on the device == the restated Camera.cpp:104-161 (bitwise), incl. row windows."""
    torch = torch_cuda
    _, b = both(oracle, miro, name)
    W, H = 97, 64
    want = oracle.eye_rays(camera_of(oracle, name), W, H, spp=spp, jitter=jitter, seed=168)
    d = torch.empty((W * H * spp, 8), dtype=torch.float32, device="cuda")
    from miro_amd import binding
    cam = camera_of(binding, name)
    n = b.gen_eye_rays(cam, W, H, d, spp=spp, jitter=jitter, seed=168)
    assert n == len(want)
    got = d.cpu().numpy().view(miro.RAY_DTYPE).reshape(-1)
    assert got.tobytes() == want.tobytes()
    n2 = b.gen_eye_rays(cam, W, H, d, y0=10, y1=20, spp=spp, jitter=jitter, seed=168)
    got2 = d[:n2].cpu().numpy().view(miro.RAY_DTYPE).reshape(-1)
    assert got2.tobytes() == want[10 * W * spp:20 * W * spp].tobytes()


@pytest.mark.parametrize("name", ["teapot", "bunny", "sponza"])
def test_shadow_ray_generator_and_hit_attrs(oracle, miro, torch_cuda, name):
    """Phong shadow rays built on the device (ballot compaction): same set of rays, bitwise, as the
    restated Phong.cpp:80-97; P and N as Triangle.cpp:160,162."""
    torch = torch_cuda
    a, b = both(oracle, miro, name)
    rays = oracle.eye_rays(camera_of(oracle, name), 150, 130)
    hits = a.trace(rays)
    light = scenes.SCENES[name]["light"]
    want, src_want = a.shadow_rays(rays, hits, light)
    n = len(rays)
    d_hits = torch.from_numpy(hits.view(np.float32).reshape(-1, 4).copy()).cuda()
    d_out = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_src = torch.empty(n, dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    b.gen_shadow_rays(None, d_hits, n, light, d_out, d_src, d_cnt)
    k = int(d_cnt.item())
    assert k == len(want)
    src = d_src[:k].cpu().numpy().astype(np.int64)
    order = np.argsort(src, kind="stable")
    assert np.array_equal(src[order], src_want.astype(np.int64))
    got = d_out[:k].cpu().numpy().view(miro.RAY_DTYPE).reshape(-1)[order]
    assert got.tobytes() == want.tobytes()
    # compaction is wave-granular: within a wave the source order is preserved
    assert (np.diff(src.reshape(-1)) != 0).all()
    P = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    N = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    b.hit_attrs(d_hits, n, P, N)
    Pw, Nw = a.hit_attrs(hits)
    assert np.array_equal(P.cpu().numpy().view(np.uint32), Pw.view(np.uint32))
    assert np.array_equal(N.cpu().numpy().view(np.uint32), Nw.view(np.uint32))


# ----------------------------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize("name,W,H,spp,closed", [("sponza", 1920, 1080, 64, True), ("bunny", 1024, 1024, 16, False)])
def test_full_size_properties(oracle, miro, torch_cuda, name, W, H, spp, closed):
    """BASELINE-size batches (config 4: 1920x1080 64 spp = 132.7 M rays; config 3: bunny 1024x1024 16 spp), both in full,
    through size-independent properties:
    (1) any prefix / permutation of the batch gives the same per-ray hits (rays are independent);
    (2) re-tracing with tMax one ulp above t hits the same primitive at the same t (idempotence), except where the
        reference itself does not (checked against the oracle);
    (3) shortening tMax below t turns every hit into a miss or a nearer-than-before impossibility;
    (4) a sub-sample agrees bit-for-bit with the oracle."""
    torch = torch_cuda
    a, b = both(oracle, miro, name)
    from miro_amd import binding
    n = W * H * spp
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.gen_eye_rays(camera_of(binding, name), W, H, d_rays, spp=spp, jitter=True, seed=168)
    b.trace_device(d_rays, n, d_hits)
    torch.cuda.synchronize()
    hits_bits = d_hits.view(torch.int32)
    n_miss = int((hits_bits[:, 1] == -1).sum())
    assert (n_miss == 0) if closed else (0 < n_miss < n // 2)      # the atrium is closed; the bunny sits on a floor
    # (1) permutation
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    d_hits2 = torch.empty_like(d_hits)
    def take_rows(x, idx, chunk=1 << 24):
        # torch 2.10+rocm7.0 returns garbage past 2^26 result rows when gathering 16-byte rows in one call
        # (tools/torch_index_probe.py); index in chunks
        return torch.cat([x[idx[c:c + chunk]] for c in range(0, len(idx), chunk)])

    b.trace_device(take_rows(d_rays, perm), n, d_hits2)
    torch.cuda.synchronize()
    assert torch.equal(d_hits2.view(torch.int32), take_rows(hits_bits, perm))
    # (2) idempotence with tMax one ulp above t (t == tMax itself loses the strict-less test, BVH.cpp:500)
    r2 = d_rays.clone()
    was_hit = hits_bits[:, 1] != -1
    r2[:, 7] = torch.where(was_hit, torch.nextafter(d_hits[:, 0], torch.full_like(d_hits[:, 0], float("inf"))), d_rays[:, 7])
    b.trace_device(r2, n, d_hits2)
    torch.cuda.synchronize()
    # The reference's boxes are not conservative for every triangle (a leaf's hit point can lie outside an ancestor's
    # padded box, so a subtree that a huge tMax lets the ray enter is culled by `minOverlap > tMax`, BVH.cpp:609, once
    # tMax is near t): 3 of the bunny's 16.7 M rays turn into misses.  That is reference behaviour -- the rows that
    # change must be few, must be misses, and must be exactly what the oracle returns for the shortened ray.
    changed = (d_hits2.view(torch.int32) != hits_bits).any(dim=1).nonzero()[:, 0]
    assert len(changed) <= max(1, n // 1_000_000)
    if len(changed):
        assert bool((d_hits2.view(torch.int32)[changed, 1] == -1).all())
        sub = r2[changed].cpu().numpy().view(oracle.RAY_DTYPE).reshape(-1)
        assert_hits_bit_exact(d_hits2[changed].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1), a.trace(sub).view(miro.HIT_DTYPE))
    # This is also the one place where MR_MATH_PRODUCT's slab arithmetic -- (corner - o) * (1/d) with a correctly
    # rounded reciprocal, within 2 ulp of the reference's quotient -- can take a different decision than the default:
    # tMax now sits one ulp above t, and a box whose entry distance ties with it is culled or kept depending on that
    # last ulp (2 of the 132.7 M sponza rays).
    d_hits3 = torch.empty_like(d_hits)
    b.trace_device(r2, n, d_hits3, flags=binding.MR_MATH_PRODUCT)
    torch.cuda.synchronize()
    tie_flips = int((d_hits3.view(torch.int32) != d_hits2.view(torch.int32)).any(dim=1).sum())
    assert tie_flips <= max(1, n // 50_000_000)
    del d_hits3
    # (3) tMax = t: no hit can be the old one
    r2[:, 7] = d_hits[:, 0]
    b.trace_device(r2, n, d_hits2)
    torch.cuda.synchronize()
    still = d_hits2.view(torch.int32)[:, 1] != -1
    assert bool((d_hits2[:, 0][still] < d_hits[:, 0][still]).all())
    # (4) sub-sample against the oracle
    idx = torch.arange(0, n, 4099, device="cuda")
    sub = d_rays[idx].cpu().numpy().view(oracle.RAY_DTYPE).reshape(-1)
    want = a.trace(sub)
    got = d_hits[idx].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    assert_hits_bit_exact(got, want.view(miro.HIT_DTYPE))


@pytest.mark.parametrize("scale", [1e-15, 1e-9, 1.0, 1e12, 1e22])
def test_exact_quotients_across_magnitudes(oracle, miro, torch_cuda, scale):
    """The default trace takes the fma correction step only for operands well inside the normal range and divides like
    the reference otherwise: the same mesh at coordinates from 1e-15 to 1e22 (irregular nodes and rays at both ends), rays
    with direction components down to 1e-20, and the empty leaves of a degenerate tree all give the oracle's hits."""
    rng = np.random.RandomState(11)
    v0, _, vi, _ = both(oracle, miro, "sphere")[0].arrays()
    v = (v0.astype(np.float64) * scale).astype(np.float32)
    v = np.concatenate([v, np.tile(v[:3], (12, 1))])                        # 12 coincident triangles: a deep chain with empty leaves
    f = np.concatenate([vi, np.arange(len(v0), len(v0) + 36, dtype=np.uint32).reshape(-1, 3)])
    n = np.tile(np.asarray([[0, 0, 1]], np.float32), (len(v), 1))
    a, b = oracle.Scene(), miro.Scene()
    for s in (a, b):
        s.add_arrays(v, n, f, f)
        s.build(4)
    lo, hi = v.min(0), v.max(0)
    rays = random_rays(oracle.RAY_DTYPE, 6000, lo, hi, seed=4, tmax=np.float32(1e30))
    tiny = rng.rand(2000) < 0.5                                             # some direction components far below 2^-40
    rays["dx"][:2000] = np.where(tiny, np.float32(1e-20), rays["dx"][:2000])
    rays["dz"][1000:3000] *= np.float32(1e-14)
    want = a.trace(rays).view(miro.HIT_DTYPE)
    # (at 1e22 the cross products of Triangle.cpp:151 overflow: every test is inf/NaN and everything misses, on both sides)
    assert (want["prim"] != oracle.MISS).any() or scale > 1e15
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), want)
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE), flags=miro.MR_MATH_PRODUCT), want)


def test_concurrent_host_threads(oracle, miro, torch_cuda):
    """mr_trace from several host threads on one scene (the reference calls the const Scene::trace from every OpenMP
    worker, Scene.cpp:112-115): host-buffer calls take turns on the staging buffers, device-buffer calls on separate
    streams run side by side; every thread gets its own rays' hits."""
    import threading
    torch = torch_cuda
    a, b = both(oracle, miro, "teapot")
    sets = [random_rays(oracle.RAY_DTYPE, 30000 + 1000 * i, (-4, 0, -4), (4, 4, 4), seed=100 + i) for i in range(6)]
    want = [a.trace(r).view(miro.HIT_DTYPE) for r in sets]
    got, errs = [None] * len(sets), []

    def host_worker(i):
        try:
            for _ in range(3):
                got[i] = b.trace(sets[i].view(miro.RAY_DTYPE))
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    def device_worker(i):
        try:
            st = torch.cuda.Stream()
            d_r = torch.from_numpy(sets[i].view(np.float32).reshape(-1, 8).copy()).cuda()
            d_h = torch.empty((len(sets[i]), 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            for _ in range(3):
                b.trace_device(d_r, len(sets[i]), d_h, stream=st)
            st.synchronize()
            got[i] = d_h.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=host_worker if i % 2 == 0 else device_worker, args=(i,)) for i in range(len(sets))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for g, w in zip(got, want):
        assert_hits_bit_exact(g, w)


def test_degenerate_triangles_and_coplanar_rays(oracle, miro, torch_cuda):
    """Zero-area triangles (collinear / repeated vertices: n = 0, so t, beta, gamma are 0/0 or x/0), rays lying in a
    triangle's plane (ddotn = 0 -> +-inf / NaN quotients) and rays through vertices of a fan: every comparison of
    Triangle.cpp:158 falls the way IEEE arithmetic makes it fall, identically on both sides."""
    rng = np.random.RandomState(7)
    v = [(0, 0, 0), (1, 0, 0), (2, 0, 0),            # collinear
         (0, 1, 0), (0, 1, 0), (1, 1, 0),            # repeated vertex
         (3, 3, 3), (3, 3, 3), (3, 3, 3),            # a point
         (-1, -1, 0), (2, -1, 0), (0.5, 2, 0),       # a proper triangle in z = 0
         (0, 0, 1), (1, 0, 1), (0, 1, 1)]            # and one in z = 1
    v = np.asarray(v, np.float32)
    n = np.tile(np.asarray([[0, 0, 1]], np.float32), (len(v), 1))
    f = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
    a, b = oracle.Scene(), miro.Scene()
    for s in (a, b):
        s.add_arrays(v, n, f, f)
        s.build(4)
    rays = np.zeros(4000, oracle.RAY_DTYPE)
    o = (rng.rand(4000, 3).astype(np.float32) - 0.3) * 4
    d = rng.randn(4000, 3).astype(np.float32)
    o[:1000, 2] = 0.0; d[:1000, 2] = 0.0              # in the plane z = 0: ddotn = 0 for the big triangle
    o[1000:1500, 2] = 1.0; d[1000:1500, 2] = 0.0      # in the plane z = 1
    tgt = v[rng.randint(0, len(v), 500)]              # straight at vertices (of degenerate triangles too)
    d[1500:2000] = tgt - o[1500:2000]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30).astype(np.float32)
    rays["ox"], rays["oy"], rays["oz"] = o[:, 0], o[:, 1], o[:, 2]
    rays["dx"], rays["dy"], rays["dz"] = d[:, 0], d[:, 1], d[:, 2]
    rays["tmax"] = 1e12
    want, ctr = a.trace(rays, counters=True)
    assert (want["prim"] != oracle.MISS).any() and (want["prim"] == oracle.MISS).any()
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE)), want.view(miro.HIT_DTYPE))
    assert_hits_bit_exact(b.trace(rays.view(miro.RAY_DTYPE), flags=miro.MR_MATH_PRODUCT), want.view(miro.HIT_DTYPE))
    b.stats()
    b.trace(rays.view(miro.RAY_DTYPE), flags=miro.MR_COUNT_STATS)
    assert b.stats() == ctr


def test_large_host_batches_are_pipelined_in_chunks(oracle, miro, torch_cuda):
    """Host-pointer batches above 2^20 rays go through the chunked upload / trace / download pipeline: same hit
    records as the device-buffer path, from pageable and from page-locked (mr_host_alloc) buffers, ragged last chunk,
    counters accumulated over the chunks, persistent kernel per chunk."""
    torch = torch_cuda
    a, b = both(oracle, miro, "teapot")
    lo, hi = scene_box(a)
    n = 2 * (1 << 20) + 12345
    rays = random_rays(miro.RAY_DTYPE, n, np.maximum(lo, -20), np.minimum(hi, 20), seed=77)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b.trace_device(d_rays, n, d_hits)
    torch.cuda.synchronize()
    want = d_hits.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    sub = np.arange(0, n, 97)
    assert_hits_bit_exact(want[sub], a.trace(rays[sub].view(oracle.RAY_DTYPE)).view(miro.HIT_DTYPE))
    assert_hits_bit_exact(b.trace(rays), want)
    pr, ph = miro.PinnedArray(n, miro.RAY_DTYPE), miro.PinnedArray(n, miro.HIT_DTYPE)
    try:
        pr.array[:] = rays
        ph.array.view(np.uint32)[:] = 0xDEADBEEF
        b.trace(pr.array, hits=ph.array)
        assert_hits_bit_exact(ph.array, want)
        ph.array.view(np.uint32)[:] = 0xDEADBEEF
        b.trace(pr.array, flags=miro.MR_TRACE_PERSISTENT, hits=ph.array)
        assert_hits_bit_exact(ph.array, want)
        b.stats()
        b.trace_device(d_rays, n, d_hits, flags=miro.MR_COUNT_STATS)
        torch.cuda.synchronize()
        ctr = b.stats()
        assert_hits_bit_exact(b.trace(pr.array, flags=miro.MR_COUNT_STATS), want)
        assert b.stats() == ctr
    finally:
        pr.close(); ph.close()


def test_bunny20_published_totals(oracle, miro, torch_cuda):
    """makeBunny20Scene (1.39 M triangles, 876 137 nodes): the device finds the 233 358 primary hits behind the
    write-up's 495 502 total rays (Readme.tex:97), every hit record equal to the oracle's, and the -DSTATS counters of
    the primary and the shadow batch equal the oracle's."""
    a, b = both(oracle, miro, "bunny20")
    rays = oracle.eye_rays(camera_of(oracle, "bunny20"), 512, 512)
    want, ctr = a.trace(rays, counters=True)
    got = b.trace(rays.view(miro.RAY_DTYPE))
    assert_hits_bit_exact(got, want.view(miro.HIT_DTYPE))
    assert 262144 + int((got["prim"] != miro.MISS).sum()) == 495502
    b.stats()
    b.trace(rays.view(miro.RAY_DTYPE), flags=miro.MR_COUNT_STATS)
    assert b.stats() == ctr
    sh, _ = a.shadow_rays(rays, want, scenes.SCENES["bunny20"]["light"])
    want_s, ctr_s = a.trace(sh, counters=True)
    assert_hits_bit_exact(b.trace(sh.view(miro.RAY_DTYPE)), want_s.view(miro.HIT_DTYPE))
    b.trace(sh.view(miro.RAY_DTYPE), flags=miro.MR_COUNT_STATS)
    assert b.stats() == ctr_s


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [1, 2, 16, 17, 18])
def test_storage_layouts_give_identical_hit_records(miro, layout):
    """mr_build_opts.layout permutes WHERE node and triangle records live in HBM (pair-aligned pre-order, breadth-first top +
    treelets, leaves padded to fewer 128-byte lines); references, visiting order and therefore every hit record are those of
    the default order -- on camera rays and on random rays, default, voting and counting kernels, triangle and sphere scenes."""
    import torch
    from miro_amd import scenes as sc_mod
    for name in ("sponza", "spiral", "cornell"):
        d = sc_mod.SCENES[name]
        built = []
        for lay in (0, layout):
            s = miro.Scene(0)
            sc_mod.populate(s, d)
            s.build(4, layout=lay)
            built.append(s)
        W, H = 160, 90
        n = W * H
        cam = miro.binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
        rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        built[0].gen_eye_rays(cam, W, H, rays)
        v = built[0].arrays()[0]
        rnd = random_rays(miro.RAY_DTYPE, 20000, v.min(0), v.max(0), seed=3)
        d_rnd = torch.from_numpy(rnd.view(np.float32).reshape(-1, 8)).cuda()
        for batch, m in ((rays, n), (d_rnd, len(rnd))):
            for flags in (0, miro.MR_TRACE_INCOHERENT, miro.MR_COUNT_STATS, miro.MR_MATH_PRODUCT, miro.MR_TRACE_PERSISTENT):
                outs = []
                for s in built:
                    o = torch.empty((m, 4), dtype=torch.float32, device="cuda")
                    s.trace_device(batch, m, o, flags)
                    torch.cuda.synchronize()
                    outs.append(o.cpu().numpy().tobytes())
                    if flags == miro.MR_COUNT_STATS:
                        outs[-1] += repr(s.stats()).encode()
                assert outs[0] == outs[1], (name, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sponza", "spiral"])
def test_grouped_trace_returns_the_plain_traces_hit_buffer(miro, name):
    """mr_trace_grouped orders a batch by direction octant inside chunks (a permutation: every index exactly once, octants
    ascending inside a chunk, original order inside an octant) and traces through the index: the hit buffer is mr_trace's byte
    for byte -- ragged batch sizes, every chunk size, default / voting / product / any-hit kernels, triangles and spheres."""
    import torch
    from miro_amd import scenes as sc_mod
    s = product_scene(miro, name)
    v = s.arrays()[0]
    for n, lg in ((1, 8), (255, 8), (4097, 8), (100003, 11), (100003, 14), (70000, 0)):
        rnd = random_rays(miro.RAY_DTYPE, n, v.min(0), v.max(0), seed=n)
        rnd["dx"][::7] = 0.0                                         # zero components sit in the "non-negative" octants
        d_r = torch.from_numpy(rnd.view(np.float32).reshape(-1, 8).copy()).cuda()
        order = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        for flags in (0, miro.MR_TRACE_INCOHERENT, miro.MR_MATH_PRODUCT, miro.MR_TRACE_ANY):
            ref = torch.empty((n, 4), dtype=torch.float32, device="cuda")
            got = torch.full((n, 4), 7.0, dtype=torch.float32, device="cuda")
            s.trace_device(d_r, n, ref, flags)
            s.trace_grouped(d_r, n, got, order, flags, chunk_log2=lg)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (n, lg, flags)
        o = order.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(o), np.arange(n))             # a permutation
        chunk = 1 << (lg or 14)
        octs = ((rnd["dx"] < 0).astype(np.int64) | ((rnd["dy"] < 0).astype(np.int64) << 1) | ((rnd["dz"] < 0).astype(np.int64) << 2))[o]
        for c0 in range(0, n, chunk):
            oc, oi = octs[c0:c0 + chunk], o[c0:c0 + chunk]
            assert (oi // chunk == c0 // chunk).all()               # ... inside each chunk
            assert (np.diff(oc) >= 0).all()                         # octants ascending
            assert (np.diff(oi)[np.diff(oc) == 0] > 0).all()        # stable inside an octant
