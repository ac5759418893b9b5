"""Row f4 of SURVEY.md section 8: specular secondary rays.  Scene::traceScene's recursion (reflect, Fresnel,
refract, depth 10, light attenuated by refractive occluders) as wavefront bounces with ballot compaction between
levels, against the oracle's restated recursion.  Tolerance parity (sinf/acosf/powf differ from libm in the last
ulps; float atomics reorder the per-pixel sums)."""
import numpy as np
import pytest

from helpers import camera_of
from miro_amd import frame as mframe
from miro_amd import scenes

pytestmark = pytest.mark.gpu

INF = float("inf")


def phong(kd, ks=(0, 0, 0), kt=(0, 0, 0), shininess=1.0, index=1.0):
    return (tuple(kd), tuple(ks), tuple(kt), shininess, index)


def clamp_like_phong_ctor(mats):
    out = []
    for kd, ks, kt, sh, ri in mats:
        kd, ks, kt = (np.asarray(x, np.float32) for x in (kd, ks, kt))
        kt = np.maximum(np.minimum(kt, np.float32(1.0) - ks), np.float32(0))
        kd = np.maximum(np.minimum(kd, np.float32(1.0) - ks - kt), np.float32(0))
        out.append(np.concatenate([kd, ks, kt, [sh, ri]]).astype(np.float32))
    return np.stack(out)


def build_both(oracle, miro):
    """teapot (mirror-ish), a glass ball (sphere.obj scaled + lifted), the floor triangle (white Lambert)"""
    ctm = np.array([[0.8, 0, 0, 2.2], [0, 0.8, 0, 1.0], [0, 0, 0.8, 1.5], [0, 0, 0, 1]], np.float32)
    a, b = oracle.Scene(), miro.Scene()
    counts = []
    for s in (a, b):
        n_teapot = s.add_obj(scenes._model("teapot.obj"))
        n_ball = s.add_obj(scenes._model("sphere.obj"), ctm)
        s.add_triangle(np.asarray(scenes.SCENES["teapot"]["floor"], np.float32).reshape(9), [0, 1, 0] * 3)
        counts.append((n_teapot, n_ball))
    assert counts[0] == counts[1]
    n_teapot, n_ball = counts[0]
    mats = [phong((0.4, 0.4, 0.5), ks=(0.6, 0.6, 0.5), shininess=INF),
            phong((1, 1, 1), kt=(0.9, 0.95, 1.0), shininess=5.0, index=1.5),
            phong((1, 1, 1))]
    prim_mat = np.array([0] * n_teapot + [1] * n_ball + [2], np.uint32)
    a.build(4)
    b.set_materials(mats, prim_mat)
    b.build(4)
    return a, b, clamp_like_phong_ctor(mats), prim_mat


@pytest.mark.parametrize("grouped", [True, False], ids=["grouped", "as-made"])
@pytest.mark.parametrize("fused", [False, True, "auto"], ids=["batched", "fused", "auto"])
def test_specular_frame_matches_oracle(oracle, miro, fused, grouped):
    """fused: every level is one launch of mr_trace_level instead of the seven batched calls; grouped: the queues of the
    levels after the first are worked through mr_order_by_octant's index"""
    import torch
    assert torch.cuda.is_available()
    a, b, mats11, prim_mat = build_both(oracle, miro)
    d = scenes.SCENES["teapot"]
    W, H, spp = 96, 72, 2
    fr = mframe.FrameRenderer(b, d, W, H, spp=spp)
    fr.generate()
    levels = fr.render_specular(depth=10, fused=fused, group_octants=grouped)
    torch.cuda.synchronize()
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), W, H, spp=spp, jitter=True, seed=168)
    want_rays, calls = a.trace_scene(mats11, prim_mat, rays, d["light"], d["wattage"], depth=10)
    want = want_rays.reshape(H * W, spp, 3).astype(np.float64).mean(axis=1)
    got = fr.d_rgb.cpu().numpy().astype(np.float64)
    # the recursion really went several levels deep and traced the same number of rays (primary + secondary + shadow)
    assert len(levels) >= 4
    assert sum(n + ns for n, ns in levels) == calls
    scale = np.abs(want).max()
    err = np.abs(got - want)
    # a ray whose Fresnel term sits at the Rs > 0.01 threshold, or whose hit flips at a silhouette after many
    # bounces of 1-ulp-different directions, may differ visibly: allow a few pixels, none of them wildly off
    assert (err.max(axis=1) <= 2e-4 * scale).mean() > 0.995
    assert np.median(err) <= 1e-6 * scale
    assert scale > 0


@pytest.mark.parametrize("grouped", [True, False], ids=["grouped", "as-made"])
@pytest.mark.parametrize("fused", [False, True], ids=["batched", "fused"])
@pytest.mark.parametrize("kinds,depth", [(3, 4), (7, 2)])
def test_path_traced_frame_matches_oracle(oracle, miro, kinds, depth, fused, grouped):
    """Scene::traceScene as the PATH_TRACING build runs it -- glossy mirror (finite shininess) and rough glass, every child
    drawn from its lobe with the counter-based generator; kinds = 7 adds the diffuse bounce of Ray::random (extension).
    Same ray totals per level as the oracle's recursion makes Scene::trace calls, pixels within the tolerance of the
    mirror-direction test (float atomics reorder the per-pixel sums; the rays themselves are bit-equal, test_path_rays.py)."""
    import torch
    a, b, _, prim_mat = build_both(oracle, miro)
    mats = [phong((0.4, 0.4, 0.5), ks=(0.6, 0.6, 0.5), shininess=30.0),
            phong((1, 1, 1), kt=(0.9, 0.95, 1.0), shininess=200.0, index=1.5),
            phong((0.8, 0.8, 0.8))]
    b.set_materials(mats, prim_mat)
    mats11 = clamp_like_phong_ctor(mats)
    d = scenes.SCENES["teapot"]
    W, H, spp = 64, 48, 2
    fr = mframe.FrameRenderer(b, d, W, H, spp=spp)
    fr.generate()
    levels = fr.render_specular(depth=depth, path_tracing=True, path_seed=99, path_kinds=kinds, fused=fused, group_octants=grouped)
    torch.cuda.synchronize()
    got = fr.d_rgb.cpu().numpy()
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), W, H, spp=spp, jitter=True, seed=168)
    per_ray, calls = a.trace_scene_pt(mats11, prim_mat, rays, d["light"], d["wattage"], depth=depth, seed=99, kinds=kinds)
    want = per_ray.reshape(H * W, spp, 3).astype(np.float64).mean(axis=1)
    # Scene::trace calls of the recursion = rays traced by the wavefront levels (primary / bounce rays + their shadow rays)
    assert calls == sum(n for n, _ in levels) + sum(s for _, s in levels)
    assert len(levels) >= 2 and levels[1][0] > 0
    scale = max(1e-6, float(np.abs(want).max()))
    err = np.abs(got - want)
    assert (err.max(axis=1) <= 2e-4 * scale).mean() > 0.995 and np.median(err) <= 1e-6 * scale


def test_default_material_equals_direct_shade(oracle, miro):
    """Without mr_scene_set_materials every triangle is the white Lambert: the general accumulate path must give
    the picture of the deterministic single-bounce path (up to float-atomic ordering -> identical here, one add per
    sample ... per pixel order may differ, so compare with a tolerance)."""
    import torch
    b = miro.Scene()
    scenes.populate(b, "bunny")
    b.build(4)
    fr = mframe.FrameRenderer(b, "bunny", 128, 96, spp=4)
    fr.generate()
    fr.step()
    ref = fr.d_rgb.clone()
    for fused in (False, True):
        levels = fr.render_specular(depth=10, fused=fused)
        torch.cuda.synchronize()
        assert len(levels) == 1                       # no specular material: no second level
        assert levels[0] == fr.ray_counts() if not fused else levels[0][0] == fr.n
        assert torch.allclose(fr.d_rgb, ref, rtol=1e-5, atol=1e-7 * float(ref.max()))


def test_material_argument_checks(miro):
    s = miro.Scene()
    s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    with pytest.raises(miro.MiroError):
        s.set_materials([phong((1, 1, 1))], np.array([3], np.uint32))      # material id out of range


@pytest.mark.parametrize("W,H,spp", [(160, 120, 1), (96, 72, 4)])
def test_direct_frame_lets_light_through_a_glass_occluder(oracle, miro, W, H, spp):
    """mr_render_direct in a scene with a material table: Phong::shade with the material of the object
    that was hit and light through the refractive ball scaled by dot(N, l) of the ball (Phong.cpp:99-113) -- the oracle's
    traceScene at depth 0 is exactly Phong::shade of the primary hit (Scene.cpp:278-284; its children return at depth -1)."""
    import torch
    from miro_amd import binding
    a, b, mats11, prim_mat = build_both(oracle, miro)
    d = scenes.SCENES["teapot"]
    fu = mframe.FusedFrame(b, d, W, H, spp=spp, keep_hits=True, tiled=False)
    fu.step()
    torch.cuda.synchronize()
    got = fu.d_rgb.cpu().numpy().astype(np.float64)
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), W, H, spp=spp, jitter=spp > 1, seed=168)
    want_rays, calls = a.trace_scene(mats11, prim_mat, rays, d["light"], d["wattage"], depth=0)
    want = want_rays.reshape(H * W, spp, 3).astype(np.float64).mean(axis=1)
    n_p, n_s = fu.ray_counts()
    assert calls == n_p + n_s                                     # one Scene::trace per eye ray and per shadow ray
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert scale > 0 and err.max() <= 2e-4 * scale and np.median(err) <= 1e-6 * scale
    # the case is live: some samples' nearest occluder is the glass ball and they are lit (an opaque ball would zero them)
    sh = fu.d_shadow_hits.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    n_teapot = int((prim_mat == 0).sum())
    through_glass = (sh["prim"] != miro.MISS) & (sh["prim"] >= n_teapot) & (sh["prim"] < n_teapot + int((prim_mat == 1).sum()))
    lit = want_rays.reshape(-1, 3).max(axis=1) > 0
    assert (through_glass & lit).sum() > 20, "no sample sees the light through the glass ball"
    # the same frame, one level of the recursion in one launch over resident eye rays (mr_trace_level, MR_LEVEL_LAST)
    fr = mframe.FrameRenderer(b, d, W, H, spp=spp)
    fr.generate()
    fr.render_specular(depth=0, fused=True)
    torch.cuda.synchronize()
    assert torch.allclose(fr.d_rgb, fu.d_rgb, rtol=1e-6, atol=1e-7 * float(scale))
    if spp == 1:
        assert torch.equal(fr.d_rgb, fu.d_rgb)                    # one sample per pixel: no summation order to differ by
    # calls that would get it wrong refuse: the flag-per-ray occlusion of mr_shade_direct, any-hit shadow rays
    with pytest.raises(miro.MiroError) as e:
        fr.step()
    assert e.value.status == binding.MR_ERR_STATE
    with pytest.raises(miro.MiroError) as e:
        mframe.FusedFrame(b, d, W, H, spp=spp, any_shadow=True).step()
    assert e.value.status == binding.MR_ERR_STATE


def test_frame_without_shadow_rays_is_the_disable_shadows_build(oracle, miro):
    """MR_FRAME_NO_SHADOWS = the reference's -DDISABLE_SHADOWS (Phong.cpp:91): BASELINE config 2, teapot 512x512 primary rays
    only -- 262 144 rays, no shadow ray traced (the write-up's count, Readme.tex:99), every hit lit as if nothing occluded it."""
    import torch
    from helpers import oracle_scene, product_scene
    name, W, H = "teapot", 512, 512
    d = scenes.SCENES[name]
    a, b = oracle_scene(oracle, name), product_scene(miro, name)
    fu = mframe.FusedFrame(b, d, W, H, spp=1, keep_hits=True, tiled=False, no_shadows=True)
    fu.step()
    torch.cuda.synchronize()
    assert fu.ray_counts() == (262144, 0)
    rays = oracle.eye_rays(camera_of(oracle, name), W, H)
    hits = a.trace(rays)
    assert fu.d_hits.cpu().numpy().tobytes() == hits.tobytes()
    assert int((hits["prim"] != oracle.MISS).sum()) == 222390
    sh = fu.d_shadow_hits.cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    assert (sh["prim"] == miro.MISS).all() and (sh["t"] == 0).all()
    want = a.shade_direct(rays, hits, np.zeros(len(rays), np.uint8), d["light"], d["wattage"], spp=1)
    got = fu.d_rgb.cpu().numpy()
    assert want.max() > 0 and np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # with materials: the unoccluded Phong::shade of the hit's own material = the shadowed frame wherever nothing occludes
    a2, b2, mats11, prim_mat = build_both(oracle, miro)
    f_on = mframe.FusedFrame(b2, d, 160, 120, spp=1, keep_hits=True, tiled=False)
    f_off = mframe.FusedFrame(b2, d, 160, 120, spp=1, tiled=False, no_shadows=True)
    f_on.step(); f_off.step()
    torch.cuda.synchronize()
    free = (f_on.d_shadow_hits[:, 1].view(torch.int32) == -1)
    assert free.any() and torch.equal(f_on.d_rgb[free], f_off.d_rgb[free])
    assert (f_off.d_rgb[~free].sum(dim=1) >= f_on.d_rgb[~free].sum(dim=1)).all()
