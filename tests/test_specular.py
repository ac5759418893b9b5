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


@pytest.mark.parametrize("fused", [False, True, "auto"], ids=["batched", "fused", "auto"])
def test_specular_frame_matches_oracle(oracle, miro, fused):
    """fused: every level is one launch of mr_trace_level instead of the seven batched calls"""
    import torch
    assert torch.cuda.is_available()
    a, b, mats11, prim_mat = build_both(oracle, miro)
    d = scenes.SCENES["teapot"]
    W, H, spp = 96, 72, 2
    fr = mframe.FrameRenderer(b, d, W, H, spp=spp)
    fr.generate()
    levels = fr.render_specular(depth=10, fused=fused)
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


@pytest.mark.parametrize("fused", [False, True], ids=["batched", "fused"])
@pytest.mark.parametrize("kinds,depth", [(3, 4), (7, 2)])
def test_path_traced_frame_matches_oracle(oracle, miro, kinds, depth, fused):
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
    levels = fr.render_specular(depth=depth, path_tracing=True, path_seed=99, path_kinds=kinds, fused=fused)
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
