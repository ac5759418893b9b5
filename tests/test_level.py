"""mr_trace_level -- one level of Scene::traceScene's recursion (Scene.cpp:270-346) in one launch -- against the batched
calls it replaces (mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_accumulate -> mr_gen_secondary_rays |
mr_gen_path_rays), level by level on the same queues: the children are the same rays, weights, pixels and ids bit for bit
(as sets: the two compactions order them differently), the ray counts are equal, the pixel sums agree up to the order of
the float atomics.  End-to-end frames against the oracle's recursion: tests/test_specular.py (fused=True)."""
import numpy as np
import pytest

from miro_amd import binding
from miro_amd import frame as mframe
from miro_amd import scenes
from test_specular import build_both, phong

pytestmark = pytest.mark.gpu


def canon(rays, w, pix, ids=None):
    """queue -> rows of uint32 in lexicographic order"""
    cols = [rays.view(np.uint32).reshape(len(rays), 8), w.view(np.uint32).reshape(len(w), 3), pix.astype(np.uint32)[:, None]]
    if ids is not None:
        cols.append(ids.astype(np.uint32)[:, None])
    m = np.concatenate(cols, axis=1)
    return m[np.lexsort(m.T[::-1])]


def batched_level(torch, sc, rays, weights, pixels, ids, n, L, W, spp, flags, children, level, seed, kinds):
    dev = rays.device
    f32 = dict(dtype=torch.float32, device=dev)
    hits = torch.empty((n, 4), **f32)
    sh_rays = torch.empty((n, 8), **f32)
    sh_hits = torch.empty((n, 4), **f32)
    src = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    rgb = torch.zeros((int(pixels.max().item()) + 1 if pixels is not None else (n + spp - 1) // spp, 3), **f32)
    sc.trace_device(rays, n, hits, flags)
    sc.gen_shadow_rays(rays, hits, n, L, sh_rays, src, cnt)
    sc.trace_indirect(sh_rays, cnt, n, sh_hits, flags)
    sc.shade_accumulate(rays, hits, weights, pixels, n, sh_rays, sh_hits, src, cnt, L, W, rgb, spp=spp)
    fan = 4
    out = (torch.empty((fan * n, 8), **f32), torch.empty((fan * n, 3), **f32), torch.empty(fan * n, dtype=torch.int32, device=dev),
           torch.empty(fan * n, dtype=torch.int32, device=dev))
    cnt2 = torch.zeros(1, dtype=torch.int64, device=dev)
    if children == binding.MR_LEVEL_PATH:
        sc.gen_path_rays(rays, hits, weights, pixels, ids, n, out[0], out[1], out[2], out[3], cnt2, spp=spp, seed=seed, bounce=level,
                         kinds=kinds)
    elif children == binding.MR_LEVEL_SPECULAR:
        sc.gen_secondary_rays(rays, hits, weights, pixels, n, out[0], out[1], out[2], cnt2, spp=spp)
    m = int(cnt2.item())
    return rgb, int(cnt.item()), [o[:m] for o in out]


def fused_level(torch, sc, rays, weights, pixels, ids, n, L, W, spp, flags, children, level, seed, kinds, n_pixels):
    dev = rays.device
    f32 = dict(dtype=torch.float32, device=dev)
    rgb = torch.zeros((n_pixels, 3), **f32)
    fan = 4
    out = (torch.empty((fan * n, 8), **f32), torch.empty((fan * n, 3), **f32), torch.empty(fan * n, dtype=torch.int32, device=dev),
           torch.empty(fan * n, dtype=torch.int32, device=dev))
    cnts = torch.zeros(3, dtype=torch.int64, device=dev)
    last = children == binding.MR_LEVEL_LAST
    sc.trace_level(rays, weights, pixels, ids, n, rgb, L, W, children=children,
                   d_out_rays=None if last else out[0], d_out_weights=None if last else out[1],
                   d_out_pixels=None if last else out[2], d_out_ids=out[3] if children == binding.MR_LEVEL_PATH else None,
                   d_out_count=None if last else cnts[2:], d_counts=cnts[:2], spp=spp, flags=flags, seed=seed, bounce=level, kinds=kinds)
    c = cnts.tolist()
    assert c[0] == n
    return rgb, c[1], [o[:c[2]] for o in out]


@pytest.mark.parametrize("mode", ["specular", "path3", "path7"])
@pytest.mark.parametrize("flags", [0, binding.MR_MATH_PRODUCT], ids=["exact", "product"])
def test_level_by_level_equals_the_batched_calls(oracle, miro, mode, flags):
    import torch
    _, sc, _, prim_mat = build_both(oracle, miro)
    if mode != "specular":
        sc.set_materials([phong((0.4, 0.4, 0.5), ks=(0.6, 0.6, 0.5), shininess=30.0),
                          phong((1, 1, 1), kt=(0.9, 0.95, 1.0), shininess=200.0, index=1.5),
                          phong((0.8, 0.8, 0.8))], prim_mat)
    children = binding.MR_LEVEL_SPECULAR if mode == "specular" else binding.MR_LEVEL_PATH
    kinds = {"specular": 3, "path3": 3, "path7": 7}[mode]
    d = scenes.SCENES["teapot"]
    W, H, spp = 80, 60, 2
    fr = mframe.FrameRenderer(sc, d, W, H, spp=spp, tiled=False)
    fr.generate()
    L, Wt = d["light"], d["wattage"]
    rays, weights, pixels, ids, n = fr.d_rays, None, None, None, fr.n
    depth = 4 if mode != "path7" else 2
    total_children = 0
    for level in range(depth + 1):
        fl = flags | (binding.MR_TRACE_INCOHERENT if level > 0 else 0)
        ch = children if level < depth else binding.MR_LEVEL_LAST
        rgb_b, ns_b, out_b = batched_level(torch, sc, rays, weights, pixels, ids, n, L, Wt, spp, fl, ch, level, 77, kinds)
        rgb_f, ns_f, out_f = fused_level(torch, sc, rays, weights, pixels, ids, n, L, Wt, spp, fl, ch, level, 77, kinds, rgb_b.shape[0])
        assert ns_b == ns_f
        scale = float(rgb_b.abs().max())
        assert torch.allclose(rgb_f, rgb_b, rtol=1e-5, atol=1e-6 * scale)
        assert scale > 0 or level > 0
        if ch == binding.MR_LEVEL_LAST:
            break
        assert len(out_f[0]) == len(out_b[0])
        with_ids = ch == binding.MR_LEVEL_PATH
        a = canon(*[o.cpu().numpy() for o in (out_b if with_ids else out_b[:3])])
        b = canon(*[o.cpu().numpy() for o in (out_f if with_ids else out_f[:3])])
        assert np.array_equal(a, b)
        total_children += len(a)
        # the next level runs on the batched queue (either would do: they are the same set)
        rays, weights, pixels, n = out_b[0].contiguous(), out_b[1].contiguous(), out_b[2].contiguous(), len(out_b[0])
        ids = out_b[3].contiguous() if with_ids else None
        if n == 0:
            break
    assert total_children > 1000


def test_level_on_a_scene_with_spheres_and_planes(oracle, miro):
    """the object dispatch (VAR bit 5) compiled into the level kernel: an analytic sphere and a plane"""
    import torch
    sc = miro.Scene()
    scenes.populate(sc, "teapot")
    sc.add_sphere([0.0, 1.0, 0.0], 0.8)
    sc.add_plane([0.0, 1.0, 0.0], [0.0, -0.5, 0.0])
    sc.build(4)
    d = scenes.SCENES["teapot"]
    fr = mframe.FrameRenderer(sc, d, 64, 48, spp=1, tiled=False)
    fr.generate()
    rgb_b, ns_b, _ = batched_level(torch, sc, fr.d_rays, None, None, None, fr.n, d["light"], d["wattage"], 1, 0, binding.MR_LEVEL_LAST, 0, 1, 3)
    rgb_f, ns_f, _ = fused_level(torch, sc, fr.d_rays, None, None, None, fr.n, d["light"], d["wattage"], 1, 0, binding.MR_LEVEL_LAST, 0, 1, 3,
                                 rgb_b.shape[0])
    assert ns_b == ns_f and ns_b > 0
    # one ray per pixel: a single addition per pixel, so the sums are the same bits
    assert torch.equal(rgb_b, rgb_f)


def test_level_argument_checks(miro):
    import torch
    sc = miro.Scene()
    scenes.populate(sc, "teapot")
    sc.build(4)
    d = scenes.SCENES["teapot"]
    rays = torch.zeros((64, 8), dtype=torch.float32, device="cuda")
    rgb = torch.zeros((64, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(miro.MiroError):          # children without an output queue
        sc.trace_level(rays, None, None, None, 64, rgb, d["light"], d["wattage"], children=binding.MR_LEVEL_SPECULAR)
    with pytest.raises(miro.MiroError):          # a flag the call does not take
        sc.trace_level(rays, None, None, None, 64, rgb, d["light"], d["wattage"], flags=binding.MR_TRACE_ANY)
    with pytest.raises(miro.MiroError):          # spp = 0
        sc.trace_level(rays, None, None, None, 64, rgb, d["light"], d["wattage"], spp=0)
    # an empty queue is fine and leaves the counters alone
    cnts = torch.zeros(3, dtype=torch.int64, device="cuda")
    sc.trace_level(rays, None, None, None, 0, rgb, d["light"], d["wattage"], d_counts=cnts[:2])
    assert cnts.tolist() == [0, 0, 0]
