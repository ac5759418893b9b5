"""mr_render_direct (the fused frame kernel, mr_frame.hip) against the batched pipeline it replaces:
mr_gen_eye_rays[_tiled] -> mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_direct [-> untile].
Both are built from the same device functions, so everything must agree bit for bit: primary hit records, the
shadow ray's hit record of every sample, the ray counts and the float framebuffer.  The batched pipeline itself is
compared with the oracle in tests/test_frame.py (test_frame_pipeline_matches_oracle)."""
import numpy as np
import pytest
import torch

from helpers import camera_of, oracle_scene, product_scene
from miro_amd import binding
from miro_amd import frame as mframe
from miro_amd import scenes

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.int32).numpy()


def _compare(miro, sc, name, W, H, spp, tiled, flags=0, any_shadow=False, jitter=None):
    ref = mframe.FrameRenderer(sc, name, W, H, spp=spp, jitter=jitter, flags=flags, tiled=tiled)
    ref.generate()
    ref.step(any_hit=any_shadow)
    fu = mframe.FusedFrame(sc, name, W, H, spp=spp, jitter=jitter, flags=flags, tiled=tiled, keep_hits=True,
                           any_shadow=any_shadow)
    assert fu.tiled == ref.tiled or spp >= 64 or spp & (spp - 1)
    fu.step()
    torch.cuda.synchronize()
    n_p, n_s = ref.ray_counts()
    assert fu.ray_counts() == (n_p, n_s)
    # primary hit records: same sample order (image order, or the tiled order of the window)
    assert np.array_equal(_bits(fu.d_hits), _bits(ref.d_hits))
    # shadow records: the batched pipeline compacts them; d_src maps a shadow ray to its sample
    src = ref.d_src[:n_s].to(torch.int64)
    got = fu.d_shadow_hits[src]
    if any_shadow:       # any-hit: which occluder is reported depends on nothing but the ray, still identical
        assert np.array_equal(_bits(got), _bits(ref.d_shadow_hits[:n_s]))
    else:
        assert np.array_equal(_bits(got), _bits(ref.d_shadow_hits[:n_s]))
    no_ray = torch.ones(ref.n, dtype=torch.bool, device=src.device)
    no_ray[src] = False
    rest = fu.d_shadow_hits[no_ray]
    assert (rest[:, 0] == 0).all() and (rest[:, 1].view(torch.int32) == -1).all()
    # the picture, in image order
    assert np.array_equal(_bits(fu.d_rgb), _bits(ref.d_rgb))
    return fu, ref


@pytest.mark.parametrize("name,W,H,spp,tiled", [
    ("teapot", 128, 96, 1, False), ("teapot", 128, 96, 1, True), ("teapot", 97, 61, 4, True), ("bunny", 80, 45, 16, True),
    ("bunny", 80, 45, 16, False), ("sponza", 64, 36, 64, False), ("sponza", 150, 101, 2, True), ("cornell", 33, 31, 32, True),
    ("sponza", 31, 17, 8, False)])
def test_fused_frame_equals_the_batched_pipeline(miro, name, W, H, spp, tiled):
    sc = product_scene(miro, name)
    _compare(miro, sc, name, W, H, spp, tiled)


@pytest.mark.parametrize("flags", ["product", "incoherent", "product+incoherent", "any"])
def test_fused_frame_modes(miro, flags):
    sc = product_scene(miro, "sponza")
    fl = 0
    if "product" in flags:
        fl |= binding.MR_MATH_PRODUCT
    if "incoherent" in flags:
        fl |= binding.MR_TRACE_INCOHERENT
    _compare(miro, sc, "sponza", 120, 67, 4, True, flags=fl, any_shadow=flags == "any")


def test_fused_frame_with_spheres_and_planes(miro):
    sc = product_scene(miro, "spiral")
    _compare(miro, sc, "spiral", 96, 64, 4, True)
    _compare(miro, sc, "spiral", 96, 64, 1, False)


@pytest.mark.parametrize("world,band", [(2, 6), (4, 6), (8, 5), (3, 8)])
def test_fused_frame_of_one_ranks_bands(miro, world, band):
    """The interleaved bands of each rank (band_rows rule of frame.band_rows), rendered by one launch per rank, put
    together give the unsharded frame byte for byte -- tiles that straddle two bands included."""
    name, W, H, spp = "sponza", 96, 54, 4
    sc = product_scene(miro, name)
    whole = mframe.FusedFrame(sc, name, W, H, spp=spp)
    whole.step()
    full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    tot = [0, 0]
    for r in range(world):
        fu = mframe.FusedFrame(sc, name, W, H, spp=spp, band=band, rank=r, world=world)
        fu.step()
        rows = torch.from_numpy(mframe.rows_of(mframe.band_rows(H, band, r, world))).to("cuda")
        assert fu.n_rows == len(rows)
        if len(rows):
            full[rows] = fu.d_rgb.view(len(rows), W, 3)
        c = fu.ray_counts()
        tot[0] += c[0]
        tot[1] += c[1]
    torch.cuda.synchronize()
    assert np.array_equal(_bits(full.view(-1, 3)), _bits(whole.d_rgb))
    assert tuple(tot) == whole.ray_counts()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_frame_shares_of_the_ranks_are_the_unsharded_frame(miro, world):
    """The bench frame (sponza 1920x1080, 64 samples per pixel) as bench.py shards it -- interleaved bands of 8 rows -- rendered
    share by share on one device: each share is a large launch of its own (a half: 259 200 chunks, an eighth: 64 800 -- both take
    the schedule with the pulled tail, mr_frame.hip), and together they are the unsharded frame byte for byte."""
    name, W, H, spp, band = "sponza", 1920, 1080, 64, 8
    sc = product_scene(miro, name)
    whole = mframe.FusedFrame(sc, name, W, H, spp=spp)
    whole.step()
    full = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda")
    for r in range(world):
        fu = mframe.FusedFrame(sc, name, W, H, spp=spp, band=band, rank=r, world=world)
        assert fu.n >= 60000 * 256
        fu.step()
        rows = torch.from_numpy(mframe.rows_of(mframe.band_rows(H, band, r, world))).to("cuda")
        full[rows] = fu.d_rgb.view(len(rows), W, 3)
        del fu
    torch.cuda.synchronize()
    assert torch.equal(full.view(-1, 3).view(torch.int32), whole.d_rgb.view(torch.int32))


@pytest.mark.parametrize("world,band", [(2, 6), (4, 5)])
def test_fused_hit_records_of_the_ranks_deinterleave_to_the_unsharded_buffer(miro, world, band):
    """The hit-buffer parity mode of SURVEY section 8e on the fused path: every rank's primary mr_hit records (image order
    inside its bands, 4*spp floats per pixel), put side by side as the gather would and de-interleaved by
    mr_deinterleave_bands, are the unsharded frame's hit buffer."""
    name, W, H, spp = "bunny", 80, 50, 4
    sc = product_scene(miro, name)
    whole = mframe.FusedFrame(sc, name, W, H, spp=spp, tiled=False, keep_hits=True)
    whole.step()
    counts = [miro.band_rows_of(H, band, r, world) for r in range(world)]
    shard_rows = max(counts)
    recv = torch.zeros((world, shard_rows * W, 4 * spp), dtype=torch.float32, device="cuda")
    for r in range(world):
        fu = mframe.FusedFrame(sc, name, W, H, spp=spp, band=band, rank=r, world=world, tiled=False, keep_hits=True)
        fu.step()
        recv[r, :counts[r] * W] = fu.d_hits.view(counts[r] * W, 4 * spp)
    full = torch.empty((H * W, 4 * spp), dtype=torch.float32, device="cuda")
    sc.deinterleave_bands(recv, full, W, H, band, world, shard_rows, 4 * spp)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(full.view(-1, 4)), _bits(whole.d_hits))


def test_fused_frame_rejects_what_it_cannot_do(miro):
    sc = product_scene(miro, "teapot")
    rgb = torch.zeros((16 * 16, 3), dtype=torch.float32, device="cuda")
    cam = binding.make_camera(*[scenes.SCENES["teapot"][k] for k in ("eye", "lookat", "up", "fov")])
    with pytest.raises(miro.MiroError):
        sc.render_direct(cam, 16, 16, rgb, (0, 1, 0), 100.0, spp=3)
    with pytest.raises(miro.MiroError):
        sc.render_direct(cam, 16, 16, rgb, (0, 1, 0), 100.0, spp=128)
    with pytest.raises(miro.MiroError):
        sc.render_direct(cam, 16, 16, rgb, (0, 1, 0), 100.0, y0=8, y1=20)
    with pytest.raises(miro.MiroError):
        sc.render_direct(cam, 16, 16, rgb, (0, 1, 0), 100.0, flags=binding.MR_COUNT_STATS)


@pytest.mark.parametrize("spp", [16, 64])
def test_fused_frame_full_size_properties(miro, spp):
    """BASELINE config 4 at full size (1920x1080; 64 spp = 132.7 M samples, the bench frame; the batched pipeline it is
    compared with keeps 13.3 GB of rays and hits resident, the fused one 25 MB): size-independent checks -- idempotence
    (two steps, same bytes), ray-count conservation (closed scene: one shadow ray per primary ray), and equality with the
    batched pipeline's picture."""
    name, W, H = "sponza", 1920, 1080
    sc = product_scene(miro, name)
    fu = mframe.FusedFrame(sc, name, W, H, spp=spp)
    fu.step()
    a = fu.d_rgb.clone()
    fu.step()
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), fu.d_rgb.view(torch.int32))
    n_p, n_s = fu.ray_counts(steps=2)
    assert n_p == W * H * spp and n_s == n_p
    ref = mframe.FrameRenderer(sc, name, W, H, spp=spp, tiled=True)
    ref.generate()
    ref.step()
    torch.cuda.synchronize()
    assert torch.equal(ref.d_rgb.view(torch.int32), fu.d_rgb.view(torch.int32))


@pytest.mark.parametrize("W,H,spp", [(1024, 1024, 64), (1023, 1031, 64), (1400, 1500, 64), (1024, 1024, 16), (1021, 1027, 16),
                                     (1920, 1080, 16), (2048, 1100, 8), (1920, 135, 64), (1000, 961, 16)])
def test_large_frame_schedule_renders_every_chunk_once(miro, W, H, spp):
    """Frames of 60 000 chunks or more are launched with the schedule of frame_schedule() (mr_frame.hip): body workgroups striding
    over their chunks, and a tail that the last workgroups pull chunk by chunk from a counter (re-armed by its last reader: the
    frame is rendered three times here).  The picture is the one narrow windows give -- those are below the threshold and take
    the plain strided schedule.  Sizes: the threshold exactly, ragged last chunks, 8 / 16 / 32 chunks per body workgroup."""
    name = "bunny"
    d = scenes.SCENES[name]
    sc = product_scene(miro, name)
    assert (W * H * spp + 255) // 256 >= 60000
    full = mframe.FusedFrame(sc, d, W, H, spp=spp, jitter=True, seed=3, tiled=False)
    for _ in range(3):
        full.d_rgb.fill_(-1.0)
        full.step()
    n_p, n_s = full.ray_counts(steps=3)
    assert n_p == W * H * spp
    parts = []
    rows = max(1, (256 * 59999) // (W * spp))                 # the tallest window below the threshold
    for y0 in range(0, H, rows):
        y1 = min(H, y0 + rows)
        assert ((y1 - y0) * W * spp + 255) // 256 < 60000
        rgb = torch.full(((y1 - y0) * W, 3), -2.0, dtype=torch.float32, device="cuda")
        sc.render_direct(full.cam, W, H, rgb, d["light"], d["wattage"], y0=y0, y1=y1, spp=spp, jitter=True, seed=3, tiled=False)
        parts.append(rgb)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts).view(torch.int32), full.d_rgb.view(torch.int32))


def test_bench_frame_window_against_the_oracle(oracle, miro):
    """The timed kernel at its own configuration, directly against the oracle: rows 536-540 of the
    bench frame -- sponza 1920x1080, 64 jittered samples per pixel, frame_kernel<794,false> in 256-lane workgroups is what
    bench.py times -- primary and shadow mr_hit records bytes-equal to the oracle's Scene::trace on the oracle's own eye and
    shadow rays (Camera.cpp:104-161, Phong.cpp:80-97), pixels within 1e-5 of the restated Phong::shade (Phong.cpp:116-156)."""
    name, W, H, spp, y0, y1 = "sponza", 1920, 1080, 64, 536, 540
    d = scenes.SCENES[name]
    sc, ref = product_scene(miro, name), oracle_scene(oracle, name)
    n = (y1 - y0) * W * spp
    f32 = dict(dtype=torch.float32, device="cuda")
    d_rgb = torch.zeros(((y1 - y0) * W, 3), **f32)
    d_h, d_s = torch.empty((n, 4), **f32), torch.empty((n, 4), **f32)
    d_counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    # a window this size (491 520 samples) would be launched in 128-lane workgroups; the bench frame runs the 256-lane
    # build, so render the window as the bench does: through the whole-frame launch of a renderer that keeps its hits
    full = mframe.FusedFrame(sc, d, W, H, spp=spp, jitter=True, seed=168, tiled=False, keep_hits=True)
    full.step()
    torch.cuda.synchronize()
    lo, hi = y0 * W * spp, y1 * W * spp
    got_h = full.d_hits[lo:hi].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    got_s = full.d_shadow_hits[lo:hi].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)
    got_rgb = full.d_rgb[y0 * W:y1 * W].cpu().numpy()
    # ... and the same rows as a window of their own (the 128-lane build): the same bytes
    sc.render_direct(full.cam, W, H, d_rgb, d["light"], d["wattage"], y0=y0, y1=y1, spp=spp, jitter=True, seed=168, tiled=False,
                     d_hits=d_h, d_shadow_hits=d_s, d_counts=d_counts)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(d_h), _bits(full.d_hits[lo:hi])) and np.array_equal(_bits(d_s), _bits(full.d_shadow_hits[lo:hi]))
    assert np.array_equal(_bits(d_rgb), _bits(full.d_rgb[y0 * W:y1 * W]))
    del full
    torch.cuda.empty_cache()

    rays = oracle.eye_rays(camera_of(oracle, name), W, H, spp=spp, jitter=True, seed=168, y0=y0, y1=y1)
    want = ref.trace(rays)
    assert got_h.tobytes() == want.tobytes(), "primary hit records differ from the oracle's"
    sh, src = ref.shadow_rays(rays, want, d["light"])
    want_s = ref.trace(sh)
    src = src.astype(np.int64)
    assert got_s[src].tobytes() == want_s.tobytes(), "shadow hit records differ from the oracle's"
    rest = np.ones(n, bool)
    rest[src] = False
    assert (got_s["prim"][rest] == oracle.MISS).all() and (got_s["t"][rest] == 0).all()
    assert d_counts.cpu().numpy().tolist() == [n, len(sh)]
    occ = np.zeros(n, np.uint8)
    occ[src] = want_s["prim"] != oracle.MISS
    want_rgb = ref.shade_direct(rays, want, occ, d["light"], d["wattage"], spp=spp)
    assert want_rgb.max() > 0 and np.abs(got_rgb - want_rgb).max() <= 1e-5 * max(1.0, np.abs(want_rgb).max())
