"""The callers either side of the hot path: wavefront frame driver, Phong shading, image-tile sharding and
the single framebuffer gather.  CPU part: sharding arithmetic + a world_size-2 gloo run of the gather.
GPU part: shading parity with the oracle and shard-count independence of the frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import camera_of, oracle_scene, product_scene
from miro_amd import binding
from miro_amd import frame as mframe
from miro_amd import scenes


# ------------------------------------------------------------------------------------------- CPU: sharding
@pytest.mark.parametrize("H,band,world", [(1080, 8, 1), (1080, 8, 2), (1080, 8, 8), (1080, 16, 3), (37, 8, 4), (5, 8, 8)])
def test_band_rows_partition_the_image(H, band, world):
    seen = np.zeros(H, np.int32)
    sizes = []
    for r in range(world):
        rows = mframe.rows_of(mframe.band_rows(H, band, r, world))
        seen[rows] += 1
        sizes.append(len(rows))
        assert (np.diff(rows) > 0).all() if len(rows) > 1 else True
    assert (seen == 1).all()                                   # every row exactly once
    assert max(sizes) - min(sizes) <= band                     # balanced to within one band


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, H, W, band, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = mframe.rows_of(mframe.band_rows(H, band, rank, world))
    # each rank "renders" its rows: value encodes (row, column, channel)
    local = torch.empty((len(rows), W, 3), dtype=torch.float32)
    for i, y in enumerate(rows):
        local[i] = (y * W + torch.arange(W, dtype=torch.float32))[:, None] * 4 + torch.arange(3, dtype=torch.float32)[None, :]
    full = mframe.gather_framebuffer(local.reshape(-1, 3), H, W, band, rank, world)
    if rank == 0:
        torch.save(full, out_path)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,band", [(37, 5, 8), (64, 3, 8)])
def test_gather_framebuffer_gloo_world2(tmp_path, H, W, band):
    """The N>1 path on CPU: two processes, gloo, one gather; rank 0 reassembles the interleaved bands."""
    out = str(tmp_path / "full.pt")
    mp.spawn(_gather_worker, args=(2, _free_port(), H, W, band, out), nprocs=2, join=True)
    full = torch.load(out, weights_only=True)
    want = (torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1) * 4 + torch.arange(3, dtype=torch.float32))
    assert torch.equal(full, want)


def _pipelined_worker(rank, world, port, H, W, band, frames, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = mframe.FrameGather(H, W, band, rank, world, torch.device("cpu"))
    rows = torch.from_numpy(mframe.rows_of(mframe.band_rows(H, band, rank, world)))
    base = (rows[:, None] * W + torch.arange(W)[None, :]).to(torch.float32).reshape(-1, 1) * 4 + torch.arange(3, dtype=torch.float32)
    got = []
    for k in range(frames):
        # the bench's step order: wait for frame k-1's gather, "shade" frame k into the send buffer, start its gather
        prev = g.wait()
        if rank == 0 and k > 0:
            got.append(prev.clone())
        g.local.copy_(base + 1000.0 * k)
        g.start()
    last = g.wait()
    if rank == 0:
        got.append(last.clone())
        torch.save(torch.stack(got), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_gather_pipelined_gloo_world2(tmp_path):
    """FrameGather as bench.py drives it at N>1: asynchronous gather started after shading, waited for before the
    next frame overwrites the send buffer; every frame arrives intact on rank 0."""
    H, W, band, frames = 37, 5, 8, 4
    out = str(tmp_path / "frames.pt")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), H, W, band, frames, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    base = torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1) * 4 + torch.arange(3, dtype=torch.float32)
    assert got.shape[0] == frames
    for k in range(frames):
        assert torch.equal(got[k], base + 1000.0 * k)


def test_gather_framebuffer_single_rank():
    H, W = 20, 7
    local = torch.arange(H * W * 3, dtype=torch.float32).reshape(-1, 3)
    full = mframe.gather_framebuffer(local, H, W, 8, 0, 1)
    assert torch.equal(full.reshape(-1, 3), local)


# ------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("name,spp", [("teapot", 1), ("bunny", 4), ("sponza", 2), ("teapot", 3), ("cornell", 64)])
def test_frame_pipeline_matches_oracle(oracle, miro, name, spp):
    """gen -> trace -> shadow gen -> indirect trace -> shade on the device vs the restated
    raytraceImage / Phong::shade on the CPU: hits bit-exact, colours within 1e-5 (powf/rounding)."""
    assert torch.cuda.is_available()
    W, H = 160, 120
    d = scenes.SCENES[name]
    a = oracle_scene(oracle, name)
    b = product_scene(miro, name)
    fr = mframe.FrameRenderer(b, d, W, H, spp=spp)
    fr.generate()
    fr.step()
    torch.cuda.synchronize()
    rays = oracle.eye_rays(camera_of(oracle, name), W, H, spp=spp, jitter=spp > 1, seed=168)
    hits = a.trace(rays)
    assert fr.d_rays.cpu().numpy().tobytes() == rays.tobytes()
    assert fr.d_hits.cpu().numpy().tobytes() == hits.tobytes()
    sh, src = a.shadow_rays(rays, hits, d["light"])
    sh_hits = a.trace(sh)
    n_p, n_s = fr.ray_counts()
    assert (n_p, n_s) == (len(rays), len(sh))
    # shadow batch: same set of (source ray, hit) pairs
    k = n_s
    got_src = fr.d_src[:k].cpu().numpy().astype(np.int64)
    order = np.argsort(got_src, kind="stable")
    assert np.array_equal(got_src[order], src.astype(np.int64))
    got_sh = fr.d_shadow_hits[:k].cpu().numpy().view(miro.HIT_DTYPE).reshape(-1)[order]
    assert got_sh.tobytes() == sh_hits.tobytes()
    occ = np.zeros(len(rays), np.uint8)
    occ[src.astype(np.int64)] = sh_hits["prim"] != oracle.MISS
    want = a.shade_direct(rays, hits, occ, d["light"], d["wattage"], spp=spp)
    got = fr.d_rgb.cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert want.max() > 0
    # tone map + 8-bit quantisation: at most one code value apart (expf vs exp rounding at a boundary)
    out = torch.empty(got.shape, dtype=torch.uint8, device="cuda")
    b.tonemap(fr.d_rgb, got.size, out)
    tm = oracle.tonemap(want)
    assert np.abs(out.cpu().numpy().astype(np.int32) - tm.astype(np.int32)).max() <= 1


@gpu
def test_any_hit_shadow_batch_gives_the_same_image(oracle, miro):
    """Opaque scene: occlusion from the any-hit query == occlusion from the closest-hit query."""
    b = product_scene(miro, "bunny")
    fr = mframe.FrameRenderer(b, "bunny", 200, 150, spp=2)
    fr.generate()
    fr.step(any_hit=False)
    ref = fr.d_rgb.clone()
    fr.step(any_hit=True)
    torch.cuda.synchronize()
    assert torch.equal(ref, fr.d_rgb)


@gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_tile_sharding_is_invisible_in_the_result(miro, world):
    """SURVEY.md section 4 tier 4: the same frame sharded 1/2/4/8 ways gives byte-identical hit buffers and
    framebuffer (here the shards run one after the other on the one GPU of the test box)."""
    W, H, spp, band = 192, 100, 2, 8
    b = product_scene(miro, "sponza")
    whole = mframe.FrameRenderer(b, "sponza", W, H, spp=spp)
    whole.generate()
    whole.step()
    torch.cuda.synchronize()
    ref_rgb = whole.d_rgb.reshape(H, W, 3)
    ref_hits = whole.d_hits.reshape(H, W * spp, 4)
    full = torch.zeros_like(ref_rgb)
    for r in range(world):
        bands = mframe.band_rows(H, band, r, world)
        fr = mframe.FrameRenderer(b, "sponza", W, H, spp=spp, bands=bands)
        fr.generate()
        fr.step()
        torch.cuda.synchronize()
        rows = torch.from_numpy(mframe.rows_of(bands)).cuda()
        assert torch.equal(fr.d_hits.reshape(len(rows), W * spp, 4).view(torch.int32), ref_hits[rows].view(torch.int32))
        part = mframe.gather_framebuffer(fr.d_rgb, H, W, band, 0, 1) if world == 1 else fr.d_rgb.reshape(len(rows), W, 3)
        full[rows] = part
    assert torch.equal(full.view(torch.int32), ref_rgb.view(torch.int32))


@gpu
def test_frame_gather_through_rccl_single_rank(miro):
    """The collective exactly as bench.py issues it at N>1 -- asynchronous dist.gather on the "nccl" (= RCCL) backend
    into views of one receive buffer, stream-level wait, de-interleave on the device -- exercised with a one-rank
    process group, which is all a one-GPU box can hold."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        H, W, band = 37, 5, 8
        # de-interleave by torch index ops (scene=None) and by the library's mr_deinterleave_bands (what bench.py passes)
        for scene in (None, product_scene(miro, "teapot")):
            g = mframe.FrameGather(H, W, band, 0, 1, dev, always_collective=True, scene=scene)
            base = (torch.arange(H * W, dtype=torch.float32, device=dev).reshape(-1, 1) * 4 + torch.arange(3, dtype=torch.float32, device=dev))
            for k in range(3):
                g.wait()
                g.local.copy_(base + 1000.0 * k)
                g.start()
            full = g.wait()
            torch.cuda.synchronize()
            assert torch.equal(full.reshape(-1, 3), base + 2000.0)
    finally:
        dist.destroy_process_group()


@gpu
@pytest.mark.parametrize("tiled", [False, True])
def test_frame_step_captured_in_a_hip_graph(miro, tiled):
    """The whole step (trace -> shadow rays -> indirect trace -> shade [-> untile]) records into a HIP graph on the
    stream it is given -- no hidden synchronisation or allocation inside the C ABI calls -- and replays to the same
    picture."""
    b = product_scene(miro, "teapot")
    fr = mframe.FrameRenderer(b, "teapot", 160, 120, spp=2, tiled=tiled)
    fr.generate()
    fr.step()
    torch.cuda.synchronize()
    ref, counts = fr.d_rgb.clone(), fr.ray_counts()
    g = fr.capture()
    for _ in range(3):
        fr.d_rgb.zero_()
        fr.d_hits.zero_()
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(fr.d_rgb, ref) and fr.ray_counts() == counts


# ---------------------------------------------------------------------------------------------- tiled ray order
@pytest.mark.parametrize("W,rows,spp", [(16, 8, 1), (10, 5, 1), (7, 3, 4), (9, 9, 16), (5, 4, 64), (13, 11, 2), (4, 4, 3),
                                        (1, 1, 1), (3, 17, 8), (64, 2, 32)])
def test_tile_pixel_map_is_a_permutation_of_the_window(W, rows, spp):
    """mr_tile_pixel_map: every pixel of the window exactly once, full blocks cover th x tw squares (one wave each)."""
    m = binding.tile_pixel_map(W, rows, spp)
    assert sorted(m.tolist()) == list(range(W * rows))
    shapes = {1: (8, 8), 2: (8, 4), 4: (4, 4), 8: (4, 2), 16: (2, 2), 32: (2, 1)}
    if spp in shapes and rows >= shapes[spp][0] and W >= shapes[spp][1]:
        th, tw = shapes[spp]
        first = m[:th * tw]
        ys, xs = first // W, first % W
        assert ys.max() == th - 1 and xs.max() == tw - 1 and len(set(first.tolist())) == th * tw
    if spp not in shapes:
        assert np.array_equal(m, np.arange(W * rows))          # image order


@pytest.mark.gpu
@pytest.mark.parametrize("name,W,H,spp,y0,y1", [("teapot", 97, 64, 1, 0, 64), ("sponza", 50, 37, 4, 3, 30), ("bunny", 33, 20, 16, 0, 20),
                                              ("teapot", 40, 24, 2, 8, 16), ("teapot", 31, 9, 64, 0, 9)])
def test_tiled_eye_rays_are_the_same_rays_in_slot_order(miro, name, W, H, spp, y0, y1):
    """mr_gen_eye_rays_tiled writes exactly the rays of mr_gen_eye_rays (bitwise, jitter included), slot p holding the
    spp rays of pixel pixel_of_slot[p]."""
    assert torch.cuda.is_available()
    b = product_scene(miro, name)
    cam = binding.make_camera(*(scenes.SCENES[name][k] for k in ("eye", "lookat", "up", "fov")))
    n = (y1 - y0) * W * spp
    lin = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    til = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    b.gen_eye_rays(cam, W, H, lin, y0=y0, y1=y1, spp=spp, jitter=spp > 1, seed=5)
    b.gen_eye_rays(cam, W, H, til, y0=y0, y1=y1, spp=spp, jitter=spp > 1, seed=5, tiled=True)
    m = torch.from_numpy(binding.tile_pixel_map(W, y1 - y0, spp).astype(np.int64)).cuda()
    want = lin.view(-1, spp, 8)[m].reshape(-1, 8)
    assert torch.equal(til.view(torch.int32), want.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("name,spp", [("sponza", 1), ("bunny", 4), ("teapot", 16)])
def test_tiled_frame_is_the_same_picture(miro, name, spp):
    """The whole step on tiled rays, scattered back with the pixel map: byte-identical framebuffer, whole frame and
    sharded in bands (ragged band heights included); the specular path and a HIP-graph replay likewise."""
    b = product_scene(miro, name)
    W, H = 150, 101
    ref = mframe.FrameRenderer(b, name, W, H, spp=spp)
    ref.generate(); ref.step()
    want = ref.d_rgb.clone()
    fr = mframe.FrameRenderer(b, name, W, H, spp=spp, tiled=True)
    assert fr.tiled
    fr.generate(); fr.step()
    torch.cuda.synchronize()
    assert torch.equal(fr.d_rgb, want)
    assert ref.ray_counts() == fr.ray_counts()
    for world, band in ((2, 8), (3, 5)):
        full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        for rank in range(world):
            bands = mframe.band_rows(H, band, rank, world)
            part = mframe.FrameRenderer(b, name, W, H, spp=spp, bands=bands, tiled=True)
            part.generate(); part.step()
            full[torch.from_numpy(mframe.rows_of(bands)).cuda()] = part.d_rgb.view(-1, W, 3)
        assert torch.equal(full.view(-1, 3), want)
    levels_ref = ref.render_specular(depth=2)
    levels = fr.render_specular(depth=2)
    torch.cuda.synchronize()
    assert levels == levels_ref
    assert torch.allclose(fr.d_rgb, ref.d_rgb, rtol=1e-5, atol=1e-7 * float(want.max()))


def _hit_gather_worker(rank, world, port, H, W, band, spp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = mframe.rows_of(mframe.band_rows(H, band, rank, world))
    # each rank's "hit buffer": spp records of 4 values per pixel, value encodes (row, column, sample, field)
    C = 4 * spp
    local = torch.empty((len(rows), W, C), dtype=torch.float32)
    for i, y in enumerate(rows):
        local[i] = (y * W + torch.arange(W, dtype=torch.float32))[:, None] * C + torch.arange(C, dtype=torch.float32)[None, :]
    full = mframe.gather_framebuffer(local.reshape(-1, C), H, W, band, rank, world)
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_hit_records_gloo_world2(tmp_path):
    """The hit-buffer parity mode of SURVEY.md section 8e: the same single gather carries the ranks' mr_hit records
    (4 * spp float32 values per pixel) instead of the framebuffer; rank 0 gets them in image order."""
    H, W, band, spp = 21, 4, 5, 2
    out = str(tmp_path / "hits.pt")
    mp.spawn(_hit_gather_worker, args=(2, _free_port(), H, W, band, spp, out), nprocs=2, join=True)
    full = torch.load(out, weights_only=True)
    C = 4 * spp
    want = torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1) * C + torch.arange(C, dtype=torch.float32)
    assert full.shape == (H, W, C) and torch.equal(full, want)


@pytest.mark.gpu
def test_sharded_hit_buffers_gather_to_the_unsharded_one(miro):
    """Two shards' hit records through FrameGather's de-interleave (one process standing in for both ranks' send
    buffers) equal the unsharded frame's hit buffer, record for record."""
    W, H, spp, band, world = 96, 50, 2, 8, 2
    b = product_scene(miro, "teapot")
    whole = mframe.FrameRenderer(b, "teapot", W, H, spp=spp)
    whole.generate(); whole.trace_primary()
    g = mframe.FrameGather(H, W, band, 0, world, torch.device("cuda"), channels=4 * spp, always_collective=False)
    parts = []
    for rank in range(world):
        bands = mframe.band_rows(H, band, rank, world)
        fr = mframe.FrameRenderer(b, "teapot", W, H, spp=spp, bands=bands)
        fr.generate(); fr.trace_primary()
        parts.append(fr.d_hits.view(-1, 4 * spp))
    for rank in range(world):                                # what dist.gather would have delivered to rank 0
        g.recv[rank, :parts[rank].shape[0]] = parts[rank]
    g.pending = True
    full = g.wait()
    torch.cuda.synchronize()
    assert torch.equal(full.view(-1, 4).view(torch.int32), whole.d_hits.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("W,rows,spp,channels", [(150, 101, 1, 3), (33, 7, 4, 3), (64, 64, 16, 1), (5, 3, 2, 8), (40, 9, 64, 3)])
def test_untile_pixels_on_the_device_follows_the_pixel_map(miro, W, rows, spp, channels):
    """mr_untile_pixels == scattering with mr_tile_pixel_map, for ragged windows and any channel count."""
    b = product_scene(miro, "testobj")
    slots = torch.arange(W * rows * channels, dtype=torch.float32, device="cuda").view(-1, channels)
    image = torch.full_like(slots, -1.0)
    b.untile_pixels(slots, image, W, rows, spp, channels=channels)
    m = torch.from_numpy(binding.tile_pixel_map(W, rows, spp).astype(np.int64)).cuda()
    want = torch.empty_like(slots)
    want[m] = slots
    torch.cuda.synchronize()
    assert torch.equal(image, want)
    with pytest.raises(miro.MiroError):
        b.untile_pixels(slots, slots, W, rows, spp, channels=channels)
