"""Wavefront frame driver over the C ABI: eye rays -> mr_trace -> shadow rays (ballot compaction) ->
mr_trace_indirect -> Phong shade, for a set of image rows; plus the image-tile sharding and the single
framebuffer gather used on a multi-GPU node.

This replaces the reference's per-pixel recursion (Scene::raytraceImage, Scene.cpp:93-212) by batches;
torch supplies device memory, streams and torch.distributed (backend "nccl" = RCCL) only.
"""
import numpy as np
import torch

from . import binding, scenes


# ------------------------------------------------------------------------------------------------ sharding
def band_rows(H, band, rank, world):
    """Image rows owned by `rank`: horizontal bands of `band` rows dealt round-robin (interleaved for load
    balance, cf. the reference's `schedule(dynamic, 2)` rows, Scene.cpp:113).  Returns [(y0, y1), ...]."""
    out = []
    nb = (H + band - 1) // band
    for b in range(rank, nb, world):
        out.append((b * band, min(H, (b + 1) * band)))
    return out


def rows_of(bands):
    if not bands:
        return np.zeros(0, np.int64)
    return np.concatenate([np.arange(y0, y1, dtype=np.int64) for y0, y1 in bands])


class FrameGather:
    """The one collective of a frame, with everything that does not change from frame to
    frame set up once: every rank owns a send buffer padded to the largest shard -- the renderer shades straight
    into its leading rows (`local`), so there is no staging copy -- and rank `dst` owns the receive buffer, the
    full frame and the row permutation that de-interleaves the bands.

    `start()` enqueues the gather asynchronously (RCCL runs it on its own stream behind the shading kernel);
    `wait()` makes the caller's stream wait for it and, on `dst`, de-interleaves.  A renderer calls `wait()` just
    before it shades the next frame, so frame k's gather overlaps frame k+1's trace launches."""

    def __init__(self, H, W, band, rank, world, device, dtype=torch.float32, group=None, dst=0, always_collective=False,
                 channels=3, scene=None):
        """always_collective: go through torch.distributed even when world == 1 (exercises the RCCL call on one GPU).
        channels: values per pixel -- 3 for the float framebuffer; 4 * spp gathers the frame's mr_hit records instead
        (the hit-buffer parity mode of SURVEY.md section 8e: 16 bytes per ray, viewed as float32)."""
        # scene: a built miro_amd.Scene on `device`; with it rank `dst` de-interleaves with the library's own kernel
        # (mr_deinterleave_bands, one launch on the current stream) instead of a torch index_select + index_copy_ pair
        self.scene = scene if dtype == torch.float32 else None
        self.channels = C = channels
        self.H, self.W, self.band, self.rank, self.world, self.group, self.dst = H, W, band, rank, world, group, dst
        self.local_only = world == 1 and not always_collective
        self.counts = [sum(y1 - y0 for y0, y1 in band_rows(H, band, r, world)) for r in range(world)]
        self.max_rows = max(self.counts)
        self.send = torch.zeros((max(self.max_rows, 1) * W, C), dtype=dtype, device=device)
        self.local = self.send[:self.counts[rank] * W]            # [n_local_rows * W, C], rows in band order
        self.work = None
        self.pending = False
        self.full = None
        if rank == dst:
            self.recv = self.send.unsqueeze(0) if self.local_only else \
                torch.empty((world, max(self.max_rows, 1) * W, C), dtype=dtype, device=device)
            self.full = torch.empty((H, W, C), dtype=dtype, device=device)
            dest = np.concatenate([rows_of(band_rows(H, band, r, world)) for r in range(world)])
            src = np.concatenate([r * max(self.max_rows, 1) + np.arange(self.counts[r], dtype=np.int64) for r in range(world)])
            self.dest_rows = torch.from_numpy(dest).to(device)
            self.src_rows = torch.from_numpy(src).to(device)

    def _staged(self):
        import torch.distributed as dist
        return not self.local_only and self.send.is_cuda and dist.get_backend(self.group) == "gloo"

    def start(self):
        import torch.distributed as dist
        if self.pending:
            self.wait()
        self.pending = True
        if self.local_only:
            return
        if self._staged():
            # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (MIRO_DIST_BACKEND=gloo): gloo has
            # no CUDA gather, so the shard is staged through the host; the production path is the RCCL branch below
            send_h = self.send.cpu()
            recv_h = [torch.empty_like(send_h) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(send_h, recv_h, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                self.recv.copy_(torch.stack(recv_h))
            return
        recv = list(self.recv.unbind(0)) if self.rank == self.dst else None
        self.work = dist.gather(self.send, recv, dst=self.dst, group=self.group, async_op=True)

    def wait(self):
        """Returns the full [H, W, channels] frame on `dst` (valid for work enqueued after this call), None elsewhere."""
        if not self.pending:
            return self.full
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.pending = False
        if self.rank != self.dst:
            return None
        if self.scene is not None and self.recv.is_cuda:
            self.scene.deinterleave_bands(self.recv, self.full, self.W, self.H, self.band, self.world, max(self.max_rows, 1),
                                          self.channels, stream=torch.cuda.current_stream(self.recv.device))
            return self.full
        rows = self.recv.reshape(-1, self.W * self.channels).index_select(0, self.src_rows)
        self.full.view(self.H, self.W * self.channels).index_copy_(0, self.dest_rows, rows)
        return self.full


def gather_framebuffer(local_rgb, H, W, band, rank, world, group=None, dst=0):
    """One-shot form of FrameGather: every rank contributes the rows it rendered ([n_local_rows*W, C] values, rows in
    band order; C = 3 for the framebuffer, 4*spp for hit records viewed as float32); rank `dst` receives them in one
    gather and de-interleaves into [H, W, C]."""
    C = local_rgb.shape[-1]
    g = FrameGather(H, W, band, rank, world, local_rgb.device, local_rgb.dtype, group, dst, channels=C)
    g.local.copy_(local_rgb.reshape(-1, C)[:g.counts[rank] * W])
    g.start()
    return g.wait()


# ------------------------------------------------------------------------------------------------ renderer
class FusedFrame:
    """One rank's share of a direct-light frame rendered by mr_render_direct: a single launch per step, no ray buffers.
    Resident: the float framebuffer shard (12 B per pixel) and, when `keep_hits`, the two hit-record buffers
    (16 + 16 B per sample) that parity tests compare with the batched pipeline's."""

    def __init__(self, scene, desc, W, H, spp=1, band=None, rank=0, world=1, jitter=None, seed=168, flags=0, rgb=None,
                 tiled=None, keep_hits=False, any_shadow=False, no_shadows=False):
        """no_shadows: MR_FRAME_NO_SHADOWS, the reference's -DDISABLE_SHADOWS build (primary rays only, BASELINE config 2).
        A scene with a material table (Scene.set_materials) is shaded with its per-object materials (Phong.cpp:99-156)."""
        if isinstance(desc, str):
            desc = scenes.SCENES[desc]
        self.scene, self.desc, self.W, self.H, self.spp = scene, desc, W, H, spp
        self.jitter = (spp > 1) if jitter is None else jitter
        self.seed = seed
        self.flags = flags | (binding.MR_TRACE_ANY if any_shadow else 0) | (binding.MR_FRAME_NO_SHADOWS if no_shadows else 0)
        self.device = torch.device("cuda", scene.device)
        self.bands = (band, rank, world) if world > 1 else None
        self.n_rows = sum(y1 - y0 for y0, y1 in band_rows(H, band, rank, world)) if world > 1 else H
        self.n_pixels = self.n_rows * W
        self.n = self.n_pixels * spp
        self.tiled = (spp < 64) if tiled is None else bool(tiled)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.d_rgb = torch.zeros((max(self.n_pixels, 1), 3), **f32) if rgb is None else rgb
        if self.d_rgb.numel() < 3 * self.n_pixels or self.d_rgb.dtype != torch.float32 or not self.d_rgb.is_contiguous():
            raise ValueError("rgb must be a contiguous float32 buffer of at least %d x 3 values" % self.n_pixels)
        self.d_hits = torch.empty((max(self.n, 1), 4), **f32) if keep_hits else None
        self.d_shadow_hits = torch.empty((max(self.n, 1), 4), **f32) if keep_hits else None
        self.d_counts = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.cam = binding.make_camera(desc["eye"], desc["lookat"], desc["up"], desc["fov"])

    def bytes_resident(self):
        own = [self.d_rgb, self.d_counts] + [t for t in (self.d_hits, self.d_shadow_hits) if t is not None]
        return sum(t.numel() * t.element_size() for t in own)

    def step(self, stream=None):
        if self.n == 0:
            return
        self.scene.render_direct(self.cam, self.W, self.H, self.d_rgb, self.desc["light"], self.desc["wattage"],
                                 bands=self.bands, spp=self.spp, jitter=self.jitter, seed=self.seed, tiled=self.tiled,
                                 flags=self.flags, d_hits=self.d_hits, d_shadow_hits=self.d_shadow_hits,
                                 d_counts=self.d_counts, stream=stream)

    def ray_counts(self, steps=1):
        """(primary, shadow) rays per step, from the device counters accumulated over `steps` steps -- synchronises."""
        c = self.d_counts.cpu().numpy()
        return int(c[0]) // max(steps, 1), int(c[1]) // max(steps, 1)


class FrameRenderer:
    """All device buffers of one rank's share of a frame, resident for the lifetime of the object."""

    def __init__(self, scene, desc, W, H, spp=1, bands=None, jitter=None, seed=168, flags=0, device=None, rgb=None,
                 tiled=None):
        if isinstance(desc, str):
            desc = scenes.SCENES[desc]
        self.scene, self.desc, self.W, self.H, self.spp = scene, desc, W, H, spp
        self.bands = bands if bands is not None else [(0, H)]
        self.jitter = (spp > 1) if jitter is None else jitter
        self.seed, self.flags = seed, flags
        self.device = torch.device("cuda", scene.device) if device is None else device
        self.n_rows = sum(y1 - y0 for y0, y1 in self.bands)
        self.n_pixels = self.n_rows * W
        self.n = self.n_pixels * spp
        n = max(self.n, 1)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.d_rays = torch.empty((n, 8), **f32)
        self.d_hits = torch.empty((n, 4), **f32)
        self.d_shadow_rays = torch.empty((n, 8), **f32)
        self.d_shadow_hits = torch.empty((n, 4), **f32)
        self.d_src = torch.empty(n, dtype=torch.int32, device=self.device)
        self.d_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        # rgb: an external [n_pixels, 3] buffer to shade into (FrameGather.local on a multi-GPU node)
        self.d_rgb = torch.zeros((max(self.n_pixels, 1), 3), **f32) if rgb is None else rgb
        self.cam = binding.make_camera(desc["eye"], desc["lookat"], desc["up"], desc["fov"])
        # Below 64 samples per pixel the rays can be generated in the tiled order of mr_gen_eye_rays_tiled (a wave covers a
        # square of pixels instead of a strip); tracing, shadow rays and shading are order-agnostic and work on pixel
        # SLOTS, and the shaded slots are scattered to image order (d_rgb) with the window's pixel map.
        # tiled=None: image order (ray k belongs to pixel k // spp in row-major order, what callers indexing d_rays /
        # d_hits expect); tiled=True is the fast choice for whole frames
        self.tiled = bool(tiled) and spp < 64 and spp & (spp - 1) == 0
        if self.tiled and self.n_pixels:
            maps, off = [], 0
            for y0, y1 in self.bands:
                maps.append(binding.tile_pixel_map(W, y1 - y0, spp).astype(np.int64) + off)
                off += (y1 - y0) * W
            self.d_slot_pixel = torch.from_numpy(np.concatenate(maps)).to(self.device)
            self.d_slots = torch.zeros((self.n_pixels, 3), **f32)
        else:
            self.tiled = False
            self.d_slots = self.d_rgb

    def bytes_resident(self):
        own = [self.d_rays, self.d_hits, self.d_shadow_rays, self.d_shadow_hits, self.d_src, self.d_count, self.d_rgb]
        if self.tiled:
            own += [self.d_slots, self.d_slot_pixel]
        return sum(t.numel() * t.element_size() for t in own)

    def generate(self, stream=None):
        """Camera::eyeRay for every owned row; the rays then stay resident in HBM."""
        off = 0
        for y0, y1 in self.bands:
            k = (y1 - y0) * self.W * self.spp
            self.scene.gen_eye_rays(self.cam, self.W, self.H, self.d_rays[off:off + k], y0=y0, y1=y1, spp=self.spp,
                                    jitter=self.jitter, seed=self.seed, stream=stream, tiled=self.tiled)
            off += k

    def trace_primary(self, stream=None):
        self.scene.trace_device(self.d_rays, self.n, self.d_hits, self.flags, stream=stream)

    def make_shadow_rays(self, stream=None):
        self.scene.gen_shadow_rays(self.d_rays, self.d_hits, self.n, self.desc["light"], self.d_shadow_rays, self.d_src,
                                   self.d_count, stream=stream)

    def trace_shadow(self, stream=None, any_hit=False):
        fl = self.flags | (binding.MR_TRACE_ANY if any_hit else 0)
        self.scene.trace_indirect(self.d_shadow_rays, self.d_count, self.n, self.d_shadow_hits, fl, stream=stream)

    def _untile(self, stream=None):
        """slot order -> image order (a no-op for frames generated in image order): the library's own scatter, band by
        band, on the caller's stream (whatever kind of handle it is)"""
        if not self.tiled:
            return
        off = 0
        for y0, y1 in self.bands:
            k = (y1 - y0) * self.W
            self.scene.untile_pixels(self.d_slots[off:off + k], self.d_rgb[off:off + k], self.W, y1 - y0, self.spp, channels=3,
                                     stream=stream)
            off += k

    def shade(self, stream=None):
        self.scene.shade_direct(self.d_rays, self.d_hits, self.n, self.d_shadow_hits, self.d_src, self.d_count,
                                self.desc["light"], self.desc["wattage"], self.d_slots, spp=self.spp, stream=stream)
        self._untile(stream)

    def final_gather(self, global_map, caustic_map=None, nphotons=500, max_dist=1e10, stream=None):
        """BASELINE config 5: the photon-map term of Scene::traceScene (Scene.cpp:285-299) for the primary hits of this
        frame, added to d_rgb after `step()`.  The scratch (48 B per ray) is allocated on first use."""
        if self.n == 0:
            return
        if getattr(self, "d_gather", None) is None:
            self.d_gather = torch.empty(12 * self.n, dtype=torch.float32, device=self.device)
        self.scene.final_gather(global_map, caustic_map, self.d_rays, self.d_hits, self.n, self.d_gather, self.d_slots,
                                max_dist=max_dist, nphotons=nphotons, spp=self.spp, stream=stream)
        self._untile(stream)

    def step(self, stream=None, any_hit=False):
        """One pass of the hot path over this rank's resident rays: primary batch, shadow batch, shade."""
        if self.n == 0:
            return
        self.trace_primary(stream)
        self.make_shadow_rays(stream)
        self.trace_shadow(stream, any_hit)
        self.shade(stream)

    def capture(self, any_hit=False):
        """One frame step recorded into a HIP graph (torch.cuda.CUDAGraph): small frames are launch-bound -- five
        kernels and a memset of tens of microseconds each -- and replay as one submission.  The step must have run
        once before (the library's grow-only scratch buffers are sized outside the capture).  Returns the graph;
        `graph.replay()` re-renders into d_rgb."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self.step(side, any_hit)            # warm-up on the capture stream
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            self.step(torch.cuda.current_stream(self.device), any_hit)
        return g

    def render_specular(self, depth=10, stream=None, path_tracing=False, path_seed=168, path_kinds=None, fused=False,
                        group_octants=False):
        """Scene::traceScene with reflective / refractive materials (Scene.cpp:270-346) as wavefront bounces: every
        level traces its queue, shades it (weight x Phong::shade added to the ray's pixel), and emits the reflect /
        Fresnel / refract children of the next level by ballot compaction.  depth = TRACE_DEPTH (Miro.h:13): rays are
        traced while depth >= 0, i.e. up to depth+1 levels.  Returns the number of rays traced per level.
        path_tracing: the PATH_TRACING build of the generators (mr_gen_path_rays: lobe-sampled children, Ray.h:149-158,
        235-239; path_kinds |= MR_PATH_DIFFUSE adds Ray::random's bounce, an extension) with stable ray ids handed down
        the levels so that every child's random numbers are those of the oracle's recursion.
        fused: every level is ONE launch of mr_trace_level (trace, shadow ray, trace, shade, children) instead of the seven
        of the batched calls -- the same rays and children; no hit or shadow-ray buffer exists.  fused="auto": the first
        level in one launch, a later level only if at least 90 % of the previous level's rays hit something (the one-launch
        form runs its second traversal and its generators at the hit rate of the queue, profiles/r02_level_probe.log).
        group_octants: the generators also write one octant byte per child, and every level after the first works on its
        queue through mr_order_by_octant's index (rays grouped by direction octant inside chunks of 16 384; the same hits and
        children).  Off by default: the bounce rays' own traversal gains 9-13 %, the frame does not (the shadow rays of grouped
        lanes are no more coherent, the pixel runs of accumulate_runs break up, profiles/r03_octant_order.log)."""
        sc, L, W = self.scene, self.desc["light"], self.desc["wattage"]
        # this driver mixes library launches (on `stream`) with torch ops and .item() read-backs: they only order against
        # each other on torch's current stream, so `stream` must be that stream (or a torch Stream, made current here)
        if stream is not None and not isinstance(stream, torch.cuda.Stream):
            raise TypeError("render_specular: pass a torch.cuda.Stream (or None for the current stream), not a raw handle")
        if stream is not None and stream != torch.cuda.current_stream(self.device):
            with torch.cuda.stream(stream):
                return self.render_specular(depth, stream, path_tracing, path_seed, path_kinds, fused, group_octants)
        self.d_slots.zero_()
        if path_kinds is None:
            path_kinds = binding.MR_PATH_MIRROR | binding.MR_PATH_REFRACT
        fan = 4 if path_tracing else 3                  # children per ray, at most
        ids = None
        octs = None                                     # octant bytes of the current queue (None: the eye rays, in image order)
        rays, weights, pixels, n = self.d_rays, None, None, self.n
        per_level = []
        f32 = dict(dtype=torch.float32, device=self.device)
        for level in range(depth + 1):
            if n == 0:
                break
            if fused is True or (fused == "auto" and (level == 0 or per_level[-1][1] >= 0.9 * per_level[-1][0])):
                last = level == depth
                order = None
                if octs is not None:                    # grouped waves share a direction sign: the default control flow
                    order = torch.empty(n, dtype=torch.int32, device=self.device)
                    sc.order_by_octant(rays, n, order, d_octants=octs, stream=stream)
                fl = (self.flags & binding.MR_MATH_PRODUCT) | (binding.MR_TRACE_INCOHERENT if level > 0 and order is None else 0)
                cnts = torch.zeros(3, dtype=torch.int64, device=self.device)        # rays, shadow rays, children
                out_rays = out_w = out_pix = out_ids = out_oct = None
                if not last:
                    out_rays = torch.empty((fan * n, 8), **f32)
                    out_w = torch.empty((fan * n, 3), **f32)
                    out_pix = torch.empty(fan * n, dtype=torch.int32, device=self.device)
                    if path_tracing:
                        out_ids = torch.empty(fan * n, dtype=torch.int32, device=self.device)
                    if group_octants:
                        out_oct = torch.empty(fan * n, dtype=torch.uint8, device=self.device)
                children = binding.MR_LEVEL_LAST if last else (binding.MR_LEVEL_PATH if path_tracing else binding.MR_LEVEL_SPECULAR)
                sc.trace_level(rays, weights, pixels, ids, n, self.d_slots, L, W, children=children, d_out_rays=out_rays,
                               d_out_weights=out_w, d_out_pixels=out_pix, d_out_ids=out_ids,
                               d_out_count=None if last else cnts[2:], d_counts=cnts[:2], spp=self.spp, flags=fl, seed=path_seed,
                               bounce=level, kinds=path_kinds, stream=stream, d_out_octants=out_oct, d_order=order)
                host = cnts.tolist()
                per_level.append((n, host[1]))
                if last:
                    break
                n = host[2]
                rays, weights, pixels = out_rays[:n], out_w[:n], out_pix[:n]
                octs = out_oct[:n] if out_oct is not None else None
                if path_tracing:
                    ids = out_ids[:n]
                continue
            hits = torch.empty((n, 4), **f32)
            sh_rays = torch.empty((n, 8), **f32)
            sh_hits = torch.empty((n, 4), **f32)
            src = torch.empty(n, dtype=torch.int32, device=self.device)
            cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
            # bounce queues are compacted in wave order, not in image order: from the first bounce on the batches carry the
            # MR_TRACE_INCOHERENT hint (voting control flow; the same hit records)
            fl = self.flags | (binding.MR_TRACE_INCOHERENT if level > 0 else 0)
            if octs is not None:
                order = torch.empty(n, dtype=torch.int32, device=self.device)
                sc.trace_grouped(rays, n, hits, order, self.flags & binding.MR_MATH_PRODUCT, d_octants=octs, stream=stream)
            else:
                sc.trace_device(rays, n, hits, fl, stream=stream)
            sc.gen_shadow_rays(rays, hits, n, L, sh_rays, src, cnt, stream=stream)
            sc.trace_indirect(sh_rays, cnt, n, sh_hits, fl, stream=stream)               # closest hit: the occluder matters
            sc.shade_accumulate(rays, hits, weights, pixels, n, sh_rays, sh_hits, src, cnt, L, W, self.d_slots,
                                spp=self.spp, stream=stream)
            n_shadow = int(cnt.item())
            per_level.append((n, n_shadow))
            if level == depth:
                break
            out_rays = torch.empty((fan * n, 8), **f32)
            out_w = torch.empty((fan * n, 3), **f32)
            out_pix = torch.empty(fan * n, dtype=torch.int32, device=self.device)
            cnt2 = torch.zeros(1, dtype=torch.int64, device=self.device)
            out_oct = torch.empty(fan * n, dtype=torch.uint8, device=self.device) if group_octants else None
            if path_tracing:
                out_ids = torch.empty(fan * n, dtype=torch.int32, device=self.device)
                sc.gen_path_rays(rays, hits, weights, pixels, ids, n, out_rays, out_w, out_pix, out_ids, cnt2, spp=self.spp,
                                 seed=path_seed, bounce=level, kinds=path_kinds, stream=stream, d_out_octants=out_oct)
            else:
                sc.gen_secondary_rays(rays, hits, weights, pixels, n, out_rays, out_w, out_pix, cnt2, spp=self.spp, stream=stream,
                                      d_out_octants=out_oct)
            n = int(cnt2.item())
            rays, weights, pixels = out_rays[:n], out_w[:n], out_pix[:n]
            octs = out_oct[:n] if out_oct is not None else None
            if path_tracing:
                ids = out_ids[:n]
        self._untile(stream)
        return per_level

    def ray_counts(self):
        """(primary, shadow) of the last step -- synchronises the whole device (the step may have run on any stream)."""
        if not self.n:
            return 0, 0
        torch.cuda.synchronize(self.device)
        return self.n, int(self.d_count.item())
