"""Quick device-side timing of mr_trace on the BASELINE scenes (development aid, not the bench contract).
usage: python tools/perf_probe.py [scene ...] [--w 1920 --h 1080 --spp 4 --reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["teapot", "bunny", "sponza"])
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--leaf", type=int, default=4)
    ap.add_argument("--incoherent", type=int, default=0, help="also time this many random rays")
    ap.add_argument("--no-host", action="store_true")
    ap.add_argument("--tiled", action="store_true", help="camera rays in the tiled order of mr_gen_eye_rays_tiled")
    a = ap.parse_args()
    stream = torch.cuda.current_stream()
    for name in a.scenes:
        d = scenes.SCENES[name]
        sc = miro_amd.Scene(0)
        t0 = time.time()
        scenes.populate(sc, d)
        info = sc.build(a.leaf)
        tb = time.time() - t0
        n = a.w * a.h * a.spp
        d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        d_sh = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        d_shh = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
        sc.gen_eye_rays(cam, a.w, a.h, d_rays, spp=a.spp, jitter=a.spp > 1, stream=stream, tiled=a.tiled)
        sc.trace_device(d_rays, n, d_hits, stream=stream)
        sc.gen_shadow_rays(d_rays, d_hits, n, d["light"], d_sh, None, d_cnt, stream=stream)
        torch.cuda.synchronize()
        ns = int(d_cnt.item())
        print("%s: %d tris, %d nodes (depth %d), build+load %.2fs, %d primary, %d shadow rays" %
              (name, info.n_triangles, info.n_nodes, info.max_depth, tb, n, ns))
        for label, flags in (("exact", 0), ("fast", miro_amd.MR_MATH_FAST)):
            for what, rays, cnt, hits, extra in (("primary", d_rays, n, d_hits, 0), ("shadow", d_sh, ns, d_shh, 0),
                                                 ("shadow-any", d_sh, ns, d_shh, miro_amd.MR_TRACE_ANY)):
                if cnt == 0:
                    continue
                sc.trace_device(rays, cnt, hits, flags | extra, stream=stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(a.reps):
                    sc.trace_device(rays, cnt, hits, flags | extra, stream=stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.reps
                print("  %-5s %-10s %8.3f ms  %9.1f Mrays/s" % (label, what, ms, cnt / ms / 1e3))
        if a.incoherent:
            # incoherent batch: random origins inside the scene box, uniform directions (divergent traversal)
            v = sc.arrays()[0]
            lo, hi = np.maximum(v.min(0), -20), np.minimum(v.max(0), 20)
            g = torch.Generator(device="cuda").manual_seed(7)
            m = a.incoherent
            o = torch.rand((m, 3), device="cuda", generator=g) * torch.tensor(hi - lo, device="cuda") + torch.tensor(lo, device="cuda")
            dd = torch.randn((m, 3), device="cuda", generator=g)
            dd = dd / dd.norm(dim=1, keepdim=True)
            r = torch.zeros((m, 8), device="cuda")
            r[:, 0:3] = o
            r[:, 4:7] = dd
            r[:, 7] = 1e12
            hh = torch.empty((m, 4), device="cuda")
            for label, flags in (("exact", 0), ("pers.", miro_amd.MR_TRACE_PERSISTENT), ("fast", miro_amd.MR_MATH_FAST)):
                sc.trace_device(r, m, hh, flags, stream=stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(a.reps):
                    sc.trace_device(r, m, hh, flags, stream=stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.reps
                print("  %-5s %-10s %8.3f ms  %9.1f Mrays/s" % (label, "incoherent", ms, m / ms / 1e3))
        if a.no_host:
            sc.close()
            continue
        # PCIe-inclusive rate of the host-buffer form of mr_trace (pageable numpy arrays, staged by the library)
        h_rays = d_rays.cpu().numpy().view(miro_amd.RAY_DTYPE).reshape(-1)
        sc.trace(h_rays[:1024])
        t0 = time.time()
        sc.trace(h_rays)
        dt = time.time() - t0
        print("  host-buffer mr_trace (H2D 32 B/ray + D2H 16 B/ray included): %8.1f ms  %9.1f Mrays/s" % (dt * 1e3, n / dt / 1e6))
        sc.close()


if __name__ == "__main__":
    main()
