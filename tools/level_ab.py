"""A/B of mr_trace_level on one queue: the diffuse bounce rays of a bunny frame (or --scene), traced by mr_trace alone, by a
level without children and by a level with path-traced children.   usage: python tools/level_ab.py [--scene bunny ...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--w", type=int, default=1024)
    ap.add_argument("--h", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    n = a.w * a.h * a.spp
    dev = "cuda"
    f32 = dict(dtype=torch.float32, device=dev)
    rays = torch.empty((n, 8), **f32)
    hits = torch.empty((n, 4), **f32)
    cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
    q = (torch.empty((n, 8), **f32), torch.empty((n, 3), **f32), torch.empty(n, dtype=torch.int32, device=dev),
         torch.empty(n, dtype=torch.int32, device=dev))
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    sc.gen_eye_rays(cam, a.w, a.h, rays, spp=a.spp, jitter=True, tiled=True)
    sc.trace_device(rays, n, hits)
    sc.gen_path_rays(rays, hits, None, None, None, n, q[0], q[1], q[2], q[3], cnt, spp=a.spp, kinds=binding.MR_PATH_DIFFUSE)
    m = int(cnt.item())
    L, W = d["light"], d["wattage"]
    rgb = torch.zeros((a.w * a.h, 3), **f32)
    out = (torch.empty((m, 8), **f32), torch.empty((m, 3), **f32), torch.empty(m, dtype=torch.int32, device=dev),
           torch.empty(m, dtype=torch.int32, device=dev))
    cnts = torch.zeros(3, dtype=torch.int64, device=dev)
    h2 = torch.empty((m, 4), **f32)
    print("%s %dx%dx%d: %d bounce rays" % (a.scene, a.w, a.h, a.spp, m))
    for name, fl in (("coherent order kernels", 0), ("voting kernels", binding.MR_TRACE_INCOHERENT)):
        t = timed(lambda: sc.trace_device(q[0], m, h2, fl), a.reps)
        nh = int((h2[:, 1].view(torch.int32) != -1).sum().item())
        print("  %-24s mr_trace alone              %7.3f ms   (%d hits)" % (name, t, nh))
        t = timed(lambda: sc.trace_level(q[0], q[1], q[2], q[3], m, rgb, L, W, children=binding.MR_LEVEL_LAST, d_counts=cnts[:2],
                                         spp=a.spp, flags=fl), a.reps)
        print("  %-24s level, no children          %7.3f ms" % (name, t))
        t = timed(lambda: sc.trace_level(q[0], q[1], q[2], q[3], m, rgb, L, W, children=binding.MR_LEVEL_SPECULAR, d_out_rays=out[0],
                                         d_out_weights=out[1], d_out_pixels=out[2], d_out_count=cnts[2:], d_counts=cnts[:2],
                                         spp=a.spp, flags=fl), a.reps)
        print("  %-24s level, specular children    %7.3f ms" % (name, t))
        t = timed(lambda: sc.trace_level(q[0], q[1], q[2], q[3], m, rgb, L, W, children=binding.MR_LEVEL_PATH, d_out_rays=out[0],
                                         d_out_weights=out[1], d_out_pixels=out[2], d_out_ids=out[3], d_out_count=cnts[2:],
                                         d_counts=cnts[:2], spp=a.spp, flags=fl, kinds=binding.MR_PATH_DIFFUSE), a.reps)
        print("  %-24s level, path-traced children %7.3f ms" % (name, t))


if __name__ == "__main__":
    main()
