"""Whole multi-level frames, batched calls against mr_trace_level (one launch per level): wall time of
FrameRenderer.render_specular either way, on BASELINE config 3 read as a path trace (bunny 1024x1024 x 16 spp, diffuse
bounces) and on the sponza stand-in.   usage: python tools/level_probe.py [--scene bunny --w 1024 --h 1024 --spp 16 --depth 2]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402
from miro_amd import frame as mframe  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--w", type=int, default=1024)
    ap.add_argument("--h", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--kinds", type=int, default=binding.MR_PATH_DIFFUSE)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    fr = mframe.FrameRenderer(sc, d, a.w, a.h, spp=a.spp, tiled=True)
    fr.generate()
    torch.cuda.synchronize()
    res = {}
    for fused, grouped in ((False, False), (False, True), (True, False), (True, True), ("auto", False), ("auto", True)):
        kw = dict(depth=a.depth, path_tracing=True, path_seed=5, path_kinds=a.kinds, fused=fused, group_octants=grouped)
        levels = fr.render_specular(**kw)   # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            levels = fr.render_specular(**kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        rays = sum(n + s for n, s in levels)
        if grouped:
            assert levels == res[fused][1], "ray counts differ with grouped queues"
            err = float((fr.d_rgb - res[fused][2]).abs().max())
            print("  grouped queues: same ray counts per level; max pixel difference %.3g (float-atomic order)" % err)
        else:
            res[fused] = (ms, levels, fr.d_rgb.clone())
        print("%s %dx%dx%d depth %d kinds %d, %-8s %-9s %8.3f ms/frame  %7.2f Grays/s  levels %s" %
              (a.scene, a.w, a.h, a.spp, a.depth, a.kinds, {False: "batched", True: "fused", "auto": "auto"}[fused],
               "grouped" if grouped else "as made", ms, rays / ms / 1e6, levels), flush=True)
    assert res[False][1] == res[True][1], "ray counts differ"
    scale = float(res[False][2].abs().max())
    err = float((res[False][2] - res[True][2]).abs().max())
    print("  same ray counts per level; max pixel difference %.3g of %.3g (float-atomic order)" % (err, scale))


if __name__ == "__main__":
    main()
