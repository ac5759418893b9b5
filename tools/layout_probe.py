"""Storage order of the device records (MR_LAYOUT_*, mr_build_opts.layout) against the incoherent batches:
the same scene built once per layout in one process; 16 M random rays and the atrium's diffuse-bounce queue traced with the
default flags and with MR_TRACE_INCOHERENT; hit buffers compared bit for bit with the default layout's.
usage: python tools/layout_probe.py [--scene sponza] [--n 16777216] [--reps 5]"""
import argparse
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402
from pmc_probe import random_rays  # noqa: E402

LAYOUTS = [(0, "dfs (default)"), (1, "pairs"), (2, "treelets"), (16, "dfs + aligned leaves"), (17, "pairs + aligned leaves"),
           (18, "treelets + aligned leaves")]


def timed(sc, rays, n, out, flags, reps):
    st = torch.cuda.current_stream()
    sc.trace_device(rays, n, out, flags, stream=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        sc.trace_device(rays, n, out, flags, stream=st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, zlib.crc32(out[:n].cpu().numpy().tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--n", type=int, default=1 << 24)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    d = scenes.SCENES[a.scene]
    ref = {}
    bounce = None
    for layout, name in LAYOUTS:
        sc = miro_amd.Scene(0)
        scenes.populate(sc, d)
        info = sc.build(4, layout=layout)
        if bounce is None:      # the bounce queue of a 1920x1080x4 frame's first level (Ray::random at every hit), made once
            W, H, spp = 1920, 1080, 4
            n0 = W * H * spp
            rays0 = torch.empty((n0, 8), dtype=torch.float32, device="cuda")
            hits0 = torch.empty((n0, 4), dtype=torch.float32, device="cuda")
            cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
            sc.gen_eye_rays(cam, W, H, rays0, spp=spp, jitter=True, tiled=True)
            sc.trace_device(rays0, n0, hits0)
            q = torch.empty((n0, 8), dtype=torch.float32, device="cuda")
            qw = torch.empty((n0, 3), dtype=torch.float32, device="cuda")
            qp = torch.empty(n0, dtype=torch.int32, device="cuda")
            qi = torch.empty(n0, dtype=torch.int32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            sc.gen_path_rays(rays0, hits0, None, None, None, n0, q, qw, qp, qi, cnt, spp=spp, kinds=binding.MR_PATH_DIFFUSE)
            nb = int(cnt.item())
            bounce = q[:nb].clone()
            del rays0, hits0, q, qw, qp, qi
            rnd = random_rays(sc, a.n)
            out = torch.empty((max(a.n, nb), 4), dtype=torch.float32, device="cuda")
            print("%s: %d random rays, %d bounce rays" % (a.scene, a.n, nb))
        print("layout %2d  %-28s device bytes %d" % (layout, name, info.device_bytes))
        for what, rays, n in (("random", rnd, a.n), ("bounce", bounce, bounce.shape[0])):
            for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
                ms, crc = timed(sc, rays, n, out, fl, a.reps)
                key = (what, fname)
                ref.setdefault(key, crc)
                print("    %-7s %-10s %8.3f ms  %6.2f Grays/s  %s" % (what, fname, ms, n / ms / 1e6,
                                                                      "same hits" if crc == ref[key] else "DIFFERENT HITS"))
        del sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
