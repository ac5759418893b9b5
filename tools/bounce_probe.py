"""BASELINE config 3 read as a path trace: bunny 1024x1024 x 16 spp primary rays, one cosine-weighted bounce per diffuse hit
(mr_gen_path_rays, Ray::random), the bounce batch traced in every control-flow mode.  Prints launch times; every mode's hit
buffer must be the default's.   usage: python tools/bounce_probe.py [--scene bunny] [--w 1024 --h 1024 --spp 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402
from ab_modes import checksum, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--w", type=int, default=1024)
    ap.add_argument("--h", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    stream = torch.cuda.current_stream()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    n = a.w * a.h * a.spp
    rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
    out = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    ow = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    op = torch.empty(n, dtype=torch.int32, device="cuda")
    oi = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    # one untimed pass: the first launch of a kernel pays for loading it
    sc.gen_eye_rays(cam, a.w, a.h, rays, spp=a.spp, jitter=True, tiled=True, stream=stream)
    sc.trace_device(rays, n, hits, stream=stream)
    sc.gen_path_rays(rays, hits, None, None, None, n, out, ow, op, oi, cnt, spp=a.spp, kinds=binding.MR_PATH_DIFFUSE, stream=stream)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(stream)
    sc.gen_eye_rays(cam, a.w, a.h, rays, spp=a.spp, jitter=True, tiled=True, stream=stream)
    e[1].record(stream)
    sc.trace_device(rays, n, hits, stream=stream)
    e[2].record(stream)
    sc.gen_path_rays(rays, hits, None, None, None, n, out, ow, op, oi, cnt, spp=a.spp, kinds=binding.MR_PATH_DIFFUSE, stream=stream)
    e[3].record(stream)
    torch.cuda.synchronize()
    m = int(cnt.item())
    print("%s %dx%dx%d: eye rays %.3f ms, primary trace %.3f ms (%.1f Grays/s), bounce generation %.3f ms -> %d bounce rays" %
          (a.scene, a.w, a.h, a.spp, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), n / e[1].elapsed_time(e[2]) / 1e6,
           e[2].elapsed_time(e[3]), m))
    ref = None
    for name, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT), ("persistent", miro_amd.MR_TRACE_PERSISTENT),
                     ("incoherent+persistent", miro_amd.MR_TRACE_INCOHERENT | miro_amd.MR_TRACE_PERSISTENT)):
        ms, cs, _ = timed(sc, out, m, fl, a.reps, stream)
        ref = cs if ref is None else ref
        print("  bounce batch, %-22s %8.3f ms  %6.2f Grays/s%s" % (name, ms, m / ms / 1e6, "" if cs == ref else "  DIFFERENT HITS"))


if __name__ == "__main__":
    main()
