"""One kind of trace launch per process, for rocprofv3 --pmc passes (tools/pmc_set.sh): the counted kernel is the only
trace_kernel instantiation of its name in the process, so per-dispatch counters need no guessing.

usage: python3 tools/pmc_probe.py --what primary|shadow|random [--spp 1] [--tiled] [--reps 3] [--n 16777216] [--flags 0]
Helper launches that must precede the counted ones (the primary trace that feeds the shadow generator) run with
MR_MATH_PRODUCT, i.e. under a different kernel name."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402


def random_rays(sc, m, seed=7):
    v = sc.arrays()[0]
    lo, hi = np.maximum(v.min(0), -20), np.minimum(v.max(0), 20)
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = torch.zeros((m, 8), device="cuda")
    r[:, 0:3] = torch.rand((m, 3), device="cuda", generator=g) * torch.tensor(hi - lo, device="cuda") + torch.tensor(lo, device="cuda")
    dd = torch.randn((m, 3), device="cuda", generator=g)
    r[:, 4:7] = dd / dd.norm(dim=1, keepdim=True)
    r[:, 7] = 1e12
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="random")
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--tiled", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--n", type=int, default=1 << 24)
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    stream = torch.cuda.current_stream()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    helper = miro_amd.MR_MATH_PRODUCT if not (a.flags & miro_amd.MR_MATH_PRODUCT) else 0
    if a.what == "random":
        n = a.n
        rays = random_rays(sc, n)
    else:
        n = a.w * a.h * a.spp
        rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
        sc.gen_eye_rays(cam, a.w, a.h, rays, spp=a.spp, jitter=a.spp > 1, stream=stream, tiled=a.tiled)
        if a.what == "shadow":
            hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
            sc.trace_device(rays, n, hits, helper, stream=stream)
            sh = torch.empty((n, 8), dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            sc.gen_shadow_rays(rays, hits, n, d["light"], sh, None, cnt, stream=stream)
            torch.cuda.synchronize()
            n = int(cnt.item())
            rays = sh
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sc.trace_device(rays, n, out, a.flags, stream=stream)
    e0.record(stream)
    for _ in range(a.reps):
        sc.trace_device(rays, n, out, a.flags, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("PMC_PROBE what=%s scene=%s spp=%d tiled=%d rays=%d launches=%d ms=%.4f mrays_s=%.1f" %
          (a.what, a.scene, a.spp, int(a.tiled), n, a.reps + 1, ms, n / ms / 1e3))


if __name__ == "__main__":
    main()
