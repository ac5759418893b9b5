"""A/B of the trace kernel's control-flow modes in ONE process: default (while-while), MR_TRACE_INCOHERENT (voting),
MR_TRACE_PERSISTENT, both -- on the bench frame's primary and shadow batches at several spp (image order and tiled)
and on random rays.  Every mode must return the default's hit buffer bit for bit (position-weighted checksum);
prints the launch times.   usage: python tools/ab_modes.py [--spp 1,4,64] [--random 16777216] [--scene sponza]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402
from pmc_probe import random_rays  # noqa: E402


def checksum(t):
    v = t.view(torch.int32).reshape(-1).to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003) + 1
    return int((v * w).sum().item())


def timed(sc, rays, n, flags, reps, stream):
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    sc.trace_device(rays, n, out, flags, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        sc.trace_device(rays, n, out, flags, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, checksum(out), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--spp", default="1,4,16,64")
    ap.add_argument("--random", type=int, default=1 << 24)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--base", type=int, default=0, help="flags common to every mode (e.g. 64 = MR_MATH_PRODUCT)")
    a = ap.parse_args()
    stream = torch.cuda.current_stream()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    modes = [("default", 0), ("vote", miro_amd.MR_TRACE_INCOHERENT), ("persistent", miro_amd.MR_TRACE_PERSISTENT),
             ("pers+vote", miro_amd.MR_TRACE_PERSISTENT | miro_amd.MR_TRACE_INCOHERENT)]
    bad = 0

    def run(label, rays, n):
        nonlocal bad
        ref = None
        line = "%-28s %10d rays:" % (label, n)
        for name, fl in modes:
            ms, cs, _ = timed(sc, rays, n, a.base | fl, a.reps, stream)
            if ref is None:
                ref = cs
            same = cs == ref
            bad += 0 if same else 1
            line += "  %s %.3f ms %.2f Grays/s%s" % (name, ms, n / ms / 1e6, "" if same else " DIFFERENT")
        print(line + "  [checksum %d]" % ref, flush=True)

    cam = binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"])
    for spp in [int(x) for x in a.spp.split(",") if x]:
        for tiled in ((False, True) if spp < 64 else (False,)):
            n = a.w * a.h * spp
            rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
            sc.gen_eye_rays(cam, a.w, a.h, rays, spp=spp, jitter=spp > 1, stream=stream, tiled=tiled)
            run("primary %dspp%s" % (spp, " tiled" if tiled else ""), rays, n)
            hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
            sc.trace_device(rays, n, hits, a.base, stream=stream)
            sh = torch.empty((n, 8), dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            sc.gen_shadow_rays(rays, hits, n, d["light"], sh, None, cnt, stream=stream)
            torch.cuda.synchronize()
            run("shadow  %dspp%s" % (spp, " tiled" if tiled else ""), sh, int(cnt.item()))
            del rays, hits, sh
            torch.cuda.empty_cache()
    if a.random:
        run("random", random_rays(sc, a.random), a.random)
    print("AB_MODES %s" % ("all identical" if bad == 0 else "%d MISMATCHES" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
