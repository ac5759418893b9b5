"""PCIe-inclusive rate of the host-buffer form of mr_trace: pageable numpy buffers against page-locked ones
(mr_host_alloc), on the bench frame's primary rays.  Prints ms and Mrays/s per variant."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    n = a.w * a.h * a.spp
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    sc.gen_eye_rays(binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), a.w, a.h, d_rays, spp=a.spp, jitter=a.spp > 1)
    rays = d_rays.cpu().numpy().view(miro_amd.RAY_DTYPE).reshape(-1)
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    sc.trace_device(d_rays, n, d_hits)
    torch.cuda.synchronize()
    want = d_hits.cpu().numpy().view(miro_amd.HIT_DTYPE).reshape(-1)

    def timed(label, r, h):
        sc.trace(r, hits=h)
        assert h.tobytes() == want.tobytes(), label
        t = time.perf_counter()
        for _ in range(a.reps):
            sc.trace(r, hits=h)
        ms = (time.perf_counter() - t) / a.reps * 1e3
        print("%-28s %8.1f ms  %8.1f Mrays/s  (%.1f GB/s over PCIe, both directions)" % (label, ms, n / ms / 1e3, n * 48 / ms / 1e6))

    print("%s: %d rays (32 B up + 16 B down each)" % (a.scene, n))
    timed("pageable numpy buffers", rays, np.empty(n, miro_amd.HIT_DTYPE))
    pr, ph = miro_amd.PinnedArray(n, miro_amd.RAY_DTYPE), miro_amd.PinnedArray(n, miro_amd.HIT_DTYPE)
    pr.array[:] = rays
    timed("pinned (mr_host_alloc)", pr.array, ph.array)
    timed("pinned rays, pageable hits", pr.array, np.empty(n, miro_amd.HIT_DTYPE))
    pr.close(); ph.close()


if __name__ == "__main__":
    main()
