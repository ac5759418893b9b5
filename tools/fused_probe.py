"""Timing of mr_render_direct variants on one frame (development aid): with / without the ray counters and hit buffers,
tiled vs image order, control-flow flags.   usage: python tools/fused_probe.py [--spp 1] [--scene sponza] [--reps 50]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import frame as mframe, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    sc = miro_amd.Scene(0)
    scenes.populate(sc, a.scene)
    sc.build(4)
    stream = torch.cuda.current_stream()
    for label, kw, counts in (("tiled, counters", dict(tiled=True), True), ("tiled, no counters", dict(tiled=True), False),
                              ("image order, counters", dict(tiled=False), True), ("tiled, hits kept", dict(tiled=True, keep_hits=True), True),
                              ("tiled, incoherent", dict(tiled=True, flags=miro_amd.MR_TRACE_INCOHERENT), True),
                              ("tiled, any-hit shadows", dict(tiled=True, any_shadow=True), True)):
        fr = mframe.FusedFrame(sc, a.scene, a.w, a.h, spp=a.spp, **kw)
        if not counts:
            fr.d_counts = None
        for _ in range(3):
            fr.step(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.reps):
            fr.step(stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print("%-26s %8.4f ms per frame  (%.2f Grays/s at 2 rays per sample)" % (label, ms, 2 * fr.n / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
