"""Launch-bound frames (BASELINE config 2: teapot 512x512, 1 spp): one frame step = 5 kernel launches + 1 memset of a
few tens of microseconds each; captured once into a HIP graph (torch.cuda.CUDAGraph over the stream the library
launches on) the step replays as a single submission.
usage: python tools/graph_probe.py [scene] [--w 512 --h 512 --spp 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
import torch
import miro_amd
from miro_amd import frame as mframe, scenes

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="teapot")
ap.add_argument("--w", type=int, default=512)
ap.add_argument("--h", type=int, default=512)
ap.add_argument("--spp", type=int, default=1)
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
sc = miro_amd.Scene(0); scenes.populate(sc, a.scene); sc.build(4)
fr = mframe.FrameRenderer(sc, a.scene, a.w, a.h, spp=a.spp)
fr.generate(); fr.step(); torch.cuda.synchronize()
ref = fr.d_rgb.clone()
n_p, n_s = fr.ray_counts()

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.reps * 1e3

eager = timed(lambda: fr.step(torch.cuda.current_stream()))
g = fr.capture()
fr.d_rgb.zero_(); g.replay(); torch.cuda.synchronize()
assert torch.equal(fr.d_rgb, ref), "graph replay differs from eager step"
graph = timed(g.replay)
print("%s %dx%d %d spp: %d rays/step; eager %.3f ms (%.0f Mrays/s), graph %.3f ms (%.0f Mrays/s)" %
      (a.scene, a.w, a.h, a.spp, n_p + n_s, eager, (n_p + n_s) / eager / 1e3, graph, (n_p + n_s) / graph / 1e3))
