"""How much would sorting an incoherent batch buy?  16 M random rays in the atrium, traced as they come and after a
sort by (direction octant, Morton code of the origin cell); the sort itself is timed with torch.sort.
usage: python tools/sort_probe.py [--n 16000000] [--bits 7]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
import numpy as np, torch
import miro_amd
from miro_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16000000)
ap.add_argument("--bits", type=int, default=7)
a = ap.parse_args()
sc = miro_amd.Scene(0); scenes.populate(sc, "sponza"); sc.build(4)
v = sc.arrays()[0]
lo, hi = torch.tensor(v.min(0), device="cuda"), torch.tensor(v.max(0), device="cuda")
g = torch.Generator(device="cuda").manual_seed(7)
m = a.n
r = torch.zeros((m, 8), device="cuda")
r[:, 0:3] = torch.rand((m, 3), device="cuda", generator=g) * (hi - lo) + lo
dd = torch.randn((m, 3), device="cuda", generator=g)
r[:, 4:7] = dd / dd.norm(dim=1, keepdim=True)
r[:, 7] = 1e12
hh = torch.empty((m, 4), device="cuda")

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def key_of(r):
    q = ((r[:, 0:3] - lo) / (hi - lo) * (1 << a.bits)).clamp(0, (1 << a.bits) - 1).to(torch.int64)
    k = torch.zeros(len(r), dtype=torch.int64, device="cuda")
    for b in range(a.bits):
        for ax in range(3):
            k |= ((q[:, ax] >> b) & 1) << (3 * b + ax)
    octant = ((r[:, 4] < 0).to(torch.int64) | ((r[:, 5] < 0).to(torch.int64) << 1) | ((r[:, 6] < 0).to(torch.int64) << 2))
    return (octant << (3 * a.bits)) | k

t_plain = timed(lambda: sc.trace_device(r, m, hh))
keys = key_of(r)
t_sort = timed(lambda: torch.sort(keys))
perm = torch.sort(keys).indices
rs = r[perm].contiguous()
t_sorted = timed(lambda: sc.trace_device(rs, m, hh))
print("%d random rays: as they come %.3f ms (%.0f Mrays/s); sorted %.3f ms (%.0f Mrays/s) + sort of the keys %.3f ms" %
      (m, t_plain, m / t_plain / 1e3, t_sorted, m / t_sorted / 1e3, t_sort))
