"""CPU model of SURVEY section 7's "2-wide -> 4-wide node collapse" on the bench scene: how many box tests, node visits and
triangle tests a closest-hit traversal of the SAME tree makes when every inner node is merged with its inner children
(its up to four grandchildren become its children), against the binary traversal the device runs.  Float64, near-first
by entry distance, culling against the running best t -- a work count, not a parity run (no GPU, no oracle).

usage: python tools/wide_bvh_model.py [--rays 3000] [--scene sponza]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
import miro_amd  # noqa: E402
from miro_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--rays", type=int, default=3000)
    a = ap.parse_args()
    s = miro_amd.Scene(0)
    scenes.populate(s, a.scene)
    s.build(4, host_only=True)
    corners, meta, leaf_prims = s.export_tree()
    v, n, vi, ni = s.arrays()
    tri = v[vi].astype(np.float64)                      # [nt, 3, 3]
    lo, hi = corners[:, :3].astype(np.float64), corners[:, 3:].astype(np.float64)
    d = scenes.SCENES[a.scene]
    # primary rays of the bench camera on a coarse grid (pixel centres)
    eye, look, up = (np.array(d[k], np.float64) for k in ("eye", "lookat", "up"))
    w = eye - look; w /= np.linalg.norm(w)
    u = np.cross(up / np.linalg.norm(up), w); u /= np.linalg.norm(u)
    vv = np.cross(w, u)
    W, H = 1920, 1080
    top = np.tan(d["fov"] * np.pi / 360.0); right = top * W / H
    rng = np.random.default_rng(1)
    px = rng.integers(0, W, a.rays); py = rng.integers(0, H, a.rays)
    dirs = (((px + 0.5) / W * 2 - 1) * right)[:, None] * u + (((py + 0.5) / H * 2 - 1) * top)[:, None] * vv - w
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)

    def box(node, o, inv):
        t0, t1 = (lo[node] - o) * inv, (hi[node] - o) * inv
        return np.minimum(t0, t1).max(), np.maximum(t0, t1).min()

    def leaf(node, o, dr, best):
        first, cnt = meta[node, 1], meta[node, 2]
        tests = 0
        for p in leaf_prims[first:first + cnt]:
            A, B, C = tri[p]
            nrm = np.cross(B - A, C - A)
            den = -dr @ nrm
            tests += 1
            if den == 0:
                continue
            t = ((o - A) @ nrm) / den
            be = (-dr @ np.cross(o - A, C - A)) / den
            ga = (-dr @ np.cross(B - A, o - A)) / den
            if be < -1e-4 or ga < -1e-4 or be + ga > 1 + 1e-4 or t < 0 or t > best:
                continue
            if t < best:
                best = t
        return best, tests

    def children4(node):
        out = []
        for c in (meta[node, 1], meta[node, 2]):
            if meta[c, 0] == 1:
                out.append(c)
            else:
                out += [meta[c, 1], meta[c, 2]]
        return out

    tot = {"bin": np.zeros(3), "wide": np.zeros(3)}      # box tests, node visits, triangle tests
    for o, dr in zip(np.repeat(eye[None], a.rays, 0), dirs):
        inv = 1.0 / dr
        for mode in ("bin", "wide"):
            best = 1e12
            stack = [0]
            c = tot[mode]
            while stack:
                node = stack.pop()
                if meta[node, 0] == 1:
                    best, k = leaf(node, o, dr, best)
                    c[2] += k
                    continue
                c[1] += 1
                kids = [meta[node, 1], meta[node, 2]] if mode == "bin" else children4(node)
                hit = []
                for k in kids:
                    mn, mx = box(k, o, inv)
                    c[0] += 1
                    if not (mn > mx or mn > best or mx < 0):
                        hit.append((mn, k))
                hit.sort(key=lambda x: -x[0])            # far first on the stack, near popped first
                stack += [k for _, k in hit]
    for mode in ("bin", "wide"):
        b, vis, t = tot[mode] / a.rays
        print("%-5s per ray: %.1f box tests, %.1f node visits, %.1f triangle tests" % (mode, b, vis, t))
    b0, v0, _ = tot["bin"] / a.rays
    b1, v1, _ = tot["wide"] / a.rays
    # VALU model of the device loop (exact mode, octant-specialised): 26 per box test + 17 control per binary visit;
    # a 4-wide visit sorts up to four children: ~40 control instructions
    print("VALU model per ray: binary %.0f, 4-wide %.0f (26 per box test; 17 / 40 control per visit)" %
          (26 * b0 + 17 * v0, 26 * b1 + 40 * v1))


if __name__ == "__main__":
    main()
