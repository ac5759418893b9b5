"""Shared helpers for the parity tests: scene assembly on both sides, ray sets, comparison rules."""
import numpy as np

from miro_amd import scenes


def oracle_scene(po, name, leaf_size=4):
    s = po.Scene()
    scenes.populate(s, name)
    s.build(leaf_size)
    return s


def product_scene(miro, name, leaf_size=4, host_only=False, device=0):
    s = miro.Scene(device)
    scenes.populate(s, name)
    s.build(leaf_size, host_only=host_only)
    return s


def camera_of(mod, name):
    d = scenes.SCENES[name]
    from_mod = mod.make_camera if hasattr(mod, "make_camera") else mod.binding.make_camera
    return from_mod(d["eye"], d["lookat"], d["up"], d["fov"])


def random_rays(dtype, n, lo, hi, seed, tmax=1e12, shell=True):
    """Seeded incoherent rays: origins in/around the box [lo,hi], directions uniform on the sphere."""
    rng = np.random.RandomState(seed)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    ext = hi - lo
    o = lo - 0.25 * ext + rng.rand(n, 3).astype(np.float32) * (1.5 * ext)
    d = rng.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays = np.zeros(n, dtype)
    rays["ox"], rays["oy"], rays["oz"] = o[:, 0], o[:, 1], o[:, 2]
    rays["dx"], rays["dy"], rays["dz"] = d[:, 0], d[:, 1], d[:, 2]
    rays["tmin"] = 0.0
    rays["tmax"] = tmax
    return rays


def assert_hits_bit_exact(got, want):
    """Exact mode: every field identical down to the bit pattern."""
    assert got.dtype == want.dtype and got.shape == want.shape
    gb = got.view(np.uint32).reshape(-1, 4)
    wb = want.view(np.uint32).reshape(-1, 4)
    bad = np.nonzero((gb != wb).any(axis=1))[0]
    assert bad.size == 0, "first mismatches at %s: got %s want %s" % (bad[:5], got[bad[:5]], want[bad[:5]])


REL_TOL = 1e-5   # BASELINE.json north_star: "within 1e-5 relative fp tolerance"


def bary_rounding_bound(mesh, rays, prims, k=16.0):
    """How far beta / gamma of Triangle.cpp:155-156 can move under ANY re-rounding of the same formula.
    beta = (-d . ((o-A) x (C-A))) / (-d . n) sums products of size |d||o-A||C-A| that cancel down to
    |d . n| ~ |edge|^2, so one ulp of each product is amplified by ~|o-A|/|edge| (1e2..1e4 for a camera
    far from a fine mesh).  Returns per-ray absolute bounds (k ulps of the un-cancelled magnitude)."""
    v, _, vi, _ = mesh
    A = v[vi[prims, 0]].astype(np.float64)
    B = v[vi[prims, 1]].astype(np.float64) - A
    Cc = v[vi[prims, 2]].astype(np.float64) - A
    o = np.stack([rays["ox"], rays["oy"], rays["oz"]], 1).astype(np.float64)
    d = np.stack([rays["dx"], rays["dy"], rays["dz"]], 1).astype(np.float64)
    p = o - A
    n = np.cross(B, Cc)
    den = np.abs((d * n).sum(1))
    eps = 2.0 ** -23

    def mag(e):   # sum of absolute values of every product in d . (p x e)
        ap, ae, ad = np.abs(p), np.abs(e), np.abs(d)
        cx = ap[:, 1] * ae[:, 2] + ap[:, 2] * ae[:, 1]
        cy = ap[:, 2] * ae[:, 0] + ap[:, 0] * ae[:, 2]
        cz = ap[:, 0] * ae[:, 1] + ap[:, 1] * ae[:, 0]
        return ad[:, 0] * cx + ad[:, 1] * cy + ad[:, 2] * cz

    # the denominator's own rounding moves the quotient too: relative error ~ eps * sum|d_i n_i| / |d.n|
    den_rel = eps * (np.abs(d) * np.abs(n)).sum(1) / den
    return (k * (eps * mag(Cc) / den + den_rel + eps), k * (eps * mag(B) / den + den_rel + eps))


def assert_hits_close(got, want, mesh, rays, rel=REL_TOL, max_flip_frac=1e-3):
    """MR_MATH_FAST: same hit/miss decision and primitive; t within `rel` relative (the tolerance
    BASELINE.json names); barycentrics within the reference formula's own rounding sensitivity
    (bary_rounding_bound) -- only the default exact mode can promise more, and it is bit-exact.
    A ray whose reference hit sits on an accept/reject boundary (edge slack, tMax) may flip; such
    rays must be rare; returns their indices for the caller to inspect."""
    assert got.shape == want.shape
    same_prim = got["prim"] == want["prim"]
    flips = np.nonzero(~same_prim)[0]
    assert flips.size <= max(2, max_flip_frac * len(want)), "too many primitive flips: %d of %d" % (flips.size, len(want))
    idx = np.nonzero(same_prim & (want["prim"] != 0xFFFFFFFF))[0]
    t_err = np.abs(got["t"][idx] - want["t"][idx]) / np.maximum(np.abs(want["t"][idx]), 1e-30)
    assert t_err.size == 0 or t_err.max() <= rel, "t relative error %g" % t_err.max()
    bb, bg = bary_rounding_bound(mesh, rays[idx], want["prim"][idx])
    for f, lim in (("beta", bb), ("gamma", bg)):
        err = np.abs(got[f][idx].astype(np.float64) - want[f][idx].astype(np.float64))
        lim = lim * np.maximum(1.0, np.abs(want[f][idx])) + rel * 1e-2
        assert err.size == 0 or (err <= lim).all(), "%s error %g over bound" % (f, (err - lim).max())
    return flips
