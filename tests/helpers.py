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


def assert_hits_close(got, want, rel=REL_TOL, max_flip_frac=2e-4):
    """Fast mode: same hit/miss decision and primitive, t within `rel` relative, barycentrics within
    `rel` absolute (they live in [-1e-4, 1+1e-4]).  A ray whose reference hit sits within tolerance of an
    accept/reject boundary (edge slack, tMax) may legitimately flip; such rays must be rare and their
    alternative hit must itself be within tolerance in t or be a different primitive at equal-within-tol t."""
    assert got.shape == want.shape
    same_prim = got["prim"] == want["prim"]
    flips = np.nonzero(~same_prim)[0]
    assert flips.size <= max(2, max_flip_frac * len(want)), "too many primitive flips: %d of %d" % (flips.size, len(want))
    idx = np.nonzero(same_prim & (want["prim"] != 0xFFFFFFFF))[0]
    t_err = np.abs(got["t"][idx] - want["t"][idx]) / np.maximum(np.abs(want["t"][idx]), 1e-30)
    assert t_err.size == 0 or t_err.max() <= rel, "t relative error %g" % t_err.max()
    # barycentrics: (x * rcp) vs x / d differ relatively; near zero compare absolutely
    for f in ("beta", "gamma"):
        err = np.abs(got[f][idx] - want[f][idx])
        lim = rel * np.maximum(1.0, np.abs(want[f][idx]))
        assert err.size == 0 or (err <= lim).all(), "%s error %g" % (f, (err - lim).max())
    return flips
