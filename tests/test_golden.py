"""Committed regression vectors (tests/golden/hits/*.npz, written by tests/golden/make_golden.py): per scene the 64x64
probe grid of the reference's counter runs, the shadow rays of its hits and 2000 random rays, with the hit records and
the -DSTATS counters the pinned oracle returned for them.  CPU: the oracle still returns them.  GPU: so does the HIP
path, bit for bit, through the C ABI -- without the oracle in the loop."""
import os

import numpy as np
import pytest

from helpers import oracle_scene, product_scene

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = ["cornell", "teapot", "bunny", "spiral"]


def load(name):
    z = np.load(os.path.join(HERE, "golden", "hits", name + ".npz"))
    return z["rays"], z["hits"], tuple(int(c) for c in z["counters"])


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden_hits(oracle, name):
    rays, hits, ctr = load(name)
    got, got_ctr = oracle_scene(oracle, name).trace(rays.view(oracle.RAY_DTYPE).reshape(-1), counters=True)
    assert np.array_equal(got.view(np.uint32).reshape(-1, 4), hits)
    assert got_ctr == ctr


def test_golden_probe_grid_is_the_counter_runs_grid(oracle):
    """The teapot file starts with eyeRay(8i, 8j, 512, 512): the grid whose counters, added to the 512x512 render's,
    give the reference's recorded -DSTATS totals (tests/golden/kat_counters.json, test_oracle_kat.py)."""
    from helpers import camera_of
    rays, _, _ = load("teapot")
    full = oracle.eye_rays(camera_of(oracle, "teapot"), 512, 512).reshape(512, 512)[::8, ::8].reshape(-1)
    assert np.array_equal(rays[:4096], full.view(np.uint32).reshape(-1, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_device_reproduces_golden_hits(miro, name):
    import torch
    assert torch.cuda.is_available()
    rays, hits, ctr = load(name)
    b = product_scene(miro, name)
    r = rays.view(miro.RAY_DTYPE).reshape(-1)
    for flags in (0, miro.MR_MATH_PRODUCT):
        got = b.trace(r, flags=flags)
        assert np.array_equal(got.view(np.uint32).reshape(-1, 4), hits)
    b.stats()
    b.trace(r, flags=miro.MR_COUNT_STATS)
    assert b.stats() == ctr
