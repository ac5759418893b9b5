"""The C-ABI library loads and exports every symbol include/miro_hip.h declares; error behaviour of the
boundary; and -- on a machine without a GPU -- that the product fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "miro_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mr_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(miro):
    L = miro.lib()
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), "libmiro_hip.so does not export %s" % name
    assert set(miro.EXPORTED_SYMBOLS) == set(declared)
    assert b"miro_hip" in L.mr_version()


def test_library_does_not_link_the_oracle(miro):
    """The product must not depend on oracle/ in any form."""
    out = subprocess.check_output(["readelf", "-d", miro.lib_path()]).decode()
    assert "oracle" not in out
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", miro.lib_path()]).decode()
    assert "orc_" not in syms


def test_struct_sizes_match_the_header(miro):
    from miro_amd import binding
    assert miro.RAY_DTYPE.itemsize == 32 and miro.HIT_DTYPE.itemsize == 16
    assert C.sizeof(binding.BuildOpts) == 32
    assert C.sizeof(binding.Camera) == 40
    assert C.sizeof(binding.SceneInfo) == 8 * 4 + 8 + 4 + 12


def test_argument_errors(miro):
    L = miro.lib()
    assert L.mr_scene_create(0, None) == -1
    assert b"NULL" in L.mr_last_error()
    h = C.c_void_p()
    assert L.mr_scene_create(-3, C.byref(h)) == -1
    s = miro.Scene()
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(miro.MiroError) as e:            # index out of range
        s.add_arrays(v, v, np.array([[0, 1, 7]], np.uint32), np.array([[0, 1, 2]], np.uint32))
    assert e.value.status == -1
    with pytest.raises(miro.MiroError) as e:            # trace before build
        s.trace(np.zeros(4, miro.RAY_DTYPE))
    assert e.value.status == -5
    with pytest.raises(miro.MiroError) as e:
        s.export_tree()
    assert e.value.status == -5


def test_round2_entry_points_fail_loudly_without_a_device(miro):
    """mr_render_direct, mr_gen_path_rays and mr_deinterleave_bands on a host_only scene: MR_ERR_STATE, never a CPU path;
    the frame descriptor's size is the header's."""
    from miro_amd import binding
    assert C.sizeof(binding.FrameDesc) == 40 + 4 * 4 + 3 * 4 + 5 * 4 + 28 + 12 + 16
    s = miro.Scene()
    s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    s.build(4, host_only=True)
    L = miro.lib()
    fd = binding.FrameDesc()
    fd.W, fd.H, fd.y1, fd.spp = 4, 4, 4, 1
    dummy = C.c_void_p(16)
    assert L.mr_render_direct(s.h, C.byref(fd), dummy, None, None, None, None) == -5
    assert b"CPU" in L.mr_last_error() or b"device" in L.mr_last_error()
    assert L.mr_gen_path_rays(s.h, dummy, dummy, None, None, None, 4, 1, 1, 0, 7, dummy, dummy, dummy, None, dummy, 16, None, None) == -5
    assert L.mr_order_by_octant(s.h, dummy, None, 4, 0, dummy, None) == -5
    assert L.mr_trace_grouped(s.h, dummy, None, 4, dummy, dummy, 0, 0, None) == -5
    assert L.mr_deinterleave_bands(s.h, dummy, dummy, 4, 4, 2, 2, 2, 3, None) == -5
    # pure host arithmetic works anywhere
    assert miro.band_rows_of(10, 4, 0, 2) == 6 and miro.band_rows_of(10, 4, 1, 2) == 4
    assert miro.band_locate(10, 4, 2, 9) == (0, 5)


def test_scene_is_immutable_after_build(miro):
    s = miro.Scene()
    s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    s.build(4, host_only=True)
    with pytest.raises(miro.MiroError) as e:
        s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    assert e.value.status == -5


def test_no_cpu_fallback(miro):
    """A host_only scene cannot be traced; and without a device the normal build reports MR_ERR_HIP."""
    import torch
    s = miro.Scene()
    s.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
    s.build(4, host_only=True)
    with pytest.raises(miro.MiroError) as e:
        s.trace(np.zeros(4, miro.RAY_DTYPE))
    assert e.value.status == -5 and "no CPU fallback" in str(e.value)
    if not torch.cuda.is_available():
        t = miro.Scene()
        t.add_triangle([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1] * 3)
        with pytest.raises(miro.MiroError) as e:
            t.build(4)
        assert e.value.status == -4 and "no CPU fallback" in str(e.value)


def test_product_sources_never_reference_the_oracle():
    pkg = os.path.join(ROOT, "cse168-raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath or os.sep + "lib" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in txt and "miro_oracle" not in txt and "orc_" not in txt, f
