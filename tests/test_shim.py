"""The C++ host shim (cse168-raytracer_amd/host/miro_shim.hpp): the reference's Scene::trace / BVH::build /
BVH::intersect / HitInfo surface over the C ABI.  CPU: it compiles and links against libmiro_hip.so.
GPU: a C++ program written like the reference's scene code traces through it; HitInfo (t, P, N, object,
material) equals the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from helpers import camera_of, oracle_scene, product_scene
from miro_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_shim_test(tmp_path, miro):
    exe = str(tmp_path / "shim_render")
    lib_dir = os.path.dirname(miro.lib_path())
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "cse168-raytracer_amd", "host"), os.path.join(ROOT, "tests", "cpp", "shim_render.cpp"),
           "-L", lib_dir, "-lmiro_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_shim_compiles_with_plain_gxx(tmp_path, miro):
    """A reference-side build needs only g++, the two headers and the shared library (no hipcc, no torch)."""
    exe = build_shim_test(tmp_path, miro)
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 2            # usage


@pytest.mark.gpu
def test_shim_hitinfo_matches_oracle(tmp_path, oracle, miro):
    exe = build_shim_test(tmp_path, miro)
    d = scenes.SCENES["teapot"]
    a = oracle_scene(oracle, "teapot")
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), 128, 96)
    rays_path, out_path = str(tmp_path / "rays.bin"), str(tmp_path / "out.bin")
    rays.tofile(rays_path)
    floor = ",".join(str(float(x)) for tri in d["floor"] for x in tri)
    r = subprocess.run([exe, scenes._model("teapot.obj"), floor, rays_path, out_path, "200"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rec = np.fromfile(out_path, dtype=np.dtype([("f", "<f4", 8), ("m", "<i4", 2)]))
    assert len(rec) == len(rays)
    want = a.trace(rays)
    hit = want["prim"] != oracle.MISS
    assert np.array_equal(rec["f"][:, 0] == 1.0, hit)
    assert np.array_equal(rec["f"][:, 1].view(np.uint32), want["t"].view(np.uint32))          # t (tMax on a miss)
    assert np.array_equal(rec["m"][hit, 0], want["prim"][hit].astype(np.int32))                # hit.object
    assert (rec["m"][hit, 1] == 7).all() and (rec["m"][~hit, 0] == -1).all()                   # hit.material
    P, N = a.hit_attrs(want)
    assert np.array_equal(rec["f"][hit, 2:5].view(np.uint32), P[hit].view(np.uint32))          # Triangle.cpp:160
    ln = np.sqrt((N[:, 0] * N[:, 0] + N[:, 1] * N[:, 1]) + N[:, 2] * N[:, 2]).astype(np.float32)
    Nn = N * (np.float32(1) / ln)[:, None]                                                     # Scene.cpp:262
    assert np.array_equal(rec["f"][hit, 5:8].view(np.uint32), Nn[hit].view(np.uint32))


@pytest.mark.gpu
def test_shim_spheres_and_planes(tmp_path, oracle, miro):
    """makeSpiralScene written against the shim's Sphere / Plane / Triangle classes (assignment1.cpp:31-72): HitInfo
    of every kind of object equals the oracle's, planes come back as the unbounded object they are."""
    exe = build_shim_test(tmp_path, miro)
    d = scenes.SCENES["spiral"]
    a = oracle_scene(oracle, "spiral")
    rays = oracle.eye_rays(camera_of(oracle, "spiral"), 128, 96)
    rays_path, out_path, obj_path = str(tmp_path / "rays.bin"), str(tmp_path / "out.bin"), str(tmp_path / "objs.txt")
    rays.tofile(rays_path)
    hexf = lambda x: float(np.float32(x)).hex()
    with open(obj_path, "w") as fh:
        for o in d["objects"]:
            if o[0] == "sphere":
                fh.write("s %s %s\n" % (" ".join(hexf(c) for c in o[1]), hexf(o[2])))
            elif o[0] == "plane":
                fh.write("p %s %s\n" % (" ".join(hexf(c) for c in o[1]), " ".join(hexf(c) for c in o[2])))
            else:
                fh.write("t %s %s\n" % (" ".join(hexf(c) for c in o[1]), " ".join(hexf(c) for c in o[2])))
    r = subprocess.run([exe, "@" + obj_path, "-", rays_path, out_path, "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rec = np.fromfile(out_path, dtype=np.dtype([("f", "<f4", 8), ("m", "<i4", 2)]))
    want = a.trace(rays)
    hit = want["prim"] != oracle.MISS
    plane = hit & ((want["prim"] & 0x80000000) != 0)
    bounded = hit & ~plane
    assert plane.any() and bounded.any()
    assert np.array_equal(rec["f"][:, 0] == 1.0, hit)
    assert np.array_equal(rec["f"][:, 1].view(np.uint32), want["t"].view(np.uint32))
    assert np.array_equal(rec["m"][bounded, 0], want["prim"][bounded].astype(np.int32))
    assert np.array_equal(rec["m"][plane, 0], -2 - (want["prim"][plane] & 0x7FFFFFFF).astype(np.int32))
    n_sph = sum(1 for o in d["objects"] if o[0] == "sphere")
    assert (rec["m"][bounded, 1] == np.where(want["prim"][bounded] < n_sph, 8, 7)).all() and (rec["m"][plane, 1] == 9).all()
    P, N = a.hit_attrs(want, rays)
    assert np.array_equal(rec["f"][hit, 2:5].view(np.uint32), P[hit].view(np.uint32))
    ln = np.sqrt((N[:, 0] * N[:, 0] + N[:, 1] * N[:, 1]) + N[:, 2] * N[:, 2]).astype(np.float32)
    Nn = N * (np.float32(1) / ln)[:, None]                                                     # Scene.cpp:262
    assert np.array_equal(rec["f"][hit, 5:8].view(np.uint32), Nn[hit].view(np.uint32))


def build_abi_frame(tmp_path, miro):
    exe = str(tmp_path / "abi_frame")
    lib_dir = os.path.dirname(miro.lib_path())
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "abi_frame.cpp"),
           "-L", lib_dir, "-lmiro_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def build_abi_frame_multi(tmp_path, miro):
    exe = str(tmp_path / "abi_frame_multi")
    lib_dir = os.path.dirname(miro.lib_path())
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "abi_frame_multi.cpp"),
           "-L", lib_dir, "-lmiro_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lrccl",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_abi_multi_gpu_frame_program_compiles_with_plain_gxx(tmp_path, miro):
    """SURVEY.md section 8e as a C++ program: one process, a scene replica and one mr_render_direct launch per device,
    ncclCommInitAll + one ncclGather + mr_deinterleave_bands (g++, the header, the library, HIP runtime, librccl)."""
    exe = build_abi_frame_multi(tmp_path, miro)
    assert subprocess.run([exe], capture_output=True).returncode == 2          # usage


@pytest.mark.parametrize("H,band,world", [(1080, 5, 8), (1080, 6, 4), (1080, 8, 1), (37, 8, 4), (5, 8, 8), (1080, 16, 3), (64, 1, 7)])
def test_band_arithmetic_of_the_c_abi_matches_the_python_sharding(miro, H, band, world):
    """mr_band_locate / mr_band_rows_of (what abi_frame_multi and the device de-interleave use) against frame.band_rows
    (what bench.py's ranks use): every image row has exactly one owner, rows keep their order inside a shard."""
    from miro_amd import frame as mframe
    owner = {}
    for r in range(world):
        rows = mframe.rows_of(mframe.band_rows(H, band, r, world))
        assert miro.band_rows_of(H, band, r, world) == len(rows)
        for local, y in enumerate(rows):
            owner[int(y)] = (r, local)
    assert sorted(owner) == list(range(H))
    for y in range(H):
        assert miro.band_locate(H, band, world, y) == owner[y]
    with pytest.raises(miro.MiroError):
        miro.band_locate(H, band, world, H)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,band,world,fpp", [(54, 96, 6, 4, 3), (37, 5, 8, 4, 3), (1080, 64, 5, 8, 3), (20, 7, 3, 2, 8)])
def test_deinterleave_bands_on_the_device(miro, H, W, band, world, fpp):
    """mr_deinterleave_bands = FrameGather's row permutation: shards filled with (row, column, channel) codes."""
    import torch
    from miro_amd import frame as mframe
    sc = product_scene(miro, "teapot")
    counts = [miro.band_rows_of(H, band, r, world) for r in range(world)]
    shard_rows = max(counts) + 1                                               # padded shards
    recv = torch.full((world, shard_rows, W, fpp), -1.0, dtype=torch.float32, device="cuda")
    want = torch.empty((H, W, fpp), dtype=torch.float32, device="cuda")
    code = (torch.arange(W, device="cuda", dtype=torch.float32)[:, None] * 16 + torch.arange(fpp, device="cuda", dtype=torch.float32)[None, :])
    for r in range(world):
        for local, y in enumerate(mframe.rows_of(mframe.band_rows(H, band, r, world))):
            recv[r, local] = code + float(y) * 4096
            want[int(y)] = code + float(y) * 4096
    full = torch.zeros((H, W, fpp), dtype=torch.float32, device="cuda")
    sc.deinterleave_bands(recv, full, W, H, band, world, shard_rows, fpp)
    torch.cuda.synchronize()
    assert torch.equal(full, want)
    with pytest.raises(miro.MiroError):
        sc.deinterleave_bands(recv, full, W, H, band, world, max(counts) - 1, fpp)


@pytest.mark.gpu
def test_c_abi_multi_gpu_frame_program_on_one_device_equals_abi_frame(tmp_path, miro):
    """abi_frame_multi with one device: the fused launch, a one-rank ncclGather and the device de-interleave give
    abi_frame's file byte for byte (five batched kernels there); the hit-record mode gathers the frame's mr_hit buffer."""
    import torch
    from miro_amd import frame as mframe
    exe1, exen = build_abi_frame(tmp_path, miro), build_abi_frame_multi(tmp_path, miro)
    d = scenes.SCENES["teapot"]
    W, H, spp = 160, 120, 2
    csv = lambda v: ",".join(str(float(x)) for x in v)
    floor = ",".join(str(float(x)) for tri in d["floor"] for x in tri)
    common = [scenes._model("teapot.obj"), floor, str(W), str(H), str(spp), csv(d["eye"]), csv(d["lookat"]), str(d["fov"]),
              csv(d["light"]), str(d["wattage"])]
    a, b, c = str(tmp_path / "one.ppm"), str(tmp_path / "multi.ppm"), str(tmp_path / "multi.hits")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([exe1] + common + [a], capture_output=True, text=True)
    r2 = subprocess.run([exen] + common + [b, "1"], capture_output=True, text=True, env=env)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr + r2.stdout
    assert open(a, "rb").read() == open(b, "rb").read()
    rays_line = lambda out: [l for l in out.splitlines() if l.startswith("rays ")]     # RCCL prints a banner of its own
    assert rays_line(r1.stdout) == rays_line(r2.stdout) and len(rays_line(r1.stdout)) == 1
    r3 = subprocess.run([exen] + common + [c, "1", "0", "hits"], capture_output=True, text=True, env=env)
    assert r3.returncode == 0, r3.stderr + r3.stdout
    got = np.fromfile(c, np.float32).reshape(-1, 4)
    fr = mframe.FrameRenderer(product_scene(miro, "teapot"), "teapot", W, H, spp=spp)
    fr.generate()
    fr.trace_primary()
    torch.cuda.synchronize()
    assert np.array_equal(got.view(np.uint32), fr.d_hits.cpu().numpy().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("world,band", [(4, 0), (8, 5), (3, 7)])
def test_c_abi_multi_gpu_frame_program_with_virtual_ranks(tmp_path, miro, world, band):
    """The N-rank flow of abi_frame_multi with all ranks on this box's one GPU ("virtual": device-to-device copies stand in
    for the ncclGather): per-rank band launches, shard layout and the device de-interleave give the one-rank program's
    picture and hit file byte for byte, for a band height that divides the image and ones that leave ragged shards."""
    exen = build_abi_frame_multi(tmp_path, miro)
    d = scenes.SCENES["teapot"]
    W, H, spp = 160, 120, 2
    csv = lambda v: ",".join(str(float(x)) for x in v)
    floor = ",".join(str(float(x)) for tri in d["floor"] for x in tri)
    common = [scenes._model("teapot.obj"), floor, str(W), str(H), str(spp), csv(d["eye"]), csv(d["lookat"]), str(d["fov"]),
              csv(d["light"]), str(d["wattage"])]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {}
    for mode in ("rgb", "hits"):
        one, many = str(tmp_path / ("one." + mode)), str(tmp_path / ("many." + mode))
        r1 = subprocess.run([exen] + common + [one, "1", "0", mode], capture_output=True, text=True, env=env)
        rn = subprocess.run([exen] + common + [many, str(world), str(band), mode, "virtual"], capture_output=True, text=True, env=env)
        assert r1.returncode == 0 and rn.returncode == 0, r1.stderr + rn.stderr + rn.stdout
        assert open(one, "rb").read() == open(many, "rb").read(), mode
        rays = lambda o: [l for l in o.splitlines() if l.startswith("rays ")]
        assert rays(r1.stdout) == rays(rn.stdout)
        out[mode] = rn.stdout
    assert ("devices %d" % world) in out["rgb"]


def test_c_abi_frame_program_compiles_with_plain_gxx(tmp_path, miro):
    """INTEGRATION.md section 3 as a program: the whole frame through the C ABI from C++ (g++, the header, the library
    and the HIP runtime for device buffers -- no hipcc, no Python)."""
    exe = build_abi_frame(tmp_path, miro)
    assert subprocess.run([exe], capture_output=True).returncode == 2          # usage


@pytest.mark.gpu
def test_c_abi_frame_program_renders_the_oracles_picture(tmp_path, oracle, miro):
    exe = build_abi_frame(tmp_path, miro)
    d = scenes.SCENES["teapot"]
    W, H, spp = 160, 120, 2
    csv = lambda v: ",".join(str(float(x)) for x in v)
    out = str(tmp_path / "frame.ppm")
    floor = ",".join(str(float(x)) for tri in d["floor"] for x in tri)
    r = subprocess.run([exe, scenes._model("teapot.obj"), floor, str(W), str(H), str(spp), csv(d["eye"]), csv(d["lookat"]),
                        str(d["fov"]), csv(d["light"]), str(d["wattage"]), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = open(out, "rb").read()
    header = ("P6\n%d %d\n255\n" % (W, H)).encode()
    assert raw.startswith(header)
    img = np.frombuffer(raw[len(header):], np.uint8).reshape(H * W, 3)
    # the oracle's picture of the same frame
    a = oracle_scene(oracle, "teapot")
    rays = oracle.eye_rays(camera_of(oracle, "teapot"), W, H, spp=spp, jitter=True, seed=168)
    hits = a.trace(rays)
    sh, src = a.shadow_rays(rays, hits, d["light"])
    occ = np.zeros(len(rays), np.uint8)
    occ[src.astype(np.int64)] = a.trace(sh)["prim"] != oracle.MISS
    want = oracle.tonemap(a.shade_direct(rays, hits, occ, d["light"], d["wattage"], spp=spp))
    assert r.stdout.split() == ["rays", str(len(rays)), str(len(sh))]
    assert np.abs(img.astype(np.int32) - want.astype(np.int32)).max() <= 1     # expf vs exp at a quantisation boundary
    assert len(np.unique(img)) > 4 and (img == img[0]).all(axis=1).mean() < 0.9   # an actual (if dim: 700 W at 15 m) picture
