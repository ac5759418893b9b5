#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the HIP intersection path on BASELINE config 4
(sponza 1920x1080, 64 spp), with the kernel's roofline and the CPU (SSE) baseline beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one frame's rays, which are generated once and stay resident in HBM:
primary batch (mr_trace) -> shadow batch built on the device (ballot compaction) -> mr_trace_indirect ->
Phong shade into the float framebuffer; with N > 1 the frame's rows are dealt to the ranks in interleaved
bands and the step ends with the single RCCL gather of the framebuffer.  Total work is fixed => "strong".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import frame as mframe  # noqa: E402
from miro_amd import scenes  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes_per_ray(V, T):
    """32 B ray in + 16 B hit out + 24 B per node visit + 36 B per triangle test, with the
    visit/test counts of the reference's scalar traversal for the ray set."""
    return 32.0 + 16.0 + 24.0 * V + 36.0 * T


def reference_counts(scene, fr_args, light):
    """V, T per ray for the bench's own ray set at 1 spp, from the device -DSTATS counters (which equal the
    reference's scalar-build counters: tests/test_gpu_parity.py::test_stats_counters_match_reference)."""
    desc, W, H, bands, seed = fr_args
    fr = mframe.FrameRenderer(scene, desc, W, H, spp=1, bands=bands, jitter=False, seed=seed,
                              flags=miro_amd.MR_COUNT_STATS)
    fr.generate()
    scene.stats()
    fr.trace_primary()
    cp = scene.stats()
    fr.make_shadow_rays()
    fr.trace_shadow()
    cs = scene.stats()
    n_p, n_s = fr.ray_counts()
    del fr
    torch.cuda.empty_cache()
    return (cp[0] / max(n_p, 1), cp[1] / max(n_p, 1)), (cs[0] / max(n_s, 1), cs[1] / max(n_s, 1)), (n_p, n_s)


def cpu_baseline(desc, label, W, H, spp, threads):
    """The reference's SSE packet path (oracle/miro_oracle_sse.c, a port: the reference itself cannot travel or
    be built here) on the host cores, on a bounded sample of the same workload; trace batches only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    s = po.Scene()
    scenes.populate(s, desc)
    s.build(8)
    cam = po.make_camera(desc["eye"], desc["lookat"], desc["up"], desc["fov"])
    rays = po.eye_rays(cam, W, H, spp=spp, jitter=spp > 1, seed=168)
    t0 = time.perf_counter()
    hits, used = s.trace_sse(rays, threads=threads)
    t1 = time.perf_counter()
    sh, _ = s.shadow_rays(rays, hits, desc["light"], sse_order=True)
    t2 = time.perf_counter()
    s.trace_sse(sh, threads=threads)
    t3 = time.perf_counter()
    n = len(rays) + len(sh)
    secs = (t1 - t0) + (t3 - t2)
    # the same path on one thread (SURVEY.md section 8d asks for both), on every spp-th ray of the sample
    r1, s1 = rays[::max(spp, 1)].copy(), sh[::max(spp, 1)].copy()
    t4 = time.perf_counter()
    s.trace_sse(r1, threads=1)
    s.trace_sse(s1, threads=1)
    t5 = time.perf_counter()
    return dict(value=round(n / secs / 1e6, 3), unit="Mrays/s", cores=int(used), kind="port",
                single_thread_mrays_s=round((len(r1) + len(s1)) / (t5 - t4) / 1e6, 3),
                sample="%s %dx%d %d spp: %d primary + %d shadow rays, SSE4.1 packet path (8 tris/leaf), OpenMP "
                       "dynamic chunks of 1024 rays, %.2f s wall" % (label, W, H, spp, len(rays), len(sh), secs))


def load_traffic(workload):
    """HBM bytes per trace launch from the committed PMC profile (profiles/*traffic*.json), or None."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith(".json") and "traffic" in f:
                try:
                    j = json.load(open(os.path.join(pdir, f)))
                except Exception:
                    continue
                if j.get("workload") == workload:
                    best = j
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--band", type=int, default=0,
                    help="rows per interleaved band when sharding the image (0: the largest height <= 8 that gives every "
                         "rank the same number of bands, else 8)")
    ap.add_argument("--fast", action="store_true", help="MR_MATH_FAST (not the parity mode; never the default)")
    ap.add_argument("--product", action="store_true",
                    help="MR_MATH_PRODUCT: slab distances as products with the rounded 1/d (about 25 %% faster; decisions can "
                         "differ from the reference's on 2-ulp ties)")
    ap.add_argument("--any-shadow", action="store_true", help="any-hit shadow batch (opaque scenes only)")
    ap.add_argument("--tiled", action="store_true",
                    help="camera rays in the tiled order of mr_gen_eye_rays_tiled (frames below 64 spp; no effect at 64)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU over RCCL.  MIRO_DIST_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks
    # (ranks then share devices and the small collectives run on the host): never a measured configuration.
    backend = os.environ.get("MIRO_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    desc = scenes.SCENES[a.scene]
    label = scenes.sponza_label() if a.scene == "sponza" else a.scene
    if a.scene == "sponza" and rank == 0:
        scenes.sponza_path()            # generate the stand-in once before the other ranks look for it
    if world > 1:
        dist.barrier()
    scene = miro_amd.Scene(local_dev)
    t_build = time.perf_counter()
    scenes.populate(scene, desc)
    info = scene.build(4)
    t_build = time.perf_counter() - t_build

    W, H, spp = a.width, a.height, a.spp
    if a.band <= 0:      # 1080 rows: 8 ranks -> 5 (27 bands each), 4 -> 6, 2 -> 6; bands of 8 would leave 17 vs 16
        a.band = next((b for b in range(8, 0, -1) if H % b == 0 and (H // b) % world == 0), 8)
    bands = mframe.band_rows(H, a.band, rank, world)
    flags = miro_amd.MR_MATH_FAST if a.fast else (miro_amd.MR_MATH_PRODUCT if a.product else 0)
    stream = torch.cuda.current_stream()

    (Vp, Tp), (Vs, Ts), _ = reference_counts(scene, (desc, W, H, bands, 168), desc["light"])
    Bp, Bs = algorithmic_bytes_per_ray(Vp, Tp), algorithmic_bytes_per_ray(Vs, Ts)

    # multi-GPU: the shard is shaded straight into the gather's send buffer; the gather of frame k is asynchronous and
    # is waited for only when frame k+1 is about to overwrite that buffer, so it overlaps k+1's trace launches
    gather = mframe.FrameGather(H, W, a.band, rank, world, dev) if world > 1 else None
    fr = mframe.FrameRenderer(scene, desc, W, H, spp=spp, bands=bands, jitter=spp > 1, seed=168, flags=flags,
                              rgb=gather.local if gather is not None and len(bands) else None, tiled=a.tiled)
    fr.generate(stream)
    torch.cuda.synchronize()

    def one_step(events=None):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if events is not None else None
        if fr.n:
            if e: e[0].record(stream)
            fr.trace_primary(stream)
            if e: e[1].record(stream)
            fr.make_shadow_rays(stream)
            if e: e[2].record(stream)
            fr.trace_shadow(stream, a.any_shadow)
            if e: e[3].record(stream)
        if gather is not None:
            gather.wait()               # frame k-1 has left the send buffer (and, on rank 0, is de-interleaved)
        if fr.n:
            fr.shade(stream)
            if e: events.append(e)
        if gather is not None:
            gather.start()

    if gather is not None:
        # RCCL sets up its point-to-point channels at the first send/recv between a pair of ranks: do that here, so
        # that a run with --warmup 0 does not time connection set-up (the buffer holds no frame yet; nothing reads it)
        gather.start()
        gather.wait()
    for _ in range(a.warmup):
        one_step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step(events)
    if gather is not None:
        gather.wait()                   # the last frame's gather and de-interleave belong to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    n_p, n_s = fr.ray_counts()
    tot = torch.tensor([float(n_p + n_s), elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        rays_all = tot[0:1].clone()
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        tmax = tot[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays_per_step, elapsed = float(rays_all.item()), float(tmax.item())
    else:
        rays_per_step = float(n_p + n_s)

    # dominant kernel (trace_kernel): live HIP-event durations of its launches on rank 0's stream
    ms_p = [e[0].elapsed_time(e[1]) for e in events]
    ms_s = [e[2].elapsed_time(e[3]) for e in events]
    trace_ms = sum(ms_p) + sum(ms_s)
    alg_bytes = a.steps * (n_p * Bp + n_s * Bs)
    achieved = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0

    if rank == 0:
        workload = "%s %dx%d %dspp primary+shadow" % (label, W, H, spp)
        traffic = load_traffic(workload)
        out = {
            "metric": "Mrays/s (primary+shadow)",
            "value": round(rays_per_step * a.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if label != "sponza" else "real",
            "config": {
                "workload": workload,
                "scene_triangles": int(info.n_triangles), "bvh_nodes": int(info.n_nodes),
                "rays_per_step": int(rays_per_step), "math": "fast" if a.fast else (
                    "exact triangle test, slab distances as products with the rounded 1/d (MR_MATH_PRODUCT)" if a.product else
                    "exact: every quotient of the reference's slab and triangle tests, bit for bit"),
                "shadow_query": "any-hit" if a.any_shadow else "closest-hit (as Phong.cpp:97)",
                "ray_order": "tiled (mr_gen_eye_rays_tiled)" if fr.tiled else "image order",
                "parallelism": "image rows in interleaved bands of %d over %d GPU(s), scene replicated, 1 RCCL gather of the framebuffer" % (a.band, world),
                "resident_bytes_per_gpu": int(fr.bytes_resident() + info.device_bytes),
                "bvh_build_s": round(t_build, 3),
                # context only (other scene size, unstated CPU): the reference's write-up, sponza 512x512 1 spp with
                # shadows, 524 288 rays in 0.166750 s (writeup/A2/Readme.tex:83,98) -- not this workload, so no vs_baseline
                "reference_writeup_mrays_s_sponza_512x512": 3.14,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "trace_kernel (primary + shadow launches, rank 0)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
                "algorithmic_bytes_per_ray": {"primary": round(Bp, 1), "shadow": round(Bs, 1)},
                "reference_visits_per_ray": {"primary": [round(Vp, 3), round(Tp, 3)], "shadow": [round(Vs, 3), round(Ts, 3)]},
                "avg_launch_ms": {"primary": round(sum(ms_p) / len(ms_p), 4), "shadow": round(sum(ms_s) / len(ms_s), 4)},
                "note": "achieved = reference-counted node/triangle bytes over kernel time; the 5 MB scene is "
                        "L2/Infinity-Cache resident, so achieved may exceed what HBM itself delivers (see traffic)",
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("MIRO_CPU_THREADS", "16")))
            out["cpu_baseline"] = cpu_baseline(desc, label, W, H, a.cpu_spp, threads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
