#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the HIP intersection path on BASELINE config 4
(sponza 1920x1080, 64 spp), with the dominant kernel's roofline and the CPU (SSE) baseline beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one frame, everything inside the timed region: mr_render_direct generates every
primary ray from the camera (Camera::eyeRay), traces it, builds and traces the shadow ray of every hit and shades the
sample, in ONE launch of frame_kernel (no ray buffers; `--batched` runs the round-1 pipeline of five kernels over
resident ray buffers instead).  With N > 1 the frame's rows are dealt to the ranks in interleaved bands and every step
ends with the single RCCL gather of the framebuffer.  Total work is fixed => "strong".  Rank 0 prints ONE JSON line.

The `roofline` object states the resource that bounds the kernel -- VALU issue -- from SQ counters: achieved = wave-level
VALU instructions per launch (SQ_INSTS_VALU, collected by a rocprofv3 --pmc pass over this same workload started from
inside this run; the committed profiles/r02_bench_pmc.json when the profiler is unavailable) / the launch duration
measured here with HIP events; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction.  The HBM figures of
SURVEY.md section 8(d) ride along, labelled as what they are: the nominal algorithmic-bytes rate (cache-resident, not a
bound) and the measured HBM traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE) as a fraction of the 8 TB/s peak.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import frame as mframe  # noqa: E402
from miro_amd import scenes  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
N_SIMD = 1024                # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4              # MI355X_MICROARCH.md: peak engine clock
VALU_PEAK_GINSTR = N_SIMD * CLOCK_GHZ / 2.0    # a wave64 VALU instruction holds its SIMD-32 for 2 cycles (guide, line 54/473)
ROUND_TAG = "r03"
PMC_FILE = next((f for f in (os.path.join(ROOT, "profiles", "%s_bench_pmc.json" % t) for t in (ROUND_TAG, "r02")) if os.path.exists(f)),
                os.path.join(ROOT, "profiles", "r02_bench_pmc.json"))
SQ_COUNTERS = "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"


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


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max or v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(desc, label, W, H, spp):
    """The reference's SSE packet path (oracle/miro_oracle_sse.c, a port: the reference itself cannot travel or
    be built here) on the host cores this process may use, on a bounded sample of the same workload; trace batches
    only.  Thread counts tried: every core of the affinity mask (capped by the cgroup's CPU quota when there is one)
    and, when that is more than 32, also 16 -- a GPU box of the pool is shared by eight jobs and its documented
    per-GPU CPU share is 16 cores; the best rate is `value`, every run is listed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    allotted = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    env_threads = int(os.environ.get("MIRO_CPU_THREADS", "0"))
    tries = [env_threads] if env_threads else ([allotted] + ([16] if allotted > 32 else []))
    s = po.Scene()
    scenes.populate(s, desc)
    s.build(8)
    cam = po.make_camera(desc["eye"], desc["lookat"], desc["up"], desc["fov"])
    rays = po.eye_rays(cam, W, H, spp=spp, jitter=spp > 1, seed=168)
    runs, sh = [], None
    for threads in tries:
        t0 = time.perf_counter()
        hits, used = s.trace_sse(rays, threads=threads)
        t1 = time.perf_counter()
        if sh is None:
            sh, _ = s.shadow_rays(rays, hits, desc["light"], sse_order=True)
        t2 = time.perf_counter()
        s.trace_sse(sh, threads=threads)
        t3 = time.perf_counter()
        secs = (t1 - t0) + (t3 - t2)
        runs.append(dict(threads=int(used), mrays_s=round((len(rays) + len(sh)) / secs / 1e6, 3), wall_s=round(secs, 3)))
    best = max(runs, key=lambda r: r["mrays_s"])
    # the same path on one thread (SURVEY.md section 8d asks for both), on every spp-th ray of the sample
    r1, s1 = rays[::max(spp, 1)].copy(), sh[::max(spp, 1)].copy()
    t4 = time.perf_counter()
    s.trace_sse(r1, threads=1)
    s.trace_sse(s1, threads=1)
    t5 = time.perf_counter()
    return dict(value=best["mrays_s"], unit="Mrays/s", cores=best["threads"], kind="port",
                threads_used=best["threads"], affinity_cores=affinity, cgroup_quota_cores=quota, host_cpus=os.cpu_count(),
                runs=runs, single_thread_mrays_s=round((len(r1) + len(s1)) / (t5 - t4) / 1e6, 3),
                sample="%s %dx%d %d spp: %d primary + %d shadow rays, SSE4.1 packet path (8 tris/leaf), OpenMP "
                       "dynamic chunks of 1024 rays; value = the best of the listed runs (%d threads, %.2f s wall)"
                       % (label, W, H, spp, len(rays), len(sh), best["threads"], best["wall_s"]))


# ------------------------------------------------------------------------------------------------ PMC (rocprofv3)
def _counter_means(out_dir, kernel_substr):
    """counter -> mean over the dispatches of the kernel whose name holds `kernel_substr`"""
    acc = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row.get("Kernel_Name", ""):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, (max((len(v) for v in acc.values()), default=0))


def live_pmc(argv_leg, kernel_substr, timeout_s=120, passes=None):
    """Three rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE -- the TCC pair does not fit one pass) over
    `python3 bench.py --pmc-leg ...`, i.e. over the same frame this run times.  Returns the counter means per launch of
    the frame kernel, or None when the profiler is missing / fails / times out."""
    passes = passes or (("sq", SQ_COUNTERS), ("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"))
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="miro_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for tag, counters in passes:
            d = os.path.join(tmp, tag)
            cmd = [exe, "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "--", sys.executable,
                                                       os.path.abspath(__file__)] + argv_leg
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            if r.returncode != 0:
                return None, "rocprofv3 pass '%s' failed: %s" % (tag, (r.stderr or r.stdout)[-300:])
            means, n = _counter_means(d, kernel_substr)
            if not means:
                return None, "rocprofv3 pass '%s' recorded no %s dispatch" % (tag, kernel_substr)
            out.update(means)
            out["dispatches_" + tag] = n
        return out, None
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 timed out"
    except Exception as e:      # a missing profiler must never take the bench line down
        return None, "rocprofv3 could not run: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def committed_pmc(workload):
    try:
        j = json.load(open(PMC_FILE))
    except Exception:
        return None
    return j if j.get("per_sample") else None      # per-sample figures: scaled to the launch at hand by the caller


# ------------------------------------------------------------------------------------------------ BASELINE config 5
PHOTON_PASSES = (("sq", SQ_COUNTERS),
                 ("lds", "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"),
                 ("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"))


def photon_cpu_baseline(po, maps_np, P, N, k, max_dist, n_sample):
    """The oracle's Photon_map::irradiance_estimate (a port of PhotonMap.cpp:81-243) on a bounded sample of the same
    queries, both maps, on the host cores this process may use (one chunk of queries per thread; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    threads = int(os.environ.get("MIRO_CPU_THREADS", "0")) or (affinity if quota is None else max(1, min(affinity, int(quota + 0.999))))
    threads = min(threads, 64)
    maps = []
    for pw, pos, d in maps_np:
        m = po.PhotonMap(len(pos) + 10)
        m.store(pw, pos, d)
        m.scale_photon_power(1.0 / len(pos))
        m.balance()
        maps.append(m)
    idx = np.linspace(0, len(P) - 1, n_sample).astype(np.int64)          # spread over the frame
    Ps, Ns = P[idx], N[idx]
    chunks = np.array_split(np.arange(n_sample), threads * 4)

    def work(c):
        for m in maps:
            m.irradiance_estimate(Ps[c], Ns[c], max_dist=max_dist, nphotons=k)
        return len(c)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        done = sum(ex.map(work, chunks))
    secs = time.perf_counter() - t0
    t1 = time.perf_counter()
    one = min(n_sample, 2000)
    for m in maps:
        m.irradiance_estimate(Ps[:one], Ns[:one], max_dist=max_dist, nphotons=k)
    s1 = time.perf_counter() - t1
    return dict(value=round(done * len(maps) / secs / 1e6, 4), unit="Mqueries/s", cores=threads, kind="port",
                threads_used=threads, affinity_cores=affinity, cgroup_quota_cores=quota, host_cpus=os.cpu_count(),
                single_thread_mqueries_s=round(one * len(maps) / s1 / 1e6, 5),
                sample="%d of the frame's %d query points x %d maps, k = %d: the oracle's restated locate_photons / "
                       "irradiance_estimate (oracle/miro_oracle_photon.c), %d threads, %.2f s wall" % (n_sample, len(P), len(maps), k, threads, secs))


def photon_main(a, world, rank, local_dev, dev, red_dev, backend):
    """BASELINE config 5: "sponza.obj + PhotonMap final-gather rays (PhotonMap.cpp kNN lookup as second HIP kernel)".
    Queries = hit point + normalised normal of every primary hit of config 4's frame at 1 spp (Scene.cpp:285-292), built
    on the device by the library; two maps of --photons photons each (Scene.h:67-68), k = PHOTON_SAMPLES = 500,
    max_dist = 1e10 (Miro.h:16-17).  A step = Photon_map::irradiance_estimate for every query on both maps
    (mr_irradiance_estimate x 2; queries resident in HBM).  value = estimates per second, whole job."""
    desc = scenes.SCENES["sponza"]
    label = scenes.sponza_label()
    if rank == 0:
        scenes.sponza_path()
    if world > 1:
        dist.barrier()
    scene = miro_amd.Scene(local_dev)
    scenes.populate(scene, desc)
    info = scene.build(4)
    W, H, k, max_dist = a.width, a.height, a.k, 1e10
    band = next((b for b in range(8, 0, -1) if H % b == 0 and (H // b) % world == 0), 8)
    bands = mframe.band_rows(H, band, rank, world)
    fr = mframe.FrameRenderer(scene, desc, W, H, spp=1, bands=bands, jitter=False)
    fr.generate()
    fr.trace_primary()
    n = fr.n
    v, _, vi, _ = scene.arrays()
    maps_np = [scenes.synthetic_photons(v, vi, a.photons, seed) for seed in (168, 169)]
    t_bal = time.perf_counter()
    maps = []
    for pw, pos, d in maps_np:
        m = miro_amd.PhotonMap(a.photons + 10, device=local_dev)
        m.store(pw, pos, d)
        m.scale_photon_power(1.0 / a.photons)
        m.balance()
        maps.append(m)
    t_bal = time.perf_counter() - t_bal
    # the library's own query construction (gather_queries_kernel): one whole mr_final_gather, whose scratch then holds
    # positions [0,3n) and normals [3n,6n); also the warm-up of the kernels
    scratch = torch.empty(12 * max(n, 1), dtype=torch.float32, device=dev)
    rgb = torch.zeros((max(fr.n_pixels, 1), 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    if n:
        scene.final_gather(maps[0], maps[1], fr.d_rays, fr.d_hits, n, scratch, rgb, max_dist=max_dist, nphotons=k, spp=1, stream=stream)
    P, N = scratch[:3 * n].view(n, 3), scratch[3 * n:6 * n].view(n, 3)
    nq = int((N[:, 0] == N[:, 0]).sum().item()) if n else 0           # NaN normal = no query (miss / non-diffuse hit)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def one_step(events=None):
        for m in maps:
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if events is not None else None
            if e: e[0].record(stream)
            if n:
                m.irradiance_estimate(P, N, n, out, max_dist=max_dist, nphotons=k, stream=stream)
            if e:
                e[1].record(stream)
                events.append(e)

    if a.pmc_leg:
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        print("PMC_LEG queries_per_launch=%d" % nq)
        return
    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step(events)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(nq * len(maps)), elapsed], dtype=torch.float64, device=red_dev)
    ranks_seen = 1
    if world > 1:
        q_all, tmax, ones = tot[0:1].clone(), tot[1:2].clone(), torch.ones(1, dtype=torch.float64, device=red_dev)
        dist.all_reduce(q_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        est_per_step, elapsed, ranks_seen = float(q_all.item()), float(tmax.item()), int(round(float(ones.item())))
    else:
        est_per_step = float(nq * len(maps))
    if rank == 0:
        ms = [e0.elapsed_time(e1) for e0, e1 in events]
        avg_ms = sum(ms) / max(len(ms), 1)
        launch_s = avg_ms * 1e-3
        # work counters of one launch per map (the counting build of the kernel), outside the timed region
        work = {}
        for name, m in zip(("global", "caustic"), maps):
            m.count_stats(True)
            m.irradiance_estimate(P, N, n, out, max_dist=max_dist, nphotons=k, stream=stream)
            work[name] = m.stats()
            m.count_stats(False)
        rec_search = sum(w["records_searched"] for w in work.values()) / len(maps)
        rec_pre = sum(w["records_prepass"] for w in work.values()) / len(maps)
        boxes = sum(w["boxes_measured"] for w in work.values()) / len(maps)
        alg_bytes = 32.0 * (rec_search + rec_pre) + 64.0 * boxes + 24.0 * nq + 12.0 * nq   # records + child boxes + query in + irradiance out
        workload = "%s %dx%d 1spp hits: %d queries x 2 photon maps of %d, k=%d" % (label, W, H, int(est_per_step / len(maps)), a.photons, k)
        pmc, pmc_source, pmc_note = None, None, None
        if world == 1 and not a.no_pmc:
            leg = ["--config", "photon", "--pmc-leg", "--no-cpu-baseline", "--no-pmc", "--width", str(W), "--height", str(H),
                   "--photons", str(a.photons), "--k", str(k), "--steps", "1", "--warmup", "0"]
            pmc, pmc_note = live_pmc(leg, "irradiance_kernel", timeout_s=300, passes=PHOTON_PASSES)
            if pmc:
                pmc_source = "live: rocprofv3 --pmc passes over this workload, started by this run"
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    json.dump({"workload": workload, "kernel": "irradiance_kernel", "queries_per_launch": nq,
                               "counters_per_launch": pmc, "work_counters": work},
                              open(os.path.join(ROOT, "gpurun_out", "%s_photon_pmc%s.json" % (ROUND_TAG, "" if (W, H, a.photons, k) == (1920, 1080, 200000, 500)
                                                                                              else "_%dx%d_%d_k%d" % (W, H, a.photons, k))), "w"), indent=1)
                except Exception:
                    pass
        if pmc is None:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "%s_photon_pmc.json" % ROUND_TAG)))
                sc_ = nq / float(rec["queries_per_launch"])
                pmc = {kk: vv * sc_ for kk, vv in rec["counters_per_launch"].items() if not kk.startswith("dispatches_")}
                pmc_source = "committed: profiles/%s_photon_pmc.json (%s), scaled per query to this launch" % (ROUND_TAG, rec.get("workload"))
            except Exception:
                pmc = None
        roof = {"kernel": "irradiance_kernel (one launch per photon map and step)", "avg_launch_ms": round(avg_ms, 4),
                "launches_per_step": len(maps), "pmc_source": pmc_source,
                "work_per_launch": {"queries": nq, "blocks_of_63_nodes": round(sum(w["blocks"] for w in work.values()) / len(maps)),
                                    "blocks_expanded": round(sum(w["expansions"] for w in work.values()) / len(maps)), "child_boxes_measured": round(boxes),
                                    "records_searched": round(rec_search), "records_prepass": round(rec_pre),
                                    "tightenings": round(sum(w["tightenings"] for w in work.values()) / len(maps)),
                                    "repeated_searches": round(sum(w["repeated_searches"] for w in work.values()) / len(maps)),
                                    "records_per_query": round((rec_search + rec_pre) / max(nq, 1), 1)},
                "algorithmic_bytes_per_launch": round(alg_bytes),
                "algorithmic_definition": "32 B per photon record examined (device counters: block search + reference-order "
                                          "pre-pass) + 64 B per child-block box pair measured + 24 B query in + 12 B irradiance out",
                "hbm": {"peak_GBps": HBM_PEAK_GBPS, "nominal_algorithmic_GBps": round(alg_bytes / launch_s / 1e9, 1) if launch_s > 0 else None,
                        "nominal_label": "algorithmic bytes over kernel time; the 9.6 MB of photon records are L2 / MALL resident, so this is not the bound"}}
        if pmc_note:
            roof["pmc_note"] = pmc_note
        if pmc and pmc.get("SQ_INSTS_VALU") and launch_s > 0:
            valu = pmc["SQ_INSTS_VALU"] / launch_s / 1e9
            roof.update({"bound": "valu_issue", "unit": "Ginstr/s", "peak": round(VALU_PEAK_GINSTR, 1), "achieved": round(valu, 1),
                         "frac": round(valu / VALU_PEAK_GINSTR, 4),
                         "valu_insts_per_query": round(pmc["SQ_INSTS_VALU"] / max(nq, 1), 1),
                         "salu_insts_per_query": round(pmc.get("SQ_INSTS_SALU", 0) / max(nq, 1), 1)})
            if pmc.get("SQ_THREAD_CYCLES_VALU"):
                roof["lane_utilisation"] = round(pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_INSTS_VALU"] / 64.0, 4)
            if pmc.get("SQ_WAVE_CYCLES"):
                wc = pmc["SQ_WAVE_CYCLES"]
                roof["wave_cycle_split"] = {"waiting": round(pmc.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stalled": round(pmc.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
            if pmc.get("SQ_INSTS_LDS") is not None:
                roof["lds"] = {"insts_per_query": round(pmc["SQ_INSTS_LDS"] / max(nq, 1), 1),
                               "bank_conflict_cycle_share": round(pmc.get("SQ_LDS_BANK_CONFLICT", 0) / max(pmc.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4),
                               "issue_stall_share_of_wave_cycles": round(pmc.get("SQ_WAIT_INST_LDS", 0) / max(pmc.get("SQ_WAVE_CYCLES", wc), 1), 4)}
        else:
            roof.update({"bound": "valu_issue", "unit": "Ginstr/s", "peak": round(VALU_PEAK_GINSTR, 1), "achieved": None, "frac": None})
        traffic = None
        if pmc and pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
            traffic = (pmc["FETCH_SIZE"] * 2.0 + pmc["WRITE_SIZE"]) * 1024.0
        roof["traffic"] = round(traffic) if traffic is not None else None
        if traffic is not None and launch_s > 0:
            roof["hbm"]["measured_GBps"] = round(traffic / launch_s / 1e9, 1)
            roof["hbm"]["hbm_measured_frac"] = round(traffic / launch_s / 1e9 / HBM_PEAK_GBPS, 4)
        outj = {"metric": "Mqueries/s (photon-map irradiance estimates, k=%d)" % k, "value": round(est_per_step * a.steps / elapsed / 1e6, 3),
                "unit": "Mqueries/s", "n_gpus": world, "rccl_ranks": ranks_seen if backend == "nccl" or world == 1 else 0,
                "dist_backend": backend if world > 1 else None, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic" if label != "sponza" else "real",
                "config": {"workload": workload, "estimates_per_step": int(est_per_step), "photons_per_map": a.photons, "k": k,
                           "max_dist": max_dist, "scene_triangles": int(info.n_triangles),
                           "step": "mr_irradiance_estimate on the global and on the caustic map for every query (queries resident in HBM)",
                           "photon_maps": "synthetic: photons uniform on the scene's surfaces, cosine-distributed directions, seeds 168 / 169 (SURVEY 8d)",
                           "balance_and_upload_s": round(t_bal, 3),
                           "parallelism": "queries of the frame's interleaved row bands of %d over %d GPU(s), maps replicated, no collective" % (band, world)},
                "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            Ph, Nh = P.cpu().numpy(), N.cpu().numpy()
            keep = Nh[:, 0] == Nh[:, 0]
            outj["cpu_baseline"] = photon_cpu_baseline(po, maps_np, Ph[keep], Nh[keep], k, max_dist, a.cpu_queries)
        print(json.dumps(outj))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relay their output and exit with the worst child status.
    The parent never touches the GPU -- no HIP call, no torch.cuda.is_available() -- and never re-execs itself;
    torch.cuda.device_count() only counts devices (it does not initialise them on this image)."""
    import socket
    backend = os.environ.get("MIRO_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and n_dev < n_ranks:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- one rank per GPU over RCCL needs %d "
                         "(MIRO_DIST_BACKEND=gloo rehearses the control flow with ranks sharing devices)" % (n_ranks, n_dev, n_ranks))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # children inherit stdout / stderr: rank 0 prints the one JSON line itself.  A rank that dies takes the job down:
    # the others would wait for it in a collective forever
    worst, alive, kill_at = 0, set(range(n_ranks)), None
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and worst == 0:
                worst = rc
                sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, rc))
                for q in alive:
                    procs[q].terminate()         # exact child PIDs, nothing by pattern
                kill_at = time.monotonic() + 15.0
        if kill_at is not None and time.monotonic() > kill_at:
            for q in alive:
                procs[q].kill()
            kill_at = None
        time.sleep(0.05)
    raise SystemExit(worst if worst > 0 else (1 if worst else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)    # default: 50 frames (50 x 17 ms: about a second) / 5 photon steps
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--band", type=int, default=0,
                    help="rows per interleaved band when sharding the image (0: the largest height <= 8 that gives every "
                         "rank the same number of bands, else 8)")
    ap.add_argument("--product", action="store_true",
                    help="MR_MATH_PRODUCT: slab distances as products with the rounded 1/d (about 25 %% faster; decisions can "
                         "differ from the reference's on 2-ulp ties)")
    ap.add_argument("--incoherent", action="store_true", help="MR_TRACE_INCOHERENT: the voting control flow")
    ap.add_argument("--any-shadow", action="store_true", help="any-hit shadow rays (opaque scenes only)")
    ap.add_argument("--mode", choices=["primary+shadow", "primary"], default="primary+shadow",
                    help="primary: no shadow rays, the reference's -DDISABLE_SHADOWS build (BASELINE config 2 is 'primary rays only')")
    ap.add_argument("--batched", action="store_true",
                    help="the round-1 pipeline: resident ray buffers, five kernels per step (eye rays generated once, outside "
                         "the step)")
    ap.add_argument("--image-order", action="store_true", help="samples in image order instead of the tiled order (< 64 spp)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 passes (use the committed profiles/rNN_bench_pmc.json)")
    ap.add_argument("--pmc-leg", action="store_true", help=argparse.SUPPRESS)   # internal: the profiled child
    ap.add_argument("--config", choices=["frame", "photon"], default="frame",
                    help="frame: BASELINE config 4 (the default); photon: BASELINE config 5, Photon_map::irradiance_estimate on the "
                         "frame's primary hits")
    ap.add_argument("--photons", type=int, default=200000, help="--config photon: photons per map (Scene.h:67-68)")
    ap.add_argument("--k", type=int, default=500, help="--config photon: PHOTON_SAMPLES (Miro.h:16)")
    ap.add_argument("--cpu-queries", type=int, default=20000, help="--config photon: queries of the CPU baseline sample")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, rendezvous, count them with an all-reduce, print {n_gpus, rccl_ranks} and stop: "
                         "checks the launch path on a box without GPUs (with MIRO_DIST_BACKEND=gloo); renders nothing")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 5 if a.config == "photon" else 50

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            spawn_ranks(a.gpus)          # never returns; nothing above this line has touched the GPU
        world, rank, local_rank = 1, 0, 0
    else:                                # under a launcher (torch.distributed.run) or as one of spawn_ranks' children
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (a.gpus, world))
    # one rank per GPU over RCCL.  MIRO_DIST_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks
    # (ranks then share devices and the small collectives run on the host): never a measured configuration.
    backend = os.environ.get("MIRO_DIST_BACKEND", "nccl")
    if a.rendezvous_only:
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if world > 1:
            dist.init_process_group(backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
        ones = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rccl_ranks": int(ones.item()) if backend == "nccl" or world == 1 else 0,
                              "ranks_counted": int(ones.item()), "dist_backend": backend if world > 1 else None, "rendezvous_only": True}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible -- refusing to wrap devices" % (rank, local_rank, n_dev))
    local_dev = local_rank % n_dev if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if a.config == "photon":
        return photon_main(a, world, rank, local_dev, dev, red_dev, backend)
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
    flags = (miro_amd.MR_MATH_PRODUCT if a.product else 0) | (miro_amd.MR_TRACE_INCOHERENT if a.incoherent else 0)
    stream = torch.cuda.current_stream()
    fused = not a.batched and spp <= 64 and spp & (spp - 1) == 0
    primary_only = a.mode == "primary"
    if primary_only and (not fused or a.product or a.incoherent or a.any_shadow):
        raise SystemExit("--mode primary runs the fused step with the default traversal (no --batched / --product / --incoherent / --any-shadow)")
    tiled = not a.image_order and spp < 64

    # multi-GPU: two send buffers, used by alternate frames, so that frame k's gather (asynchronous, RCCL's own stream)
    # overlaps frame k+1's launch -- the kernel of frame k+1 writes the other buffer
    gathers = [mframe.FrameGather(H, W, a.band, rank, world, dev, scene=scene) for _ in range(2)] if world > 1 else None
    if fused:
        frs = [mframe.FusedFrame(scene, desc, W, H, spp=spp, band=a.band, rank=rank, world=world, jitter=spp > 1, seed=168,
                                 flags=flags, rgb=g.local if g is not None and len(bands) else None, tiled=tiled,
                                 any_shadow=a.any_shadow, no_shadows=primary_only) for g in (gathers or [None])]
    else:
        frs = [mframe.FrameRenderer(scene, desc, W, H, spp=spp, bands=bands, jitter=spp > 1, seed=168, flags=flags,
                                    rgb=g.local if g is not None and len(bands) else None, tiled=tiled)
               for g in (gathers or [None])]
        for fr in frs[:1]:
            fr.generate(stream)
        if len(frs) > 1:        # the second renderer shares the first one's ray / hit buffers, only its rgb target differs
            for k in ("d_rays", "d_hits", "d_shadow_rays", "d_shadow_hits", "d_src", "d_count"):
                setattr(frs[1], k, getattr(frs[0], k))
    fr0 = frs[0]
    torch.cuda.synchronize()
    frame_no = [0]

    def one_step(events=None):
        k = frame_no[0] % len(frs)
        frame_no[0] += 1
        fr, g = frs[k], (gathers[k] if gathers else None)
        if g is not None:
            g.wait()                    # this buffer's previous gather (two frames ago) has left it
        if fr.n:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if events is not None else None
            if fused:
                if e: e[0].record(stream)
                fr.step(stream)
                if e: e[1].record(stream)
            else:
                if e: e[0].record(stream)
                fr.trace_primary(stream)
                if e: e[1].record(stream)
                fr.make_shadow_rays(stream)
                if e: e[2].record(stream)
                fr.trace_shadow(stream, a.any_shadow)
                if e: e[3].record(stream)
                fr.shade(stream)
            if e: events.append(e)
        if g is not None:
            g.start()

    if a.pmc_leg:
        # the profiled child of live_pmc(): the same frame, one untimed + two counted launches, nothing else
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        print("PMC_LEG samples_per_launch=%d" % fr0.n)
        return

    if gathers is not None:
        # RCCL sets up its point-to-point channels at the first send/recv between a pair of ranks: do that here, so
        # that a run with --warmup 0 does not time connection set-up (the buffers hold no frame yet; nothing reads them)
        for g in gathers:
            g.start()
            g.wait()
    for _ in range(a.warmup):
        one_step()
    if gathers is not None:
        for g in gathers:
            g.wait()
    torch.cuda.synchronize()
    if fused:
        for fr in frs:
            fr.d_counts.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step(events)
    if gathers is not None:
        for g in gathers:
            g.wait()                    # the last frames' gathers and de-interleaves belong to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if fused:
        c = sum(fr.d_counts.cpu().numpy().astype(np.int64) for fr in frs)
        n_p, n_s = int(c[0]) // max(a.steps, 1), int(c[1]) // max(a.steps, 1)
    else:
        n_p, n_s = fr0.ray_counts()
    tot = torch.tensor([float(n_p + n_s), elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        rays_all = tot[0:1].clone()
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        tmax = tot[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays_per_step, elapsed = float(rays_all.item()), float(tmax.item())
        ones = torch.ones(1, dtype=torch.float64, device=red_dev)       # every rank that took part adds one
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        if ranks_seen != dist.get_world_size() or ranks_seen != a.gpus:
            raise SystemExit("rank %d: %d ranks answered the all-reduce, --gpus %d" % (rank, ranks_seen, a.gpus))
    else:
        rays_per_step = float(n_p + n_s)
        ranks_seen = 1

    if rank == 0:
        workload = "%s %dx%d %dspp %s" % (label, W, H, spp, a.mode)
        # ---- dominant kernel: live HIP-event durations of its launches on rank 0's stream
        if fused:
            ms_k = [e[0].elapsed_time(e[1]) for e in events]
            kernel_name, launches_per_step = "frame_kernel (eye ray + primary trace + shadow trace + shade, one launch per step)", 1
            avg_ms = {"frame": round(sum(ms_k) / max(len(ms_k), 1), 4)}
            kernel_ms = sum(ms_k)
        else:
            ms_p = [e[0].elapsed_time(e[1]) for e in events]
            ms_s = [e[2].elapsed_time(e[3]) for e in events]
            kernel_name, launches_per_step = "trace_kernel (primary + shadow launches)", 2
            avg_ms = {"primary": round(sum(ms_p) / max(len(ms_p), 1), 4), "shadow": round(sum(ms_s) / max(len(ms_s), 1), 4)}
            kernel_ms = sum(ms_p) + sum(ms_s)
        kernel_s_per_step = kernel_ms * 1e-3 / max(a.steps, 1)
        samples_rank0 = fr0.n

        # ---- SQ / TCC counters of that kernel: live rocprofv3 passes over this workload, else the committed profile
        pmc, pmc_source, pmc_note = None, None, None
        if fused and world == 1 and not a.no_pmc:
            leg = ["--pmc-leg", "--no-cpu-baseline", "--no-pmc", "--scene", a.scene, "--width", str(W), "--height", str(H),
                   "--spp", str(spp), "--steps", "2", "--warmup", "1"] + (["--product"] if a.product else []) + \
                  (["--incoherent"] if a.incoherent else []) + (["--any-shadow"] if a.any_shadow else []) + \
                  (["--image-order"] if a.image_order else []) + ["--mode", a.mode]
            pmc, pmc_note = live_pmc(leg, "frame_kernel")
            if pmc:
                pmc_source = "live: rocprofv3 --pmc passes over this workload, started by this run"
                per = {k: v / samples_rank0 for k, v in pmc.items() if not k.startswith("dispatches_")}
                rec = {"workload": workload, "kernel": "frame_kernel", "samples_per_launch": samples_rank0,
                       "counters_per_launch": {k: v for k, v in pmc.items()}, "per_sample": per,
                       "flags": {"product": a.product, "incoherent": a.incoherent, "any_shadow": a.any_shadow, "tiled": tiled}}
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    default_frame = (a.scene, W, H, spp) == ("sponza", 1920, 1080, 64) and not (a.product or a.incoherent or a.any_shadow or primary_only)
                    name = "%s_bench_pmc.json" % ROUND_TAG if default_frame else "%s_bench_pmc_%s_%dx%d_%dspp.json" % (ROUND_TAG, a.scene, W, H, spp)
                    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
                except Exception:
                    pass
        if pmc is None and fused:        # the committed counters are the frame kernel's: they say nothing about --batched
            rec = committed_pmc(workload)
            if rec:
                pmc = {k: v * samples_rank0 for k, v in rec["per_sample"].items()}
                pmc_source = "committed: profiles/%s (%s), scaled per sample to this launch" % (os.path.basename(PMC_FILE), rec.get("workload"))
        roof = {"bound": "valu_issue", "kernel": kernel_name, "peak": round(VALU_PEAK_GINSTR, 1), "unit": "Ginstr/s",
                "peak_definition": "%d SIMD-32 x %.1f GHz / 2 cycles per wave64 VALU instruction" % (N_SIMD, CLOCK_GHZ),
                "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step, "pmc_source": pmc_source}
        if pmc_note:
            roof["pmc_note"] = pmc_note
        if pmc and pmc.get("SQ_INSTS_VALU"):
            achieved = pmc["SQ_INSTS_VALU"] / kernel_s_per_step / 1e9
            roof.update({"achieved": round(achieved, 1), "frac": round(achieved / VALU_PEAK_GINSTR, 4),
                         "valu_insts_per_launch": round(pmc["SQ_INSTS_VALU"]),
                         "valu_insts_per_64_samples": round(pmc["SQ_INSTS_VALU"] * 64.0 / samples_rank0, 1)})
            if pmc.get("SQ_THREAD_CYCLES_VALU"):
                roof["lane_utilisation"] = round(pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_INSTS_VALU"] / 64.0, 4)
            if pmc.get("SQ_BUSY_CYCLES"):
                # the profiler's own clock: SQ_BUSY_CYCLES / 32 = kernel cycles under the counter pass
                roof["frac_at_profiled_clock"] = round(pmc["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * pmc["SQ_BUSY_CYCLES"] / 32.0), 4)
            if pmc.get("SQ_WAVE_CYCLES"):
                wc = pmc["SQ_WAVE_CYCLES"]
                roof["wave_cycle_split"] = {"waiting": round(pmc.get("SQ_WAIT_ANY", 0) / wc, 3),
                                            "issue_stalled": round(pmc.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
        else:
            roof.update({"achieved": None, "frac": None})
        # HBM: measured traffic (guide: FETCH_SIZE is in KiB and reports half of a wide streaming read on gfx950 -> x2;
        # WRITE_SIZE exact) and the nominal algorithmic figure of SURVEY.md 8(d)
        traffic = None
        if pmc and pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
            traffic = (pmc["FETCH_SIZE"] * 2.0 + pmc["WRITE_SIZE"]) * 1024.0
        roof["traffic"] = round(traffic) if traffic is not None else None
        (Vp, Tp), (Vs, Ts), _ = reference_counts(scene, (desc, W, H, bands, 168), desc["light"])
        Bp, Bs = algorithmic_bytes_per_ray(Vp, Tp), algorithmic_bytes_per_ray(Vs, Ts)
        nominal = (n_p * Bp + n_s * Bs) / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
        roof["hbm"] = {
            "peak_GBps": HBM_PEAK_GBPS,
            "nominal_algorithmic_GBps": round(nominal, 1),
            "nominal_label": "SURVEY 8(d) bytes (32+16+24V+36T per ray, reference-counted V/T) over kernel time: the 5 MB scene "
                             "is cache-resident, so this is NOT a bound and may exceed the HBM peak",
            "algorithmic_bytes_per_ray": {"primary": round(Bp, 1), "shadow": round(Bs, 1)},
            "reference_visits_per_ray": {"primary": [round(Vp, 3), round(Tp, 3)], "shadow": [round(Vs, 3), round(Ts, 3)]},
            "measured_GBps": round(traffic / kernel_s_per_step / 1e9, 1) if traffic is not None else None,
            "hbm_measured_frac": round(traffic / kernel_s_per_step / 1e9 / HBM_PEAK_GBPS, 4) if traffic is not None else None,
        }
        out = {
            "metric": "Mrays/s (%s)" % a.mode,
            "value": round(rays_per_step * a.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "rccl_ranks": ranks_seen if backend == "nccl" or world == 1 else 0,     # 0: a gloo rehearsal, not RCCL
            "dist_backend": backend if world > 1 else None,
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
                "rays_per_step": int(rays_per_step),
                "step": "mr_render_direct: eye rays generated, traced, shadow rays built and traced, samples shaded in one "
                        "launch" if fused else "batched: mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_direct "
                        "over resident rays",
                "math": "exact triangle test, slab distances as products with the rounded 1/d (MR_MATH_PRODUCT)" if a.product else
                        "exact: every quotient of the reference's slab and triangle tests, bit for bit",
                "control_flow": "voting (MR_TRACE_INCOHERENT)" if a.incoherent else "while-while",
                "shadow_query": "none (-DDISABLE_SHADOWS, Phong.cpp:91)" if primary_only else ("any-hit" if a.any_shadow else "closest-hit (as Phong.cpp:97)"),
                "ray_order": "tiled" if (fr0.tiled if hasattr(fr0, "tiled") else False) else "image order",
                "parallelism": "image rows in interleaved bands of %d over %d GPU(s), scene replicated, 1 RCCL gather of the framebuffer" % (a.band, world),
                "resident_bytes_per_gpu": int(sum(fr.bytes_resident() for fr in (frs if fused else frs[:1])) + info.device_bytes),
                "bvh_build_s": round(t_build, 3),
                # context only (other scene size, unstated CPU): the reference's write-up, sponza 512x512 1 spp with
                # shadows, 524 288 rays in 0.166750 s (writeup/A2/Readme.tex:83,98) -- not this workload, so no vs_baseline
                "reference_writeup_mrays_s_sponza_512x512": 3.14,
            },
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, label, W, H, a.cpu_spp)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
