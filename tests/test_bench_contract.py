"""bench.py's output contract: ONE JSON line on stdout with the driver's fields, the `roofline` object of the dominant
kernel and the `cpu_baseline` object -- checked on a small frame so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--width", "320", "--height", "184", "--spp", "4", "--cpu-spp", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["rccl_ranks"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "strong" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert j["value"] > 0 and j["ms_per_step"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    # the bound is the resource that limits the kernel (VALU issue, from SQ counters), so the fraction is a fraction
    assert rf["bound"] == "valu_issue" and rf["unit"] == "Ginstr/s" and rf["peak"] == 1228.8
    assert rf["pmc_source"], rf
    assert rf["frac"] is not None and 0.0 < rf["frac"] <= 1.0, rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    assert 0.0 < rf["lane_utilisation"] <= 1.0
    hb = rf["hbm"]
    assert hb["peak_GBps"] == 8000.0 and hb["nominal_algorithmic_GBps"] > 0 and "NOT a bound" in hb["nominal_label"]
    if rf["traffic"] is not None:
        assert 0.0 < hb["hbm_measured_frac"] <= 1.0
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and cb["sample"]
    assert cb["threads_used"] == cb["cores"] and cb["affinity_cores"] >= 1 and cb["host_cpus"] >= 1 and cb["runs"]
    assert j["config"]["resident_bytes_per_gpu"] < 64 << 20        # no ray buffers: framebuffer + scene only
    # rays per step = primary + shadow of the frame actually traced
    assert j["config"]["rays_per_step"] >= 320 * 184 * 4


@pytest.mark.gpu
def test_bench_batched_pipeline_still_runs():
    """`--batched`: the round-1 step (five kernels over resident rays) stays available for comparison; its line carries the
    per-launch times of both trace launches and no VALU fraction (the committed counters belong to the fused kernel)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batched",
                        "--width", "320", "--height", "184", "--spp", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["value"] > 0 and j["config"]["step"].startswith("batched")
    assert set(j["roofline"]["avg_launch_ms"]) == {"primary", "shadow"} and j["roofline"]["frac"] is None
    assert j["config"]["resident_bytes_per_gpu"] > 320 * 184 * 4 * 100


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """The N > 1 flow of bench.py -- rendezvous, interleaved row bands, per-rank frames, the pipelined framebuffer gather,
    max-over-ranks timing, one line from rank 0 -- with two ranks sharing this box's GPU.  The collectives run over gloo
    here (MIRO_DIST_BACKEND, host-staged gather): a rehearsal of the control flow, not a measurement."""
    env = dict(os.environ, MIRO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # the plain command, no launcher: bench.py starts its two ranks itself (before anything touches the GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "184", "--spp", "4"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["dist_backend"] == "gloo" and j["rccl_ranks"] == 0       # a rehearsal says so: no RCCL rank took part
    assert j["config"]["rays_per_step"] >= 320 * 184 * 4          # both ranks' rays are counted
    assert "cpu_baseline" not in j                                # rank 0 times the CPU only at N = 1


def _plain_env(**extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_bench_gpus_n_starts_n_ranks_without_a_launcher():
    """`python bench.py --gpus N` with WORLD_SIZE unset must run N ranks, never one.  No GPU here, so the
    ranks only rendezvous (gloo) and count themselves with an all-reduce."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=_plain_env(MIRO_DIST_BACKEND="gloo"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3 and j["ranks_counted"] == 3 and j["dist_backend"] == "gloo"


def test_bench_gpus_n_refuses_to_measure_fewer_gpus():
    """With the RCCL backend (the measured configuration) and fewer visible GPUs than --gpus the command fails loudly
    instead of wrapping devices or silently printing n_gpus: 1; and a launcher whose rank count disagrees with --gpus is
    an error too."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_plain_env(MIRO_DIST_BACKEND="nccl"))
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr, (r.stdout + r.stderr)[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{") and json.loads(l).get("n_gpus") == 1], r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=_plain_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr, (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_bench_photon_config_prints_one_contract_line():
    """`bench.py --config photon` (BASELINE config 5) on a small frame and small maps: one JSON line with the driver's fields,
    the irradiance kernel's `roofline` (live PMC or the committed counters, device work counters, algorithmic bytes) and the
    oracle-timed `cpu_baseline`."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "photon", "--width", "160", "--height", "90",
                        "--photons", "20000", "--k", "50", "--steps", "1", "--warmup", "1", "--cpu-queries", "400"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["unit"] == "Mqueries/s" and j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["value"] > 0 and j["dtype"] == "f32"
    assert j["config"]["estimates_per_step"] == 2 * 160 * 90 and j["config"]["k"] == 50      # the atrium is closed: every ray hits
    rf = j["roofline"]
    assert rf["bound"] == "valu_issue" and rf["peak"] == 1228.8 and rf["avg_launch_ms"] > 0 and rf["launches_per_step"] == 2
    w = rf["work_per_launch"]
    assert w["queries"] == 160 * 90 and w["blocks_of_63_nodes"] > w["queries"] and w["records_searched"] > 50 * w["queries"]
    assert rf["algorithmic_bytes_per_launch"] >= 32 * (w["records_searched"] + w["records_prepass"])
    if rf["frac"] is not None:
        assert 0.0 < rf["frac"] <= 1.0
    cb = j["cpu_baseline"]
    assert cb["unit"] == "Mqueries/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"


@pytest.mark.gpu
def test_bench_primary_only_mode():
    """`--mode primary` = the -DDISABLE_SHADOWS build: rays counted = primary rays, no shadow ray traced."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "teapot", "--width", "128", "--height", "128", "--spp", "1",
                        "--mode", "primary", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["metric"] == "Mrays/s (primary)" and j["config"]["rays_per_step"] == 128 * 128
    assert j["config"]["shadow_query"].startswith("none")
