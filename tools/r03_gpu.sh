#!/bin/bash
# Round-3 GPU sessions (run through gpurun from the repo root): tools/r03_gpu.sh <stage>.  Output under gpurun_out/r03_<stage>/.
set -o pipefail
stage=$1
out=gpurun_out/r03_$stage
mkdir -p $out
LIB=$PWD/cse168-raytracer_amd
case $stage in
  tests)
    python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 6 $out/pytest.log ;;
  level_ab)
    for v in "" _w4; do
      echo "== lib$v" >> $out/level_ab.log
      MIRO_LIB=$LIB/lib$v/libmiro_hip.so python tools/level_probe.py >> $out/level_ab.log 2>&1
      MIRO_LIB=$LIB/lib$v/libmiro_hip.so python tools/level_probe.py --scene sponza --w 1920 --h 1080 --spp 4 >> $out/level_ab.log 2>&1
    done; cat $out/level_ab.log ;;
  primary)
    python bench.py --scene teapot --width 512 --height 512 --spp 1 --mode primary --steps 200 --no-cpu-baseline > $out/teapot_primary.json 2> $out/err.log
    python bench.py --mode primary --steps 20 --no-cpu-baseline > $out/sponza_primary.json 2>> $out/err.log
    cut -c1-300 $out/teapot_primary.json $out/sponza_primary.json ;;
  bench)
    python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-700 $out/bench.json ;;
  photon)
    python -m pytest tests/test_photon.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 4 $out/pytest.log
    python bench.py --config photon > $out/photon.json 2> $out/photon.err; cat $out/photon.json; tail -n 3 $out/photon.err ;;
  photon_ab)
    for v in ${PHOTON_VARIANTS:-default _nopf default _nopf}; do
      d=lib$v; [ "$v" = default ] && d=lib
      echo "== $d" >> $out/photon_ab.log
      MIRO_LIB=$LIB/$d/libmiro_hip.so python tools/photon_probe.py $PHOTON_ARGS >> $out/photon_ab.log 2>&1
    done; grep -v amdgpu.ids $out/photon_ab.log ;;
  layouts)
    python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k layouts > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 4 $out/pytest.log
    python tools/layout_probe.py > $out/layout_probe.log 2>&1; grep -v amdgpu.ids $out/layout_probe.log ;;
  layout_pmc)
    # L1 accesses / L2 requests per ray of the random-ray launch under three storage orders
    export TMPDIR=/tmp
    for lay in 0 2 18; do
      export MIRO_LAYOUT=$lay
      rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/lay$lay -- \
        python3 tools/pmc_probe.py --what random --flags 0 > $out/lay$lay.log 2>&1
    done
    unset MIRO_LAYOUT
    python3 - <<'PY'
import csv, glob, re
for lay in (0, 2, 18):
    d = "gpurun_out/r03_layout_pmc/lay%d" % lay
    log = open(d + ".log", errors="replace").read()
    m = re.search(r"PMC_PROBE.*rays=(\d+).*ms=([0-9.]+) mrays_s=([0-9.]+)", log)
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    rays = int(m.group(1)) if m else 1
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    print("layout %2d: %s Mrays/s (under the profiler); per ray: L1 accesses %.2f  L2 read requests %.2f  L2 hit rate %.3f" % (
        lay, m.group(3) if m else "?", mean.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / rays, mean.get("TCP_TCC_READ_REQ_sum", 0) / rays,
        mean.get("TCC_HIT_sum", 0) / max(mean.get("TCC_HIT_sum", 0) + mean.get("TCC_MISS_sum", 0), 1)))
PY
    ;;
  children_pmc)
    export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bounce_probe.py > $out/trace.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/pmc1 -- python3 tools/bounce_probe.py > $out/pmc1.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 tools/bounce_probe.py > $out/pmc2.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc3 -- python3 tools/bounce_probe.py > $out/pmc3.log 2>&1
    python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r03_children_pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "children_kernel" in k or "trace_kernel" in k:
                acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(d, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()}, "dispatches", max(len(v) for v in c.values()))
for f in glob.glob("gpurun_out/r03_children_pmc/trace/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.reader(open(f))):
        if i < 8: print(r[:6])
PY
    ;;
  children_ab)
    for v in ${CHILD_VARIANTS:-default _cw4 _cw6 default _cw4 _cw6}; do
      d=lib$v; [ "$v" = default ] && d=lib
      echo "== $d" >> $out/children_ab.log
      MIRO_LIB=$LIB/$d/libmiro_hip.so python tools/bounce_probe.py 2>&1 | grep "bounce generation" >> $out/children_ab.log
      MIRO_LIB=$LIB/$d/libmiro_hip.so python tools/bounce_probe.py --scene sponza --w 1920 --h 1080 --spp 4 2>&1 | grep "bounce generation" >> $out/children_ab.log
    done; cat $out/children_ab.log ;;
  fuzz)
    # fresh seeds on the round-3 kernels (the hot loop's asm block changed): every mode against the oracle, fused frames and
    # level kernels against the batched calls
    MIRO_FUZZ_BASE=${FUZZ_BASE:-90000} MIRO_FUZZ_SEEDS=${FUZZ_SEEDS:-500} python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > $out/fuzz.log 2>&1
    tail -n 5 $out/fuzz.log ;;
  bench_ab)
    for v in ${BENCH_VARIANTS:-default _not default _not}; do
      d=lib$v; [ "$v" = default ] && d=lib
      echo "== $d" >> $out/bench_ab.log
      MIRO_LIB=$LIB/$d/libmiro_hip.so python bench.py --no-cpu-baseline --no-pmc --steps 60 $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['config']['workload'])" >> $out/bench_ab.log
    done; cat $out/bench_ab.log ;;
  profiles)
    # everything profiles/r03_* is made from: the two bench lines, rocprofv3 summaries of both commands, the other configs
    python bench.py > $out/r03_bench_line.json 2> $out/bench.err
    python bench.py --config photon > $out/r03_photon_line.json 2> $out/photon.err
    bash tools/collect_profiles.sh r03 > $out/collect_r03.log 2>&1
    bash tools/collect_profiles.sh r03_photon --config photon --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $out/collect_r03_photon.log 2>&1
    : > $out/r03_other_configs.jsonl
    for args in "--scene teapot --width 512 --height 512 --spp 1" "--scene teapot --width 512 --height 512 --spp 1 --mode primary" \
                "--scene bunny --width 1024 --height 1024 --spp 16" "--spp 1" "--spp 4" "--spp 16" "--width 512 --height 512 --spp 1" \
                "--scene bunny20 --spp 16" "--scene spiral --width 1024 --height 1024 --spp 16" "--scene cornell --width 256 --height 256 --spp 1" \
                "--mode primary" "--product"; do
      python bench.py $args --steps 40 --no-cpu-baseline --no-pmc 2>/dev/null >> $out/r03_other_configs.jsonl
    done
    python3 -c "
import json
for l in open('$out/r03_other_configs.jsonl'):
    j = json.loads(l); print('%-60s %10.1f %s  %8.4f ms' % (j['config']['workload'], j['value'], j['unit'], j['ms_per_step']))"
    cut -c1-300 $out/r03_bench_line.json; cut -c1-300 $out/r03_photon_line.json ;;
  photon_l2)
    export TMPDIR=/tmp
    rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/g1 -- python3 tools/photon_probe.py --reps 1 > $out/g1.log 2>&1
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/g2 -- python3 tools/photon_probe.py --reps 1 > $out/g2.log 2>&1
    python3 - <<'PY'
import csv, glob
for g in ("g1", "g2"):
    acc = {}
    for f in glob.glob("gpurun_out/r03_photon_l2/%s/**/*counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(f)):
            if "irradiance_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(g, {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()}, "dispatches", max([len(v) for v in acc.values()] or [0]))
PY
    ;;
  frame_pmc)
    # where the frame kernel's waiting goes: SMEM / VMEM / LDS instruction cycles, scalar-cache hit rate, instruction fetch
    export TMPDIR=/tmp
    LEG="bench.py --pmc-leg --no-cpu-baseline --no-pmc --steps 2 --warmup 1 $FRAME_ARGS"
    i=0
    for G in "SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
             "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SMEM_NORM"; do
      i=$((i+1))
      rocprofv3 --pmc $G --output-format csv -d $out/g$i -- python3 $LEG > $out/g$i.log 2>&1 || tail -n 3 $out/g$i.log
    done
    python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/r03_frame_pmc/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frame_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print("%-32s %.5g" % (k, m[k]))
wc = m.get("SQ_WAVE_CYCLES", 1)
print("share of wave-cycles: waiting %.3f ; SMEM instruction cycles %.3f ; VMEM_RD %.3f ; SALU %.3f" % (
    m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_INST_CYCLES_SMEM", 0) / wc, m.get("SQ_INST_CYCLES_VMEM_RD", 0) / wc, m.get("SQ_INST_CYCLES_SALU", 0) / wc))
if m.get("SQC_DCACHE_REQ"): print("scalar cache: hit rate %.3f, misses per request %.3f" % (m["SQC_DCACHE_HITS"] / m["SQC_DCACHE_REQ"], m["SQC_DCACHE_MISSES"] / m["SQC_DCACHE_REQ"]))
if m.get("SQ_INSTS_SMEM"): print("cycles per SMEM instruction: %.1f" % (m.get("SQ_INST_CYCLES_SMEM", 0) / m["SQ_INSTS_SMEM"]))
PY
    ;;
  photon_diff)
    for v in default _kd; do
      d=lib$v; [ "$v" = default ] && d=lib
      MIRO_LIB=$LIB/$d/libmiro_hip.so python tools/photon_probe.py --reps 1 --random-queries 20000 --dump $out/q$v.npz 2>&1 | grep "checksums\|queries, k"
    done
    python3 - <<'PY'
import numpy as np
a = np.load("gpurun_out/r03_photon_diff/qdefault.npz"); b = np.load("gpurun_out/r03_photon_diff/q_kd.npz")
bad = np.nonzero(a["r2"].view(np.uint32) != b["r2"].view(np.uint32))[0]
print("r2 differs on", len(bad), "of", len(a["r2"]), "; found differs on", int((a["found"] != b["found"]).sum()))
for i in bad[:12]: print(i, i % 64, "boxed", a["r2"][i], a["found"][i], "kd", b["r2"][i], b["found"][i], "ratio", a["r2"][i] / b["r2"][i])
PY
    ;;
  *) echo "unknown stage $stage"; exit 2 ;;
esac
