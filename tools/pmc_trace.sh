#!/bin/bash
# Runs on the GPU box: SQ / cache counter passes over a short perf_probe run of the sponza frame, to see where
# trace_kernel's wave-cycles go (issue vs wait vs memory).  One rocprofv3 --pmc pass per counter group.
# usage: bash tools/pmc_trace.sh <tag> [perf_probe args]
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:-"sponza --spp 16 --reps 2 --no-host"}
OUT=gpurun_out/pmc_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS"
G3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d "$OUT/g$i" -- python3 tools/perf_probe.py $ARGS > "$OUT/g$i.log" 2>&1 || { tail -5 "$OUT/g$i.log"; }
done
python3 - "$OUT" <<'EOF'
import csv, glob, os, sys
d = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(d, "g*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "trace_kernel" not in k and "trace_persistent" not in k:
            continue
        name = k[k.index("trace_"):][:60]
        acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, c in acc.items():
    print("==", k)
    for n in sorted(c):
        v = c[n]
        print("  %-32s n=%3d  mean=%.4g  max=%.4g" % (n, len(v), sum(v) / len(v), max(v)))
EOF
