#!/bin/bash
# Runs on the GPU box: instruction-cache counters of the bench frame's kernel (one rocprofv3 --pmc pass over bench.py --pmc-leg).
# usage: bash tools/icache_probe.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/icache_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc ${COUNTERS:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU} --output-format csv -d "$OUT/p" -- python3 bench.py --pmc-leg --no-cpu-baseline --no-pmc "$@" > "$OUT/log.txt" 2>&1 || tail -5 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "p", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "frame_kernel" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print("  %-30s %.5g" % (k, m[k]))
if m.get("SQC_DCACHE_REQ"):
    print("  scalar D-cache miss rate %.4f ; requests per 1000 VALU %.2f" % (m["SQC_DCACHE_MISSES"] / m["SQC_DCACHE_REQ"], 1000 * m["SQC_DCACHE_REQ"] / m["SQ_INSTS_VALU"]))
if m.get("SQC_ICACHE_REQ"):
    print("  I-cache miss rate %.4f ; misses per 1000 VALU %.3f" % (m["SQC_ICACHE_MISSES"] / m["SQC_ICACHE_REQ"], 1000 * m["SQC_ICACHE_MISSES"] / m["SQ_INSTS_VALU"]))
PY
