#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes over the bench command.
#   pass 1: --kernel-trace --stats          (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE                (TCC slots: FETCH_SIZE needs 3 of 4 -> own pass)
#   pass 3: --pmc WRITE_SIZE
# Counter passes never combine with sys/hip/hsa trace domains (the pool refuses that).
# usage: tools/collect_profiles.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline --no-pmc"}   # --no-pmc: bench.py must not start its own rocprofv3 child under the profiler
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -5 "$OUT/pmc_write.log"; exit 1; }
python3 tools/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
