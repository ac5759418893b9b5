#!/bin/bash
# Runs on the GPU box: three rocprofv3 --pmc passes (SQ issue/wait, SQ instruction mix, vector-L1 / L2) over ONE kind of
# trace launch (tools/pmc_probe.py), summarised per counter into gpurun_out/pmc_<tag>/summary.txt.
# usage: bash tools/pmc_set.sh <tag> <pmc_probe args...>
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/pmc_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"
G3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
G4="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_BUSY_CYCLES"
G5="TA_TA_BUSY_sum TA_FLAT_LOAD_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum"
G6="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d "$OUT/g$i" -- python3 tools/pmc_probe.py "$@" > "$OUT/g$i.log" 2>&1 || { tail -5 "$OUT/g$i.log"; }
done
python3 tools/pmc_summarize.py "$OUT" "$@" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
