#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:  bash tools/collect_sq.sh <tag> [bench args]
# Three --pmc passes of SQ counters over bench.py; leaves gpurun_out/sq_<tag>.json (per-kernel means per dispatch).
set -e -o pipefail
ROOT=$(pwd); TAG=$1; shift
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$TAG/p$i -o p -- python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-graph "$@" > /tmp/sq_$TAG.log 2>&1
done
python3 "$ROOT/tools/pmc_counters.py" $(find /tmp/sq_$TAG -name "*.db" | sort) > "$ROOT/gpurun_out/sq_$TAG.json"
