#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:  bash tools/collect_sq_cmd.sh <tag> <program> [args...]
# Three --pmc passes of SQ counters + two TCC passes (FETCH_SIZE / WRITE_SIZE) over ANY program (put the program itself after
# the tag: python3 script.py ...); leaves gpurun_out/sq_<tag>.json and gpurun_out/traffic_<tag>.json (per-kernel means per dispatch).
set -e -o pipefail
ROOT=$(pwd); TAG=$1; shift
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_$TAG /tmp/tf_$TAG /tmp/tw_$TAG
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$TAG/p$i -o p -- "$@" > /tmp/sq_$TAG.log 2>&1
done
python3 "$ROOT/tools/pmc_counters.py" $(find /tmp/sq_$TAG -name "*.db" | sort) > "$ROOT/gpurun_out/sq_$TAG.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tf_$TAG -o f -- "$@" > /tmp/tf_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tw_$TAG -o w -- "$@" > /tmp/tw_$TAG.log 2>&1
python3 "$ROOT/tools/pmc_to_json.py" traffic "$(find /tmp/tf_$TAG -name '*.db' | head -1)" "$(find /tmp/tw_$TAG -name '*.db' | head -1)" > "$ROOT/gpurun_out/traffic_$TAG.json"
