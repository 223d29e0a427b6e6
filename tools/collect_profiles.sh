#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh [extra bench args]
# Leaves under gpurun_out/collect/: bench.json (the default bench line), trace/ (kernel trace + stats of the same
# command), fetch/ and write/ (the two PMC passes).  tools/pmc_to_json.py turns the .db files into profiles/ summaries.
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
rm -rf "$OUT" && mkdir -p "$OUT"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f -- python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w -- python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline "$@" > "$OUT/write.log" 2>&1
ls -R "$OUT" | head -30
