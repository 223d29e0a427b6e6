#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:  bash tools/collect_r05.sh [part ...]   (parts: bench stats traffic scaling; default all)
# Collects round 5's measurements into gpurun_out/r05c/ (copy the files you want judged into profiles/ as r05_*).
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05c
mkdir -p "$OUT"
PARTS=${*:-bench stats traffic scaling}
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <name> <program + args...>: rocprofv3 kernel trace of a command -> <name>_kernel_stats.csv
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.trace.log" 2>&1
  python3 "$ROOT/tools/pmc_to_json.py" stats "$(find /tmp/prof_$name -name '*.db' | head -1)" > "$OUT/${name}_kernel_stats.csv"
}
traffic() { # traffic <name> <program + args...>: two PMC passes -> <name>_traffic_pmc.json
  local name=$1; shift
  rm -rf /tmp/pmcf_$name /tmp/pmcw_$name
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcf_$name -o f -- "$@" > "$OUT/$name.fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcw_$name -o w -- "$@" > "$OUT/$name.write.log" 2>&1
  python3 "$ROOT/tools/pmc_to_json.py" traffic "$(find /tmp/pmcf_$name -name '*.db' | head -1)" "$(find /tmp/pmcw_$name -name '*.db' | head -1)" > "$OUT/${name}_traffic_pmc.json"
}
B="python3 $ROOT/bench.py"
for part in $PARTS; do case $part in
bench)
  $B > "$OUT/bench.json" 2> "$OUT/bench.err"
  $B --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2>> "$OUT/bench.err"
  $B --angles 180 --no-cpu-baseline --no-modes > "$OUT/bench_angles180.json" 2>> "$OUT/bench.err"
  $B --mode n512 --no-cpu-baseline > "$OUT/bench_moden512.json" 2>> "$OUT/bench.err"
  python3 "$ROOT/tools/time_modes.py" --shapes headline,a180,b400,n512 --n 100 --json "$OUT/modes.json" > "$OUT/modes.txt" 2>&1
  ;;
stats)
  stats modes python3 "$ROOT/tools/time_modes.py" --shapes headline --n 200
  stats modes_n512 python3 "$ROOT/tools/time_modes.py" --shapes n512 --n 20
  stats bench python3 "$ROOT/bench.py" --no-cpu-baseline --no-modes
  stats moden512 python3 "$ROOT/bench.py" --no-cpu-baseline --mode n512
  ;;
traffic)
  traffic modes python3 "$ROOT/tools/time_modes.py" --shapes headline --n 50
  traffic bench python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --no-modes
  ;;
scaling)
  $B --project-scaling --angles 180 --total-batch 400 > "$OUT/projected_scaling.json" 2>> "$OUT/bench.err"
  $B --project-scaling --angles 20 --total-batch 400 > "$OUT/projected_scaling_a20.json" 2>> "$OUT/bench.err"
  ;;
esac; done
ls -la "$OUT"
