#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:  bash tools/collect_r04.sh [part ...]   (parts: bench stats traffic sq times; default all)
# Collects everything under profiles/r04_* into gpurun_out/r04c/ (copy the files you want judged into profiles/).
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04c
mkdir -p "$OUT"
PARTS=${*:-bench stats traffic sq times}
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
  $B --cold --no-cpu-baseline > "$OUT/bench_cold.json" 2>> "$OUT/bench.err"
  $B --angles 180 --no-cpu-baseline > "$OUT/bench_angles180.json" 2>> "$OUT/bench.err"
  $B --mode n512 --no-cpu-baseline > "$OUT/bench_moden512.json" 2>> "$OUT/bench.err"
  $B --mode siddon --no-cpu-baseline > "$OUT/bench_modesiddon.json" 2>> "$OUT/bench.err"
  $B --mode train --steps 100 --no-cpu-baseline > "$OUT/bench_modetrain.json" 2>> "$OUT/bench.err"
  ;;
stats)
  stats bench python3 "$ROOT/bench.py" --no-cpu-baseline
  stats angles180 python3 "$ROOT/bench.py" --no-cpu-baseline --angles 180
  stats moden512 python3 "$ROOT/bench.py" --no-cpu-baseline --mode n512
  stats training_call python3 "$ROOT/tools/trace_training_call.py" 50
  stats setup_path python3 "$ROOT/tools/time_recon.py"
  ;;
traffic)
  traffic bench python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline
  traffic angles180 python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --angles 180
  traffic angles180_compact python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --angles 180 --plan-format compact
  traffic n512 python3 "$ROOT/bench.py" --no-cpu-baseline --mode n512
  traffic training_call python3 "$ROOT/tools/trace_training_call.py" 50
  ;;
sq)
  cd "$ROOT"
  bash tools/collect_sq.sh r04_a20 && bash tools/collect_sq.sh r04_angles180 --angles 180 && bash tools/collect_sq.sh r04_n512 --mode n512
  cp gpurun_out/sq_r04_a20.json "$OUT/sq_a20_counters.json"; cp gpurun_out/sq_r04_angles180.json "$OUT/sq_angles180_counters.json"
  cp gpurun_out/sq_r04_n512.json "$OUT/sq_n512_counters.json"
  cd /tmp
  ;;
times)
  python3 "$ROOT/tools/time_recon.py" > "$OUT/time_recon.txt" 2>&1
  python3 "$ROOT/tools/time_sel.py" 10 > "$OUT/time_sel.txt" 2>&1
  python3 "$ROOT/tools/time_compact_shapes.py" > "$OUT/time_compact_shapes.txt" 2>&1
  python3 "$ROOT/tools/time_affine.py" > "$OUT/time_affine.txt" 2>&1
  python3 "$ROOT/tools/time_tile_modes.py" > "$OUT/time_tile_modes.txt" 2>&1
  python3 "$ROOT/tools/time_fold.py" > "$OUT/time_fold.txt" 2>&1
  python3 "$ROOT/tools/sweep_bwd_paths.py" > "$OUT/sweep_bwd_paths.txt" 2>&1
  python3 "$ROOT/tools/ab_compare.py" > "$OUT/ab_current.txt" 2>&1
  python3 "$ROOT/tools/profile_trainer_kernels.py" > "$OUT/profile_trainer_kernels.txt" 2>&1 || true
  ;;
esac; done
ls -la "$OUT"
