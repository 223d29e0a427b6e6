#!/bin/bash
# Developer tool, run ON THE GPU BOX: GPU tests, then the hot launches with the in-tree library and with the session-start build
# (tools/ab_compare.py, alternating on one box).  Output under gpurun_out/r04/<tag>_*.
set -o pipefail
TAG=${1:-x}
mkdir -p gpurun_out/r04
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/r04/${TAG}_gputests.log 2>&1
  tail -3 gpurun_out/r04/${TAG}_gputests.log
  grep -q passed gpurun_out/r04/${TAG}_gputests.log && ! grep -q failed gpurun_out/r04/${TAG}_gputests.log || exit 1
fi
python3 tools/ab_compare.py > gpurun_out/r04/${TAG}_ab_new.txt 2>&1 || exit 1
CTPVAE_VARIANT_LIB=tools/libctpvae_radon_prev.bin python3 tools/ab_compare.py > gpurun_out/r04/${TAG}_ab_start.txt 2>&1 || exit 1
python3 tools/ab_compare.py > gpurun_out/r04/${TAG}_ab_new2.txt 2>&1 || exit 1
paste -d'\n' gpurun_out/r04/${TAG}_ab_start.txt gpurun_out/r04/${TAG}_ab_new.txt gpurun_out/r04/${TAG}_ab_new2.txt | grep -v amdgpu.ids
