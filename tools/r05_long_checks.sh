#!/bin/bash
# the long checks of earlier rounds on the round-5 build (the parser's symbol decoders, coefficient loop and record arrays were rewritten):
#   bash tools/r05_long_checks.sh <part>      part 1: feature sweep + sb128 sweep + live-stream stress;  part 2: long fuzz + soak
cd "$(dirname "$0")/.."
if [ "$1" = "1" ]; then
  python tools/dbg/feature_sweep.py 64 > gpurun_out/r05_l_feature_sweep.log 2>&1; tail -2 gpurun_out/r05_l_feature_sweep.log
  python tools/dbg/sb128_sweep.py 48 > gpurun_out/r05_l_sb128_sweep.log 2>&1; tail -2 gpurun_out/r05_l_sb128_sweep.log
else
  EA1_FUZZ_TRIALS=3000 EA1_DIFF_FUZZ=600 python -m pytest tests/test_video_gpu.py -m gpu -q --timeout 1000 -k "fuzz or differential or mutat" > gpurun_out/r05_l_long_fuzz.log 2>&1; tail -2 gpurun_out/r05_l_long_fuzz.log
  bash tools/dbg/soak.sh > gpurun_out/r05_l_soak.txt 2>&1; tail -3 gpurun_out/r05_l_soak.txt
fi
