#!/bin/bash
# round 5: generated streams (super-resolution, scaled references, switch frames, 4:4:4 / 4:2:2, ...) through the GPU path against dav1d
cd "$(dirname "$0")/.."
python tools/dbg/gen_gpu_sweep.py encode /tmp/ea1_gen_sweep ${1:-672} > gpurun_out/r05_s_gen_encode.log 2>&1 || { tail -5 gpurun_out/r05_s_gen_encode.log; exit 1; }
python tools/dbg/gen_gpu_sweep.py check /tmp/ea1_gen_sweep > gpurun_out/r05_s_gen_gpu_sweep.log 2>&1; tail -6 gpurun_out/r05_s_gen_gpu_sweep.log
exit 0
