#!/bin/bash
# round 5: fresh libaom 4:4:4 / 4:2:2 streams through the GPU path against dav1d (encode + dav1d on the CPU first, then one GPU process)
cd "$(dirname "$0")/.."
python tools/dbg/chroma_live_sweep.py encode /tmp/ea1_chroma_sweep > gpurun_out/r05_n_chroma_live_encode.log 2>&1 || { tail -5 gpurun_out/r05_n_chroma_live_encode.log; exit 1; }
python tools/dbg/chroma_live_sweep.py check /tmp/ea1_chroma_sweep > gpurun_out/r05_n_chroma_live_sweep.log 2>&1; rc=$?; tail -4 gpurun_out/r05_n_chroma_live_sweep.log
[ $rc -ne 0 ] && grep -v " OK$" gpurun_out/r05_n_chroma_live_sweep.log | head -40
exit 0
