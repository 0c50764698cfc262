#!/bin/bash
# hand-over rate of 16 library threads under different settings of the copy path (diagnostic knobs of av1_recon.cpp)
for W in 4k 1080p; do
  for cfg in "EA1_NO_PULL=1" "EA1_PULL_GRID=192" "EA1_PULL_GRID=1024" "EA1_PULL_GRID=4096" "EA1_PULL_MAX_KB=4096" "EA1_PULL_JOBS=2" "EA1_PULL_JOBS=32"; do
    echo "== $W $cfg"
    env $cfg python tools/time_job_create.py $W 16 2>&1 | grep "16 library threads, coefficients in pinned"
  done
done
