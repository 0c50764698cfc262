#!/bin/bash
# does the 16-CPU quota throttle the pipeline (16 helpers + stage + caller + runtime threads > 16 runnable)?  nr_throttled / throttled_usec of the cgroup
# around runs with 12 .. 16 helpers, and with the tile threads off.     bash tools/r05_threads_probe.sh > gpurun_out/<name>.txt
cd "$(dirname "$0")/.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 192))
PY
export LD_LIBRARY_PATH=$PWD/easyav1_amd:/opt/rocm/lib
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for v in "X=1" "EASYAV1_AMD_THREADS=15" "EASYAV1_AMD_THREADS=14" "EASYAV1_AMD_THREADS=13" "EASYAV1_AMD_THREADS=12" "EASYAV1_AMD_TILE_THREADS=1" "EASYAV1_AMD_THREADS=14 EASYAV1_AMD_TILE_THREADS=1" "EASYAV1_AMD_THREADS=15 EASYAV1_AMD_GOPS=15"; do
  echo "== $v"
  echo "before: $(stat)"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 256 2>&1 | grep -v "^Wall\|^With\|^Decoding\|^Video\|^Slowest\|^Decoded" | cut -c1-700
  echo "after:  $(stat)"
done
rm -f /dev/shm/real4k.webm
