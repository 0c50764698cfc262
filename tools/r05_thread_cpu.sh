#!/bin/bash
# where the process's CPU seconds go, by thread, on the real 4K and 1080p bench streams through the public interface: bash tools/r05_thread_cpu.sh [repeats of the GOP]
set -e
cd "$(dirname "$0")/.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", ${1:-192}))
print(make_stream.write_real_av1_stream("/dev/shm/real1080.webm", "tests/golden/av1_stream_bench_hd_gop16.npz", ${1:-192}))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for f in real4k real1080; do
  echo "== $f"
  EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/$f.webm --json --threads --skip 256 2>&1 | grep "sink stats\|threads\]\|fps_wall" | sed 's/.*\[sink stats\]/[sink stats]/' | cut -c1-400
done
rm -f /dev/shm/real4k.webm /dev/shm/real1080.webm
