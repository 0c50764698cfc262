#!/bin/bash
# does the 16-CPU quota throttle the pipeline?  nr_throttled / throttled_usec of the cgroup around runs with different helper counts, each variant twice,
# interleaved (boxes differ by several per cent: only runs of one call compare).     bash tools/r05_threads_probe.sh > gpurun_out/<name>.txt
cd "$(dirname "$0")/.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 192))
PY
export LD_LIBRARY_PATH=$PWD/easyav1_amd:/opt/rocm/lib
stat() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for round in 1 2; do
for v in ${VARIANTS:-"X=1" "EASYAV1_AMD_THREADS=16" "EASYAV1_AMD_THREADS=15" "EASYAV1_AMD_THREADS=13" "EASYAV1_AMD_THREADS=14,EASYAV1_AMD_TILE_THREADS=1" "EASYAV1_AMD_THREADS=16,EASYAV1_AMD_TILE_THREADS=1"}; do
  echo "== $v (round $round)   before: $(stat)"
  env ${v//,/ } EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 256 2>&1 | grep "^{\|sink stats" | cut -c1-420
  echo "   after:  $(stat)"
done
done
rm -f /dev/shm/real4k.webm
