#!/bin/bash
# where the public-interface pipeline spends its time on the committed real 4K GOP (EASYAV1_AMD_STATS): bash tools/sink_stats.sh [repeats]
set -e
cd "$(dirname "$0")/.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", ${1:-96}))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
VARIANTS=("X=1" "EASYAV1_AMD_LOOKAHEAD=256" "EASYAV1_AMD_LOOKAHEAD=128" "EASYAV1_AMD_LOOKAHEAD=256 EASYAV1_AMD_THREADS=32 EASYAV1_AMD_GOPS=32" "EASYAV1_AMD_LOOKAHEAD=512 EASYAV1_AMD_THREADS=32 EASYAV1_AMD_GOPS=32" "EASYAV1_AMD_LOOKAHEAD=256 EASYAV1_AMD_THREADS=24 EASYAV1_AMD_GOPS=16")
for v in "${VARIANTS[@]}"; do
  echo "== $v"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 256 2>&1 | grep -v "^Wall\|^With" | cut -c1-900
done
rm -f /dev/shm/real4k.webm
