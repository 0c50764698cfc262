#!/bin/bash
# real 4K stream through the public interface against the helper threads and GOP chains (the default: 16 / 16): bash tools/dbg/helpers_sweep.sh
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 192))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for v in "X=1" "EASYAV1_AMD_THREADS=8 EASYAV1_AMD_GOPS=8" "EASYAV1_AMD_THREADS=24 EASYAV1_AMD_GOPS=24" "EASYAV1_AMD_THREADS=32 EASYAV1_AMD_GOPS=32"; do
  echo "== $v"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 512 2>&1 | grep "sink stats\|fps_wall" | cut -c1-330
done
rm -f /dev/shm/real4k.webm
