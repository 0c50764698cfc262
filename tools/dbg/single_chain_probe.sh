#!/bin/bash
# one GOP chain (what a stream without closed GOPs gets): real 4K and 1080p through the public interface: bash tools/dbg/single_chain_probe.sh
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 24))
print(make_stream.write_real_av1_stream("/dev/shm/real1080.webm", "tests/golden/av1_stream_bench_hd_gop16.npz", 48))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for f in real4k real1080; do
for v in "EASYAV1_AMD_GOPS=1" "EASYAV1_AMD_GOPS=2" "EASYAV1_AMD_GOPS=4"; do
  echo "== $f $v"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/$f.webm --json --skip 32 2>&1 | grep "sink stats\|fps_wall" | cut -c1-420
done
done
rm -f /dev/shm/real4k.webm /dev/shm/real1080.webm
