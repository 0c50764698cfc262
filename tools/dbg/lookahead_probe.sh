#!/bin/bash
# does a long item queue cost the sink anything (its helpers scan it under the lock)?  the real 4K stream with look-aheads of 256 / 1024 (default) / 4096 packets
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 384))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for v in "EASYAV1_AMD_LOOKAHEAD=256" "X=1" "EASYAV1_AMD_LOOKAHEAD=4096"; do
  echo "== $v"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 1024 2>&1 | grep "sink stats\|fps_wall" | cut -c1-300
done
rm -f /dev/shm/real4k.webm
