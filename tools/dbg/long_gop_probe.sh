#!/bin/bash
# a look-ahead shorter than the GOPs (what long GOPs do to the default look-ahead), with and without the shim reading on until every chain has a GOP:
#   bash tools/dbg/long_gop_probe.sh        (EASYAV1_AMD_LOOKAHEAD=10 on the GOP-16 1080p stream; EASYAV1_AMD_FIXED_LOOKAHEAD=1: the old rule)
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real1080.webm", "tests/golden/av1_stream_bench_hd_gop16.npz", 96))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for v in "EASYAV1_AMD_LOOKAHEAD=10 EASYAV1_AMD_FIXED_LOOKAHEAD=1" "EASYAV1_AMD_LOOKAHEAD=10" "X=1"; do
  echo "== $v"
  env $v tools/bin/easyav1_bench /dev/shm/real1080.webm --json --skip 256 2>&1 | tail -1 | cut -c1-260
done
rm -f /dev/shm/real1080.webm
