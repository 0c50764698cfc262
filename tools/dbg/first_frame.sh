#!/bin/bash
# time to the first pictures of a real 4K stream, with and without deferred code-object loading:  bash tools/dbg/first_frame.sh
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 24))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for v in "X=1" "HIP_ENABLE_DEFERRED_LOADING=0" "X=2"; do
  echo "== $v"
  env $v EASYAV1_AMD_TRACE=1 EASYAV1_BENCH_TIMELINE=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json 2>&1 | grep "timeline\] picture \(1\|16\|64\|256\) \|context\|ticket 0 \|oldest ticket 0 \|open_ms\|code objects" | cut -c1-200
done
rm -f /dev/shm/real4k.webm
