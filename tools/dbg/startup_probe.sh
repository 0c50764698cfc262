#!/bin/bash
# what the first second of a real 4K stream through the public interface consists of: pictures 1, 2, 4, ... by wall time (EASYAV1_BENCH_TIMELINE) beside the
# sink's trace (EASYAV1_AMD_TRACE: context creation, first tickets prepared / committed / submitted) and its stats:  bash tools/dbg/startup_probe.sh [repeats]
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", ${1:-96}))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for v in "X=1" "EASYAV1_AMD_GOPS=4" ; do
  echo "== $v (first run of the process: code objects load)"
  env $v EASYAV1_AMD_STATS=1 EASYAV1_AMD_TRACE=1 EASYAV1_BENCH_TIMELINE=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json 2>&1 | grep -v "^Wall\|^With\|Decoding" | grep -v "ticket [0-9]* waited" | cut -c1-400 | head -120
done
rm -f /dev/shm/real4k.webm
