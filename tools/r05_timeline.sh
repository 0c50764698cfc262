#!/bin/bash
# Round 5, verdict item 1(a): what stops the real 4K stream through libeasyav1.so?  (1) how many host cores the box gives one process,
# (2) the sink's own phase times at 16 / 24 / 32 helpers, (3) a rocprofv3 kernel + memory-copy timeline of the same decode, summarised
# by tools/timeline_summary.py.      bash tools/r05_timeline.sh <tag> [repeats]    -> gpurun_out/<tag>_*.txt
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05_a}
REP=${2:-96}
mkdir -p gpurun_out tools/bin
g++ -O2 -std=c++17 -pthread tools/cpu_scale_probe.cpp -o tools/bin/cpu_scale_probe
tools/bin/cpu_scale_probe > gpurun_out/${TAG}_cpu_scale.txt 2>&1
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", ${REP}))
PY
export LD_LIBRARY_PATH=$PWD/easyav1_amd:/opt/rocm/lib
{
for v in "X=1" "EASYAV1_AMD_THREADS=24" "EASYAV1_AMD_THREADS=32" ${EXTRA_VARIANTS}; do
  echo "== $v"
  env $v EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 256 2>&1 | grep -v "^Wall\|^With\|^Decoding" | cut -c1-1200
done
} > gpurun_out/${TAG}_sink_stats.txt 2>&1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG}_trace
rm -rf $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tl -- tools/bin/easyav1_bench /dev/shm/real4k.webm --json --skip 256 > gpurun_out/${TAG}_trace_run.txt 2>&1
python tools/timeline_summary.py $OUT 12533760 > gpurun_out/${TAG}_timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -exec gzip -9 {} \;
du -sh $OUT
rm -f /dev/shm/real4k.webm
cat gpurun_out/${TAG}_cpu_scale.txt gpurun_out/${TAG}_sink_stats.txt gpurun_out/${TAG}_timeline.txt
