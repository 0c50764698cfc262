#!/bin/bash
# kernel timeline of the first pictures of a real 4K stream (rocprofv3 --kernel-trace): bash tools/dbg/first_frame_trace.sh
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 8))
PY
export LD_LIBRARY_PATH=$ROOT/easyav1_amd:/opt/rocm/lib
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ff_trace
EASYAV1_AMD_TRACE=1 EASYAV1_BENCH_TIMELINE=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ff_trace -- $ROOT/tools/bin/easyav1_bench /dev/shm/real4k.webm --json 2>&1 | grep "timeline\] picture \(1\|16\|64\) \|context\|ticket 0 \|oldest ticket 0 \|code objects" | cut -c1-160
python3 - <<PY
import csv, glob
rows = []
for p in glob.glob("/tmp/ff_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for p in glob.glob("/tmp/ff_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
t0 = rows[0][0]
print("%d records; first at 0" % len(rows))
prev_end = t0
for i, (a, b, n) in enumerate(rows[:4000]):
    gap = (a - prev_end) / 1e6
    if gap > 3.0 or i < 12 or (b - a) / 1e6 > 3.0:
        print("%6d  start %9.3f ms  dur %8.3f ms  gap before %8.3f ms  %s" % (i, (a - t0) / 1e6, (b - a) / 1e6, gap, n))
    prev_end = max(prev_end, b)
PY
cp $(find /tmp/ff_trace -name "*kernel_trace.csv" | head -1) $ROOT/gpurun_out/ff_kernel_trace.csv; cp $(find /tmp/ff_trace -name "*memory_copy_trace.csv" | head -1) $ROOT/gpurun_out/ff_copy_trace.csv
rm -f /dev/shm/real4k.webm
