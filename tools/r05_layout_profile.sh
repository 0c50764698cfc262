#!/bin/bash
# round 5: where the GPU time of a 1080p 4:4:4 / 4:2:2 stream goes (rocprofv3 --kernel-trace --stats of the decode through libeasyav1.so)
#     bash tools/r05_layout_profile.sh <tag>      streams: tools/bin/sweep/av1_stream_hd444_q45_gop16.npz, ..._hd422_..., ..._hd420_... (tools/make_sweep_streams.py)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05_q}
export LD_LIBRARY_PATH=$PWD/easyav1_amd:/opt/rocm/lib
export TMPDIR=/tmp
for L in 444 422 420; do
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/hd$L.webm", "tools/bin/sweep/av1_stream_hd${L}_q45_gop16.npz", 96))
PY
OUT=$PWD/gpurun_out/${TAG}_trace_$L
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tl -- tools/bin/easyav1_bench /dev/shm/hd$L.webm --json --skip 128 > gpurun_out/${TAG}_run_$L.txt 2>&1
python - <<PY > gpurun_out/${TAG}_kernels_$L.txt
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("1080p 4:$L stream, 96 GOPs of 16: kernel time %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print("%-70s %6d calls %9.1f us avg %6.2f %%" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $OUT /dev/shm/hd$L.webm
cat gpurun_out/${TAG}_kernels_$L.txt; grep -o '"fps_wall": [0-9.]*' gpurun_out/${TAG}_run_$L.txt | tail -1
done
