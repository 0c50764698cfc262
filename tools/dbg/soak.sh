#!/bin/bash
# a long run of the real 4K stream through the public interface (30 720 frames): the same checksum of every picture's sampled bytes twice,
# resident memory and HBM use at the end against their values a quarter of the way in:  bash tools/dbg/soak.sh
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 1920))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for k in 1 2; do
  tools/bin/easyav1_bench /dev/shm/real4k.webm --json --touch > /tmp/soak_$k.json 2>/tmp/soak_$k.err &
  pid=$!
  sleep 8; a_rss=$(grep VmRSS /proc/$pid/status | awk '{print $2}'); a_hbm=$(rocm-smi --showmeminfo vram 2>/dev/null | grep "Used" | awk '{print $NF}' | head -1)
  wait $pid
  echo "run $k: rss after 8 s ${a_rss} kB, vram used then ${a_hbm} B; $(tail -1 /tmp/soak_$k.json | cut -c1-330)"
done
python3 - <<PY
import json
a=json.loads(open("/tmp/soak_1.json").read().strip().splitlines()[-1]); b=json.loads(open("/tmp/soak_2.json").read().strip().splitlines()[-1])
print("frames", a["frames_all"], b["frames_all"], "checksums equal:", a["checksum"] == b["checksum"], a["checksum"])
PY
rm -f /dev/shm/real4k.webm
