#!/bin/bash
# RECORD OF AN EXPERIMENT (the switch is not in the tree): events made with hipEventBlockingSync (a waiting host thread sleeps; variant X=1) against HIP's default
# spinning wait (EASYAV1_AMD_EVENT_SPIN=1) on the real 4K and 1080p bench
# streams through the public interface, interleaved rounds: frames/s and the process's CPU seconds.   bash tools/r05_event_wait_probe.sh [repeats of the GOP] [rounds]
set -e
cd "$(dirname "$0")/.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", ${1:-192}))
print(make_stream.write_real_av1_stream("/dev/shm/real1080.webm", "tests/golden/av1_stream_bench_hd_gop16.npz", ${1:-192}))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
for r in $(seq 1 ${2:-3}); do
for f in real4k real1080; do
for v in "X=1" "EASYAV1_AMD_EVENT_SPIN=1"; do
  echo -n "round $r $f $v: "
  env $v tools/bin/easyav1_bench /dev/shm/$f.webm --json --skip 256 2>&1 | grep fps_wall | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steady %.1f whole %.1f frames/s, cpu %.2f s, cpus busy %.2f' % (d['fps_wall'], d['fps_wall_all'], d['cpu_seconds'], d['cpus_busy']))"
done
done
done
rm -f /dev/shm/real4k.webm /dev/shm/real1080.webm
