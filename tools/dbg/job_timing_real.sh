#!/bin/bash
# phases of ea1_job_create on the frames of a real 4K stream through the public interface (EA1_JOB_TIMING): bash tools/dbg/job_timing_real.sh
set -e
cd "$(dirname "$0")/../.."
python - <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_stream
print(make_stream.write_real_av1_stream("/dev/shm/real4k.webm", "tests/golden/av1_stream_bench_uhd_4cols_gop16.npz", 24))
PY
export LD_LIBRARY_PATH=easyav1_amd:/opt/rocm/lib
EA1_JOB_TIMING=1 EASYAV1_AMD_STATS=1 tools/bin/easyav1_bench /dev/shm/real4k.webm --json 2> /tmp/job_timing.txt | tail -1 | cut -c1-200
python3 - <<PY
import re
v=[];l=[];p=[]
for line in open("/tmp/job_timing.txt"):
    m=re.search(r"validate ([0-9.]+) ms, lists ([0-9.]+) ms, pack \+ stage ([0-9.]+) ms", line)
    if m: v.append(float(m.group(1))); l.append(float(m.group(2))); p.append(float(m.group(3)))
n=len(v)
print("%d jobs: validate %.3f ms, lists %.3f ms, pack + stage %.3f ms on average (sums %.1f / %.1f / %.1f ms)" % (n, sum(v)/n, sum(l)/n, sum(p)/n, sum(v), sum(l), sum(p)))
big=[i for i in range(n) if v[i]+l[i]+p[i] > 5]
print("jobs above 5 ms:", len(big), "avg of the rest: %.3f ms" % (sum(v[i]+l[i]+p[i] for i in range(n) if i not in set(big))/max(1,n-len(big))))
for line in open("/tmp/job_timing.txt"):
    if "sink stats" in line: print(line.strip()[:500])
PY
rm -f /dev/shm/real4k.webm
