#!/bin/bash
# Host half of the hand-over (ea1_job_prepare: validation, the lists the host makes) on the parsed frames of a real stream, on the CPU alone
# (the context exists only as a size: the fuzzers' hook; the call ends at the device allocation, after every host phase):
# per-frame time, the phases (EA1_JOB_TIMING), and a SIGPROF sample profile by source line.
#   bash tools/host_prepare/run.sh [fixture] [frames] [repetitions]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
FIX="${1:-bench_uhd_4cols_gop16}"; N="${2:-16}"; REPS="${3:-20}"
OUT="${TMPDIR:-/tmp}/ea1_host_prepare"; mkdir -p "$OUT"
CSRC="$ROOT/easyav1_amd/csrc"
python "$ROOT/tools/host_prepare/make_packets.py" "$FIX" "$N" "$OUT/frames.bin" | tail -3
g++ -std=c++17 -O2 -g -no-pie -DEA1_FUZZ_HOOK -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" -I"$ROOT/tools/host_prepare" \
    "$ROOT/tools/host_prepare/time_job_prepare.cpp" "$CSRC/av1_recon.cpp" "$CSRC/common.cpp" "$CSRC/wedge.cpp" \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -o "$OUT/time_job_prepare"
EA1_JOB_TIMING=1 "$OUT/time_job_prepare" "$OUT/frames.bin" 2 2>&1 | grep -m3 "ea1_job_create" || true
PROF="$OUT/samples.txt" "$OUT/time_job_prepare" "$OUT/frames.bin" "$REPS"
python - "$OUT" <<'PY'
import collections, subprocess, sys
out = sys.argv[1]
rows = [l.split() for l in open(out + "/samples.txt")]
lines = subprocess.run(["addr2line", "-e", out + "/time_job_prepare"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
h, tot = collections.Counter(), 0
for (a, c), l in zip(rows, lines):
    h[l.split(" ")[0].split("/")[-1]] += int(c)
    tot += int(c)
print("%d samples; the ten hottest source lines:" % tot)
for k, v in h.most_common(10):
    print("  %5.1f %%  %s" % (100.0 * v / tot, k))
PY
