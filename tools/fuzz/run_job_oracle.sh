#!/bin/bash
# The job fuzzer with the CPU oracle behind it (ASan + UBSan on both): whatever ea1_job_create accepts, the checker must be able to decode --
# 4:2:0 frames of the synthetic generator and, with <dir>/layout_frames.bin (tools/fuzz/make_layout_seeds.py), 4:4:4 / 4:2:2 frames.
#   bash tools/fuzz/run_job_oracle.sh <dir> [iterations per seed frame]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
OUT="${TMPDIR:-/tmp}/ea1_fuzz_job_oracle"
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
for f in av1_recon av1_itx av1_grain; do
    gcc -std=gnu99 $SAN -ffp-contract=off -I"$ROOT/include" -c "$ROOT/oracle/$f.c" -o "${TMPDIR:-/tmp}/ea1o_$f.o"
done
g++ -std=c++17 $SAN -DEA1_FUZZ_HOOK -DEA1_FUZZ_ORACLE -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
    "$ROOT/tools/fuzz/fuzz_job.cpp" "$CSRC/av1_recon.cpp" "$CSRC/common.cpp" "$ROOT/tests/native/synth_stream.cpp" -I"$ROOT/tests/native" "$CSRC/wedge.cpp" \
    "${TMPDIR:-/tmp}"/ea1o_av1_recon.o "${TMPDIR:-/tmp}"/ea1o_av1_itx.o "${TMPDIR:-/tmp}"/ea1o_av1_grain.o \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lm -o "$OUT"
if [ -f "$1/layout_frames.bin" ]; then
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT" "${2:-300}" "$1/layout_frames.bin"
else
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT" "${2:-300}"
fi
