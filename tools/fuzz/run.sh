#!/bin/bash
# ASan + UBSan build of the host-side parsers (CPU only) and a mutation run over the seed corpus.
#   bash tools/fuzz/run.sh <corpus dir> [iterations per file]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
OUT="${TMPDIR:-/tmp}/ea1_fuzz_parsers"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
    "$ROOT/tools/fuzz/fuzz_parsers.cpp" "$CSRC/obu_parse.cpp" "$CSRC/obu_frame_header.cpp" "$CSRC/webm_reader.cpp" \
    "$CSRC/vorbis_front.cpp" "$CSRC/common.cpp" \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o "$OUT"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT" "$1" "${2:-500}"
