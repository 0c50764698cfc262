#!/bin/bash
# ASan + UBSan build of the host-side parsers (CPU only) and a mutation run over the seed corpus.
#   bash tools/fuzz/run.sh <corpus dir> [iterations per file]          the bitstream / container parsers
#   bash tools/fuzz/run.sh <corpus dir> [iterations per seed frame] job  the host part of ea1_job_create
#   bash tools/fuzz/run.sh <corpus dir> [iterations per stream] av1     the block-level AV1 parser (corpus: tools/fuzz/make_av1_corpus.py)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
OUT="${TMPDIR:-/tmp}/ea1_fuzz_parsers"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
    "$ROOT/tools/fuzz/fuzz_parsers.cpp" "$CSRC/obu_parse.cpp" "$CSRC/obu_frame_header.cpp" "$CSRC/webm_reader.cpp" \
    "$CSRC/vorbis_front.cpp" "$CSRC/common.cpp" \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o "$OUT"
if [ "${3:-parsers}" = "av1" ]; then
    # the block-level AV1 parser (symbol decoder, syntax, contexts, frame driver), every source of it instrumented
    AV1="${TMPDIR:-/tmp}/ea1_fuzz_av1"
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
        -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
        "$ROOT/tools/fuzz/fuzz_av1_blocks.cpp" "$CSRC/av1_frame_parse.cpp" "$CSRC/av1_cdfs.cpp" "$CSRC/obu_parse.cpp" "$CSRC/obu_frame_header.cpp" \
        "$CSRC/msac.cpp" "$CSRC/common.cpp" \
        -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o "$AV1"
    ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$AV1" "$1" "${2:-300}"
    exit 0
fi
if [ "${3:-parsers}" = "job" ]; then
    # the host part of ea1_job_create (validation + schedule derivation) on frames of the synthetic generator
    JOB="${TMPDIR:-/tmp}/ea1_fuzz_job"
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -DEA1_FUZZ_HOOK \
        -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
        "$ROOT/tools/fuzz/fuzz_job.cpp" "$CSRC/av1_recon.cpp" "$CSRC/common.cpp" "$ROOT/tests/native/synth_stream.cpp" -I"$ROOT/tests/native" "$CSRC/wedge.cpp" \
        -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o "$JOB"
    # (frames of the other chroma layouts when the corpus directory holds them: tools/fuzz/make_layout_seeds.py)
    if [ -f "$1/layout_frames.bin" ]; then
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$JOB" "${2:-2000}" "$1/layout_frames.bin"
    else
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$JOB" "${2:-2000}"
    fi
    exit 0
fi
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT" "$1" "${2:-500}"
