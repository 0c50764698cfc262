#!/bin/bash
# ThreadSanitizer build of the HOST shim (easyav1_api.cpp, easyav1_video_sink.cpp, every parser) against a device-less stand-in for the GPU half
# (tools/tsan/device_stub.cpp), driven by tools/tsan/tsan_shim.cpp on a plain-AV1 WebM file.  The reference builds Debug with
# -fsanitize=thread,undefined (/root/reference/CMakeLists.txt:50-58).   bash tools/tsan/run.sh <file.webm>
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
OUT="${TMPDIR:-/tmp}/ea1_tsan_shim"
if [ ! -x "$OUT" ] || [ -n "$(find "$CSRC" "$ROOT/include" "$ROOT/tools/tsan" -newer "$OUT" -type f | head -1)" ]; then
    g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
        "$ROOT/tools/tsan/tsan_shim.cpp" "$ROOT/tools/tsan/device_stub.cpp" "$CSRC/easyav1_api.cpp" "$CSRC/easyav1_video_sink.cpp" \
        "$CSRC/av1_frame_parse.cpp" "$CSRC/av1_cdfs.cpp" "$CSRC/obu_parse.cpp" "$CSRC/obu_frame_header.cpp" "$CSRC/msac.cpp" "$CSRC/ir_packet.cpp" \
        "$CSRC/webm_reader.cpp" "$CSRC/audio_ring.cpp" "$CSRC/common.cpp" -lpthread -ldl -o "$OUT"
fi
EASYAV1_AMD_AV1_TABLES="${EASYAV1_AMD_AV1_TABLES:-$ROOT/easyav1_amd/data/av1_spec_tables.bin}" TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" "$OUT" "$1"
