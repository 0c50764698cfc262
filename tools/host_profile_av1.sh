#!/bin/bash
# builds the block-level AV1 parser with line info and samples where its time goes (source lines):
#   bash tools/host_profile_av1.sh <table blob> <units file> [repeats] [lines shown]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
mkdir -p /tmp/hp
g++ -std=c++17 -O3 -g -no-pie -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
    "$ROOT/tools/host_profile_av1.cpp" "$CSRC/av1_frame_parse.cpp" "$CSRC/av1_cdfs.cpp" "$CSRC/obu_parse.cpp" "$CSRC/obu_frame_header.cpp" "$CSRC/msac.cpp" "$CSRC/common.cpp" \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -o /tmp/hp/host_profile_av1
/tmp/hp/host_profile_av1 "$1" "$2" "${3:-10}"
addr2line -e /tmp/hp/host_profile_av1 -i -f -C < /tmp/hp/pcs_av1.txt 2>/dev/null | paste - - | awk '{print $NF}' | sed 's/ (discriminator.*//' | sed 's|.*/||' | sort | uniq -c | sort -rn | head -${4:-45}
