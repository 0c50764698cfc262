#!/bin/bash
# builds the host part of ea1_job_create without a device and samples where its time goes (source lines)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/easyav1_amd/csrc"
mkdir -p /tmp/hp
g++ -std=c++17 -O3 -g -no-pie -fno-omit-frame-pointer -DEA1_FUZZ_HOOK -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$CSRC" \
    "$ROOT/tools/host_profile_job.cpp" "$CSRC/av1_recon.cpp" "$CSRC/common.cpp" "$ROOT/tests/native/synth_stream.cpp" -I"$ROOT/tests/native" "$CSRC/wedge.cpp" \
    -L"$ROOT/easyav1_amd" -leasyav1_amd -Wl,-rpath,"$ROOT/easyav1_amd" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -o /tmp/hp/host_profile_job
/tmp/hp/host_profile_job "${1:-1080p}" "${2:-100}"
addr2line -e /tmp/hp/host_profile_job -i -f -C < /tmp/hp/pcs.txt 2>/dev/null | paste - - | awk '{print $NF}' | sed 's/ (discriminator.*//' | sort | uniq -c | sort -rn | head -${3:-40}
