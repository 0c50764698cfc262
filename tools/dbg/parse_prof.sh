#!/bin/bash
# flat profile of the block-level parser on a committed stream (build container): bash tools/dbg/parse_prof.sh [fixture] [repeats]
set -e
cd "$(dirname "$0")/../.."
FX=${1:-tests/golden/av1_stream_bench_uhd_4cols_gop16.npz}
REP=${2:-3}
python tools/dbg/make_parse_stream.py $FX /tmp/parse_prof.stream
SRCS="easyav1_amd/csrc/av1_frame_parse.cpp easyav1_amd/csrc/av1_cdfs.cpp easyav1_amd/csrc/obu_parse.cpp easyav1_amd/csrc/obu_frame_header.cpp easyav1_amd/csrc/msac.cpp easyav1_amd/csrc/common.cpp tools/dbg/prof_stub.cpp"
g++ -O3 -g -fno-omit-frame-pointer -std=c++17 -pthread -no-pie -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ieasyav1_amd/csrc tools/dbg/parse_prof.cpp $SRCS -o /tmp/parse_prof ${EXTRA}
/tmp/parse_prof /tmp/parse_prof.stream $REP > /tmp/parse_prof.samples
python - <<'PY'
import subprocess, collections
rows = [l.split() for l in open("/tmp/parse_prof.samples")]
addrs = [r[0] for r in rows]
out = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", "/tmp/parse_prof"] + addrs, capture_output=True, text=True).stdout
# without -i each address gives 2 lines; with -i inlined frames add pairs: re-run per address in bulk without -i for the function, with -i for lines
fn = subprocess.run(["addr2line", "-f", "-C", "-e", "/tmp/parse_prof"] + addrs, capture_output=True, text=True).stdout.splitlines()
byfn, byline = collections.Counter(), collections.Counter()
total = 0
for i, r in enumerate(rows):
    n = int(r[1]); total += n
    f, l = fn[2 * i], fn[2 * i + 1]
    byfn[f[:100]] += n
    byline[l.split("/")[-1].split(" ")[0]] += n
print("samples", total)
print("-- by innermost (inlined) function")
for f, n in byfn.most_common(40): print("%5.1f%%  %s" % (100.0 * n / total, f))
print("-- by line")
for f, n in byline.most_common(40): print("%5.1f%%  %s" % (100.0 * n / total, f))
PY
