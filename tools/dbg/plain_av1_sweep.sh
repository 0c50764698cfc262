#!/bin/bash
# plain AV1 through the public interface against GOP chains in flight / helper threads (generated 1080p file)
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT"
python - <<PY
import sys
sys.path.insert(0, "tools")
import make_stream
print(make_stream.write_plain_av1_stream("/tmp/plain.webm", "/tmp/plain_tables.bin", 1920, 1080, 12, 40, 1, policy=dict(skip=55, all_zero=60, inter=80, max_mv_class=4, split_bias=25)))
PY
export LD_LIBRARY_PATH=$ROOT/easyav1_amd:/opt/rocm/lib:$LD_LIBRARY_PATH EASYAV1_AMD_AV1_TABLES=/tmp/plain_tables.bin
for cfg in "8 8" "8 16" "12 16" "16 16" "16 24"; do
    set -- $cfg
    echo "gops $1 threads $2: $(EASYAV1_AMD_GOPS=$1 EASYAV1_AMD_THREADS=$2 tools/bin/easyav1_bench /tmp/plain.webm --json --repeat 2 --skip 96 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["fps_wall_all"]), round(d["fps_wall"]))')"
done
nproc
