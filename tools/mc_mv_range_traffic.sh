#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the inter prediction kernels against the range of the synthetic motion vectors:
#   tools/mc_mv_range_traffic.sh <tag>
# EA1_SYN_MV_SHIFT=k divides the generator's random +-64 px vectors by 2^k.  One counter per pass, never with --stats.
set -e -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/mvrange_$TAG
mkdir -p $O
P="bench.py --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-two-contexts --no-end-to-end --no-api --no-host-parse --steps 1 --warmup 0"
for SHIFT in 0 2 4 6; do
    export EA1_SYN_MV_SHIFT=$SHIFT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$SHIFT -- python3 $P > /dev/null
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$SHIFT -- python3 $P > /dev/null
    python3 tools/pmc_traffic.py $O/f$SHIFT $O/w$SHIFT --workload 4k --chains 32 --source "mv_shift $SHIFT" --out $O/traffic_shift$SHIFT.json > /dev/null
    python3 bench.py --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-two-contexts --no-end-to-end --no-api --no-host-parse > $O/bench_shift$SHIFT.json
    rm -rf $O/f$SHIFT $O/w$SHIFT
    echo "shift $SHIFT done"
done
python3 - <<PY > $O/summary.txt
import json
print("# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes, corrected as tools/pmc_traffic.py) and duration of the inter prediction")
print("# kernels per launch of 32 4K frames against the range of the synthetic motion vectors (EA1_SYN_MV_SHIFT=k: +-64 px >> k)")
for sh in (0, 2, 4, 6):
    t = json.load(open("$O/traffic_shift%d.json" % sh))["entries"][-1]["kernels"]
    b = json.loads(open("$O/bench_shift%d.json" % sh).read().strip().splitlines()[-1])
    us = {k["name"].replace(" ", ""): k["avg_us"] for k in b["roofline"]["kernels"]}
    for name in ("mc_tile_kernel<8,64,0,false>", "mc_tile_kernel<16,64,1,false>", "mc_tile_kernel<32,256,2,false>"):
        k = t[name]
        print("mv +-%-3d px  %-34s %7.1f us  fetch_raw %6.1f MB  write %6.1f MB  traffic %6.1f .. %6.1f MB" % (64 >> sh, name, us.get(name, 0), k["fetch_size_raw_bytes"] / 1e6, k["write_size_bytes"] / 1e6, k["traffic_low_bytes"] / 1e6, k["traffic_bytes"] / 1e6))
    print("mv +-%-3d px  value %.1f frames/s" % (64 >> sh, b["value"]))
PY
cat $O/summary.txt
