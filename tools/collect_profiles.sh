#!/bin/bash
# Collects the per-round profile evidence of the default bench workload on the GPU box:
#   tools/collect_profiles.sh <tag> [extra bench.py flags]
# kernel-trace stats, then FETCH_SIZE / WRITE_SIZE and the SQ counters in SEPARATE --pmc passes
# (never combined with --stats or a sys trace), summaries under gpurun_out/prof_<tag>/.
set -e -o pipefail
TAG=$1; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
# the roofline's schedule: ONE context holding all the chains (the kernel's own duration; the default headline splits them over three contexts whose
# kernels overlap, which stretches every per-kernel duration in a trace)
B="bench.py --contexts 1 --chains 33 --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-two-contexts --no-end-to-end --no-api --no-host-parse --no-real-av1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/bench_under_rocprof.json
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
echo "stats done"
P="$B --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $P > /dev/null
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $P > /dev/null
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq/p1 -- python3 $P > /dev/null
echo "sq1 done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq/p2 -- python3 $P > /dev/null
echo "sq2 done"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq/p3 -- python3 $P > /dev/null
echo "sq3 done"
python3 tools/pmc_summary.py $O/sq > $O/sq_counters.txt
python3 $B > $O/bench.json
# derived utilisation counters: a counter this rocprofv3 does not know only loses this one summary
if rocprofv3 --pmc VALUBusy MfmaUtil --kernel-trace --output-format csv -d $O/util -- python3 $P > /dev/null; then
    python3 tools/pmc_summary.py $O/util > $O/util_counters.txt
    echo "util done"
else
    echo "util pass failed"
fi
# per-kernel HBM traffic of the default workload (4K, 33 chains per launch), merged into a copy of profiles/pmc_traffic.json
python3 tools/pmc_traffic.py $O/fetch $O/write --workload 4k --chains 33 --source "profiles/${TAG}_4k_default_pmc_traffic.json" \
    --merge profiles/pmc_traffic.json --out $O/pmc_traffic.json
# the raw traces are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf $O/stats $O/fetch $O/write $O/sq $O/util
ls -la $O
