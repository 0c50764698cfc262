#!/bin/bash
# Profile evidence of the AUDIO workload (BASELINE configs[4]'s synthesis: 16 streams x 5.1, blocksizes 256 / 2048, one launch) on the GPU box:
#   tools/collect_audio_profiles.sh <tag>
# kernel-trace stats, then the counters in SEPARATE --pmc passes (never combined with --stats or a sys trace); summaries under gpurun_out/prof_<tag>/
set -e -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
B="bench.py --workload audio --steps 5 --warmup 2"
python3 $B > $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/bench_under_rocprof.json
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
P="bench.py --workload audio --steps 1 --warmup 0"
mkdir -p $O/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python3 $P > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -- python3 $P > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc/p1 -- python3 $P > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc/p2 -- python3 $P > /dev/null
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc/p3 -- python3 $P > /dev/null
python3 tools/pmc_summary.py $O/pmc > $O/counters.txt
rm -rf $O/stats $O/pmc
ls -la $O
