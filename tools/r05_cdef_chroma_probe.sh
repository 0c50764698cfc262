#!/bin/bash
# chroma CDEF: the per-lane-direction kernel (default) against the direction-sorted one (EA1_CDEF_CHROMA_SORTED=1), which since round 5 lists a tile's blocks by
# superblock column and skips the range tracking in one-class steps -- per-kernel averages of the graded 4K batch, two interleaved rounds
cd "$(dirname "$0")/.."
B="bench.py --contexts 1 --chains 33 --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-two-contexts --no-end-to-end --no-api --no-host-parse --no-real-av1 --no-audio"
for r in 1 2; do
for v in "X=1" "EA1_CDEF_CHROMA_SORTED=1"; do
  echo "round $r $v"
  env $v python3 $B | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('  value %.1f frames/s' % d['value'])
for k in d['roofline']['kernels']:
    if 'cdef' in k['name']: print('  %-34s %8.1f us' % (k['name'], k['avg_us']))
"
done
done
