#!/bin/bash
# `value` (synthetic 4K, default schedule) against the intra wavefront's launch threshold: bash tools/dbg/intra_wide_sweep.sh
cd "$(dirname "$0")/../.."
for w in 96 16 32 256 1024; do
  echo "== EA1_INTRA_WIDE=$w"
  EA1_INTRA_WIDE=$w python bench.py --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-end-to-end --no-api --no-host-parse --no-real-av1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 'one_context', (j.get('one_context') or {}).get('value'))"
done
