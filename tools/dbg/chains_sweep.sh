#!/bin/bash
# the default schedule (three contexts) against the number of GOP chains per GPU: bash tools/dbg/chains_sweep.sh
cd "$(dirname "$0")/../.."
for c in 33 48 66 96; do
  echo "== --chains $c"
  python bench.py --chains $c --no-cpu-baseline --no-d2h --no-handover --no-1080p --no-end-to-end --no-api --no-host-parse --no-real-av1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 'ms/step', j['ms_per_step'], 'one_context', (j.get('one_context') or {}).get('value'), j['config'].get('chains_per_gpu', j['config']))" | cut -c1-300
done
