#!/bin/bash
# with_handover of bench.py under the diagnostic knobs of the copy path (av1_recon.cpp)
for cfg in "EA1_DEFAULT=1" "EA1_PULL=1" "EA1_COPY_STREAMS=2"; do
  env $cfg python bench.py --no-cpu-baseline --no-d2h 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', 'value', d['value'], '4k handover', d['with_handover']['value'], '1080p', d['also_1080p']['value'], 'handover', d['also_1080p']['with_handover']['value'])"
done
