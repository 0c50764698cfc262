set -e
for cfg in "1 32" "2 16" "3 11" "4 8" "2 32" "4 16"; do
  set -- $cfg
  python bench.py --contexts $1 --chains $2 --steps 2 --warmup 1 --no-cpu-baseline --no-api --no-end-to-end --no-d2h --no-host-parse --no-handover --no-1080p --no-two-contexts > gpurun_out/ctx_$1_$2.json 2> gpurun_out/ctx_$1_$2.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ctx_$1_$2.json').read().strip().splitlines()[-1])
print('contexts $1 chains $2:', d['value'], d['ms_per_step'], d['config'].get('frames_per_step'))"
done
