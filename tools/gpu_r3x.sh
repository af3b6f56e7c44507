#!/bin/bash
o="--no-cpu-baseline --no-roofline --force-collectives --steps 16 --warmup 4"
for rep in 1 2 3; do
  for cfg in "1 1" "2 0" "2 1"; do set -- $cfg
    ADAMML_SYNC_GROUPS=$1 ADAMML_SYNC_DIRECT=$2 timeout 600 python bench.py $o 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('groups=$1 direct=$2 B=72', d['value'], d['ms_per_step'], d['ms_per_step_median_hipevent'], 'host', d['host_issue_ms'])"
  done
done
python bench.py --no-cpu-baseline --no-roofline --steps 16 --warmup 4 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plain B=72', d['value'], d['ms_per_step'], d['ms_per_step_median_hipevent'], 'host', d['host_issue_ms'])"
