#!/bin/bash
# SyncBN exchange groups: lock-step (1) vs alternating (2) with the configs[2] choreography forced on one rank
out=gpurun_out/r3s; mkdir -p $out
o="--no-cpu-baseline --no-roofline --force-collectives"
for g in 1 2; do
  ADAMML_SYNC_GROUPS=$g timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_syncbn_gpu.py tests/test_launch_plan_gpu.py -x -q 2>&1 | tail -1
done
for rep in 1 2; do for g in 1 2; do
  for b in 72 36 18; do
    ADAMML_SYNC_GROUPS=$g timeout 600 python bench.py $o --batch $b --steps 12 --warmup 4 2>/dev/null | grep '"metric"' > $out/b${b}_g${g}_$rep.json
    python -c "
import json; d=json.loads(open('$out/b${b}_g${g}_$rep.json').read()); print('groups=$g B=$b', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"
  done
  ADAMML_SYNC_GROUPS=$g timeout 600 python bench.py $o --batch 9 --steps 20 --warmup 6 --launch-plan 2>/dev/null | grep '"metric"' > $out/b9_g${g}_$rep.json
  python -c "
import json; d=json.loads(open('$out/b9_g${g}_$rep.json').read()); print('groups=$g B=9 plan', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"
done; done
