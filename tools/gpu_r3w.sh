#!/bin/bash
# single-job exchange groups: collective on the job's own stream (ADAMML_SYNC_DIRECT) vs through the communication stream
out=gpurun_out/r3w; mkdir -p $out
o="--no-cpu-baseline --no-roofline --force-collectives"
for rep in 1 2; do for dct in 0 1; do
  for b in 72 18; do
    ADAMML_SYNC_DIRECT=$dct timeout 600 python bench.py $o --batch $b --steps 12 --warmup 4 2>/dev/null | grep '"metric"' > $out/b${b}_d${dct}_$rep.json
    python -c "
import json; d=json.loads(open('$out/b${b}_d${dct}_$rep.json').read()); print('direct=$dct B=$b', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"
  done
  ADAMML_SYNC_DIRECT=$dct timeout 600 python bench.py $o --batch 9 --steps 20 --warmup 6 --launch-plan 2>/dev/null | grep '"metric"' > $out/b9_d${dct}_$rep.json
  python -c "
import json; d=json.loads(open('$out/b9_d${dct}_$rep.json').read()); print('direct=$dct B=9 plan', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"
done; done
ADAMML_SYNC_DIRECT=1 timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_syncbn_gpu.py tests/test_launch_plan_gpu.py -x -q > $out/pytest.log 2>&1; echo "rc=$?"; grep "passed\|failed" $out/pytest.log | tail -1
