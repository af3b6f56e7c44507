#!/bin/bash
# Round 3: launch plans -- parity tests and the B = 9 lines (plain / forced one-rank RCCL) with and without plans.
out=gpurun_out/r3h; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_launch_plan_gpu.py -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -v "^$\|socket.cpp\|amdgpu" $out/pytest.log | tail -30
o="--no-cpu-baseline --no-roofline --batch 9 --steps 20 --warmup 6"
for v in "" "--launch-plan"; do for f in "" "--force-collectives"; do
  timeout 600 python bench.py $o $v $f 2>$out/err.txt | grep '"metric"' > $out/b.json && python -c "
import json; d=json.loads(open('$out/b.json').read()); print('B=9 [$v] [$f]', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'], d['peak_mem_gib'])" || tail -5 $out/err.txt
done; done
