#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_launch_plan_gpu.py tests/test_models_gpu.py -x -q -s -k "plan or skipping or eval" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -v "^$\|socket.cpp\|amdgpu" $out/pytest.log | tail -15
o="--no-cpu-baseline --no-roofline --stage infer --steps 30 --warmup 8"
for b in 1 4 8; do for v in "" "--launch-plan"; do
  timeout 600 python bench.py $o --batch $b $v 2>$out/err.txt | grep '"metric"' > $out/b.json && python -c "
import json; d=json.loads(open('$out/b.json').read()); print('infer B=$b [$v]', d['value'], 'clips/s', d['ms_per_step'], 'ms, host', d['host_issue_ms'], 'mem', d['peak_mem_gib'], d['config']['executed_clips_per_modality'])" || tail -5 $out/err.txt
done; done
