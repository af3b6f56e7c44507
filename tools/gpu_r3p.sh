#!/bin/bash
# per-GPU shares 36 / 18 of the global batch of 72 with the forced one-rank choreography, then the full GPU suite
out=gpurun_out/r03; mkdir -p $out
o="--no-cpu-baseline --no-roofline"
timeout 600 python bench.py $o --batch 36 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b36_forced_collectives.json
timeout 600 python bench.py $o --batch 18 --steps 12 --warmup 4 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b18_forced_collectives.json
timeout 600 python bench.py $o --batch 18 --steps 12 --warmup 4 --force-collectives --launch-plan 2>/dev/null | grep '"metric"' > $out/bench_b18_forced_collectives_launch_plan.json
for f in $out/bench_b36*.json $out/bench_b18*.json; do python -c "
import json; d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], 'host', d.get('host_issue_ms'), d['peak_mem_gib'])"; done
bash tools/gpu_r3i.sh
