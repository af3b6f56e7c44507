#!/bin/bash
# refresh of the forced-choreography lines of gpurun_out/r03 (tools/gpu_round3.sh has them too)
out=gpurun_out/r03; mkdir -p $out
o="--no-cpu-baseline --no-roofline"
b9="$o --batch 9 --steps 20 --warmup 6"
timeout 600 python bench.py $b9 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b9_forced_collectives.json
timeout 600 python bench.py $b9 --force-collectives --launch-plan 2>/dev/null | grep '"metric"' > $out/bench_b9_forced_collectives_launch_plan.json
timeout 600 python bench.py $o --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced_collectives.json
timeout 600 python bench.py $o --batch 36 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b36_forced_collectives.json
timeout 600 python bench.py $o --batch 18 --steps 12 --warmup 4 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b18_forced_collectives.json
ADAMML_SYNC_GROUPS=1 timeout 600 python bench.py $o --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced_collectives_one_group.json
for f in $out/bench_b*forced*.json; do python -c "
import json; d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], 'host', d.get('host_issue_ms'), d['peak_mem_gib'])"; done
