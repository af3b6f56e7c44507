#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out
timeout 600 python tools/host_time.py 9 > $out/host_time_b9.txt 2>&1; grep -v "amdgpu" $out/host_time_b9.txt | head -75
o="--no-cpu-baseline --no-roofline"
timeout 600 python bench.py $o --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced.json
timeout 600 python bench.py $o --batch 9 --steps 20 --warmup 5 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b9_forced.json
for f in bench_b72_forced bench_b9_forced; do python -c "
import json; d=json.loads(open('$out/$f.json').read()); print('$f', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"; done
