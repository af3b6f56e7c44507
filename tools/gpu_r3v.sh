#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout 600 python bench.py --no-cpu-baseline --no-roofline --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced_collectives.json
python -c "
import json; d=json.loads(open('$out/bench_b72_forced_collectives.json').read()); print('b72 forced', d['value'], d['ms_per_step'], d['host_issue_ms'])"
