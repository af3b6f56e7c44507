#!/bin/bash
# cost of deterministic mode on the benchmark step
out=gpurun_out/r03; mkdir -p $out
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default mode', d['value'], d['ms_per_step'], d['deterministic'])"
  ADAMML_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 2>/dev/null | grep '"metric"' > $out/bench_deterministic.json
  python -c "
import json; d=json.loads(open('$out/bench_deterministic.json').read()); print('deterministic', d['value'], d['ms_per_step'], d['deterministic'])"
done
