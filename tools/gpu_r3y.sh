#!/bin/bash
# kernel statistics of the per-GPU share of the reference recipe (B = 9), launch plans on
out=gpurun_out/r3y; mkdir -p $out; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b9 -- python bench.py --batch 9 --steps 20 --warmup 6 --launch-plan --no-cpu-baseline --no-roofline > $out/bench_b9.json 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_b9.csv
rm -rf $out/prof
grep '"metric"' $out/bench_b9.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=9 plan under rocprof', d['value'], d['ms_per_step'], d['host_issue_ms'])"
