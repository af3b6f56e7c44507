#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
grep '"metric"' $out/bench_default.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
