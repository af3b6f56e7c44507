#!/bin/bash
# Quick iteration call on the GPU box: default bench line, per-layer conv table, single-stream rocprofv3 kernel stats (no full pytest).
# Usage (through gpurun): bash tools/gpu_quick.sh <tag> [pytest-args...]
tag=${1:-q}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$1" ]; then timeout 1500 python -m pytest "$@" -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log; fi
timeout 600 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.loads(open('$out/bench_default.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['peak_mem_gib'])
for k,v in list(d['kernel_breakdown'].items())[:14]: print(' ', k, v['launches'], v['ms'], v['tflops'], v['gbs'])"
timeout 600 python tools/bench_conv.py > $out/bench_conv.txt 2>&1; cat $out/bench_conv.txt | cut -c1-230
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ss -- python bench.py --single-stream --no-cpu-baseline > $out/bench_ss.json 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
head -40 $out/kernel_stats.csv | cut -c1-160
