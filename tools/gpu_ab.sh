#!/bin/bash
# A/B of one environment switch on the default bench line, after an optional pytest subset.
# Usage (through gpurun): bash tools/gpu_ab.sh <tag> <VAR> <value-list, e.g. "0 1"> [pytest-args...]
tag=$1; var=$2; vals=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$1" ]; then timeout 1800 python -m pytest "$@" -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log; fi
for rep in 1 2; do for v in $vals; do
  env $var=$v timeout 600 python bench.py --no-cpu-baseline > $out/bench_${var}_${v}_$rep.json 2> $out/bench.err
  python -c "
import json; d=json.loads(open('$out/bench_${var}_${v}_$rep.json').read()); print('$var=$v', d['value'], d['ms_per_step'])"
done; done
