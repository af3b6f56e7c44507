#!/bin/bash
# A/B of environment settings on the default bench line, alternating, one box.  Usage: bash tools/gpu_ab_env.sh <tag> <pairs> "ENV_A" "ENV_B" ...
# (each ENV_x: a quoted list of VAR=value assignments, "" = defaults)
tag=$1; pairs=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for r in $(seq 1 $pairs); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout 600 python bench.py --no-cpu-baseline > $out/bench_${i}_$r.json 2> $out/bench_${i}_$r.err
    python -c "
import json; d=json.loads(open('$out/bench_${i}_$r.json').read()); print('[$e]', d['value'], d['ms_per_step'], d['peak_mem_gib'])"
  done
done
