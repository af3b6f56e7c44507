#!/bin/bash
# Round-6 iteration call: tests of the wide 1x1 streaming kernels and their per-layer A/B (ADAMML_WIDE_STREAM=1 / 0 in one box).
# Usage (through gpurun): bash tools/gpu_wide1.sh <tag> [full]
tag=${1:-w1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide or conv_fwd_bwd or groups_equal" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
for on in 1 0 1 0; do
  ADAMML_WIDE_STREAM=$on timeout 300 python tools/bench_conv.py 72 "l3 c" > $out/bench_conv_l3_wide$on.txt 2>&1
  echo "== wide=$on"; cat $out/bench_conv_l3_wide$on.txt | cut -c1-200
done
if [ "$2" = "full" ]; then
for on in 1 0; do
ADAMML_WIDE_STREAM=$on timeout 600 python bench.py --no-cpu-baseline > $out/bench_wide$on.json 2> $out/bench_wide$on.err; python -c "
import json; d=json.loads(open('$out/bench_wide$on.json').read()); print('wide=$on', d['value'], d['ms_per_step'], d['roofline']['frac'], d['peak_mem_gib'])"
done
fi
