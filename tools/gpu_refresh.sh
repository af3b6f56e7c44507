#!/bin/bash
# Round profile refresh on the GPU box: parity tests, default bench line, single-stream rocprofv3 kernel stats.
# Usage (from the repo root, through gpurun): bash tools/gpu_refresh.sh <tag>
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ss -- python bench.py --single-stream --no-cpu-baseline > $out/bench_ss.json 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -30 $out/kernel_stats.csv | cut -c1-150
