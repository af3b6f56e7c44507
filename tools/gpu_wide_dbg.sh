#!/bin/bash
tag=${1:-wd}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for dbg in 0 1 2 3; do
  echo "== dbg=$dbg"; ADAMML_WIDE_DBG=$dbg timeout 300 python tools/bench_conv.py 72 "l3 c3" 2>&1 | grep "l3 c3" | cut -c1-200
  echo "== dbg=$dbg nostats"; ADAMML_WIDE_DBG=$dbg timeout 300 python tools/bench_conv.py 72 "l3 c3" nostats 2>&1 | grep "l3 c3" | cut -c1-140
done
