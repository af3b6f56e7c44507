#!/bin/bash
out=gpurun_out/r3f; mkdir -p $out
for rep in 1 2 3; do for v in 0 1; do
  echo "WSWZ=$v"; ADAMML_C64_WSWZ=$v timeout 600 python tools/bench_conv.py 72 "l1 c2" 2>&1 | grep "l1 c2"
done; done
