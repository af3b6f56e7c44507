#!/bin/bash
# conv3x3_c64 variant A/B (tools/c64_ab.py): arguments = flag sets
out=gpurun_out/r3m; mkdir -p $out
timeout 900 python tools/c64_ab.py "$@" 2>&1 | grep -v amdgpu | tee -a $out/ab.txt
