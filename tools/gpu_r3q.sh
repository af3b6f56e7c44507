#!/bin/bash
out=gpurun_out/r3q; mkdir -p $out
timeout 900 python tools/stem_ab.py "$@" 2>&1 | grep -v amdgpu | tee -a $out/ab.txt
