#!/bin/bash
out=gpurun_out/r3o; mkdir -p $out
timeout 600 python tools/launch_table.py resnet 72 10 all 2>&1 | grep -v amdgpu > $out/launch_table_resnet.txt
grep "64->64 k3" $out/launch_table_resnet.txt
head -12 $out/launch_table_resnet.txt | cut -c1-150
