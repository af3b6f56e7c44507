#!/bin/bash
# conv3x3_c64 phase timing (tools/c64_phase_probe.py), one run per flag set given as arguments ("" = the shipped configuration)
out=gpurun_out/r3l; mkdir -p $out
i=0
for flags in "$@"; do
  echo "######## flags: [$flags]" | tee -a $out/phases.txt
  C64_PROBE_RAW=1 C64_PROBE_DIR=/tmp/c64_probe_$i timeout 600 python tools/c64_phase_probe.py $flags 2>&1 | grep -v amdgpu | tee -a $out/phases.txt
  i=$((i+1))
done
