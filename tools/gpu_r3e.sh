#!/bin/bash
# Round 3, fifth GPU call: conv3x3 / 64 kernel -- weight-row swizzle and the 160-byte patch pitch (A/B), parity of both.
out=gpurun_out/r3e; mkdir -p $out
export TMPDIR=/tmp
for pp in 144 160; do
  ADAMML_C64_PITCH=$pp timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "c64 or conv_fwd_bwd or groups_equal" > $out/pytest_$pp.log 2>&1; echo "pitch $pp rc=$?"; tail -2 $out/pytest_$pp.log
  ADAMML_C64_PITCH=$pp timeout 600 python tools/bench_conv.py 72 "l1 c2" 2>&1 | grep -v amdgpu | tee $out/bench_conv_$pp.txt
done
o="--no-cpu-baseline --no-roofline"
for rep in 1 2; do for pp in 144 160; do
  ADAMML_C64_PITCH=$pp timeout 600 python bench.py $o 2>/dev/null | grep '"metric"' > $out/bench_pp${pp}_$rep.json
  python -c "
import json; d=json.loads(open('$out/bench_pp${pp}_$rep.json').read()); print('C64_PITCH=$pp', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'])"
done; done
