#!/bin/bash
# Round 3, fourth GPU call: temporal pool fused into the conv3 + bn3 + add kernel -- kernel parity, block / model / full-size parity,
# trajectory, bench A/B.
out=gpurun_out/r3d; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "tpool or temporal_pool or fwd_bn_add or stem1 or dwconv" > $out/pytest_k.log 2>&1; echo "rc=$?" >> $out/pytest_k.log; tail -15 $out/pytest_k.log
timeout 1700 python -m pytest tests/test_blocks_gpu.py tests/test_models_gpu.py tests/test_parity_fullsize_gpu.py tests/test_train_trajectory_gpu.py -x -q -s > $out/pytest_m.log 2>&1; echo "rc=$?" >> $out/pytest_m.log; grep -v "^$" $out/pytest_m.log | grep "traject\|per-step\|passed\|failed\|rc=\|Error\|assert" | tail -20
o="--no-cpu-baseline --no-roofline"
for rep in 1 2; do for v in "0 0" "1 0" "1 1"; do set -- $v
  ADAMML_FUSE_TPOOL=$1 ADAMML_STEM1_F32=$2 timeout 600 python bench.py $o 2>/dev/null | grep '"metric"' > $out/bench_tp$1_st$2_$rep.json
  python -c "
import json; d=json.loads(open('$out/bench_tp$1_st$2_$rep.json').read()); print('FUSE_TPOOL=$1 STEM1_F32=$2', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'], d['peak_mem_gib'])"
done; done
