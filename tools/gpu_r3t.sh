#!/bin/bash
out=gpurun_out/r3t; mkdir -p $out
for g in 2 1; do
  ADAMML_SYNC_GROUPS=$g timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_syncbn_gpu.py tests/test_launch_plan_gpu.py tests/test_reference_loop_gpu.py tests/test_train_gpu.py -x -q > $out/pytest_g$g.log 2>&1
  echo "groups=$g rc=$?"; grep "passed\|failed" $out/pytest_g$g.log | tail -1
done
