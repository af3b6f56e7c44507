#!/bin/bash
# margin survey: the chaotic whole-network tests three times, printed quantities only
out=gpurun_out/r4b; mkdir -p $out
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/test_parity_fullsize_gpu.py tests/test_train_trajectory_gpu.py tests/test_models_gpu.py tests/test_syncbn_gpu.py tests/test_reference_loop_gpu.py -q -s > $out/run$i.log 2>&1
  echo "run $i rc=$?"; grep "passed\|failed" $out/run$i.log | tail -1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke:" | sed "s/^/run $i /"
done
