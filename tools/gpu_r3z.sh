#!/bin/bash
# 16-byte split reduce: correctness + A/B against the previous build of the library (adamml_amd/libadamml_hip_base.so)
out=gpurun_out/r3z; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_rccl_gpu.py tests/test_syncbn_gpu.py tests/test_parity_fullsize_gpu.py -x -q -k "bn or finalize or rccl or syncbn or deterministic or block or sound or policy" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $out/pytest.log | tail -1
for rep in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export ADAMML_HIP_LIB=$PWD/adamml_amd/libadamml_hip_base.so; else unset ADAMML_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-roofline --steps 16 --warmup 4 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib B=72', d['value'], d['ms_per_step'])"
    python bench.py --no-cpu-baseline --no-roofline --batch 9 --steps 30 --warmup 6 --launch-plan 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib B=9 plan', d['value'], d['ms_per_step'])"
  done
done
