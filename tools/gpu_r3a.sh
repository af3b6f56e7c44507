#!/bin/bash
# Round 3, first GPU call: the changed host paths (RCCL one-rank step, launcher spawn, stock-DDP detection, K > 256 head), the default
# bench line, and the per-GPU share of the reference recipe (B = 9) with its host issue time.
out=gpurun_out/r3a; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_train_gpu.py tests/test_reference_loop_gpu.py tests/test_head_gpu.py tests/test_syncbn_gpu.py -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -25 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout 600 python bench.py --no-cpu-baseline --batch 9 --steps 20 --warmup 5 > $out/bench_b9.json 2> $out/bench_b9.err
for f in $out/bench_default.json $out/bench_b9.json; do python -c "
import json; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'], d['peak_mem_gib'])"; done
timeout 600 python tools/host_time.py 9 > $out/host_time_b9.txt 2>&1; head -40 $out/host_time_b9.txt
