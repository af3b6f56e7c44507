#!/bin/bash
# Round 3, third GPU call: coroutine scheduler of the SyncBN exchange -- parity (one-rank RCCL, 2-rank gloo) and its host cost at B = 9 / B = 72.
out=gpurun_out/r3c; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_syncbn_gpu.py tests/test_reference_loop_gpu.py -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -v "^$" $out/pytest.log | tail -12
o="--no-cpu-baseline --no-roofline"
timeout 600 python bench.py $o --batch 9 --steps 20 --warmup 5 --force-collectives > $out/bench_b9_forced.json 2> $out/bench_b9_forced.err
timeout 600 python bench.py $o --batch 9 --steps 20 --warmup 5 > $out/bench_b9.json 2> $out/bench_b9.err
timeout 600 python bench.py $o --force-collectives > $out/bench_b72_forced.json 2> $out/bench_b72_forced.err
timeout 600 python bench.py $o > $out/bench_default.json 2> $out/bench_default.err
for f in bench_b9_forced bench_b9 bench_b72_forced bench_default; do python -c "
import json; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'], d['peak_mem_gib'])" || tail -5 $out/$f.err; done
