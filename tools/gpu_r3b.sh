#!/bin/bash
# Round 3, second GPU call: loss-trajectory test (first measurement), B = 9 with the forced one-rank RCCL choreography, per-role roofline,
# full launch tables of the three MobileNetV2 backbones and the ResNet.
out=gpurun_out/r3b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_trajectory_gpu.py -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -v "^$" $out/pytest.log | tail -20
timeout 600 python bench.py --no-cpu-baseline --batch 9 --steps 20 --warmup 5 --force-collectives > $out/bench_b9_forced.json 2> $out/bench_b9_forced.err
timeout 600 python bench.py --no-cpu-baseline --batch 9 --steps 20 --warmup 5 > $out/bench_b9.json 2> $out/bench_b9.err
timeout 600 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
for f in bench_b9_forced bench_b9 bench_default; do python -c "
import json; d=json.loads(open('$out/$f.json').read()); print('$f', d['value'], d['ms_per_step'], 'host', d['host_issue_ms'], d['peak_mem_gib'])
for k, v in (d['roofline'] or {}).get('per_role', {}).items(): print('   ', k, v['bound'], v['frac'], v['ms_per_step'], v['c_abi_launches_per_step'])"; done
for n in sound policy_rgb policy_sound resnet; do timeout 600 python tools/launch_table.py $n 72 10 all 2>&1 | grep -v amdgpu > $out/launch_table_$n.txt; done
grep -A28 "per entry point" $out/launch_table_sound.txt
