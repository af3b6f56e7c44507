#!/bin/bash
# Whole evidence refresh of a round in ONE gpurun call: parity tests, default bench line, single-stream rocprofv3 kernel stats, PMC
# traffic passes, non-headline bench lines, per-layer tables.  Usage: bash tools/gpu_round.sh <tag>
tag=${1:-r02}
mkdir -p gpurun_out/$tag
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$tag/pytest.log; tail -3 gpurun_out/$tag/pytest.log
bash tools/gpu_final.sh $tag          # PMC passes first, so that the default bench line carries the traffic of THIS build
out=gpurun_out/${tag}v
mkdir -p $out
o="--no-cpu-baseline --no-roofline --steps 6"
timeout 600 python bench.py $o --stage policy 2>/dev/null | grep '"metric"' > $out/bench_policy_stage.json
timeout 600 python bench.py $o --stage infer 2>/dev/null | grep '"metric"' > $out/bench_inference_skipping.json
timeout 900 python bench.py $o --modalities rgb flow rgbdiff 2>/dev/null | grep '"metric"' > $out/bench_c4_rgb_flow_rgbdiff_b72.json
timeout 900 python bench.py $o --modalities rgb sound flow rgbdiff --batch 48 2>/dev/null | grep '"metric"' > $out/bench_c5_four_modalities_b48.json
for f in $out/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['peak_mem_gib'])"; done
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $out/bench_conv.txt
timeout 600 python tools/bench_fused.py 2>&1 | grep -v amdgpu > $out/bench_fused.txt
timeout 600 python tools/bench_dw.py 2>&1 | grep -v amdgpu > $out/bench_dw.txt
timeout 600 python tools/bench_elementwise.py 2>&1 | grep -v amdgpu > $out/bench_elementwise.txt
for L in 1 2; do timeout 600 python tools/explore_stream.py 72 $L 2>&1 | grep -v amdgpu; done > $out/explore_stream.txt
timeout 600 python tools/launch_table.py resnet 72 60 2>&1 | grep -v amdgpu > $out/launch_table_resnet.txt
timeout 600 python tools/launch_table.py sound 72 40 2>&1 | grep -v amdgpu > $out/launch_table_sound.txt
timeout 600 python tools/bench_nets.py 2>&1 | grep -v amdgpu > $out/bench_nets.txt
tail -3 $out/bench_conv.txt; cat $out/bench_nets.txt
