#!/bin/bash
# Non-headline bench lines of a round (policy stage, policy-gated inference, BASELINE.json configs[3] / configs[4] on one GPU).
tag=${1:-r02}
out=gpurun_out/${tag}v
mkdir -p $out
o="--no-cpu-baseline --no-roofline --steps 6"
timeout 600 python bench.py $o --stage policy 2>/dev/null | grep '"metric"' > $out/bench_policy_stage.json
timeout 600 python bench.py $o --stage infer 2>/dev/null | grep '"metric"' > $out/bench_inference_skipping.json
timeout 900 python bench.py $o --modalities rgb flow rgbdiff 2>/dev/null | grep '"metric"' > $out/bench_c4_rgb_flow_rgbdiff_b72.json
timeout 900 python bench.py $o --modalities rgb sound flow rgbdiff --batch 48 2>/dev/null | grep '"metric"' > $out/bench_c5_four_modalities_b48.json
for f in $out/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['peak_mem_gib'])"; done
