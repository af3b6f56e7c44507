#!/bin/bash
# Whole evidence refresh of a round in ONE gpurun call (the full `pytest -m gpu` runs in its own call): PMC traffic passes first (so that
# the default bench line carries the traffic of THIS build), the default bench line, single-stream
# rocprofv3 kernel stats, the per-GPU share of the reference recipe (B = 9: plain / launch plan / forced one-rank RCCL choreography),
# B = 72 with the forced choreography, the non-headline lines, per-layer and per-launch tables.
# Usage (through gpurun): bash tools/gpu_round.sh [tag] [round, e.g. r04]; then, in the build container: bash tools/install_profiles.sh <tag> <round>
tag=${1:-r04}; round=${2:-r04}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_pmc.sh ${tag}pmc $round
cp gpurun_out/${tag}pmc/${round}_pmc_hbm_traffic.json profiles/${round}_pmc_hbm_traffic.json
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ss -- python bench.py --single-stream --no-cpu-baseline > $out/bench_ss.json 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
o="--no-cpu-baseline --no-roofline"
b9="$o --batch 9 --steps 20 --warmup 6"
timeout 600 python bench.py $b9 2>/dev/null | grep '"metric"' > $out/bench_b9.json
timeout 600 python bench.py $b9 --launch-plan 2>/dev/null | grep '"metric"' > $out/bench_b9_launch_plan.json
timeout 600 python bench.py $b9 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b9_forced_collectives.json
timeout 600 python bench.py $b9 --force-collectives --launch-plan 2>/dev/null | grep '"metric"' > $out/bench_b9_forced_collectives_launch_plan.json
timeout 600 python bench.py $o --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced_collectives.json
ADAMML_SYNC_GROUPS=1 timeout 600 python bench.py $o --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b72_forced_collectives_one_group.json
timeout 600 python bench.py $o --batch 36 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b36_forced_collectives.json
timeout 600 python bench.py $o --batch 18 --steps 12 --warmup 4 --force-collectives 2>/dev/null | grep '"metric"' > $out/bench_b18_forced_collectives.json
timeout 600 python bench.py $o --steps 6 --stage policy 2>/dev/null | grep '"metric"' > $out/bench_policy_stage.json
timeout 600 python bench.py $o --steps 6 --stage infer 2>/dev/null | grep '"metric"' > $out/bench_inference_skipping.json
timeout 900 python bench.py $o --steps 6 --modalities rgb flow rgbdiff 2>/dev/null | grep '"metric"' > $out/bench_c4_rgb_flow_rgbdiff_b72.json
timeout 900 python bench.py $o --steps 6 --modalities rgb sound flow rgbdiff --batch 48 2>/dev/null | grep '"metric"' > $out/bench_c5_four_modalities_b48.json
for f in $out/bench_*.json; do python -c "
import json; d=json.loads([l for l in open('$f') if '\"metric\"' in l][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], 'host', d.get('host_issue_ms'), d['peak_mem_gib'])"; done
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $out/bench_conv.txt
timeout 600 python tools/bench_fused.py 2>&1 | grep -v amdgpu > $out/bench_fused.txt
timeout 600 python tools/bench_dw.py 2>&1 | grep -v amdgpu > $out/bench_dw.txt
timeout 600 python tools/bench_elementwise.py 2>&1 | grep -v amdgpu > $out/bench_elementwise.txt
timeout 300 python tools/bench_dw_bwd.py 2>&1 | grep -v amdgpu > $out/bench_dw_bwd.txt
timeout 300 python tools/bench_fadd_next.py 2>&1 | grep -v amdgpu > $out/bench_fadd_next.txt
timeout 600 python tools/launch_table.py resnet 72 60 2>&1 | grep -v amdgpu > $out/launch_table_resnet.txt
timeout 600 python tools/launch_table.py sound 72 40 2>&1 | grep -v amdgpu > $out/launch_table_sound.txt
timeout 600 python tools/launch_table.py policy_rgb 72 400 2>&1 | grep -v "amdgpu\|Warning\|warnings.warn" > $out/launch_table_policy_rgb.txt
timeout 600 python tools/launch_table.py policy_sound 72 400 2>&1 | grep -v "amdgpu\|Warning\|warnings.warn" > $out/launch_table_policy_sound.txt
timeout 600 python tools/bench_nets.py 2>&1 | grep -v amdgpu > $out/bench_nets.txt
tail -3 $out/bench_conv.txt | cut -c1-200; cat $out/bench_nets.txt
