#!/bin/bash
# Last call of a round: default bench line + single-stream kernel stats + PMC traffic of the FINAL build (no pytest; tools/gpu_round.sh ran it).
tag=${1:-r03f}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_pmc.sh ${tag}pmc
cp gpurun_out/${tag}pmc/r03_pmc_hbm_traffic.json profiles/r03_pmc_hbm_traffic.json       # so that the bench line below reports the traffic of THIS build
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 400 $out/bench_default.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ss -- python bench.py --single-stream --no-cpu-baseline > $out/bench_ss.json 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $out/bench_conv.txt
timeout 600 python tools/launch_table.py resnet 72 60 2>&1 | grep -v amdgpu > $out/launch_table_resnet.txt
timeout 600 python tools/bench_nets.py 2>&1 | grep -v amdgpu > $out/bench_nets.txt
python -c "
import json; d=json.loads(open('$out/bench_default.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
