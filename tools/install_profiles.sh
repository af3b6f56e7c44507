#!/bin/bash
# Copies the evidence of one tools/gpu_round.sh call (gpurun_out/<tag>, <tag>pmc, <tag>v) into profiles/ under the round's names and
# regenerates profiles/README.md.  Usage (in the build container): bash tools/install_profiles.sh <tag> <round, e.g. r02>
tag=$1; r=${2:-r03}; g=gpurun_out
cp $g/$tag/bench_default.json profiles/${r}_bench_default.json
cp $g/$tag/bench_ss.json profiles/${r}_bench_single_stream_under_rocprof.json
cp $g/$tag/kernel_stats.csv profiles/${r}_bench_single_stream_kernel_stats.csv
[ -f $g/$tag/pytest.log ] && cp $g/$tag/pytest.log profiles/${r}_gpu_tests.txt
cp $g/${tag}pmc/r03_pmc_hbm_traffic.json profiles/${r}_pmc_hbm_traffic.json
v=$g/${tag}v
if [ -d $v ]; then
  for n in policy_stage inference_skipping c4_rgb_flow_rgbdiff_b72 c5_four_modalities_b48; do cp $v/bench_$n.json profiles/${r}_bench_$n.json; done
  cp $v/bench_conv.txt profiles/${r}_per_layer_bench_conv.txt
  cp $v/bench_dw.txt profiles/${r}_per_layer_bench_dw.txt
  cp $v/bench_fused.txt profiles/${r}_per_layer_bench_fused.txt
  cp $v/bench_elementwise.txt profiles/${r}_bench_elementwise.txt
  cp $v/explore_stream.txt profiles/${r}_streaming_kernels_layer1_2.txt
  cp $v/launch_table_resnet.txt profiles/${r}_launch_table_resnet.txt
  cp $v/launch_table_sound.txt profiles/${r}_launch_table_sound.txt
  cp $v/bench_nets.txt profiles/${r}_bench_nets.txt
else
  cp $g/$tag/bench_conv.txt profiles/${r}_per_layer_bench_conv.txt
  cp $g/$tag/launch_table_resnet.txt profiles/${r}_launch_table_resnet.txt
  cp $g/$tag/bench_nets.txt profiles/${r}_bench_nets.txt
fi
python tools/profile_readme.py $r
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/${r}_pmc_hbm_traffic.json"))
print("stamp json", d["_source_stamp"], "HEAD build", bench.source_stamp(), "MATCH" if d["_source_stamp"] == bench.source_stamp() else "MISMATCH")
PY
