#!/bin/bash
# Copies the evidence of one tools/gpu_round.sh call (gpurun_out/<tag>, <tag>pmc) and of the full pytest call (gpurun_out/<tag>t/pytest.log)
# into profiles/ under the round's names and regenerates profiles/README.md.  Usage (build container): bash tools/install_profiles.sh <tag> [r04]
tag=${1:-r04}; r=${2:-r04}; g=gpurun_out
grep '"metric"' $g/$tag/bench_default.json | tail -1 > profiles/${r}_bench_default.json
grep '"metric"' $g/$tag/bench_ss.json | tail -1 > profiles/${r}_bench_single_stream_under_rocprof.json
cp $g/$tag/kernel_stats.csv profiles/${r}_bench_single_stream_kernel_stats.csv
[ -f $g/${tag}t/pytest.log ] && cp $g/${tag}t/pytest.log profiles/${r}_gpu_tests.txt
cp $g/${tag}pmc/${r}_pmc_hbm_traffic.json profiles/${r}_pmc_hbm_traffic.json
for n in b9 b9_launch_plan b9_forced_collectives b9_forced_collectives_launch_plan b72_forced_collectives b72_forced_collectives_one_group b36_forced_collectives b18_forced_collectives policy_stage inference_skipping \
         c4_rgb_flow_rgbdiff_b72 c5_four_modalities_b48; do cp $g/$tag/bench_$n.json profiles/${r}_bench_$n.json; done
cp $g/$tag/bench_conv.txt profiles/${r}_per_layer_bench_conv.txt
cp $g/$tag/bench_dw.txt profiles/${r}_per_layer_bench_dw.txt
cp $g/$tag/bench_fused.txt profiles/${r}_per_layer_bench_fused.txt
cp $g/$tag/bench_elementwise.txt profiles/${r}_bench_elementwise.txt
[ -f $g/$tag/bench_dw_bwd.txt ] && cp $g/$tag/bench_dw_bwd.txt profiles/${r}_per_layer_bench_dw_bwd.txt
[ -f $g/$tag/bench_fadd_next.txt ] && cp $g/$tag/bench_fadd_next.txt profiles/${r}_per_layer_bench_fadd_next.txt
cp $g/$tag/launch_table_resnet.txt profiles/${r}_launch_table_resnet.txt
cp $g/$tag/launch_table_sound.txt profiles/${r}_launch_table_sound.txt
cp $g/$tag/launch_table_policy_rgb.txt profiles/${r}_launch_table_policy_rgb.txt
cp $g/$tag/launch_table_policy_sound.txt profiles/${r}_launch_table_policy_sound.txt
cp $g/$tag/bench_nets.txt profiles/${r}_bench_nets.txt
python tools/profile_readme.py $r
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/${r}_pmc_hbm_traffic.json"))
print("stamp json", d["_source_stamp"], "HEAD build", bench.source_stamp(), "MATCH" if d["_source_stamp"] == bench.source_stamp() else "MISMATCH")
PY
