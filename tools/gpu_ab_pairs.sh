out=gpurun_out/r5g; mkdir -p $out
for rep in 1 2 3; do for which in prev new; do
  lib=adamml_amd/libadamml_hip.so; [ $which = prev ] && lib=adamml_amd/libadamml_hip_prev.so
  ADAMML_HIP_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep '"metric"' > $out/bench_${which}_$rep.json
  python -c "
import json; d=json.loads(open('$out/bench_${which}_$rep.json').read()); print('$which', d['value'], d['ms_per_step'], d['ms_per_step_median_hipevent'])"
done; done
