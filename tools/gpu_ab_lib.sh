#!/bin/bash
# A/B of two BUILDS of the library on the default bench line (ADAMML_HIP_LIB), two alternating pairs on one box, after an optional
# pytest subset with the build under test.  Usage (through gpurun): bash tools/gpu_ab_lib.sh <tag> <previous .so> [pytest-args...]
tag=$1; prev=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$1" ]; then timeout 2400 python -m pytest "$@" -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log; fi
for rep in 1 2; do for which in prev new; do
  lib=adamml_amd/libadamml_hip.so; [ $which = prev ] && lib=$prev
  ADAMML_HIP_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline > $out/bench_${which}_$rep.json 2> $out/bench.err
  python -c "
import json; d=json.loads(open('$out/bench_${which}_$rep.json').read()); print('$which', d['value'], d['ms_per_step'])"
done; done
for which in prev new; do
  lib=adamml_amd/libadamml_hip.so; [ $which = prev ] && lib=$prev
  ADAMML_HIP_LIB=$PWD/$lib timeout 300 python tools/bench_dual_mb.py > $out/dual_$which.txt 2>&1
  ADAMML_HIP_LIB=$PWD/$lib timeout 300 python tools/explore_stream.py 72 1 alg 10 > $out/stream_$which.txt 2>&1
  ADAMML_HIP_LIB=$PWD/$lib timeout 300 python tools/bench_nets.py > $out/nets_$which.txt 2>&1
  ADAMML_HIP_LIB=$PWD/$lib ADAMML_ALG_STREAM=0 timeout 300 python tools/bench_alg.py > $out/alg1_$which.txt 2>&1
  ADAMML_HIP_LIB=$PWD/$lib timeout 300 python tools/bench_alg.py 2 > $out/alg2_$which.txt 2>&1
  ADAMML_HIP_LIB=$PWD/$lib timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $out/conv_$which.txt
done
tail -n 30 $out/dual_prev.txt $out/dual_new.txt $out/stream_prev.txt $out/stream_new.txt $out/nets_prev.txt $out/nets_new.txt $out/alg1_prev.txt $out/alg1_new.txt $out/alg2_prev.txt $out/alg2_new.txt
paste -d'\n' $out/conv_prev.txt $out/conv_new.txt | cut -c1-200 | tail -34
