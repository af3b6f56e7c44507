#!/bin/bash
# HBM traffic per kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950), kernel-trace only.
out=gpurun_out/${1:-pmc}; round=${2:-r04}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- python bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/f.log 2>&1
timeout 1500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- python bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/w.log 2>&1
f=$(find $out/f -name '*counter_collection.csv' | head -1); w=$(find $out/w -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $f $w $out/${round}_pmc_hbm_traffic.json | tee $out/summary.txt
rm -rf $out/f $out/w     # the raw per-dispatch CSVs are hundreds of MB
