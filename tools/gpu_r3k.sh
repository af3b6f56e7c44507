#!/bin/bash
# SQ / LDS counters of conv3x3_c64 (layer-1 conv2, forward + BatchNorm-fused data gradient as tools/bench_conv.py runs them) for the three
# LDS layouts of round 3: unswizzled weight rows (round 2), swizzled weight rows (default), swizzled + 160-byte patch pitch.
out=gpurun_out/r3k; mkdir -p $out
export TMPDIR=/tmp
: > $out/summary.txt
for cfg in "0 144" "1 144" "1 160"; do set -- $cfg
  key=wswz$1_pitch$2
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ADAMML_C64_WSWZ=$1 ADAMML_C64_PITCH=$2 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p -o p -- python tools/bench_conv.py 72 "l1 c2" > $out/${key}_p$i.log 2>&1
    f=$(find $out/p -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp $f $out/${key}_pass$i.csv
    rm -rf $out/p
  done
  echo "== conv3x3_c64, layer-1 conv2 (tools/bench_conv.py 72 'l1 c2'), ADAMML_C64_WSWZ=$1 ADAMML_C64_PITCH=$2" >> $out/summary.txt
  grep "^l1 c2 " $out/${key}_p1.log | cut -c1-230 >> $out/summary.txt
  python tools/pmc_summary.py $out/${key}_pass*.csv | grep "conv3x3_c64" >> $out/summary.txt
  rm -f $out/${key}_pass*.csv
done
cat $out/summary.txt | cut -c1-420
