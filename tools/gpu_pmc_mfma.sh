#!/bin/bash
# SQ counter study of the MFMA-bound 3x3 layers (forward / data gradient / weight gradient kernels as tools/bench_conv.py runs them):
# one --pmc pass per counter group and layer, kernel-trace only.  Usage: bash tools/gpu_pmc_mfma.sh <tag> [B]
tag=${1:-mfma}; B=${2:-72}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
: > $out/summary.txt
for layer in "l1 c2" "l2 c2" "l3 c2" "l4 c2"; do
  key=$(echo $layer | tr ' ' '_')
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p -o p -- python tools/bench_conv.py $B "$layer" > $out/${key}_p$i.log 2>&1
    f=$(find $out/p -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp $f $out/${key}_pass$i.csv
    rm -rf $out/p
  done
  echo "== $layer (x s1; tools/bench_conv.py $B '$layer')" >> $out/summary.txt
  grep "^$layer " $out/${key}_p1.log | cut -c1-230 >> $out/summary.txt
  python tools/pmc_summary.py $out/${key}_pass*.csv | grep -v "at::native\|rocclr\|pack_\|reduce\|elementwise\|fill" >> $out/summary.txt
  rm -f $out/${key}_pass*.csv
done
cat $out/summary.txt | cut -c1-400
