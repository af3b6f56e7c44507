#!/bin/bash
# SQ issue / stall breakdown of the kernels tools/explore_stream.py runs (one --pmc pass per counter group, kernel-trace only).
# Usage: bash tools/gpu_pmc_sq.sh <tag> [B] [layer] [only]
tag=${1:-sq}; B=${2:-72}; L=${3:-2}; only=${4:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p -- python tools/explore_stream.py $B $L "$only" 2 > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
  rm -rf $out/p$i
done
python tools/pmc_summary.py $out/pass*.csv | grep -v "at::native\|rocclr\|pack_conv" | tee $out/summary.txt
