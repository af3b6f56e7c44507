#!/bin/bash
# Counter study of the wide 1x1 kernels as tools/bench_conv.py runs them (one --pmc pass per counter group, kernel-trace only):
# fabric reads / writes (FETCH_SIZE x2 per the gfx950 note), L2 hits, SQ issue / stall breakdown.
# Usage: bash tools/gpu_pmc_wide.sh <tag> [only-substring] [B]
tag=${1:-pw}; only=${2:-l3 c3}; B=${3:-72}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p -- python tools/bench_conv.py $B "$only" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
  rm -rf $out/p$i
done
python tools/pmc_summary.py $out/pass*.csv | grep -v "at::native\|rocclr\|pack_conv\|elementwise\|fill" | tee $out/summary.txt
