#!/bin/bash
# Counter passes over tools/explore_stream.py (kernel-trace + --pmc only; one counter group per pass).
# Usage: bash tools/gpu_pmc_stream.sh <tag> [B] [layer] [only]
tag=${1:-pmcs}; B=${2:-72}; L=${3:-1}; only=${4:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python tools/explore_stream.py $B $L $only 5 2>&1 | grep -v amdgpu.ids | tee $out/times.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p -- python tools/explore_stream.py $B $L $only 2 > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
  rm -rf $out/p$i
done
python tools/pmc_summary.py $out/pass*.csv | tee $out/summary.txt
