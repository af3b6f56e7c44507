#!/bin/bash
# Scaling curve of the headline workload on ONE multi-GPU node (BASELINE.json configs[2]): `bench.py --gpus N` for N = 1, 2, 4, 8
# (capped at the GPUs visible), weak scaling (72 videos per GPU) and strong scaling (the reference's own semantics: a GLOBAL batch of 72
# split over the ranks, train_adamml.py:122), one JSON line per run on stdout -- the same line bench.py prints, nothing computed here
# (efficiency is for the reader to derive from the per-N `value`s).  bench.py spawns its own ranks, one process per GPU over RCCL.
# Usage: bash tools/scale8.sh [steps] [warmup] > scale.jsonl
steps=${1:-8}; warmup=${2:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
for scaling in weak strong; do
  for n in 1 2 4 8; do
    [ "$n" -le "$ngpu" ] || continue
    [ "$scaling" = strong ] && [ "$n" = 1 ] && continue          # (identical to the weak N = 1 line)
    python bench.py --gpus $n --steps $steps --warmup $warmup --scaling $scaling --no-cpu-baseline --no-roofline 2> /dev/null | tail -1
  done
done
