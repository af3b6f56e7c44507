#!/bin/bash
# 2 gloo ranks on ONE GPU: exercises the N>1 code path (SyncBN exchange, bucketed gradient all-reduce, interleaving).
# usage: tools/gloo2.sh <batch> [extra env assignments...]
b=$1; shift
env "$@" ADAMML_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 --steps 3 --warmup 2 --batch $b ${GLOO2_ARGS:---no-roofline} 2>/tmp/gloo2.err | grep '"metric"' | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'loss', d['loss'], 'mem', d['peak_mem_gib'])" || tail -20 /tmp/gloo2.err
