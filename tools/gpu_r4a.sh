#!/bin/bash
for i in 1 2 3; do for r in 0 1; do
  ADAMML_REDUCE4=$r timeout 900 python -m pytest tests/test_parity_fullsize_gpu.py -q -s -k "train_policy" 2>&1 | grep "290 gradient tensors\|passed\|failed" | sed "s/^/reduce4=$r /" | cut -c1-230
done; done
