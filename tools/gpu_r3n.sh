#!/bin/bash
out=gpurun_out/r3n; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -k "c64 or conv_fwd_bwd or groups_equal or 3x3" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 2>$out/err.txt | grep '"metric"' > $out/b$i.json; python -c "
import json; d=json.loads(open('$out/b$i.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['host_issue_ms'])"; done
