#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
