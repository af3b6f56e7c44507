timeout 2400 python -m pytest tests/test_parity_fullsize_gpu.py -x -q -s -k deterministic 2>&1 | grep -v amdgpu.ids | tail -40
