python tools/_wg_check.py 2>&1 | grep -v amdgpu | tail -7
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -3
for g in 1 0; do ADAMML_WGRAD_GLDS=$g timeout 600 python bench.py --no-cpu-baseline --steps 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']; print('glds=$g', d['value'], d['ms_per_step'], 'wgrad ms', kb['adamml_conv_bwd_weight']['ms'], kb['adamml_conv_bwd_weight_grouped']['ms'])"; done
ADAMML_WGRAD_GLDS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('glds=1', d['value'], d['ms_per_step'])"
ADAMML_WGRAD_GLDS=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('glds=0', d['value'], d['ms_per_step'])"
