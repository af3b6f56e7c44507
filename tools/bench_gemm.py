"""fp32 GEMM of the policy head (adamml_gemm_f32) at the benchmark's shapes -- M = 72 videos x 5 segments = 360 rows; joint FC 2560 -> 2048,
LSTM input gates 2048 -> 2048, a classifier-sized product -- with the matrix-core kernel (ADAMML_GEMM_MFMA=1, the default for
K-contiguous operands) and the 64 x 64 VALU tiles it replaced (=0), each in its own process (the switch is read once), next to torch.matmul.
GPU box only; nothing here is part of the product path.   usage: python tools/bench_gemm.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from adamml_amd.runtime import gemm_f32
    dev = torch.device("cuda:0")
    for M, N, K in ((360, 2048, 2560), (360, 2048, 2048), (360, 1024, 2048), (2880, 31, 2048), (72, 2048, 512)):
        a, b, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        res = []
        for fn in (lambda: gemm_f32(a, b, out=out, bias=bias, act=1), lambda: torch.addmm(bias, a, b.t())):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print("  M %5d N %5d K %5d: adamml_gemm_f32 %7.1f us (%5.1f TFLOP/s)   torch.addmm %7.1f us" % (M, N, K, res[0], 2.0 * M * N * K / res[0] * 1e-6, res[1]))
else:
    for v in ("0", "1"):
        print("ADAMML_GEMM_MFMA=%s" % v)
        sys.stdout.flush()
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ADAMML_GEMM_MFMA=v))
