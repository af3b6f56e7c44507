"""Last bottleneck of ResNet-50 layer 1 at the benchmark shape (5 groups x 72 clips x 8 frames of 56 x 56): conv3 + bn3 + add + ReLU + temporal
max-pool in one kernel (adamml_conv_fwd_bn_add_tpool).  ADAMML_FADD_TPOOL_STREAM=0 selects conv_gemm_kernel's TP instance, the default
the streaming kernel of csrc/conv1x1_fadd_next.hip (one process per form: the switch is read once).  GPU box only."""
import os
import sys
from ctypes import byref

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402
from adamml_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
G, clips, T, H, Cin, Cout = 5, 72, 8, 56, 64, 256
N = clips * T
x = (torch.randn(G * N, H, H, Cin, device=dev) * 1.5).to(torch.bfloat16)
xvec = torch.rand(G, 4, Cin, device=dev) + 0.5
w3 = (torch.randn(Cout, Cin, device=dev) * 0.1).to(torch.bfloat16)
vec = torch.rand(G, 4, Cout, device=dev) + 0.5
idn = torch.randn(G * N, H, H, Cout, device=dev).to(torch.bfloat16)
pooled = torch.empty(G * N // 2, H, H, Cout, dtype=torch.bfloat16, device=dev)
code = torch.empty(G * N // 2, H, H, Cout // 8, dtype=torch.int16, device=dev)
d = hip.ConvDesc(N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cin)


def fn():
    call("adamml_conv_fwd_bn_add_tpool", byref(d), ptr(x), ptr(w3), ptr(xvec[0, 0]), ptr(xvec[0, 1]), ptr(vec), ptr(idn), None, None, 0, 1, T, ptr(pooled), ptr(code))


gb = G * N * H * H * (Cin + Cout + Cout // 2 + Cout // 16) * 2 / 1e9
for r in range(3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("round %d  ADAMML_FADD_TPOOL_STREAM=%s  %.3f ms   (%.2f GB: %.0f GB/s)" % (r, os.environ.get("ADAMML_FADD_TPOOL_STREAM", "1"), ms, gb, gb / ms * 1e3))
