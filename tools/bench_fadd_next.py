"""Layer-1 bottleneck boundary at the benchmark shape (5 groups x 576 frames of 56 x 56): conv3 + bn3 + add + ReLU followed by the next block's
conv1 as two launches (adamml_conv_fwd_bn_add, adamml_conv_fwd) against the one streaming kernel that keeps the block-output tile in LDS
(adamml_conv_fwd_bn_add_next, csrc/conv1x1_fadd_next.hip).  GPU box only; nothing here is part of the product path."""
import os
import sys
from ctypes import byref

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402
from adamml_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
G, N, H, Cin, Cout, Cn = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 576, 56, 64, 256, 64
x = (torch.randn(G * N, H, H, Cin, device=dev) * 1.5).to(torch.bfloat16)
xvec = torch.rand(G, 4, Cin, device=dev) + 0.5
w3 = (torch.randn(Cout, Cin, device=dev) * 0.1).to(torch.bfloat16)
w1 = (torch.randn(Cn, Cout, device=dev) * 0.05).to(torch.bfloat16)
vec = torch.rand(G, 4, Cout, device=dev) + 0.5
idn = torch.randn(G * N, H, H, Cout, device=dev).to(torch.bfloat16)
out = torch.empty_like(idn)
mask = torch.empty(G * N, H, H, Cout // 8, dtype=torch.uint8, device=dev)
y = torch.empty(G * N, H, H, Cn, dtype=torch.bfloat16, device=dev)
st = torch.zeros(G, 32, 2 * Cn, dtype=torch.float64, device=dev)
d3 = hip.ConvDesc(N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cin)
d1 = hip.ConvDesc(N, H, H, Cout, H, H, Cn, 1, 1, 1, 0, 1, 0, 0, G, 0)


def pair():
    call("adamml_conv_fwd_bn_add", byref(d3), ptr(x), ptr(w3), ptr(xvec[0, 0]), ptr(xvec[0, 1]), ptr(vec), ptr(idn), None, None, 0, 1, ptr(out), ptr(mask))
    call("adamml_conv_fwd", byref(d1), ptr(out), ptr(w1), None, None, ptr(y), ptr(st))


def fused():
    call("adamml_conv_fwd_bn_add_next", byref(d3), ptr(x), ptr(w3), ptr(xvec[0, 0]), ptr(xvec[0, 1]), ptr(vec), ptr(idn), None, None, 0, 1, ptr(out), ptr(mask),
         ptr(w1), ptr(y), ptr(st))


def fadd_only():
    call("adamml_conv_fwd_bn_add", byref(d3), ptr(x), ptr(w3), ptr(xvec[0, 0]), ptr(xvec[0, 1]), ptr(vec), ptr(idn), None, None, 0, 1, ptr(out), ptr(mask))


gb = G * N * H * H * (Cin + 2 * Cout + Cout // 16 + Cn) * 2 / 1e9
for r in range(3):
    for name, fn in (("two launches", pair), ("conv_fwd_bn_add alone", fadd_only), ("one streaming kernel", fused)):
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
        print("round %d  %-24s %.3f ms   (%.2f GB touched by the fused form: %.0f GB/s)" % (r, name, ms, gb, gb / ms * 1e3))
