"""Backward of the stride-1 depthwise convs of the Sound-MobileNetV2 at the benchmark's shapes (5 groups x 72 spectrograms): the per-layer
form (adamml_bn_bwd_apply, adamml_dwconv_bwd_weight, adamml_dwconv_bwd_data_bn: eight passes over the block's widest tensor) against the
one-pass kernel of csrc/dwconv_bwd_fused.hip (four).  GPU box only; nothing here is part of the product path."""
import os
import sys
from ctypes import byref

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402
from adamml_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
G, N = 5, 72
SHAPES = [(128, 32, 1), (64, 144, 1), (32, 192, 1), (16, 384, 1), (16, 576, 1), (8, 960, 1), (128, 96, 2), (64, 144, 2), (32, 192, 2), (16, 576, 2)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]            # H x C x stride
tot = {"per-layer": 0.0, "fused": 0.0}
for H, C, st in SHAPES:
    OH = (H - 1) // st + 1
    d = hip.ConvDesc(N, H, H, C, OH, OH, C, 3, 3, st, 1, 1, 2, 0, G, 4 * C)
    g = torch.randn(G * N, OH, OH, C, device=dev).to(torch.bfloat16)
    z = torch.randn(G * N, OH, OH, C, device=dev).to(torch.bfloat16)
    x = (torch.randn(G * N, H, H, C, device=dev) * 2).to(torch.bfloat16)
    vec = torch.rand(G, 4, C, device=dev) + 0.5
    xvec = torch.rand(G, 4, C, device=dev) + 0.5
    coef = torch.randn(G, 3, C, device=dev) * 0.3
    aff = torch.randn(G, 3, C, device=dev) * 0.3
    wp = torch.randn(9, C, device=dev) * 0.3
    dz, dx = torch.empty_like(g), torch.empty_like(x)
    dw = torch.zeros(C, 1, 3, 3, device=dev)
    sums = torch.zeros(G, 32, 2 * C, dtype=torch.float64, device=dev)
    ws1 = hip.wgrad_workspace(d, 0, dev, depthwise=True)
    ws2 = hip.scratch(hip.load().adamml_dwconv_bwd_fused_workspace(byref(d)), dev)
    P = N * OH * OH

    def per_layer():
        call("adamml_bn_bwd_apply", ptr(g), ptr(z), ptr(vec), 2, ptr(coef), ptr(dz), P, C, G)
        call("adamml_dwconv_bwd_weight", byref(d), ptr(dz), ptr(x), ptr(xvec[0, 0]), ptr(xvec[0, 1]), ptr(dw), ptr(ws1), ws1.numel() * 4)
        call("adamml_dwconv_bwd_data_bn", byref(d), ptr(dz), ptr(wp), ptr(dx), ptr(x), ptr(xvec), 2, ptr(sums))

    def fused():
        call("adamml_dwconv_bwd_fused", byref(d), ptr(g), ptr(z), ptr(aff), ptr(wp), ptr(x), ptr(xvec), 2, ptr(dx), ptr(sums), ptr(dw), ptr(ws2),
             ws2.numel() * 4)

    gb = 2 * G * N * (H * H + OH * OH) * C * 2 / 1e9
    for name, fn in (("per-layer", per_layer), ("fused", fused)):
        best = 1e9
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
            best = min(best, e0.elapsed_time(e1) / 10)
        tot[name] += best
        print("%3d x %3d x %4d s%d  %-10s %.3f ms   (g, z, x, dx once = %.2f GB: %.0f GB/s)" % (H, H, C, st, name, best, gb, gb / best * 1e3), flush=True)
print("sum  per-layer %.3f ms   fused %.3f ms" % (tot["per-layer"], tot["fused"]))
