"""Backward of the temporal pool behind ResNet-50 stage 1 at the benchmark shape (5 groups x 72 clips x 8 frames of 56 x 56, 256 channels):
adamml_temporal_pool_bwd_code followed by the product g2^T a (adamml_conv_bwd_weight_grouped, which reads g2 back) against the one pass of
csrc/tpool_bwd_prod.hip (adamml_temporal_pool_bwd_code_prod).  GPU box only; nothing here is part of the product path."""
import os
import sys
from ctypes import byref

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402
from adamml_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
G, clips, T, H, C, Cin = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 72, 8, 56, 256, 64
To, Q = T // 2, H * H
gy = torch.randn(G * clips * To, H, H, C, device=dev).to(torch.bfloat16)
code = torch.randint(0, 1 << 16, (G * clips * To, H, H, C // 8), device=dev, dtype=torch.int32).to(torch.int16)
a = (torch.randn(G * clips * T, H, H, Cin, device=dev) * 1.5).to(torch.bfloat16)
avec = torch.rand(G, 4, Cin, device=dev) + 0.5
g2 = torch.empty(G * clips * T, H, H, C, dtype=torch.bfloat16, device=dev)
s = torch.zeros(G, 32, 2 * C, dtype=torch.float64, device=dev)
P = torch.empty(G, C, Cin, device=dev)
d = hip.ConvDesc(clips * T, H, H, Cin, H, H, C, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cin)
ws = hip.wgrad_workspace(d, Cin, dev)
ws2 = hip.scratch(max(hip.load().adamml_temporal_pool_bwd_code_prod_workspace(clips, T, Q, C, Cin, G), 4), dev)


def expand():
    call("adamml_temporal_pool_bwd_code", ptr(gy), ptr(code), ptr(g2), ptr(s), clips, T, Q, C, G)


def product():
    call("adamml_conv_bwd_weight_grouped", byref(d), ptr(g2), None, None, 0, 0, ptr(a), ptr(avec[0, 0]), ptr(avec[0, 1]), ptr(P), Cin, ptr(ws), ws.numel() * 4)


def two():
    expand()
    product()


def fused():
    call("adamml_temporal_pool_bwd_code_prod", ptr(gy), ptr(code), ptr(g2), ptr(s), ptr(a), ptr(avec[0, 0]), ptr(avec[0, 1]), 4 * Cin, 1, ptr(P),
         ptr(ws2), ws2.numel() * 4, clips, T, Q, C, Cin, G)


gb = G * clips * Q * (To * C * 2 + To * C // 4 + T * C * 2 + T * Cin * 2) / 1e9
for r in range(3):
    for name, fn in (("expand", expand), ("product", product), ("two launches", two), ("one pass", fused)):
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
        print("round %d  %-14s %.3f ms   (the one pass touches %.2f GB: %.0f GB/s)" % (r, name, ms, gb, gb / ms * 1e3), flush=True)
