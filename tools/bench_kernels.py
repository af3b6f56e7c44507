#!/usr/bin/env python3
"""Per-signature micro-benchmark of the conv kernels (GPU box): ResNet-50 / MobileNetV2 layer shapes at a given number
of frames; prints time, TFLOP/s, GB/s (algorithmic) and the binding roofline fraction per entry point."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from adamml_amd.hip import ConvDesc, call, ptr

DEV = "cuda"
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 64          # frames (N*T) at the stem; temporal pooling halves it per stage
WHICH = sys.argv[2] if len(sys.argv) > 2 else "resnet"

# (Cin, Cout, k, s, T-divisor, Hin, count)
RESNET = [(8, 64, 7, 2, 1, 224, 1), (64, 64, 1, 1, 1, 56, 1), (64, 64, 3, 1, 1, 56, 3), (64, 256, 1, 1, 1, 56, 4), (256, 64, 1, 1, 1, 56, 2),
          (256, 128, 1, 1, 2, 56, 1), (128, 128, 3, 2, 2, 56, 1), (128, 512, 1, 1, 2, 28, 4), (256, 512, 1, 2, 2, 56, 1),
          (512, 128, 1, 1, 2, 28, 3), (128, 128, 3, 1, 2, 28, 3), (512, 256, 1, 1, 4, 28, 1), (256, 256, 3, 2, 4, 28, 1),
          (256, 1024, 1, 1, 4, 14, 6), (512, 1024, 1, 2, 4, 28, 1), (1024, 256, 1, 1, 4, 14, 5), (256, 256, 3, 1, 4, 14, 5),
          (1024, 512, 1, 1, 8, 14, 1), (512, 512, 3, 2, 8, 14, 1), (512, 2048, 1, 1, 8, 7, 3), (1024, 2048, 1, 2, 8, 14, 1),
          (2048, 512, 1, 1, 8, 7, 2), (512, 512, 3, 1, 8, 7, 2)]
MBV2 = [(8, 32, 3, 2, 1, 256, 1), (32, 16, 1, 1, 1, 128, 1), (16, 96, 1, 1, 1, 128, 1), (96, 24, 1, 1, 1, 64, 1), (24, 144, 1, 1, 1, 64, 2),
        (144, 24, 1, 1, 1, 64, 1), (144, 32, 1, 1, 1, 32, 1), (32, 192, 1, 1, 1, 32, 3), (192, 32, 1, 1, 1, 32, 2), (192, 64, 1, 1, 1, 16, 1),
        (64, 384, 1, 1, 1, 16, 4), (384, 64, 1, 1, 1, 16, 3), (384, 96, 1, 1, 1, 16, 1), (96, 576, 1, 1, 1, 16, 3), (576, 96, 1, 1, 1, 16, 2),
        (576, 160, 1, 1, 1, 8, 1), (160, 960, 1, 1, 1, 8, 3), (960, 160, 1, 1, 1, 8, 2), (960, 320, 1, 1, 1, 8, 1), (320, 1280, 1, 1, 1, 8, 1)]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    floor = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-26s %8s | %-26s | %-26s | %-26s" % ("sig (Cin Cout k s N H)", "GFLOP", "fwd ms TF GB/s", "dgrad ms TF GB/s", "wgrad ms TF GB/s"))
    only = [int(v) for v in os.environ.get("ONLY", "").split(",") if v]
    table = RESNET if WHICH == "resnet" else MBV2
    if only:
        table = [table[i] for i in only]
    for cin, cout, k, s, tdiv, h, cnt in table:
        n = max(1, NF // tdiv)
        p = (k - 1) // 2
        oh = (h + 2 * p - k) // s + 1
        d = ConvDesc(n, h, h, cin, oh, oh, cout, k, k, s, p, 1, 1, 0)
        x = torch.randn(n, h, h, cin, device=DEV).bfloat16()
        w = torch.randn(cout, k * k * cin, device=DEV).bfloat16()
        wd = torch.randn(cin, k * k * cout, device=DEV).bfloat16()
        y = torch.empty(n, oh, oh, cout, device=DEV, dtype=torch.bfloat16)
        dz = torch.randn(n, oh, oh, cout, device=DEV).bfloat16()
        dx = torch.empty_like(x)
        dw = torch.zeros(cout, cin, k, k, device=DEV)
        stats = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=DEV)
        sc, sh = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV)
        flops = 2.0 * n * oh * oh * cout * k * k * cin
        by = 2.0 * (n * h * h * cin + n * oh * oh * cout)
        t_f = timeit(lambda: call("adamml_conv_fwd", byref(d), ptr(x), ptr(w), ptr(sc), ptr(sh), ptr(y), ptr(stats)))
        t_d = timeit(lambda: call("adamml_conv_bwd_data", byref(d), ptr(dz), ptr(wd), ptr(dx), 0))
        from adamml_amd import hip as _hip
        ws = _hip.wgrad_workspace(d, cin, dz.device)
        t_w = timeit(lambda: call("adamml_conv_bwd_weight", byref(d), ptr(dz), ptr(x), ptr(sc), ptr(sh), ptr(dw), cin, ptr(ws), ws.numel() * 4))
        row = "%4d %4d %d %d %4d %3d x%d" % (cin, cout, k, s, n, h, cnt)
        cells = []
        for key, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[key] += t * cnt
            floor[key] += cnt * max(flops / 2.5e15, by / 8e12) * 1e3
            cells.append("%7.3f %6.1f %7.1f" % (t, flops / t / 1e9, by / t / 1e6))
        print("%-26s %8.2f | %-26s | %-26s | %-26s" % (row, flops / 1e9, *cells))
    print("TOTAL ms (x count): fwd %.2f dgrad %.2f wgrad %.2f | roofline floors: fwd %.2f dgrad %.2f wgrad %.2f" %
          (tot["fwd"], tot["dgrad"], tot["wgrad"], floor["fwd"], floor["dgrad"], floor["wgrad"]))


if __name__ == "__main__":
    main()
