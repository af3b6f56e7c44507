"""Micro-benchmark of the depthwise kernels on the MobileNetV2 shapes of the B=72 x 5-segment step (G=5 groups of 72 images)."""
import sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd import hip
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV = "cuda"

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

G, N = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 72
tot = [0, 0, 0, 0]
for (H, C, s) in [(128, 32, 1), (128, 96, 2), (64, 144, 1), (64, 144, 2), (32, 192, 1), (32, 192, 2), (16, 384, 1), (16, 576, 1), (16, 576, 2), (8, 960, 1)]:
    OH = (H - 1) // s + 1
    x = torch.randn(G * N, H, H, C, device=DEV).to(torch.bfloat16)
    dz = torch.randn(G * N, OH, OH, C, device=DEV).to(torch.bfloat16)
    y = torch.empty_like(dz); dx = torch.empty_like(x)
    w = torch.randn(9, C, device=DEV)
    vec = torch.rand(G, 4, C, device=DEV) + 0.5
    st = torch.zeros(G, STAT_SLOTS, 2 * C, dtype=torch.float64, device=DEV)
    d = ConvDesc(N, H, H, C, OH, OH, C, 3, 3, s, 1, 1, 2, 0, G, 4 * C)
    dwt = torch.zeros(C, 1, 3, 3, device=DEV)
    ws = hip.wgrad_workspace(d, 0, DEV, depthwise=True)
    gb = (x.numel() + y.numel()) * 2 / 1e9
    t1 = timeit(lambda: call("adamml_dwconv_fwd", byref(d), ptr(x), ptr(w), ptr(vec[0, 0]), ptr(vec[0, 1]), ptr(y), ptr(st) if "nostats" not in sys.argv else None))
    t2 = timeit(lambda: call("adamml_dwconv_bwd_data", byref(d), ptr(dz), ptr(w), ptr(dx), 0))
    t3 = timeit(lambda: call("adamml_dwconv_bwd_weight", byref(d), ptr(dz), ptr(x), ptr(vec[0, 0]), ptr(vec[0, 1]), ptr(dwt), ptr(ws), ws.numel() * 4))
    t4 = timeit(lambda: call("adamml_dwconv_bwd_data_bn", byref(d), ptr(dz), ptr(w), ptr(dx), ptr(x), ptr(vec), 2, ptr(st)))
    tot[0] += t1; tot[1] += t2; tot[2] += t3; tot[3] += t4
    print("H=%3d C=%3d s=%d  %.2f GB | fwd %.3f ms %5.0f GB/s | bwd_data %.3f ms %5.0f GB/s | bwd_weight %.3f ms %5.0f GB/s | bwd_data_bn %.3f ms %5.0f GB/s"
          % (H, C, s, gb, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3, t4, (gb + x.numel() * 2 / 1e9) / t4 * 1e3))
print("sum: fwd %.2f ms, bwd_data %.2f ms, bwd_weight %.2f ms, bwd_data_bn %.2f ms" % tuple(tot))
