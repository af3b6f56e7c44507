"""DUAL data gradient (BatchNorm-backward apply folded into the loader of a 1x1 data gradient) on the Sound-MobileNetV2 expand
convs of the B=72 x 5-segment step: time with / without the dz side output and the BatchNorm-fused epilogue."""
import sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV, G, B = "cuda", 5, 72


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


bf = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)
for H, Cin, Cexp in [(128, 16, 96), (64, 24, 144), (32, 32, 192), (16, 64, 384)]:
    N = B
    w = torch.randn(Cexp, Cin, 1, 1, device=DEV) * 0.05
    wd = torch.empty(Cin, Cexp, dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w), ptr(wd), Cexp, Cin, Cin, 1, 1, 1)
    d = ConvDesc(N, H, H, Cin, H, H, Cexp, 1, 1, 1, 0, 1, 0, 0, G, 0)
    g, z, dzs = bf(G * N, H, H, Cexp), bf(G * N, H, H, Cexp), bf(G * N, H, H, Cexp)
    dx, zin = bf(G * N, H, H, Cin), bf(G * N, H, H, Cin)
    aff = torch.rand(G, 3, Cexp, device=DEV)
    vin = torch.rand(G, 4, Cin, device=DEV) + 0.5
    sm = torch.zeros(G, STAT_SLOTS, 2 * Cin, dtype=torch.float64, device=DEV)
    X = g.numel() * 2 / 1e9
    m = dx.numel() * 2 / 1e9
    full = timeit(lambda: call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(z), ptr(aff), ptr(dzs), ptr(wd), ptr(dx), 0, ptr(zin), ptr(vin), 0, ptr(sm)))
    noside = timeit(lambda: call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(z), ptr(aff), None, ptr(wd), ptr(dx), 0, ptr(zin), ptr(vin), 0, ptr(sm)))
    noepi = timeit(lambda: call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(z), ptr(aff), ptr(dzs), ptr(wd), ptr(dx), 0, None, None, 0, None))
    bare = timeit(lambda: call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(z), ptr(aff), None, ptr(wd), ptr(dx), 0, None, None, 0, None))
    coef = torch.rand(G, 3, Cexp, device=DEV)
    vec = torch.rand(G, 4, Cexp, device=DEV) + 0.5
    ap = timeit(lambda: call("adamml_bn_bwd_apply", ptr(g), ptr(z), ptr(vec), 2, ptr(coef), ptr(dzs), N * H * H, Cexp, G))
    print("H=%3d %3d->%3d  %.2f GB | full %.3f ms %4.0f GB/s | no side %.3f %4.0f | no epilogue %.3f %4.0f | bare %.3f %4.0f | bn_bwd_apply alone %.3f %4.0f"
          % (H, Cexp, Cin, 3 * X + 2 * m, full, (3 * X + 2 * m) / full * 1e3, noside, (2 * X + 2 * m) / noside * 1e3, noepi, (3 * X + m) / noepi * 1e3,
             bare, (2 * X + m) / bare * 1e3, ap, 3 * X / ap * 1e3), flush=True)
