"""Micro-benchmark of the fused data-gradient variants (RES epilogue, DUAL loader) on the ResNet-50 shapes of the B=72 x
5-segment step: achieved HBM GB/s over the tensors each launch really touches."""
import sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd import hip
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV = "cuda"
G = 5
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def bf(*shape):
    return torch.randn(*shape, device=DEV).to(torch.bfloat16)


# (name, N per group, H, Cmid, Cbig): conv1 is Cbig -> Cmid (its dgrad has the RES epilogue), conv3 is Cmid -> Cbig (DUAL loader)
for name, N, H, Cm, Cb in [("layer1", B * 8, 56, 64, 256), ("layer2", B * 4, 28, 128, 512), ("layer3", B * 2, 14, 256, 1024), ("layer4", B, 7, 512, 2048)]:
    big = lambda: bf(G * N, H, H, Cb)
    mid = lambda: bf(G * N, H, H, Cm)
    X = G * N * H * H * Cb * 2 / 1e9
    m = X * Cm / Cb
    # ---- RES: dgrad of conv1 (Cb -> Cm) accumulating onto the identity gradient, mask from out, sums with z3
    w1 = torch.randn(Cm, Cb, 1, 1, device=DEV) * 0.05
    wd1 = torch.empty(Cb, Cm, dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w1), ptr(wd1), Cm, Cb, Cb, 1, 1, 1)
    d1 = ConvDesc(N, H, H, Cb, H, H, Cm, 1, 1, 1, 0, 1, 0, 0, G, 0)
    dz1, gid, out, z3 = mid(), big(), big().clamp_min(0), big()
    vec = torch.rand(G, 4, Cb, device=DEV) + 0.5
    sa = torch.zeros(G, STAT_SLOTS, 2 * Cb, dtype=torch.float64, device=DEV)
    t_res = timeit(lambda: call("adamml_conv_bwd_data_res", byref(d1), ptr(dz1), ptr(wd1), ptr(gid), 1, ptr(out), None, 1, ptr(z3), ptr(vec), ptr(sa),
                                None, None, None))
    t_acc = timeit(lambda: call("adamml_conv_bwd_data", byref(d1), ptr(dz1), ptr(wd1), ptr(gid), 1))
    g2 = torch.empty_like(gid)
    t_rb = timeit(lambda: call("adamml_residual_bwd", ptr(gid), ptr(out), 1, ptr(g2), ptr(z3), ptr(vec), ptr(sa), None, None, None, N * H * H, Cb, G))
    b_res = 4 * X + m
    print("%s RES  dgrad %4d->%4d: %.3f ms %5.0f GB/s (%.1f GB) | unfused: dgrad+acc %.3f ms %5.0f GB/s + residual_bwd %.3f ms %5.0f GB/s"
          % (name, Cm, Cb, t_res, b_res / t_res * 1e3, b_res, t_acc, (2 * X + m) / t_acc * 1e3, t_rb, 4 * X / t_rb * 1e3))
    del gid, out, g2
    # ---- DUAL: dgrad of conv3 (Cm -> Cb) reading (g, z3), side output dz, BatchNorm epilogue for a2
    w3 = torch.randn(Cb, Cm, 1, 1, device=DEV) * 0.05
    wd3 = torch.empty(Cm, Cb, dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w3), ptr(wd3), Cb, Cm, Cm, 1, 1, 1)
    d3 = ConvDesc(N, H, H, Cm, H, H, Cb, 1, 1, 1, 0, 1, 0, 0, G, 0)
    g, dzs, dx, zin = big(), big(), mid(), mid()
    aff = torch.rand(G, 3, Cb, device=DEV)
    coef = torch.rand(G, 3, Cb, device=DEV)
    vin = torch.rand(G, 4, Cm, device=DEV) + 0.5
    sm = torch.zeros(G, STAT_SLOTS, 2 * Cm, dtype=torch.float64, device=DEV)
    if not hip.load().adamml_conv_bwd_data_dual_supported(byref(d3)):
        print("%s DUAL dgrad %4d->%4d: not supported by the dual loader (Cout > 512: apply + data gradient is faster, see conv_gemm.hip)" % (name, Cb, Cm))
        del g, dzs, dx, zin, z3, dz1
        continue
    t_dual = timeit(lambda: call("adamml_conv_bwd_data_dual", byref(d3), ptr(g), ptr(z3), ptr(aff), ptr(dzs), ptr(wd3), ptr(dx), 0, ptr(zin),
                                 ptr(vin), 1, ptr(sm)))
    t_ap = timeit(lambda: call("adamml_bn_bwd_apply", ptr(g), ptr(z3), ptr(vec), 0, ptr(coef), ptr(dzs), N * H * H, Cb, G))
    t_bn = timeit(lambda: call("adamml_conv_bwd_data_bn", byref(d3), ptr(dzs), ptr(wd3), ptr(dx), ptr(zin), ptr(vin), 1, ptr(sm)))
    b_dual = 3 * X + 2 * m
    print("%s DUAL dgrad %4d->%4d: %.3f ms %5.0f GB/s (%.1f GB) | unfused: apply %.3f ms %5.0f GB/s + dgrad_bn %.3f ms %5.0f GB/s"
          % (name, Cb, Cm, t_dual, b_dual / t_dual * 1e3, b_dual, t_ap, 3 * X / t_ap * 1e3, t_bn, (X + 2 * m) / t_bn * 1e3))
    del g, dzs, dx, zin, z3, dz1
