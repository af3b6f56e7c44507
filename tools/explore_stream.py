"""Layer-1/2 streaming kernels in the form the net runs them (FADD, RES with mask, algebraic data gradient, plain 1x1 forward,
BatchNorm-backward apply, elementwise references), one timing line each; `only` selects kernels by substring so the same script
serves rocprofv3 --pmc passes (tools/gpu_pmc_stream.sh).  Usage: python tools/explore_stream.py [B] [layer] [only] [reps]"""
import sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd import hip
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV = "cuda"
G = 5
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
LAYER = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ONLY = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "all" else None
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
N, H, Cm, Cb = {1: (B * 8, 56, 64, 256), 2: (B * 4, 28, 128, 512), 3: (B * 2, 14, 256, 1024)}[LAYER]


def timeit(fn, n=REPS):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def bf(*shape):
    return torch.randn(*shape, device=DEV).to(torch.bfloat16)


def run(name, gb, fn):
    if ONLY and ONLY not in name:
        return
    t = timeit(fn)
    print("L%d %-34s %.3f ms  %5.0f GB/s  (%.2f GB)" % (LAYER, name, t, gb / t * 1e3, gb), flush=True)


P = N * H * H
X = G * P * Cb * 2 / 1e9           # block-output sized tensor
m = X * Cm / Cb
big = lambda: bf(G * N, H, H, Cb)
mid = lambda: bf(G * N, H, H, Cm)
xb, xb2, ob = big(), big(), torch.empty(G * N, H, H, Cb, dtype=torch.bfloat16, device=DEV)
xm, om = mid(), torch.empty(G * N, H, H, Cm, dtype=torch.bfloat16, device=DEV)
mask = torch.randint(0, 255, (G * P * Cb // 8,), dtype=torch.uint8, device=DEV)
vecb = torch.rand(G, 4, Cb, device=DEV) + 0.5
vecm = torch.rand(G, 4, Cm, device=DEV) + 0.5
coef = torch.rand(G, 3, Cb, device=DEV)
stb = torch.zeros(G, STAT_SLOTS, 2 * Cb, dtype=torch.float64, device=DEV)
stm = torch.zeros(G, STAT_SLOTS, 2 * Cm, dtype=torch.float64, device=DEV)

run("copy big", 2 * X, lambda: ob.copy_(xb))
run("torch add big", 3 * X, lambda: torch.add(xb, xb2, out=ob))
run("bn_bwd_apply big", 3 * X, lambda: call("adamml_bn_bwd_apply", ptr(xb), ptr(xb2), ptr(vecb), 1, ptr(coef), ptr(ob), P, Cb, G))
run("bn_act_add_mask big", 3 * X + X / 16, lambda: call("adamml_bn_act_add_mask", ptr(xb), ptr(vecb[0, 0]), ptr(vecb[0, 1]), 4 * Cb, 1, ptr(xb2), None, None, 0,
                                                        ptr(ob), ptr(mask), P, Cb, G))

# conv3: Cm -> Cb
w3 = torch.randn(Cb, Cm, 1, 1, device=DEV) * 0.05
wf3 = torch.empty(Cb, Cm, dtype=torch.bfloat16, device=DEV)
wd3 = torch.empty(Cm, Cb, dtype=torch.bfloat16, device=DEV)
call("adamml_pack_conv_weight", ptr(w3), ptr(wf3), Cb, Cm, Cm, 1, 1, 0)
call("adamml_pack_conv_weight", ptr(w3), ptr(wd3), Cb, Cm, Cm, 1, 1, 1)
d3l = ConvDesc(N, H, H, Cm, H, H, Cb, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cm)       # lazy input (act relu)
d3 = ConvDesc(N, H, H, Cm, H, H, Cb, 1, 1, 1, 0, 1, 0, 0, G, 0)
run("conv3 fwd lazy + stats", X + m, lambda: call("adamml_conv_fwd", byref(d3l), ptr(xm), ptr(wf3), ptr(vecm[0, 0]), ptr(vecm[0, 1]), ptr(ob), ptr(stb)))
run("conv3 FADD lazy + id + mask", 2 * X + m + X / 16,
    lambda: call("adamml_conv_fwd_bn_add", byref(d3l), ptr(xm), ptr(wf3), ptr(vecm[0, 0]), ptr(vecm[0, 1]), ptr(vecb), ptr(xb), None, None, 0, 1,
                 ptr(ob), ptr(mask)))
# conv1: Cb -> Cm
w1 = torch.randn(Cm, Cb, 1, 1, device=DEV) * 0.05
wf1 = torch.empty(Cm, Cb, dtype=torch.bfloat16, device=DEV)
wd1 = torch.empty(Cb, Cm, dtype=torch.bfloat16, device=DEV)
call("adamml_pack_conv_weight", ptr(w1), ptr(wf1), Cm, Cb, Cb, 1, 1, 0)
call("adamml_pack_conv_weight", ptr(w1), ptr(wd1), Cm, Cb, Cb, 1, 1, 1)
d1 = ConvDesc(N, H, H, Cb, H, H, Cm, 1, 1, 1, 0, 1, 0, 0, G, 0)
run("conv1 fwd plain + stats", X + m, lambda: call("adamml_conv_fwd", byref(d1), ptr(xb), ptr(wf1), None, None, ptr(om), ptr(stm)))
# RES as the net runs it with the algebraic backward: accumulate onto the identity gradient in ob, 1-bit mask, sum(g') only
run("conv1 RES dgrad acc + mask", 2 * X + m + X / 16,
    lambda: call("adamml_conv_bwd_data_res", byref(d1), ptr(xm), ptr(wd1), ptr(ob), 1, ptr(xb2), ptr(mask), 1, None, ptr(vecb), ptr(stb), None, None, None))
# algebraic data gradient of conv3: dx[Cm] = W_g [g' | a] + c, BatchNorm-fused epilogue for the lazily normalised a
walg = bf(G, Cm, Cb + Cm) * 0.05
eadd = torch.rand(G, Cm, device=DEV)
run("conv3 alg dgrad [g'|a]", X + 3 * m,
    lambda: call("adamml_conv_bwd_data_alg", byref(d3l), ptr(xb), ptr(xm), ptr(vecm[0, 0]), ptr(vecm[0, 1]), ptr(walg), ptr(eadd), ptr(om), 0, ptr(xm),
                 ptr(vecm), 1, ptr(stm)))
# grouped products of the algebraic backward: P = g'^T a (a lazy) and the Gram matrix a^T a
pout = torch.empty(G, Cb, Cm, device=DEV)
gout = torch.empty(G, Cm, Cm, device=DEV)
ws = hip.wgrad_workspace(d3, Cm, DEV)
dg = ConvDesc(N, H, H, Cm, H, H, Cm, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cm)
wsg = hip.wgrad_workspace(dg, Cm, DEV)
run("grouped P = g'^T a", X + m, lambda: call("adamml_conv_bwd_weight_grouped", byref(d3l), ptr(xb), None, None, 0, 0, ptr(xm), ptr(vecm[0, 0]), ptr(vecm[0, 1]),
                                               ptr(pout), Cm, ptr(ws), ws.numel() * 4))
run("grouped Gram a^T a", m, lambda: call("adamml_conv_bwd_weight_grouped", byref(dg), ptr(xm), ptr(vecm[0, 0]), ptr(vecm[0, 1]), 1, 4 * Cm, ptr(xm),
                                           ptr(vecm[0, 0]), ptr(vecm[0, 1]), ptr(gout), Cm, ptr(wsg), wsg.numel() * 4))
if hip.load().adamml_gram_colsum_supported(Cm):
    wgc = hip.scratch(hip.load().adamml_gram_colsum_workspace(P, Cm, G), DEV)
    scol2 = torch.empty(G, Cm, device=DEV)
    run("gram_colsum kernel (G and s)", m, lambda: call("adamml_gram_colsum", ptr(xm), ptr(vecm[0, 0]), ptr(vecm[0, 1]), 4 * Cm, 1, ptr(gout), ptr(scol2), P, Cm, G,
                                                         ptr(wgc), wgc.numel() * 4))
scol = torch.empty(G, Cm, device=DEV)
run("lazy_colsum a", m, lambda: call("adamml_lazy_colsum", ptr(xm), ptr(vecm[0, 0]), ptr(vecm[0, 1]), 4 * Cm, 1, ptr(scol), P, Cm, G))
run("bn_bwd_apply mid", 3 * m, lambda: call("adamml_bn_bwd_apply", ptr(xm), ptr(xm), ptr(vecm), 1, ptr(coef[:, :, :Cm].contiguous()), ptr(om), P, Cm, G))
run("conv1 wgrad dz1^T X", X + m, lambda: call("adamml_conv_bwd_weight", byref(d1), ptr(xm), ptr(xb), None, None, ptr(torch.zeros_like(w1)), Cb,
                                                ptr(hip.wgrad_workspace(d1, Cb, DEV)), hip.wgrad_workspace(d1, Cb, DEV).numel() * 4))
