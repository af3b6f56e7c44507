"""Micro-benchmark of the dense conv kernels on the ResNet-50 shapes of the B=72 x 5-segment step (G=5 groups)."""
import sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd import hip
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV = "cuda"

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

G = 5
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
# (name, N per group, H, Cin, Cout, k, stride, pad, lazy-input, count per net)
L = [("stem 7x7", B * 8, 224, 3, 64, 7, 2, 3, 0, 1),
     ("l1 c1 first", B * 8, 56, 64, 64, 1, 1, 0, 0, 1), ("l1 c2", B * 8, 56, 64, 64, 3, 1, 1, 0, 3), ("l1 c3", B * 8, 56, 64, 256, 1, 1, 0, 1, 3),
     ("l1 ds", B * 8, 56, 64, 256, 1, 1, 0, 0, 1), ("l1 c1", B * 8, 56, 256, 64, 1, 1, 0, 0, 2),
     ("l2 c1 first", B * 4, 56, 256, 128, 1, 1, 0, 0, 1), ("l2 c2 s2", B * 4, 56, 128, 128, 3, 2, 1, 0, 1), ("l2 ds", B * 4, 56, 256, 512, 1, 2, 0, 0, 1),
     ("l2 c3", B * 4, 28, 128, 512, 1, 1, 0, 1, 4), ("l2 c1", B * 4, 28, 512, 128, 1, 1, 0, 0, 3), ("l2 c2", B * 4, 28, 128, 128, 3, 1, 1, 0, 3),
     ("l3 c1 first", B * 2, 28, 512, 256, 1, 1, 0, 0, 1), ("l3 c2 s2", B * 2, 28, 256, 256, 3, 2, 1, 0, 1), ("l3 ds", B * 2, 28, 512, 1024, 1, 2, 0, 0, 1),
     ("l3 c3", B * 2, 14, 256, 1024, 1, 1, 0, 1, 6), ("l3 c1", B * 2, 14, 1024, 256, 1, 1, 0, 0, 5), ("l3 c2", B * 2, 14, 256, 256, 3, 1, 1, 0, 5),
     ("l4 c1 first", B, 14, 1024, 512, 1, 1, 0, 0, 1), ("l4 c2 s2", B, 14, 512, 512, 3, 2, 1, 0, 1), ("l4 ds", B, 14, 1024, 2048, 1, 2, 0, 0, 1),
     ("l4 c3", B, 7, 512, 2048, 1, 1, 0, 1, 3), ("l4 c1", B, 7, 2048, 512, 1, 1, 0, 0, 2), ("l4 c2", B, 7, 512, 512, 3, 1, 1, 0, 2)]
only = sys.argv[2] if len(sys.argv) > 2 else None
tot = [0.0, 0.0, 0.0]
for name, N, H, Cin, Cout, k, s, p, lazy, cnt in L:
    if only and only not in name:
        continue
    cp = (Cin + 7) // 8 * 8
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(G * N, H, H, cp, device=DEV).to(torch.bfloat16)
    y = torch.empty(G * N, OH, OH, Cout, dtype=torch.bfloat16, device=DEV)
    dz = torch.randn(G * N, OH, OH, Cout, device=DEV).to(torch.bfloat16)
    dx = torch.empty_like(x)
    w = torch.randn(Cout, Cin, k, k, device=DEV) * 0.05
    wf = torch.empty(Cout, k * k * cp, dtype=torch.bfloat16, device=DEV)
    wd = torch.empty(cp, k * k * Cout, dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w), ptr(wf), Cout, Cin, cp, k, k, 0)
    call("adamml_pack_conv_weight", ptr(w), ptr(wd), Cout, Cin, cp, k, k, 1)
    vec = torch.rand(G, 4, cp, device=DEV) + 0.5
    st = torch.zeros(G, STAT_SLOTS, 2 * Cout, dtype=torch.float64, device=DEV)
    sm = torch.zeros(G, STAT_SLOTS, 2 * cp, dtype=torch.float64, device=DEV)
    d = ConvDesc(N, H, H, cp, OH, OH, Cout, k, k, s, p, 1, 1 if lazy else 0, 0, G, 4 * cp if lazy else 0)
    sc, sh = (ptr(vec[0, 0]), ptr(vec[0, 1])) if lazy else (None, None)
    dw = torch.zeros_like(w)
    ws = hip.wgrad_workspace(d, Cin, DEV)
    gb = (x.numel() * Cin / cp + y.numel()) * 2 / 1e9
    fl = 2.0 * y.numel() * Cin * k * k / 1e12
    if name.startswith("stem") and hip.load().adamml_conv_stem_supported(byref(d)):
        wst = torch.empty(Cout, 224, dtype=torch.bfloat16, device=DEV)
        call("adamml_pack_stem_weight", ptr(w), ptr(wst), Cout, Cin)
        t1 = timeit(lambda: call("adamml_conv_stem_fwd", byref(d), ptr(x), ptr(wst), ptr(y), ptr(st)))
    else:
        t1 = timeit(lambda: call("adamml_conv_fwd", byref(d), ptr(x), ptr(wf), sc, sh, ptr(y), ptr(st) if "nostats" not in sys.argv else None))
    if name.startswith("stem"):
        t2 = 0.0
    elif k == 3 or "c3" in name:   # sole-consumer data gradients carry the BatchNorm-backward reduction
        t2 = timeit(lambda: call("adamml_conv_bwd_data_bn", byref(d), ptr(dz), ptr(wd), ptr(dx), ptr(x), ptr(vec), 1, ptr(sm)))
    else:
        acc = 1 if (name.endswith(" c1") or name.endswith(" ds")) else 0      # as used in the net: accumulates into the identity-path gradient
        t2 = timeit(lambda: call("adamml_conv_bwd_data", byref(d), ptr(dz), ptr(wd), ptr(dx), acc))
    if name.startswith("stem") and hip.load().adamml_conv_stem_supported(byref(d)):
        wss = hip.wgrad_workspace(d, Cin, DEV, stem=True)
        t3 = timeit(lambda: call("adamml_conv_stem_bwd_weight", byref(d), ptr(dz), ptr(x), ptr(dw), Cin, ptr(wss), wss.numel() * 4))
    else:
        t3 = timeit(lambda: call("adamml_conv_bwd_weight", byref(d), ptr(dz), ptr(x), sc, sh, ptr(dw), Cin, ptr(ws), ws.numel() * 4))
    tot[0] += t1 * cnt; tot[1] += t2 * cnt; tot[2] += t3 * cnt
    print("%-12s x%d N=%4d H=%3d %4d->%4d k%d s%d  %.2f GB %.2f TF | fwd %.3f ms %5.0f GB/s %4.0f TF/s | dgrad %.3f ms %5.0f GB/s | wgrad %.3f ms %5.0f GB/s %4.0f TF/s"
          % (name, cnt, N, H, Cin, Cout, k, s, gb, fl, t1, gb / t1 * 1e3, fl / t1 * 1e3, t2, gb / max(t2, 1e-9) * 1e3, t3, gb / t3 * 1e3, fl / t3 * 1e3))
    del x, y, dz, dx
print("net sum (x count): fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % tuple(tot))
