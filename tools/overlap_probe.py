"""Do an HBM-bound streaming conv and an MFMA-bound weight gradient overlap when issued on two HIP streams?  Times each alone and both together
(B = 72 shapes of layer 1: conv1 256 -> 64 forward, conv2 3x3 64 -> 64 weight gradient; and the bn_bwd_apply elementwise pass as a low-footprint
partner).  If the pair takes the SUM of its parts, the two kernels do not co-reside on a CU (LDS / registers of the first fill it)."""
import sys, time, torch
sys.path.insert(0, ".")
from ctypes import byref
from adamml_amd import hip
from adamml_amd.hip import ConvDesc, call, ptr, STAT_SLOTS
DEV = "cuda"
G, N, H = 5, 576, 56


def pack(w, cp, mode):
    cout, cin, k, _ = w.shape
    out = torch.empty((cout, k * k * cp) if mode == 0 else (cp, k * k * cout), dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w), ptr(out), cout, cin, cp, k, k, mode)
    return out


x256 = torch.randn(G * N, H, H, 256, device=DEV).to(torch.bfloat16)
y64 = torch.empty(G * N, H, H, 64, dtype=torch.bfloat16, device=DEV)
w1 = torch.randn(64, 256, 1, 1, device=DEV) * 0.05
w1p = pack(w1, 256, 0)
st1 = torch.zeros(G, STAT_SLOTS, 128, dtype=torch.float64, device=DEV)
d1 = ConvDesc(N, H, H, 256, H, H, 64, 1, 1, 1, 0, 1, 0, 0, G, 0)
x64 = torch.randn(G * N, H, H, 64, device=DEV).to(torch.bfloat16)
dz64 = torch.randn(G * N, H, H, 64, device=DEV).to(torch.bfloat16)
w2 = torch.randn(64, 64, 3, 3, device=DEV) * 0.05
dw2 = torch.zeros_like(w2)
d2 = ConvDesc(N, H, H, 64, H, H, 64, 3, 3, 1, 1, 1, 0, 0, G, 0)
vec = torch.rand(G, 4, 64, device=DEV) + 0.5
coef = torch.rand(G, 3, 64, device=DEV)
dzo = torch.empty_like(x64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s2):
    ws = hip.wgrad_workspace(d2, 64, DEV)


def conv1():
    call("adamml_conv_fwd", byref(d1), ptr(x256), ptr(w1p), None, None, ptr(y64), ptr(st1))


def wgrad2():
    call("adamml_conv_bwd_weight", byref(d2), ptr(dz64), ptr(x64), None, None, ptr(dw2), 64, ptr(ws), ws.numel() * 4)


def apply_():
    call("adamml_bn_bwd_apply", ptr(dz64), ptr(x64), ptr(vec), 0, ptr(coef), ptr(dzo), N * H * H, 64, G)


def timed(fa, fb, n=20):
    torch.cuda.synchronize()
    for _ in range(3):
        if fa:
            with torch.cuda.stream(s1):
                fa()
        if fb:
            with torch.cuda.stream(s2):
                fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if fa:
            with torch.cuda.stream(s1):
                fa()
        if fb:
            with torch.cuda.stream(s2):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b, c = timed(conv1, None), timed(None, wgrad2), timed(apply_, None)
print("alone: conv1 forward (HBM-bound) %.3f ms, conv2 3x3 weight gradient (MFMA-bound) %.3f ms, bn_bwd_apply (elementwise) %.3f ms" % (a, b, c))
print("two streams: conv1 + wgrad %.3f ms (sum %.3f, max %.3f)" % (timed(conv1, wgrad2), a + b, max(a, b)))
print("two streams: bn_bwd_apply + wgrad %.3f ms (sum %.3f, max %.3f)" % (timed(apply_, wgrad2), c + b, max(c, b)))
print("two streams: conv1 + conv1 %.3f ms (sum %.3f)" % (timed(conv1, conv1), 2 * a))
