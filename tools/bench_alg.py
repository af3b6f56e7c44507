"""Algebraic data gradient at the benchmark shapes (B=72 x 5 segments): layer 1 (Cout 256 -> Cin 64: streaming kernel vs the CAT instance of
conv_gemm_kernel, run with ADAMML_ALG_STREAM=0 / 1) or, `python tools/bench_alg.py 2`, layer 2 (Cout 512 -> Cin 128: CAT instance)."""
import os, sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
DEV, G = "cuda", 5
N, H, Cin, Cout = (288, 28, 128, 512) if len(sys.argv) > 1 and sys.argv[1] == "2" else (576, 56, 64, 256)
bf = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)
g, a, dx = bf(G * N, H, H, Cout), bf(G * N, H, H, Cin), bf(G * N, H, H, Cin)
vin = torch.rand(G, 4, Cin, device=DEV) + 0.5
w_alg = bf(G, Cin, Cout + Cin)
cadd = torch.rand(G, Cin, device=DEV)
sm = torch.zeros(G, STAT_SLOTS, 2 * Cin, dtype=torch.float64, device=DEV)
d = ConvDesc(N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1, 1, 0, G, 4 * Cin)
def run(mode):
    if mode == "bn":
        call("adamml_conv_bwd_data_alg", byref(d), ptr(g), ptr(a), ptr(vin[0, 0]), ptr(vin[0, 1]), ptr(w_alg), ptr(cadd), ptr(dx), 0, ptr(a), ptr(vin), 1, ptr(sm))
    else:
        call("adamml_conv_bwd_data_alg", byref(d), ptr(g), ptr(a), ptr(vin[0, 0]), ptr(vin[0, 1]), ptr(w_alg), ptr(cadd), ptr(dx), 1, None, None, 0, None)
for mode in ("bn", "acc"):
    run(mode); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(5):
        run(mode)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5
    X, m = g.numel() * 2 / 1e9, a.numel() * 2 / 1e9
    gb = X + 3 * m if mode == "bn" else X + 3 * m
    print("ADAMML_ALG_STREAM=%s %s: %.3f ms  %.0f GB/s (%.1f GB)" % (os.environ.get("ADAMML_ALG_STREAM", "1"), mode, t, gb / t * 1e3, gb))
