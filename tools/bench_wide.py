"""Wide 1x1 convs of ResNet-50 layer 3 at the benchmark shape (5 groups x 144 frames x 14^2, 256 -> 1024): the forward with a lazy / plain
input, with / without statistics, and the plain / accumulating data gradient, by conv_gemm_kernel (ADAMML_WIDE_STREAM=0) and by the activation-stationary kernel of csrc/conv1x1_wide.hip (default) -- each in its own process."""
import os
import subprocess
import sys

CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
from ctypes import byref
from adamml_amd import hip
from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
G, N, H, K, C = 5, 144, 14, 256, 1024
dev = "cuda"
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(G * N, H, H, K, device=dev).to(torch.bfloat16)
y = torch.empty(G * N, H, H, C, dtype=torch.bfloat16, device=dev)
w = torch.randn(C, K, 1, 1, device=dev) * 0.05
wf = torch.empty(C, K, dtype=torch.bfloat16, device=dev)
call("adamml_pack_conv_weight", ptr(w), ptr(wf), C, K, K, 1, 1, 0)
vec = torch.rand(G, 4, K, device=dev) + 0.5
st = torch.zeros(G, STAT_SLOTS, 2 * C, dtype=torch.float64, device=dev)
out = []
for lazy in (1, 0):
    for stats in (1, 0):
        d = ConvDesc(N, H, H, K, H, H, C, 1, 1, 1, 0, 1, 1 if lazy else 0, 0, G, 4 * K if lazy else 0)
        sc, sh = (ptr(vec[0, 0]), ptr(vec[0, 1])) if lazy else (None, None)
        t = timeit(lambda: call("adamml_conv_fwd", byref(d), ptr(x), ptr(wf), sc, sh, ptr(y), ptr(st) if stats else None))
        out.append("fwd %s %s %.0f us" % ("lazy" if lazy else "plain", "stats" if stats else "nostats", t))
# data gradient of the conv 1024 -> 256: dz [.., 256] -> dx [.., 1024]
wr = torch.randn(K, C, 1, 1, device=dev) * 0.05
wd = torch.empty(C, K, dtype=torch.bfloat16, device=dev)
call("adamml_pack_conv_weight", ptr(wr), ptr(wd), K, C, C, 1, 1, 1)
dr = ConvDesc(N, H, H, C, H, H, K, 1, 1, 1, 0, 1, 0, 0, G, 0)
for acc in (0, 1):
    t = timeit(lambda: call("adamml_conv_bwd_data", byref(dr), ptr(x), ptr(wd), ptr(y), acc))
    out.append("dgrad %s %.0f us" % ("acc" if acc else "plain", t))
print("RESULT " + " | ".join(out))
'''
for name, env in (("conv_gemm", {"ADAMML_WIDE_STREAM": "0"}), ("wide", {})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    print("%-10s %s" % (name, line[0][7:] if line else "FAILED " + r.stderr[-300:]), flush=True)
