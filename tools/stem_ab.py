"""A/B timing of conv_stem build variants in ONE process (see tools/c64_ab.py): builds csrc/conv_stem.hip alone once per flag set, then
times the ResNet stem of the benchmark (5 groups x 576 frames of 224x224x4 -> 112x112x64), forward with statistics and weight gradient,
variants interleaved.  GPU box only.   usage: python tools/stem_ab.py "" "-DSTEM_STAGGER=60" ..."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402  (ConvDesc only)

cs = os.path.join(ROOT, "adamml_amd", "csrc")
out = "/tmp/stem_ab"
os.makedirs(out, exist_ok=True)
stub = os.path.join(out, "stub.hip")
open(stub, "w").write("""#include <hip/hip_runtime.h>
#include <stdio.h>
int adamml_set_error(int code, const char* fmt, ...) { fprintf(stderr, "stem probe: error %d: %s\\n", code, fmt); return code; }
int adamml_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\\n", what, hipGetErrorString(e)); return -3; } return 0; }
""")
P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
variants = sys.argv[1:] or [""]
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-shared", "-o",
                           os.path.join(out, "lib%d.so" % i), os.path.join(cs, "conv_stem.hip"), stub] + f.split()) for i, f in enumerate(variants)]
libs = []
for i, pr in enumerate(procs):
    assert pr.wait() == 0
    lib = ctypes.CDLL(os.path.join(out, "lib%d.so" % i))
    lib.adamml_conv_stem_fwd.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P]
    lib.adamml_conv_stem_bwd_weight.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, I, P, Z, P]
    lib.adamml_conv_stem_bwd_weight_workspace.argtypes = [ctypes.POINTER(hip.ConvDesc)]
    lib.adamml_conv_stem_bwd_weight_workspace.restype = Z
    libs.append(lib)

dev = torch.device("cuda:0")
G, N, H, W = 5, 576, 224, 224
d = hip.ConvDesc(N, H, W, 4, 112, 112, 64, 7, 7, 2, 3, 1, 0, 0, G, 0)
x = torch.randn(G * N, H, W, 4, device=dev).bfloat16()
w = (torch.randn(64, 7 * 8 * 4, device=dev) * 0.05).bfloat16()
y = torch.empty(G * N, 112, 112, 64, dtype=torch.bfloat16, device=dev)
dz = torch.randn(G * N, 112, 112, 64, device=dev).bfloat16()
stats = torch.zeros(G * 64 * 128, dtype=torch.float64, device=dev)
dw = torch.zeros(64, 3, 7, 7, device=dev)
need = libs[0].adamml_conv_stem_bwd_weight_workspace(ctypes.byref(d))
ws = torch.empty(need // 4 + 16, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
forms = {
    "fwd+stats": lambda lib: lib.adamml_conv_stem_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), st),
    "wgrad": lambda lib: lib.adamml_conv_stem_bwd_weight(ctypes.byref(d), dz.data_ptr(), x.data_ptr(), dw.data_ptr(), 3, ws.data_ptr(), ws.numel() * 4, st),
}
ROUNDS, REP = 5, 10
res = {(f, i): [] for f in forms for i in range(len(libs))}
for lib in libs:
    for _ in range(30):
        assert forms["fwd+stats"](lib) == 0
torch.cuda.synchronize()
for r in range(ROUNDS):
    for f, fn in forms.items():
        for i, lib in enumerate(libs):
            for _ in range(2):
                fn(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                fn(lib)
            e1.record()
            torch.cuda.synchronize()
            res[(f, i)].append(e0.elapsed_time(e1) / REP)
print("%-40s %s" % ("variant", "   ".join("%-22s" % f for f in forms)))
for i, flags in enumerate(variants):
    print("%-40s %s" % ("[%s]" % flags, "   ".join("%.3f (min %.3f)      " % (np.median(res[(f, i)]), min(res[(f, i)])) for f in forms)))
