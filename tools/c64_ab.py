"""A/B timing of conv3x3_c64 build variants in ONE process (boxes and clock states differ by up to 10 % from run to run): builds
csrc/conv3x3_c64.hip alone once per flag set given on the command line (one quoted argument each, "" = the shipped configuration), then
times the layer-1 conv2 shape of the benchmark in its four forms, the variants interleaved over several rounds; prints the median and
the minimum per variant.  GPU box only; nothing here is part of the product path.

usage: python tools/c64_ab.py "" "-DC64_STAGGER=94" ..."""
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
out = "/tmp/c64_ab"
os.makedirs(out, exist_ok=True)
stub = os.path.join(out, "stub.hip")
open(stub, "w").write("""#include <hip/hip_runtime.h>
#include <stdio.h>
int adamml_set_error(int code, const char* fmt, ...) { fprintf(stderr, "c64 probe: error %d: %s\\n", code, fmt); return code; }
int adamml_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\\n", what, hipGetErrorString(e)); return -3; } return 0; }
""")
P, I = ctypes.c_void_p, ctypes.c_int
variants = sys.argv[1:] or [""]
libs = []
procs = []
for i, flags in enumerate(variants):
    lp = os.path.join(out, "libc64_%d.so" % i)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-DC64_PHASE_TIMING",
                                   "-shared", "-o", lp, os.path.join(cs, "conv3x3_c64.hip"), stub] + flags.split()))
for i, pr in enumerate(procs):
    assert pr.wait() == 0
    lib = ctypes.CDLL(os.path.join(out, "libc64_%d.so" % i))
    lib.c64_probe_launch.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P, P, P, P, I, P]
    lib.c64_probe_wgrad_launch.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P, P]
    lib.c64_probe_wgrad_blocks.argtypes = [ctypes.POINTER(hip.ConvDesc), ctypes.POINTER(I)]
    libs.append(lib)

dev = torch.device("cuda:0")
G, N, H, W, C = 5, 576, 56, 56, 64
d = hip.ConvDesc(N, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1, 0, G, 4 * C)
d0 = hip.ConvDesc(N, H, W, C, H, W, C, 3, 3, 1, 1, 1, 0, 0, G, 0)
x = torch.randn(G * N, H, W, C, device=dev).bfloat16()
w = (torch.randn(C, 9, C, device=dev) * 0.05).bfloat16()
vec = torch.randn(G, 4, C, device=dev).abs().float() + 0.5
y = torch.empty_like(x)
z = torch.randn_like(x)
stats = torch.zeros(G * 64 * 128, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
tpb = I(0)
nblk = libs[0].c64_probe_wgrad_blocks(ctypes.byref(d), ctypes.byref(tpb))
ws = torch.empty(nblk * C * 9 * C, dtype=torch.float32, device=dev)
sc, sh = vec.data_ptr(), vec.data_ptr() + 4 * C

forms = {
    "fwd+stats": lambda lib: lib.c64_probe_launch(ctypes.byref(d), x.data_ptr(), w.data_ptr(), sc, sh, y.data_ptr(), stats.data_ptr(), None, None, 0, st),
    "fwd raw": lambda lib: lib.c64_probe_launch(ctypes.byref(d0), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(), None, None, 0, st),
    "dgrad+bn": lambda lib: lib.c64_probe_launch(ctypes.byref(d0), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(), z.data_ptr(), vec.data_ptr(), 1, st),
    "wgrad": lambda lib: lib.c64_probe_wgrad_launch(ctypes.byref(d), z.data_ptr(), x.data_ptr(), sc, sh, ws.data_ptr(), st),
}
ROUNDS, REP = 5, 20
res = {(f, i): [] for f in forms for i in range(len(libs))}
for lib in libs:                                      # clock ramp
    for _ in range(50):
        forms["fwd+stats"](lib)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for f, fn in forms.items():
        for i, lib in enumerate(libs):
            for _ in range(3):
                fn(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                fn(lib)
            e1.record()
            torch.cuda.synchronize()
            res[(f, i)].append(e0.elapsed_time(e1) / REP)
flop = 2.0 * G * N * H * W * C * C * 9
print("%-44s %s" % ("variant", "   ".join("%-22s" % f for f in forms)))
for i, flags in enumerate(variants):
    cells = []
    for f in forms:
        v = np.array(res[(f, i)])
        cells.append("%.3f (min %.3f) %4.0fTF" % (np.median(v), v.min(), flop / np.median(v) * 1e-9))
    print("%-44s %s" % ("[%s]" % flags, "   ".join(cells)))
