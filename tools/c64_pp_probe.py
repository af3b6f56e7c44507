"""Where the two-phase conv3x3_c64 kernel (conv3x3_c64_pp_kernel) spends its time: builds csrc/conv3x3_c64.hip ALONE with
-DC64_PHASE_TIMING; every wave of one workgroup sums the shader-clock cycles of its role segments over all phases -- compute role:
MFMA loop, wait at the phase barrier; stage role: patch -> LDS, epilogue from the accumulators, requests of the strip after next, wait
at the barrier -- and the tool prints cycles per phase and wave for the layer-1 conv2 shape of the benchmark (5 groups x 576 frames of
56x56x64), forward with statistics and BatchNorm-fused data gradient.  GPU box only; nothing here is part of the product path.

usage: python tools/c64_pp_probe.py [extra hipcc -D flags ...]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402  (ConvDesc only)

out = os.environ.get("C64_PROBE_DIR", "/tmp/c64_pp_probe")
os.makedirs(out, exist_ok=True)
lib_path = os.path.join(out, "libc64.so")
cs = os.path.join(ROOT, "adamml_amd", "csrc")
stub = os.path.join(out, "stub.hip")
open(stub, "w").write("""#include <hip/hip_runtime.h>
#include <stdio.h>
int adamml_set_error(int code, const char* fmt, ...) { fprintf(stderr, "c64 probe: error %d: %s\\n", code, fmt); return code; }
int adamml_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\\n", what, hipGetErrorString(e)); return -3; } return 0; }
""")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-DC64_PHASE_TIMING", "-shared",
                       "-o", lib_path, os.path.join(cs, "conv3x3_c64.hip"), stub] + sys.argv[1:])
lib = ctypes.CDLL(lib_path)
P, I = ctypes.c_void_p, ctypes.c_int
lib.c64_probe_launch.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P, P, P, P, I, P]
lib.adamml_c64_set_phase_buffer.argtypes = [P]
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
dbg = torch.zeros(64, dtype=torch.int32, device=dev)
assert lib.adamml_c64_set_phase_buffer(dbg.data_ptr()) == 0
st = torch.cuda.current_stream().cuda_stream
sc, sh = vec.data_ptr(), vec.data_ptr() + 4 * C
NAMES = ["compute: MFMA loop", "compute: barrier wait", "stage: patch -> LDS", "stage: epilogue", "stage: requests", "stage: barrier wait"]


def run(label, fn):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    t = dbg.cpu().numpy().reshape(8, 8).astype(np.int64) & 0xffffffff
    n = int(t[0, 6])
    print("== %s: %.3f ms per launch (%.0f TFLOP/s); workgroup of %d strips, cycles per PHASE PAIR (one strip per group) and wave:" % (label, ms, 2.0 * G * N * H * W * C * C * 9 / ms * 1e-9, n))
    per = t[:, :6] / max(1.0, (n + 1) / 2.0)
    for k, nm in enumerate(NAMES):
        print("   %-24s %s" % (nm, " ".join("%6d" % v for v in per[:, k])))
    print("   %-24s %s" % ("sum", " ".join("%6d" % v for v in per.sum(1))))


run("forward + statistics", lambda: lib.c64_probe_launch(ctypes.byref(d), x.data_ptr(), w.data_ptr(), sc, sh, y.data_ptr(), stats.data_ptr(), None, None, 0, st))
run("forward raw", lambda: lib.c64_probe_launch(ctypes.byref(d0), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(), None, None, 0, st))
run("data gradient + bn sums", lambda: lib.c64_probe_launch(ctypes.byref(d0), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), stats.data_ptr(), z.data_ptr(), vec.data_ptr(), 1, st))
