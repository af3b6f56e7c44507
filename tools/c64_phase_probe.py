"""Where a tile of conv3x3_c64 spends its time: builds csrc/conv3x3_c64.hip ALONE with -DC64_PHASE_TIMING (every wave of one workgroup
stamps the shader clock at the phase boundaries of its first 16 tiles), runs the layer-1 conv2 shape of the benchmark (5 groups x 576
frames of 56x56x64) in its three forms -- forward with statistics, BatchNorm-fused data gradient, weight gradient -- and prints the
cycles per phase (median over tiles 2..15, per wave min / max).  GPU box only; nothing here is part of the product path.

usage: python tools/c64_phase_probe.py [extra hipcc -D flags ...]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import hip  # noqa: E402  (ConvDesc only)

out = os.environ.get("C64_PROBE_DIR", "/tmp/c64_probe")
os.makedirs(out, exist_ok=True)
lib_path = os.path.join(out, "libc64.so")
cs = os.path.join(ROOT, "adamml_amd", "csrc")
stub = os.path.join(out, "stub.hip")
open(stub, "w").write("""#include <hip/hip_runtime.h>
#include <stdio.h>
int adamml_set_error(int code, const char* fmt, ...) { fprintf(stderr, "c64 probe: error %d: %s\\n", code, fmt); return code; }
int adamml_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\\n", what, hipGetErrorString(e)); return -3; } return 0; }
""")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-DC64_PHASE_TIMING", "-shared",
       "-o", lib_path, os.path.join(cs, "conv3x3_c64.hip"), stub] + sys.argv[1:]
subprocess.check_call(cmd)
lib = ctypes.CDLL(lib_path)
P, I = ctypes.c_void_p, ctypes.c_int
lib.c64_probe_launch.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P, P, P, P, I, P]
lib.c64_probe_wgrad_launch.argtypes = [ctypes.POINTER(hip.ConvDesc), P, P, P, P, P, P]
lib.c64_probe_wgrad_blocks.argtypes = [ctypes.POINTER(hip.ConvDesc), ctypes.POINTER(I)]
lib.adamml_c64_set_phase_buffer.argtypes = [P]

dev = torch.device("cuda:0")
G, N, H, W, C = 5, 576, 56, 56, 64
d = hip.ConvDesc(N, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1, 0, G, 4 * C)
x = torch.randn(G * N, H, W, C, device=dev).bfloat16()
w = (torch.randn(C, 9, C, device=dev) * 0.05).bfloat16()
vec = torch.randn(G, 4, C, device=dev).abs().float() + 0.5


class _Ptr:
    def __init__(self, v):
        self.v = v

    def data_ptr(self):
        return self.v


scale, shift = _Ptr(vec.data_ptr()), _Ptr(vec.data_ptr() + 4 * C)      # [G][4][C]: rows 0 / 1 of each group, group stride 4C
y = torch.empty_like(x)
z = torch.randn_like(x)
stats = torch.zeros(G * 64 * 128, dtype=torch.float64, device=dev)
dbg = torch.zeros(16 * 8 * 16, dtype=torch.int32, device=dev)
assert lib.adamml_c64_set_phase_buffer(dbg.data_ptr()) == 0
st = torch.cuda.current_stream().cuda_stream
tpb = I(0)
nblk = lib.c64_probe_wgrad_blocks(ctypes.byref(d), ctypes.byref(tpb))
ws = torch.empty(nblk * C * 9 * C, dtype=torch.float32, device=dev)

NAMES_F = ["stage patch -> LDS", "wait barrier A", "issue next loads + indices", "MFMA loop", "wait barrier B", "acc -> LDS staging",
           "wait barrier C", "store / epilogue", "wait barrier D", "loop end (+ stamp flush)"]
NAMES_W = ["stage patch+dz -> LDS", "wait barrier A", "issue next loads", "MFMA loop", "wait barrier B", "loop end (+ stamp flush)"]


def run(label, fn, names, nst):
    for _ in range(100):           # (clock ramp: the first launches of a fresh process run ~30 % slower)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(16, 8, 16).astype(np.int64) & 0xffffffff
    print("== %s: %.3f ms per launch (%.0f TFLOP/s), stamped launch %.3f ms" % (label, ms, 2.0 * G * N * H * W * C * C * 9 / ms * 1e-9, e0.elapsed_time(e1)))
    tiles = range(2, 15)
    if os.environ.get("C64_PROBE_RAW"):
        base = t[5, :, 0].min()
        for wv in range(8):
            print("   raw tile 5 wave %d: %s | next tile stamp 0: %d" % (wv, " ".join("%6d" % (v - base) for v in t[5, wv, :nst]), t[6, wv, 0] - base))
    # phase k = stamp k -> stamp k+1; the last phase of a tile ends at stamp 0 of the next tile
    tot = np.median([t[i + 1, :, 0] - t[i, :, 0] for i in tiles], axis=0)
    print("   tile period (cycles, per wave): min %d max %d" % (tot.min(), tot.max()))
    for k in range(nst):
        if k + 1 < nst:
            seg = np.array([t[i, :, k + 1] - t[i, :, k] for i in tiles])
        else:
            seg = np.array([t[i + 1, :, 0] - t[i, :, k] for i in tiles])      # (includes the stamp flush of the probe)
        med = np.median(seg, axis=0)
        print("   %-24s median over waves %7d   (wave min %7d, max %7d)   %5.1f %%" % (names[k], np.median(med), med.min(), med.max(),
                                                                                        100.0 * np.median(med) / np.median(tot)))


run("forward + statistics", lambda: lib.c64_probe_launch(ctypes.byref(d), x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                                     y.data_ptr(), stats.data_ptr(), None, None, 0, st), NAMES_F, 10)
run("forward, input not lazily normalised", lambda: lib.c64_probe_launch(ctypes.byref(d), x.data_ptr(), w.data_ptr(), None, None,
                                                                                     y.data_ptr(), stats.data_ptr(), None, None, 0, st), NAMES_F, 10)
d0 = hip.ConvDesc(N, H, W, C, H, W, C, 3, 3, 1, 1, 1, 0, 0, G, 0)
run("data gradient + BatchNorm-backward sums", lambda: lib.c64_probe_launch(ctypes.byref(d0), x.data_ptr(), w.data_ptr(), None, None,
                                                                                        y.data_ptr(), stats.data_ptr(), z.data_ptr(), vec.data_ptr(), 1, st), NAMES_F, 10)
run("weight gradient", lambda: lib.c64_probe_wgrad_launch(ctypes.byref(d), z.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                                     ws.data_ptr(), st), NAMES_W, 6)
