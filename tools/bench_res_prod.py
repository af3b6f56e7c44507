"""Time adamml_conv_bwd_data_res_prod at the layer-1 geometry of the benchmark (2880 frames of 56 x 56: the data gradient of a
bottleneck's conv1 onto the identity-path gradient + P = g'^T a), streaming kernel (csrc/res_prod_stream.hip) against the tile kernel of
csrc/conv_gemm.hip (ADAMML_RES_PROD_STREAM=0, read at every call).  GB/s over the algorithmic bytes of the launch."""
import os
import sys
import time
from ctypes import byref

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from adamml_amd import hip  # noqa: E402
from adamml_amd.hip import STAT_SLOTS, ConvDesc, call, ptr  # noqa: E402

DEV = torch.device("cuda:0")


def run(G, N, H, stream, iters=8):
    os.environ["ADAMML_RES_PROD_STREAM"] = str(stream)
    Cb, Cm, Ca = 256, 64, 64
    P = N * H * H
    d = ConvDesc(N, H, H, Cb, H, H, Cm, 1, 1, 1, 0, 1, 0, 0, G, 0)
    assert hip.load().adamml_conv_bwd_data_res_prod_supported(byref(d), Ca)
    torch.manual_seed(3)
    dz = (torch.randn(G * N, H, H, Cm, device=DEV) * 0.5).to(torch.bfloat16)
    w = torch.randn(Cm, Cb, 1, 1, device=DEV) * (2.0 / Cb) ** 0.5
    wd = torch.empty(Cb * Cm, dtype=torch.bfloat16, device=DEV)
    call("adamml_pack_conv_weight", ptr(w), ptr(wd), Cm, Cb, Cb, 1, 1, 1)
    dx = torch.randn(G * N, H, H, Cb, device=DEV).to(torch.bfloat16)
    mask = torch.randint(0, 256, (G * P * Cb // 8,), dtype=torch.uint8, device=DEV)
    a = (torch.randn(G * N, H, H, Ca, device=DEV) * 1.5).to(torch.bfloat16)
    avec = torch.rand(G, 4, Ca, device=DEV) + 0.5
    s = torch.zeros(G, STAT_SLOTS, 2 * Cb, dtype=torch.float64, device=DEV)
    Pf = torch.empty(G, Cb, Ca, device=DEV)
    need = hip.load().adamml_conv_bwd_data_res_prod_workspace(byref(d))
    wsp = torch.empty(need // 4 + 1, device=DEV)

    def once():
        call("adamml_conv_bwd_data_res_prod", byref(d), ptr(dz), ptr(wd), ptr(dx), ptr(mask), 1, ptr(s), ptr(a), ptr(avec[0, 0]), ptr(avec[0, 1]), 1,
             4 * Ca, Ca, ptr(Pf), ptr(wsp), wsp.numel() * 4)

    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = G * P * (2 * Cb * 2 + Cm * 2 + Ca * 2 + Cb // 8)
    print(f"G={G} N={N} H={H} stream={stream}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s over {nbytes / 1e9:.2f} GB", flush=True)
    return ms


if __name__ == "__main__":
    shapes = [(2, 1440, 56)]
    for G, N, H in shapes:
        for stream in (1, 0, 1, 0):
            run(G, N, H, stream)
