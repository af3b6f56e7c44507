import sys, torch
sys.path.insert(0, ".")
from ctypes import byref
from adamml_amd import hip
from adamml_amd.hip import call, ptr, ConvDesc
torch.manual_seed(0)
out = {}
for (G, N, H, Cin, Cout, k, s, p) in [(2, 3, 14, 256, 256, 3, 1, 1), (1, 2, 28, 512, 128, 1, 1, 0), (3, 2, 28, 128, 128, 3, 2, 1), (2, 5, 7, 512, 512, 3, 1, 1),
                                      (1, 3, 13, 256, 384, 1, 1, 0), (2, 2, 9, 128, 256, 3, 1, 1)]:
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(G * N, H, H, Cin, device="cuda").to(torch.bfloat16)
    dz = torch.randn(G * N, OH, OH, Cout, device="cuda").to(torch.bfloat16)
    d = ConvDesc(N, H, H, Cin, OH, OH, Cout, k, k, s, p, 1, 0, 0, G, 0)
    dw = torch.zeros(Cout, Cin, k, k, device="cuda")
    ws = hip.wgrad_workspace(d, Cin, "cuda")
    call("adamml_conv_bwd_weight", byref(d), ptr(dz), ptr(x), None, None, ptr(dw), Cin, ptr(ws), ws.numel() * 4)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dz.float().permute(0, 3, 1, 2), stride=s, padding=p)
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    print((G, N, H, Cin, Cout, k, s, p), "rel err vs torch %.2e" % err, "sum %.6f" % dw.double().sum().item())
    assert err < 2e-3
