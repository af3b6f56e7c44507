"""Micro-benchmark of the elementwise / BatchNorm passes against a plain device copy (achievable HBM bandwidth)."""
import sys, torch
sys.path.insert(0, ".")
from adamml_amd.hip import call, ptr, STAT_SLOTS
DEV = "cuda"

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for (P, C, G) in [(576 * 56 * 56, 256, 5), (288 * 28 * 28, 512, 5), (576 * 56 * 56, 64, 5), (144 * 14 * 14, 1024, 5), (72 * 7 * 7, 2048, 5)]:
    n = G * P * C
    g = torch.randn(n, device=DEV).to(torch.bfloat16)
    z = torch.randn(n, device=DEV).to(torch.bfloat16)
    o = torch.empty_like(g)
    vec = torch.rand(G, 4, C, device=DEV) + 0.5
    coef = torch.rand(G, 3, C, device=DEV)
    sums = torch.zeros(G, STAT_SLOTS, 2 * C, dtype=torch.float64, device=DEV)
    sums2 = torch.zeros_like(sums)
    gb = n * 2 / 1e9
    t = timeit(lambda: o.copy_(g)); print("P=%d C=%d  copy            %.3f ms  %.0f GB/s" % (P, C, t, 2 * gb / t * 1e3))
    t = timeit(lambda: torch.add(g, z, out=o)); print("   torch add (3 passes)      %.3f ms  %.0f GB/s" % (t, 3 * gb / t * 1e3))
    t = timeit(lambda: call("adamml_bn_bwd_apply", ptr(g), ptr(z), ptr(vec), 1, ptr(coef), ptr(o), P, C, G))
    print("   bn_bwd_apply (3 passes)   %.3f ms  %.0f GB/s" % (t, 3 * gb / t * 1e3))
    t = timeit(lambda: call("adamml_bn_act_add", ptr(g), ptr(vec[0, 0]), ptr(vec[0, 1]), 4 * C, 1, ptr(z), None, None, 0, ptr(o), P, C, G))
    print("   bn_act_add (3 passes)     %.3f ms  %.0f GB/s" % (t, 3 * gb / t * 1e3))
    t = timeit(lambda: call("adamml_bn_act_add", ptr(g), ptr(vec[0, 0]), ptr(vec[0, 1]), 4 * C, 1, None, None, None, 0, ptr(o), P, C, G))
    print("   materialize (2 passes)    %.3f ms  %.0f GB/s" % (t, 2 * gb / t * 1e3))
    z2 = torch.randn(n, device=DEV).to(torch.bfloat16)        # distinct tensors: block output and raw conv3 output
    t = timeit(lambda: call("adamml_residual_bwd", ptr(g), ptr(z2), 1, ptr(o), ptr(z), ptr(vec), ptr(sums), None, None, None, P, C, G))
    print("   residual_bwd 1 op (4 p)   %.3f ms  %.0f GB/s" % (t, 4 * gb / t * 1e3))
    t = timeit(lambda: call("adamml_bn_bwd_reduce", ptr(g), ptr(z), ptr(vec), 1, ptr(sums), P, C, G))
    print("   bn_bwd_reduce (2 passes)  %.3f ms  %.0f GB/s" % (t, 2 * gb / t * 1e3))
