"""Where a slab step of wide_all_kernel (csrc/conv1x1_wide.hip) spends its time: builds that file ALONE with -DWIDE_PROBE (every wave of
workgroup 0 sums the shader-clock ticks of each phase of its slab steps) into a second library, runs the layer-3 conv3 shape of the
benchmark (5 groups x 144 frames x 14^2, 256 -> 1024) and prints the per-wave phase table.  GPU box; build container compiles."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "adamml_amd", "libadamml_hip_probe.so")


def build():
    import glob
    import __graft_entry__ as ge
    objdir = os.path.join(ROOT, "build", "obj")
    o = os.path.join(objdir, "conv1x1_wide_probe.o")
    subprocess.check_call([ge.HIPCC] + ge.FLAGS + ["-DWIDE_PROBE", "-c", os.path.join(ge.CSRC, "conv1x1_wide.hip"), "-o", o])
    objs = [x for x in sorted(glob.glob(os.path.join(objdir, "*.o"))) if not x.endswith(("conv1x1_wide.o", "conv1x1_wide_probe.o"))] + [o]
    subprocess.check_call([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(ge.CSRC, "exports.map"), "-o", LIB] + objs)


def run():
    os.environ["ADAMML_HIP_LIB"] = LIB
    import torch
    from ctypes import byref
    from adamml_amd.hip import call, ptr, STAT_SLOTS, ConvDesc
    G, N, H, K, C = 5, 144, 14, 256, 1024
    x = torch.randn(G * N, H, H, K, device="cuda").to(torch.bfloat16)
    y = torch.empty(G * N, H, H, C, dtype=torch.bfloat16, device="cuda")
    w = torch.randn(C, K, 1, 1, device="cuda") * 0.05
    wf = torch.empty(C, K, dtype=torch.bfloat16, device="cuda")
    call("adamml_pack_conv_weight", ptr(w), ptr(wf), C, K, K, 1, 1, 0)
    vec = torch.rand(G, 4, K, device="cuda") + 0.5
    names = ["between steps", "barrier 1 wait", "MFMA phase", "barrier 2 wait", "stage + statistics", "read-back + stores", "weight write + request"]
    for lazy in (1, 0):
        st = torch.zeros(G, STAT_SLOTS, 2 * C, dtype=torch.float64, device="cuda")
        d = ConvDesc(N, H, H, K, H, H, C, 1, 1, 1, 0, 1, 1 if lazy else 0, 0, G, 4 * K if lazy else 0)
        sc, sh = (ptr(vec[0, 0]), ptr(vec[0, 1])) if lazy else (None, None)
        for _ in range(2):
            st.zero_()
            call("adamml_conv_fwd", byref(d), ptr(x), ptr(wf), sc, sh, ptr(y), ptr(st))
        torch.cuda.synchronize()
        t = st.flatten()[:64].view(8, 8).cpu()
        print("forward, %s input: ticks per slab step and wave (workgroup 0; %d steps)" % ("lazy" if lazy else "plain", int(t[0, 7])))
        for i, n in enumerate(names):
            print("  %-24s " % n + " ".join("%7.0f" % (t[w, i] / max(t[w, 7], 1)) for w in range(8)))
        print("  %-24s " % "sum" + " ".join("%7.0f" % (t[w, :7].sum() / max(t[w, 7], 1)) for w in range(8)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run()
