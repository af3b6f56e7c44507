"""Per-launch table of one backbone's forward + backward at the benchmark shape (single stream, HIP events around every C-ABI launch):
time, algorithmic GB/s / TFLOP/s where the caller supplies them, and the EXCESS over a 5.3 TB/s / 800 TFLOP/s floor -- where the time
above the roofs sits.  Usage: python tools/launch_table.py [resnet|sound] [B] [top]"""
import sys, torch
sys.path.insert(0, ".")
from adamml_amd import adamml, synth, hip
from adamml_amd.adamml import _frames
which = sys.argv[1] if len(sys.argv) > 1 else "resnet"
B, S = int(sys.argv[2]) if len(sys.argv) > 2 else 72, 5
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 45
dev = torch.device("cuda")
m = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
           causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.5, pooling_method="max",
           fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=1234)); m.to(dev); m.train()
m._flat_policy.ensure(dev); m._flat_main.ensure(dev); m._flat_main.ensure_grads(); m._flat_policy.ensure_grads()
rgb = torch.randn(B, S * 24, 224, 224, device=dev)
snd = torch.randn(B, S, 256, 256, device=dev)
p_x, m_x, _ = m.data_layer([rgb, snd], S)
del rgb
res, sound = m.main_net.nets
import os
os.environ["ADAMML_WGRAD_STREAM"] = "0"
if which in ("policy_rgb", "policy_sound"):                # frozen policy backbones of the main-net stage: forward only, train-mode BatchNorm
    m.freeze_policy_net()
    pr, ps = m.policy_net.joint_net.nets
    net, x = (pr, _frames(p_x[0])) if which == "policy_rgb" else (ps, _frames(p_x[1]))
    net.train()

    def step():
        with torch.no_grad():
            net.call(x, S)
else:
    net, x = (res, _frames(m_x[0])) if which == "resnet" else (sound, _frames(m_x[1]))

    def step():
        out = net.forward_nhwc(x, S)
        out.sum().backward()


step(); step(); torch.cuda.synchronize()
# shape of every launch that takes a conv descriptor (first argument), recorded beside the profiler's records
from adamml_amd import runtime as _rt
shapes = []
_orig_call = hip.call


def _call(name, *args):
    d = getattr(args[0], "_obj", None) if args else None
    shapes.append("%dx%dx%d %d->%d k%d s%d g%d" % (d.N, d.H, d.W, d.Cin, d.Cout, d.KH, max(d.stride, d.up), d.groups) if isinstance(d, hip.ConvDesc) else "")
    return _orig_call(name, *args)


_rt.call = hip.call = _call
hip.profiler = hip.LaunchProfiler()
step()
torch.cuda.synchronize()
recs = hip.profiler.records
hip.profiler = None
_rt.call = hip.call = _orig_call
rows = []
for i, (name, s, e, meta) in enumerate(recs):
    ms = s.elapsed_time(e)
    fl, by = meta[0], meta[1]
    kern = meta[2] if len(meta) > 2 and meta[2] else ""
    floor = max(by / 5.3e9, fl / 8e11) if (by or fl) else 0.0        # ms
    rows.append((ms - floor, i, name.replace("adamml_", ""), kern, ms, by / ms / 1e6 if ms and by else 0.0, fl / ms / 1e9 if ms and fl else 0.0, by / 1e9))
tot = sum(r[4] for r in rows)
print("%s: %d launches, %.1f ms of launch time; top %d by excess over the 5.3 TB/s / 800 TFLOP/s floor (no figure: caller supplies no bytes)" % (which, len(rows), tot, TOP))
for ex, i, name, kern, ms, gbs, tfs, gb in sorted(rows, reverse=True)[:TOP]:
    print("#%4d %-28s %-18s %7.3f ms %6.0f GB/s %5.0f TF/s %6.2f GB  excess %6.3f" % (i, name[:28], kern[:18], ms, gbs, tfs, gb, ex))

if len(sys.argv) > 4 and sys.argv[4] == "all":
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, s_, e, meta in recs:
        a = agg[name.replace("adamml_", "")]
        a[0] += 1
        a[1] += s_.elapsed_time(e)
    print("per entry point:")
    for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-32s %4d launches %8.3f ms" % (k, c, ms))
    print("chronological:")
    for i, (name, s_, e, meta) in enumerate(recs):
        ms = s_.elapsed_time(e)
        print("@%4d %-28s %-26s %7.3f ms %6.2f GB %6.0f GB/s" % (i, name.replace("adamml_", "")[:28], shapes[i] if i < len(shapes) else "", ms, meta[1] / 1e9,
                                                             meta[1] / ms / 1e6 if ms and meta[1] else 0))
