"""Device time of each backbone alone at the benchmark shapes (B=72 x 5 segments), single stream: forward, and
forward+backward for the trainable main nets -- shows how the step's device time splits between the networks."""
import sys, time, torch
sys.path.insert(0, ".")
from adamml_amd import adamml, synth, hip
from adamml_amd.adamml import _frames
from adamml_amd.runtime import clip_to_nhwc
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 72, 5
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


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def fwd_bwd(net, x):
    def f():
        out = net.forward_nhwc(x, S)
        out.sum().backward()
    return f


def fwd(net, x):
    def f():
        with torch.no_grad():
            net.call(x, S)
    return f


res, sound = m.main_net.nets
print("ResNet-50 main      fwd+bwd %.1f ms" % timed(fwd_bwd(res, _frames(m_x[0]))))
print("Sound-MBv2 main     fwd+bwd %.1f ms" % timed(fwd_bwd(sound, _frames(m_x[1]))))
m.freeze_policy_net()
pr, ps = m.policy_net.joint_net.nets
for net in (pr, ps):
    net.train()
print("policy MBv2 rgb     fwd     %.1f ms" % timed(fwd(pr, _frames(p_x[0]))))
print("policy MBv2 sound   fwd     %.1f ms" % timed(fwd(ps, _frames(p_x[1]))))

if len(sys.argv) > 2 and sys.argv[2] == "profile-sound":
    f = fwd_bwd(sound, _frames(m_x[1]))
    f(); torch.cuda.synchronize()
    hip.profiler = hip.LaunchProfiler()
    f()
    agg = hip.profiler.summary()
    hip.profiler = None
    tot = sum(a["ms"] for a in agg.values())
    print("sound net per entry point (total %.1f ms):" % tot)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:14]:
        print("  %-34s %4d launches %7.2f ms  %6.0f GB/s (algorithmic)" % (k, a["launches"], a["ms"], a["bytes"] / (a["ms"] * 1e6) if a["ms"] else 0))
    recs = [(n, s.elapsed_time(e), meta) for n, s, e, meta in []]
