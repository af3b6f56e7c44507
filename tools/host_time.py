"""Host-side issue time of one training step (time until step() returns, GPU not synchronised) vs the synchronised step
time: tells whether the Python / ctypes launch path is a co-bottleneck."""
import sys, time, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
from adamml_amd import adamml, synth
from adamml_amd.optim import FlatSGD
B, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 72), 5
dev = torch.device("cuda")
m = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
           causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.5, pooling_method="max",
           fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=1234)); m.to(dev); m.freeze_policy_net(); m.train()
x = [torch.randn(B, S * 24, 224, 224, device=dev), torch.randn(B, S, 256, 256, device=dev)]
t = torch.randint(0, 31, (B,), device=dev)
opt = None
def step():
    global opt
    out, sel = m(x)
    F.cross_entropy(out, t).backward()
    if opt is None:
        opt = FlatSGD(m._flat_main, lr=0.001, momentum=0.9, weight_decay=5e-4)
    opt.step(); opt.zero_grad()
for _ in range(3):
    step()
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.time(); step(); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("host issue %.1f ms, step complete %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
