"""Which Python call sites issue the small device-to-device copies / fills of a training step (rocprofv3 counts ~135
__amd_rocclr_copyBuffer and ~30 fill kernels per step)?  torch.profiler with stacks over two steps of the benchmark workload; prints the
copy / fill events grouped by their innermost adamml_amd / bench frame.  GPU box only.   usage: python tools/copy_sites.py [batch]"""
import collections
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from adamml_amd.optim import FlatSGD  # noqa: E402

sys.argv = [sys.argv[0], "--batch", sys.argv[1] if len(sys.argv) > 1 else "72"]
args = bench.parse()
dev = torch.device("cuda:0")
model = bench.build(args, dev)
model.freeze_policy_net()
model.train()
images, target = bench.synth_batch(args, args.batch, dev, 0)
opt = [None]


def step():
    out, sel = model(images)
    F.cross_entropy(out, target).backward()
    if opt[0] is None:
        opt[0] = FlatSGD(model._flat_main, lr=0.001, momentum=0.9, weight_decay=5e-4)
    opt[0].step()
    opt[0].zero_grad()


for _ in range(4):
    step()
torch.cuda.synchronize()
NSTEP = 2
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(NSTEP):
        step()
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    n = ev.name
    if not (n.startswith("aten::copy_") or n.startswith("aten::fill_") or n.startswith("aten::zero_") or "Memcpy" in n or "Memset" in n):
        continue
    frame = next((f for f in (ev.stack or []) if "adamml_amd" in f or "bench.py" in f or "tools/" in f), "(no repo frame)")
    sites[(n, frame.strip()[:150])] += 1
for (n, f), c in sites.most_common(40):
    print("%6.1f / step  %-22s %s" % (c / NSTEP, n, f))
