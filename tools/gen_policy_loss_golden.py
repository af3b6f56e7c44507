"""Golden vectors of the reference's compute_policy_loss (utils/utils.py:166-184) on hand-made selections where the
classifier is partly RIGHT -- the whole-network goldens have random 31-way logits whose top-1 is almost always wrong, so
they do not pin the `correctness * pl` term, whose [N] x [N,1] broadcast in the reference averages over an N x N outer
product (mean(correct) * mean(usage), not mean(correct * usage)).

Run here (needs /root/reference): python tools/gen_policy_loss_golden.py -> tests/golden/policy_loss_cases.npz"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
tvt.Compose = lambda ts: None
tvt.CenterCrop = tvt.Resize = lambda *a, **k: None
tv.transforms = tvt
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
from utils.utils import compute_policy_loss  # noqa: E402

rng = np.random.Generator(np.random.PCG64(20260929))
out = {}
for ci, (N, S, M) in enumerate([(6, 5, 2), (4, 3, 3), (8, 5, 4), (1, 5, 2)]):
    sel = torch.from_numpy((rng.random((N, S, M)) > 0.4).astype(np.float32))
    logits = torch.from_numpy(rng.standard_normal((N, 7)).astype(np.float32))
    target = logits.argmax(-1).clone()
    wrong = torch.from_numpy(rng.random(N) > 0.5)
    target[wrong] = (target[wrong] + 1) % 7                     # about half of the predictions are correct
    cw = torch.from_numpy(rng.random(M).astype(np.float32))
    gam = torch.tensor(10.0)
    out["c%d.sel" % ci], out["c%d.logits" % ci], out["c%d.target" % ci], out["c%d.cw" % ci] = \
        sel.numpy(), logits.numpy(), target.numpy(), cw.numpy()
    for pt in ("blockdrop", "mean"):
        out["c%d.%s" % (ci, pt)] = compute_policy_loss(pt, sel, cw, gam, logits, target).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "policy_loss_cases.npz"), **out)
print({k: v for k, v in out.items() if k.endswith(("blockdrop", "mean"))})
