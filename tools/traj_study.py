"""CPU study (oracle only): the 20-step trajectory of tests/golden/adamml_c2_traj.npz taken by the oracle with its bf16-STORAGE
emulation (oracle.QUANT: every tensor the HIP path keeps in bf16 is rounded, arithmetic stays fp32).  Shows how much of the HIP
path's distance from the fp32 reference curve is intrinsic to bf16 activation / gradient storage, and (emulate = 1) writes that curve
to tests/golden/adamml_c2_traj_bf16emu.npz: the second comparator of tests/test_train_trajectory_gpu.py.  Usage: python tools/traj_study.py [emulate 0|1]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adamml_amd import synth  # noqa: E402
from oracle import adamml_oracle as O  # noqa: E402
from tests.golden_cases import CASES  # noqa: E402
from tests.oracle_harness import manifest, case_inputs, load_golden  # noqa: E402

emulate = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
c = CASES["adamml_c2"]
traj = load_golden("adamml_c2_traj")
sd = O.make_leaf_state(synth.synth_state_dict(manifest(c), seed=1234), ("main_net.",))
xs, target = case_inputs(c)
expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=int(traj["gumbel_seed"]))
opt = torch.optim.SGD([v for v in sd.values() if v.requires_grad], float(traj["lr"]), momentum=float(traj["momentum"]),
                      weight_decay=float(traj["weight_decay"]))
O.QUANT = O.bf16_straight_through if emulate else None
losses = []
for it in range(int(traj["steps"])):
    logits, sel, _ = O.adamml_forward(sd, xs, c["modality"], c["S"], c["groups"], 50, 5.0, expo, "lstm", "max", False, 0.0, True)
    loss = F.cross_entropy(logits, target)
    opt.zero_grad()
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
    print("step %2d loss %.4f ref %.4f rel %.4f" % (it, losses[-1], traj["loss"][it], abs(losses[-1] - traj["loss"][it]) / traj["loss"][it]), flush=True)
rel = np.abs(np.array(losses) - traj["loss"]) / traj["loss"]
print("emulate_bf16=%s: max rel %.4f at step %d, mean %.4f" % (emulate, rel.max(), int(rel.argmax()), rel.mean()))
if emulate:
    # fixture for tests/test_train_trajectory_gpu.py: the curve a bf16-STORAGE pipeline with exact fp32 arithmetic follows
    path = os.path.join(ROOT, "tests", "golden", "adamml_c2_traj_bf16emu.npz")
    np.savez_compressed(path, loss=np.array(losses, dtype=np.float64), steps=np.array(len(losses)))
    print(path)
