"""Pins the oracle (oracle/adamml_oracle.py) against golden vectors captured from the
real reference by tools/gen_golden.py (CPU fp32).  Runs without GPU and without
/root/reference."""
import pytest

from tests.golden_cases import CASES
from tests.oracle_harness import oracle_case, load_golden, compare_records


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_oracle_matches_reference_golden(name):
    got = oracle_case(CASES[name])
    compare_records(got, load_golden(name))


def test_policy_loss_matches_reference_on_partly_correct_predictions():
    """compute_policy_loss (utils/utils.py:166-184) incl. its [N] x [N,1] broadcast in the blockdrop term, on cases where
    about half of the top-1 predictions are right (tools/gen_policy_loss_golden.py ran the reference function)."""
    import os
    import numpy as np
    import torch
    from oracle import adamml_oracle as O
    from adamml_amd.train import compute_policy_loss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy_loss_cases.npz"))
    for ci in range(4):
        sel, lg, tg, cw = [torch.from_numpy(g["c%d.%s" % (ci, k)]) for k in ("sel", "logits", "target", "cw")]
        for pt in ("blockdrop", "mean"):
            want = float(g["c%d.%s" % (ci, pt)])
            assert abs(float(O.policy_loss(pt, sel, cw, torch.tensor(10.0), lg, tg)) - want) <= 1e-6 * max(1.0, abs(want))
            assert abs(float(compute_policy_loss(pt, sel, cw, torch.tensor(10.0), lg, tg)) - want) <= 1e-6 * max(1.0, abs(want))


def test_oracle_training_trajectory_follows_the_reference():
    """tests/golden/adamml_c2_traj.npz (tools/gen_golden_traj.py: the REAL reference taking 20 main-net-stage optimizer steps on the
    adamml_c2 batch, utils/utils.py:359-400 + torch.optim.SGD as train_adamml.py:251-257): the oracle, stepped with the same
    optimizer, reproduces the first steps' losses and logits -- i.e. its gradients are the reference's all the way into the update,
    and its running statistics / frozen-policy behaviour carry from step to step.  (3 of the 20 steps here: CPU time; the HIP path
    is held to all 20 in tests/test_train_trajectory_gpu.py.)"""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from adamml_amd import synth
    from oracle import adamml_oracle as O
    from tests.oracle_harness import manifest, case_inputs
    c = CASES["adamml_c2"]
    traj = load_golden("adamml_c2_traj")
    sd = O.make_leaf_state(synth.synth_state_dict(manifest(c), seed=1234), ("main_net.",))
    xs, target = case_inputs(c)
    expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=int(traj["gumbel_seed"]))
    params = [v for k, v in sd.items() if v.requires_grad]
    opt = torch.optim.SGD(params, float(traj["lr"]), momentum=float(traj["momentum"]), weight_decay=float(traj["weight_decay"]))
    for it in range(3):
        logits, sel, _ = O.adamml_forward(sd, xs, c["modality"], c["S"], c["groups"], 50, 5.0, expo, "lstm", "max", False, 0.0, True)
        loss = F.cross_entropy(logits, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert np.array_equal(np.round(sel.detach().numpy()), np.round(traj["decisions"]))
        lv = float(loss.detach())
        assert abs(lv - float(traj["loss"][it])) <= 2e-4 * float(traj["loss"][it]), (it, lv, float(traj["loss"][it]))
        np.testing.assert_allclose(logits.detach().numpy(), traj["logits"][it], rtol=2e-3, atol=2e-4)
