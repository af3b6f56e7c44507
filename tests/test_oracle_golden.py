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
