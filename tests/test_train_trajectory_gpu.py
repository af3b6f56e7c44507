"""Does it TRAIN: the HIP path takes the 20 optimizer steps of tests/golden/adamml_c2_traj.npz (tools/gen_golden_traj.py: the REAL
fp32 reference, main-net stage of utils/utils.py:359-400 with torch.optim.SGD(lr 0.01, momentum 0.9, weight decay 1e-4) as
train_adamml.py:251-257 builds it, on the adamml_c2 batch: RGB+Audio AdaMML, B = 4 videos, 5 segments, 224^2 / 256^2) with its own
fused flat SGD (adamml_sgd_step) and must follow the reference's loss curve step by step -- the statement that the bf16-stored
gradients are USEFUL, not merely plausible: every step's gradients feed the next step's weights, BatchNorm running statistics and
momentum buffers, so an error in any of them compounds over the 20 steps instead of averaging out.

Two comparators:
  (1) the fp32 reference curve itself.  Training in bf16 STORAGE converges slightly more slowly on this 4-video batch: the loss falls
      3.49 -> 0.37 in the reference and 3.49 -> 0.43 here, a lag of about one step in twenty; relative to the (exponentially falling)
      reference loss that is 0.1 % at step 0, 3 % at step 8, 18 % at step 19.
  (2) the curve of the ORACLE WITH ITS bf16-STORAGE EMULATION (tests/golden/adamml_c2_traj_bf16emu.npz, tools/traj_study.py: every
      tensor the HIP path keeps in bf16 is rounded, all arithmetic fp32, same 20 steps on the CPU).  That curve shows the same lag
      (18.2 % at step 19): the distance in (1) is the price of bf16 activations / gradients, not of the kernels.  The HIP path must
      follow THIS curve tightly at every step.
Measured on MI355X (printed by the test): vs the emulation max 2.0 % / mean 0.8 % per step (the HIP path sits on the reference's side of
the emulation since its MobileNetV2 stems read the fp32 spectrogram unrounded; with bf16 input it followed the emulation within 0.9 %);
vs the fp32 reference max 15.8 % (step 19) / mean 5.0 %; final logits 4.3e-2 of scale vs the reference; ResNet fc weight update over the
20 steps 2.5e-2 rel L2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES  # noqa: E402
from tests.oracle_harness import manifest, load_golden, case_inputs  # noqa: E402
from tests.test_parity_fullsize_gpu import build, rel_max  # noqa: E402

DEV = "cuda"


def test_twenty_steps_follow_the_reference_loss_curve():
    from adamml_amd import hip
    from adamml_amd.optim import FlatSGD
    c = CASES["adamml_c2"]
    traj = load_golden("adamml_c2_traj")
    steps = int(traj["steps"])
    model = build(c)
    model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
    model.to(DEV)
    xs, target = case_inputs(c)
    xs, target = [t.to(DEV) for t in xs], target.to(DEV)
    expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=int(traj["gumbel_seed"])).to(DEV)
    model.freeze_policy_net()
    model.unfreeze_main_net()
    model.train()
    opt = None
    losses, worst_logit = [], 0.0
    for it in range(steps):
        logits, sel = model(xs, gumbel_exponential=expo)
        loss = F.cross_entropy(logits, target)
        loss.backward()
        if opt is None:
            opt = FlatSGD(model._flat_main, lr=float(traj["lr"]), momentum=float(traj["momentum"]), weight_decay=float(traj["weight_decay"]))
        opt.step()
        opt.zero_grad()
        assert np.array_equal(np.round(sel.detach().cpu().numpy()), np.round(traj["decisions"])), "decisions differ at step %d" % it
        losses.append(float(loss.item()))
        worst_logit = max(worst_logit, rel_max(logits.detach().cpu().numpy(), traj["logits"][it]))
    with torch.no_grad():
        final_logits, _ = model(xs, gumbel_exponential=expo)
    ref = traj["loss"]
    emu = load_golden("adamml_c2_traj_bf16emu")["loss"]
    rel = np.abs(np.array(losses) - ref) / ref
    rel_emu = np.abs(np.array(losses) - emu) / emu
    e_final = rel_max(final_logits.cpu().numpy(), traj["final_logits"])
    e_fc = float(np.linalg.norm(model.main_net.nets[0].fc.weight.detach().cpu().numpy() - traj["final_fc_weight"]) /
                 np.linalg.norm(traj["final_fc_weight"] - synth.synth_state_dict(manifest(c), seed=1234)["main_net.nets.0.fc.weight"].numpy()))
    print("adamml_c2 trajectory: loss HIP  %s" % np.round(losses, 4))
    print("                               loss ref  %s" % np.round(ref, 4))
    print("                          loss bf16-emu  %s" % np.round(emu, 4))
    print("  per-step |loss - bf16-storage emulation| / emulation: max %.4f (step %d), mean %.4f" % (rel_emu.max(), int(rel_emu.argmax()), rel_emu.mean()))
    print("  per-step |loss - reference| / reference: max %.4f (step %d), mean %.4f; logits along the way max %.4f of scale; after the "
          "last update: logits %.4f of scale, ResNet fc weight UPDATE (w20 - w0) rel L2 %.4f" % (rel.max(), int(rel.argmax()), rel.mean(),
                                                                                               worst_logit, e_final, e_fc))
    assert losses[-1] < 0.2 * losses[0]                   # it trains: 3.49 -> 0.37 in the reference
    # Every per-channel sum is order-fixed (csrc/common.h), so the trajectory is a reproducible sequence of numbers (both modes print the
    # same curve to the last digit shown); the bounds live in tests/parity_bounds.json (1.3 x ONE measured value, below the stated
    # tolerance of the category) and are re-based with the others by tools/rebase_bounds.py -- a kernel that reorders an fp32
    # accumulation moves the late steps of a 20-step run by a few parts in a thousand (round 5: 1.56 % -> 2.16 % against the emulation
    # when the depthwise backward formed dz in its loader) and must not be vetoed by a literal here.
    from tests.parity_bounds import check
    check("traj_c2.loss_vs_emu_max", rel_emu.max(), "step %d" % int(rel_emu.argmax()), "traj_emu")      # the curve bf16 storage allows
    # vs the fp32 reference: the bf16-storage lag of about one step in twenty (the emulation's own lag is 0.17 / 0.056)
    check("traj_c2.loss_vs_ref_max", rel.max(), "step %d" % int(rel.argmax()), "traj_ref_max")
    check("traj_c2.loss_vs_ref_mean", rel.mean(), "", "traj_ref_mean")
    check("traj_c2.final_logits", e_final, "", "traj_logits")
    check("traj_c2.fc_update", e_fc, "", "traj_fc")
