"""__graft_entry__.smoke(): one small AdaMML (RGB+Audio, LSTM policy) main-net-stage training step on cuda:0 through
the HIP hot path, checked against the CPU oracle (bf16-storage emulation) and the reference golden decisions."""
import numpy as np
import torch
import torch.nn.functional as F


def run_smoke():
    from adamml_amd import adamml, synth, hip
    from adamml_amd.optim import FlatSGD
    from tests.golden_cases import CASES, CH
    from tests.oracle_harness import manifest, load_golden, case_inputs, case_gumbel, oracle_case
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X (no CPU fallback)")
    hip.load()
    name = "adamml_rgb_sound"
    c = CASES[name]
    gold = load_golden(name)
    mod = c["modality"]
    model = adamml(groups=c["groups"], modality=mod, input_channels=[CH[m] for m in mod], num_segments=c["S"], rng_policy=False,
                   rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0,
                   pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
    model.to("cuda:0")
    xs, target = case_inputs(c)
    xs, target = [t.to("cuda:0") for t in xs], target.to("cuda:0")
    expo = case_gumbel(c).to("cuda:0")
    model.freeze_policy_net()
    model.train()
    logits, sel = model(xs, gumbel_exponential=expo)
    loss = F.cross_entropy(logits, target)
    loss.backward()
    FlatSGD(model._flat_main, lr=0.01, momentum=0.9, weight_decay=5e-4).step()
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(model._flat_main.flat).all()
    assert np.array_equal(np.round(sel.detach().cpu().numpy()), np.round(gold["train_main.decisions"])), "decisions differ from the reference"
    emu = oracle_case(c, emulate_bf16=True, modes=["train_main"])
    a, b = logits.detach().cpu().numpy(), emu["train_main.logits"]
    g = gold["train_main.logits"]
    e_emu = np.abs(a - b).max() / np.abs(b).max()
    e_ref = np.abs(a - g).max() / np.abs(g).max()
    d_emu = np.abs(b - g).max() / np.abs(g).max()
    print("smoke: loss %.4f | logits |HIP-emulation| %.4f |HIP-reference| %.4f (|emulation-reference| %.4f) | decisions match"
          % (float(loss), e_emu, e_ref, d_emu))
    assert e_emu <= max(3e-2, 1.5 * d_emu) and e_ref <= max(3e-2, 3.0 * d_emu)
