"""__graft_entry__.smoke(): AdaMML (RGB+Audio, LSTM policy) main-net-stage training steps on cuda:0 through the HIP hot path:
(1) a small case checked against the CPU oracle run here (bf16-storage emulation) and the reference golden decisions -- B = 2 at 96 px
    leaves 4-18 samples per BatchNorm channel in the deep layers, so its logit distance to the fp32 reference (~0.14) is that of ANY
    bf16-storage pipeline at this size (the oracle's own emulation: ~0.11);
(2) the BASELINE.json configs[1] workload at B = 4 videos (5 segments, 224^2 / 256^2: >= 196 samples per channel everywhere) against the
    golden logits / decisions the REAL reference produced (tests/golden/adamml_c2.npz, tools/gen_golden.py; the oracle is pinned to the
    same file by tests/test_oracle_golden.py): the well-conditioned number."""
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
          % (float(loss.detach()), e_emu, e_ref, d_emu))
    # fixed numbers, 1.3 x measured (0.0555 / 0.1357; reproducible: order-fixed sums, csrc/common.h) -- this 96-pixel case is ill-conditioned
    # by construction (the bf16-storage emulation itself sits 0.163 from the fp32 reference); the tight statement is the full-size one below
    assert e_emu <= 0.072 and e_ref <= 0.176, (e_emu, e_ref)
    del model
    # ---- (2) full-size, well-conditioned: the configs[1] workload at B = 4 against the reference golden
    c2 = CASES["adamml_c2"]
    gold2 = load_golden("adamml_c2")
    mod = c2["modality"]
    model = adamml(groups=c2["groups"], modality=mod, input_channels=[CH[m] for m in mod], num_segments=c2["S"], rng_policy=False,
                   rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0,
                   pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    model.load_state_dict(synth.synth_state_dict(manifest(c2), seed=1234))
    model.to("cuda:0")
    xs, target = case_inputs(c2)
    xs, target = [t.to("cuda:0") for t in xs], target.to("cuda:0")
    model.freeze_policy_net()
    model.train()
    logits, sel = model(xs, gumbel_exponential=case_gumbel(c2).to("cuda:0"))
    F.cross_entropy(logits, target).backward()
    torch.cuda.synchronize()
    assert np.array_equal(np.round(sel.detach().cpu().numpy()), np.round(gold2["train_main.decisions"])), "full-size decisions differ from the reference"
    g2 = gold2["train_main.logits"]
    e2 = np.abs(logits.detach().cpu().numpy() - g2).max() / np.abs(g2).max()
    ep = np.abs(model.last_policy_logits.detach().cpu().numpy() - gold2["train_main.policy_logits"]).max() / np.abs(gold2["train_main.policy_logits"]).max()
    print("smoke: full size (B=4, S=5, 224^2 / 256^2) logits |HIP-reference| %.4f, policy logits %.4f of scale | decisions match" % (e2, ep))
    # 1.3 x measured (logits 3.09e-2; policy logits, two 52-layer random-weight MobileNetV2 stacks in front of them, 5.60e-2): reproducible
    assert e2 <= 4.0e-2 and ep <= 7.3e-2, (e2, ep)
