"""Whole-network parity on the MI355X.

What can be stated (measured in this repo, see DESIGN.md "Parity"):
  * oracle fp32 == reference golden (CPU, tests/test_oracle_golden.py, 1e-4) -- the pin;
  * every HIP kernel == the torch fp32 operator it replaces on identical bf16 operands (tests/test_kernels_gpu.py,
    1e-2 of the output scale) -- the tight, per-launch parity;
  * whole network, train mode: a randomly initialised 50-layer train-mode-BatchNorm net is chaotic -- perturbing the
    fp32 oracle's weights by 1e-7 moves its own stem gradients by 0.6 %, and merely STORING conv outputs in bf16
    (oracle.QUANT emulation, fp32 arithmetic) moves logits by ~1.5 % and gradients by 20-45 % relative L2, for any
    pipeline.  So the whole-network statement is relative to that emulation:
        logits:    |HIP - fp32| <= max(3e-2, 3 x |emulation - fp32|)  and  |HIP - emulation| <= max(3e-2, 1.5 x |emulation - fp32|)
                   (the emulation has to store what the HIP path stores: while it still rounded the spectrogram and the MobileNetV2
                   stem weights to bf16 -- the HIP stems read both in fp32 since round 3 -- the ill-conditioned 96-pixel RGB+Audio
                   case measured 1.40 x and 1.58 x in two runs of one build; with the stem emulated as built: 0.46 x)
        gradients: relL2(HIP, fp32) <= max(0.35, 2.2 x relL2(emulation, fp32)), cosine(HIP, fp32) >= 0.6
                   (tensors whose EMULATION already sits > 0.5 relL2 from fp32 carry no information and are skipped)
        running statistics: relL2(HIP, emulation) <= 6e-2
  * eval mode with calibrated running statistics (BatchNorm = fixed affine, no chaos): logits 4e-2 vs fp32.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES, grad_probe, stat_probe  # noqa: E402
from tests.oracle_harness import manifest, load_golden, case_inputs, oracle_case  # noqa: E402

DEV = "cuda"
LOGIT_TIGHT, GRAD_FLOOR, STAT_TIGHT = 3e-2, 0.35, 6e-2   # GRAD_FLOOR: two equivalent bf16 pipelines (emulation vs emulation with 1e-7 weight noise) already differ by 0.22-0.27 relL2


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def rel_l2(a, b):
    b = b.to(a.device)
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def probe_errors(named_probes, gold, mode):
    names = list(gold[mode + ".grad_names"])
    ref = gold[mode + ".grad_probe"]
    floor = 1e-3 * ref[:, 1].max()
    errs = {}
    for i, k in enumerate(names):
        got = named_probes[k]
        l2 = max(ref[i, 1], floor)
        errs[k] = max(abs(got[1] - ref[i, 1]) / l2, np.abs(got[2:] - ref[i, 2:]).max() / l2)
    return errs


def check_against_emulation(model, emu, ref, mode, logits):
    """emu / ref: oracle runs on the GPU with and without the bf16-storage emulation (keep_grads=True)."""
    e = rel_err(logits, emu[mode + ".logits"])
    e_ef = rel_err(emu[mode + ".logits"], ref[mode + ".logits"])
    print("  [%s] logits: |HIP-emulation| %.4f, |emulation-fp32| %.4f" % (mode, e, e_ef))
    assert e <= max(LOGIT_TIGHT, 1.5 * e_ef), e
    if mode + ".grads" not in emu:
        return
    g_emu, g_ref = emu[mode + ".grads"], ref[mode + ".grads"]
    params = dict(model.named_parameters())
    gmax = max(g.norm().item() for g in g_ref.values())
    worst, worst_cos, n, skipped = (0.0, None, 0.0, 0.0), (1.0, None), 0, 0
    for k, gr in g_ref.items():
        assert params[k].grad is not None, "no grad for " + k
        if gr.norm().item() < 1e-4 * gmax:      # analytically-zero gradients (bias before a BatchNorm)
            continue
        gh = params[k].grad
        assert torch.isfinite(gh).all(), k
        d_hf, d_ef = rel_l2(gh, gr), rel_l2(g_emu[k], gr)
        if d_ef > 0.5 or gr.numel() < 16:
            # the emulation itself is decorrelated from fp32 here (chaotic regime, MobileNet stacks) or the tensor
            # is a handful of scalars: no information in a comparison -- covered by kernel / block-level tests
            skipped += 1
            continue
        ratio = d_hf / max(GRAD_FLOOR, 2.2 * d_ef)
        cos = F.cosine_similarity(gh.flatten().double(), gr.to(gh.device).flatten().double(), dim=0).item()
        n += 1
        if ratio > worst[0]:
            worst = (ratio, k, d_hf, d_ef)
        if cos < worst_cos[0]:
            worst_cos = (cos, k)
    print("  [%s] %d gradient tensors compared (%d in the chaotic regime skipped): worst relL2(HIP,fp32)=%.3f vs relL2(emulation,fp32)=%.3f at %s (%.2f of bound); "
          "min cosine %.3f at %s" % (mode, n, skipped, worst[2], worst[3], worst[1], worst[0], worst_cos[0], worst_cos[1]))
    assert worst[0] <= 1.0, worst
    assert worst_cos[0] >= 0.6, worst_cos
    st, st_ref = emu[mode + ".state"], ref[mode + ".state"]
    sd = model.state_dict()
    ws = (0.0, None)
    for k, v in st.items():
        if k.endswith(("running_mean", "running_var")):
            r = rel_l2(sd[k], v) / max(STAT_TIGHT, 2.0 * rel_l2(v, st_ref[k]))
            if r > ws[0]:
                ws = (r, k)
        elif k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
    print("  [%s] running stats: worst relL2(HIP,emulation) at %.2f of max(6e-2, 2 x relL2(emulation,fp32)) (%s)"
          % (mode, ws[0], ws[1]))
    assert ws[0] <= 1.0, ws


# Fixed-number gates of the reduced cases: 1.3 x the logit error measured on MI355X against the reference golden (reproducible: every
# per-channel sum is order-fixed).  These cases are ill-conditioned BY CONSTRUCTION (B = 1-2 at 64-96 px: 4-18 samples per BatchNorm
# channel in the deep layers) -- the bf16-storage emulation of the oracle itself sits as far from fp32 (second column of the printout) --
# so the numbers are large; they pin the plumbing of each variant, the tight statements are the full-size ones (test_parity_fullsize_gpu.py).
LOGIT_BOUND = {("resnet50_train", "train"): 0.0236, ("resnet50_avg", "train"): 0.0120, ("resnet50_flow", "train"): 0.0370,
               ("sound_mbv2", "train"): 0.156, ("adamml_rgb_sound", "train_main"): 0.177, ("adamml_rgb_sound", "train_policy"): 0.177,
               ("adamml_rgb_sound_nolstm", "train_policy"): 0.567, ("adamml_rgb_flow_rgbdiff", "train_main"): 0.0471,
               ("adamml_4mod", "train_policy"): 0.0331}


def check_against_golden(model, gold, emu, mode, logits, k=3.0, name=None):
    e_hip, e_emu = rel_err(logits, gold[mode + ".logits"]), rel_err(emu[mode + ".logits"], gold[mode + ".logits"])
    print("  [%s] vs fp32 golden: HIP logits %.4f, emulation %.4f" % (mode, e_hip, e_emu))
    assert e_hip <= max(3e-2, k * e_emu)
    if (name, mode) in LOGIT_BOUND:
        assert e_hip <= LOGIT_BOUND[(name, mode)], (name, mode, e_hip)
    if mode + ".grad_names" not in gold:
        return
    names = list(gold[mode + ".grad_names"])
    params = dict(model.named_parameters())
    hip_e = probe_errors({kn: grad_probe(kn, params[kn].grad) for kn in names}, gold, mode)
    enames = list(emu[mode + ".grad_names"])
    emu_e = probe_errors({kn: emu[mode + ".grad_probe"][i] for i, kn in enumerate(enames)}, gold, mode)
    med = float(np.median(list(emu_e.values())))
    worst = max((hip_e[kn] / max(8e-2, k * max(emu_e[kn], med)), kn) for kn in names if emu_e[kn] < 0.3)
    # 4-sample probes are too noisy in the chaotic regime to gate on; reported for the record
    print("  [%s] vs fp32 golden: worst gradient probe at %.2f of 3x the emulation's probe error (%s); emulation median "
          "probe err %.4f" % (mode, worst[0], worst[1], med))


def calibrated_state(sd, run_oracle_train):
    """Synthetic running statistics are arbitrary, so eval-mode activations grow geometrically through the
    residual stack and logits become differences of 1e4-sized features (ill-conditioned for ANY reduced
    precision).  For eval-mode parity the running statistics are first set to the batch statistics of the test
    input by one train-mode oracle pass with momentum 1 -- the regime real checkpoints are in."""
    from oracle import adamml_oracle as O
    cal = {k: v.clone() for k, v in sd.items()}
    old = O.BN_MOMENTUM
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            run_oracle_train(cal)
    finally:
        O.BN_MOMENTUM = old
    for k in cal:
        if k.endswith("num_batches_tracked"):
            cal[k].zero_()
    return cal


def check_eval(y, cal, run_oracle_eval):
    """Eval mode (BatchNorm = fixed affine): HIP vs the fp32 oracle, relative to the bf16-storage emulation's own
    distance from fp32 (a random 52-layer MobileNetV2 expands ANY perturbation ~1.09x per layer even in eval mode:
    BatchNorm removes the post-ReLU mean, i.e. a third of the signal energy but none of the perturbation's)."""
    from oracle import adamml_oracle as O
    with torch.no_grad():
        ref = run_oracle_eval({k: v.clone() for k, v in cal.items()})
        O.QUANT = O.bf16_straight_through
        try:
            emu = run_oracle_eval({k: v.clone() for k, v in cal.items()})
        finally:
            O.QUANT = None
    e, e_emu = rel_err(y.cpu().numpy(), ref.numpy()), rel_err(emu.numpy(), ref.numpy())
    print("  [eval, calibrated] logits: |HIP-fp32| %.4f, |emulation-fp32| %.4f" % (e, e_emu))
    assert e <= max(4e-2, 2.0 * e_emu)


@pytest.mark.parametrize("name", ["resnet50_train", "resnet50_avg", "resnet50_flow"])
def test_resnet50(name):
    from adamml_amd.resnet import resnet
    from oracle import adamml_oracle as O
    c = CASES[name]
    gold = load_golden(name)
    emu = oracle_case(c, emulate_bf16=True, modes=["train"], device="cpu", keep_grads=True)
    ref = oracle_case(c, emulate_bf16=False, modes=["train"], device="cpu", keep_grads=True)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    model = resnet(depth=50, num_classes=31, without_t_stride=False, groups=c["groups"], dropout=0.0,
                   pooling_method=c.get("pooling", "max"), input_channels={"rgb": 3, "flow": 10}[c["modality"][0]],
                   imagenet_pretrained=False)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.to(DEV)
    x_cpu, target = case_inputs(c)
    x, target = x_cpu.to(DEV), target.to(DEV)
    pool = c.get("pooling", "max")
    model.train()
    model.zero_grad()
    y = model(x)
    F.cross_entropy(y, target).backward()
    print(name)
    check_against_emulation(model, emu, ref, "train", y.detach().cpu().numpy())
    check_against_golden(model, gold, emu, "train", y.detach().cpu().numpy(), name=name)
    if name != "resnet50_train":
        return          # the reduced B=1 / 64x64 cases leave 4 samples per layer4 channel: calibration is meaningless
    # eval mode with calibrated running statistics: BatchNorm is a fixed affine map -> plain bf16 tolerance vs fp32
    cal = calibrated_state(sd, lambda s: O.resnet_forward(s, "", x_cpu, c["groups"], 50, pool, False, 0.0, True))
    model.load_state_dict(cal)
    model.eval()
    with torch.no_grad():
        y = model(x)
    check_eval(y, cal, lambda s: O.resnet_forward(s, "", x_cpu, c["groups"], 50, pool, False, 0.0, False))


def test_sound_mobilenet_v2():
    from adamml_amd.sound_mobilenet_v2 import sound_mobilenet_v2
    from oracle import adamml_oracle as O
    name = "sound_mbv2"
    c = CASES[name]
    gold = load_golden(name)
    emu = oracle_case(c, emulate_bf16=True, modes=["train"], device="cpu", keep_grads=True)
    ref = oracle_case(c, emulate_bf16=False, modes=["train"], device="cpu", keep_grads=True)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    model = sound_mobilenet_v2(num_classes=31, input_channels=1, dropout=0.0, imagenet_pretrained=False)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.to(DEV)
    x_cpu, target = case_inputs(c)
    x, target = x_cpu.to(DEV), target.to(DEV)
    model.train()
    model.zero_grad()
    y = model(x)
    F.cross_entropy(y, target).backward()
    print(name)
    check_against_emulation(model, emu, ref, "train", y.detach().cpu().numpy())
    check_against_golden(model, gold, emu, "train", y.detach().cpu().numpy(), name="sound_mbv2")
    cal = calibrated_state(sd, lambda s: O.sound_mbv2_forward(s, "", x_cpu, 0.0, True))
    model.load_state_dict(cal)
    model.eval()
    with torch.no_grad():
        y = model(x)
    check_eval(y, cal, lambda s: O.sound_mbv2_forward(s, "", x_cpu, 0.0, False))


def build_adamml(c):
    from adamml_amd import adamml
    from tests.golden_cases import CH
    mod = c["modality"]
    return adamml(groups=c["groups"], modality=mod, input_channels=[CH[m] for m in mod], num_segments=c["S"],
                  rng_policy=False, rng_threshold=0.5, causality_modeling=c.get("causality", "lstm"), num_classes=31,
                  depth=50, without_t_stride=False, dropout=0.0, pooling_method="max", fusion_point="logits",
                  unimodality_pretrained=[], learnable_lf_weights=True)


@pytest.mark.parametrize("name", ["adamml_rgb_sound", "adamml_rgb_sound_nolstm", "adamml_rgb_flow_rgbdiff", "adamml_4mod"])
def test_adamml(name):
    """AdaMML forward + backward in both freeze stages: decisions must equal the reference's (golden margins are
    > 0.25 by construction of the Gumbel seed, far above the policy-logit error), logits / gradients as above."""
    from tests.oracle_harness import case_gumbel
    c = CASES[name]
    gold = load_golden(name)
    train_modes = [m for m in c["modes"] if m != "eval"]
    emu = oracle_case(c, emulate_bf16=True, modes=train_modes, device="cpu", keep_grads=True)
    ref = oracle_case(c, emulate_bf16=False, modes=train_modes, device="cpu", keep_grads=True)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    model = build_adamml(c)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.to(DEV)
    xs, target = case_inputs(c)
    xs, target = [t.to(DEV) for t in xs], target.to(DEV)
    expo = case_gumbel(c).to(DEV)
    print(name, "golden min decision margin %.3f" % float(gold["min_decision_margin"]))
    for mode in train_modes:
        model.load_state_dict(sd)
        model.unfreeze_policy_net()
        model.unfreeze_main_net()
        if mode == "train_main":
            model.freeze_policy_net()
        else:
            model.freeze_main_net()
        model.train()
        model.zero_grad()
        logits, sel = model(xs, gumbel_exponential=expo)
        pl_err = rel_err(model.last_policy_logits.detach().cpu().numpy(), gold[mode + ".policy_logits"])
        print("  [%s] policy logits rel err vs golden %.4f" % (mode, pl_err))
        # reduced fixtures (B = 1-2, 64-96 px: 4-18 samples per BatchNorm channel in the deep layers) are ill-conditioned by
        # construction -- measured 2e-2 .. 0.35 here; the asserted bound for policy logits is in tests/test_parity_fullsize_gpu.py
        # (full size: 5e-2 measured, 1e-1 asserted; forced-forward replay 1e-6).  Here: plumbing-level sanity only.
        assert pl_err <= 0.6, pl_err
        assert np.array_equal(np.round(sel.detach().cpu().numpy()), np.round(gold[mode + ".decisions"])), "decisions differ"
        loss = F.cross_entropy(logits, target)
        if model.update_policy_net:
            from oracle import adamml_oracle as O
            cw = torch.tensor([1.0] * sel.shape[-1], device=DEV)
            loss = loss + O.policy_loss("blockdrop", sel, cw, torch.tensor(10.0, device=DEV), logits, target)
        loss.backward()
        check_against_emulation(model, emu, ref, mode, logits.detach().cpu().numpy())
        check_against_golden(model, gold, emu, mode, logits.detach().cpu().numpy(), name=name)


def test_eval_skipping_equals_masking():
    """Inference with decision-driven compaction (main nets only run the selected (segment, video) clips) returns exactly
    the logits of the compute-everything-then-mask forward of the reference (models/adamml.py:81-86)."""
    from adamml_amd import adamml
    torch.manual_seed(0)
    S, B = 3, 4
    model = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
                   causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.5, pooling_method="max",
                   fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True).to(DEV).eval()
    xs = [t.to(DEV) for t in synth.synth_inputs(["rgb", "sound"], B, S, 8, 64, seed=5)]
    expo = synth.synth_gumbel_exponential(S, 2, B, seed=11).to(DEV)
    with torch.no_grad():
        model.skip_unselected = False
        ref, dec_ref = model(xs, gumbel_exponential=expo)
        model.skip_unselected = True
        got, dec = model(xs, gumbel_exponential=expo)
    assert torch.equal(dec, dec_ref)
    st = model.last_skip_stats
    assert st["clips"] == S * B and all(0 <= n <= S * B for n in st["executed_per_modality"])
    assert st["executed_per_modality"] == [int(dec[:, :, m].sum().item()) for m in range(2)]
    print("  executed clips per modality:", st["executed_per_modality"], "of", st["clips"])
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (got - ref).abs().max().item()
