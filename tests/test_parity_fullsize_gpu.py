"""Whole-network parity at FULL size on the MI355X.  Forward figures (logits, policy logits, running statistics, head gradients) are
gated RELATIVE TO THE ORACLE'S bf16-STORAGE EMULATION of the same step (tests/golden/*_bf16emu.npz, tools/gen_golden_emu.py;
tests/parity_bounds.emu_gate) with the fixed numbers of rounds 4-5 printed beside them as regression figures; the forced-forward replay,
inference and reproducibility statements keep FIXED numbers (tests/parity_bounds.json).

Cases (tests/golden_cases.py, fixtures generated from the real reference by tools/gen_golden.py):
  resnet50_c1  BASELINE.json configs[0]: unimodal RGB ResNet-50, 8 frames, 224^2, b = 4 (models/resnet.py:195-223)
  adamml_c2    the configs[1] workload at B = 4 videos: RGB+Audio AdaMML, 5 segments, 224^2 / 256^2, both freeze stages
               (models/adamml.py:69-91)

Two comparators:
  (1) the fp32 reference golden.  Forward quantities are well conditioned at full size (>= 196 samples per BatchNorm channel)
      and are asserted directly: logits, policy logits, EVERY running statistic, num_batches_tracked, hard decisions, the
      classifier-head gradients.  Deep gradients are NOT comparable element-wise between an fp32 and ANY bf16-storage pipeline:
      rounding moves pre-activations across 0 (ReLU) and across each other (max-pool), every such flip changes the gradient of
      that element by 100 %, and the relative L2 distance grows like sqrt(fraction flipped) -- tools/conditioning_study.py shows
      the sqrt(eps) law on the oracle alone (a 1e-3 / 1e-4 / 1e-5 relative weight perturbation moves stem gradients by
      22 % / 7 % / 1.8 %).  Against the golden they are held to norm agreement only.
  (2) FORCED-FORWARD REPLAY (tests/oracle_harness.oracle_case_forced): the fp32 oracle re-runs the step with its conv outputs
      replaced by the tensors the HIP forward stored, so both take identical ReLU / max-pool decisions; what remains is the
      arithmetic of the backward pass (bf16-stored gradients vs fp32).  This is the tight whole-network gradient statement.

Bounds: ONE table, tests/parity_bounds.json -- every figure this file asserts has an entry {measured, bound = min(1.3 x measured,
stated tolerance of its category)}; `python tools/rebase_bounds.py` (GPU box) re-measures all of them in one run and rewrites the table
(tests/parity_bounds.py; the stated tolerances are held by a CPU test).  The policy-logit figures are those of the random-weight
MobileNetV2 stacks, which amplify any perturbation ~1.09x per layer: the oracle's own bf16-storage emulation sits at 5.0e-2 / 5.7e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES, CH, is_head  # noqa: E402
from tests.parity_bounds import check, emu_gate  # noqa: E402  (ONE table: tests/parity_bounds.json, re-based by tools/rebase_bounds.py; emulation-relative gates)
from tests.oracle_harness import (manifest, load_golden, case_inputs, case_gumbel, oracle_case_forced,  # noqa: E402
                                  calibrated_state)

DEV = "cuda"


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def build(c):
    if c["kind"] == "resnet":
        from adamml_amd.resnet import resnet
        return resnet(depth=50, num_classes=31, without_t_stride=False, groups=c["groups"], dropout=0.0, pooling_method="max",
                      input_channels=3, imagenet_pretrained=False)
    from adamml_amd import adamml
    mod = c["modality"]
    return adamml(groups=c["groups"], modality=mod, input_channels=[CH[m] for m in mod], num_segments=c["S"], rng_policy=False,
                  rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0,
                  pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)


def nets_of(model):
    return model.backbones() if hasattr(model, "backbones") else [model]


def hip_train_step(model, c, mode, sd, after_backward=None):
    """One train-mode forward + backward through the HIP path with the conv outputs captured (test hook NetRT.capture).
    after_backward: optional callable run between loss.backward() and the device synchronisation (data-parallel reduction)."""
    xs, target = case_inputs(c)
    model.load_state_dict(sd)
    model.to(DEV)
    cap = {}
    for n in nets_of(model):
        n.rt.capture = cap
    try:
        model.train()
        if c["kind"] == "adamml":
            model.unfreeze_policy_net()
            model.unfreeze_main_net()
            (model.freeze_policy_net if mode == "train_main" else model.freeze_main_net)()
            model.zero_grad()
            logits, sel = model([t.to(DEV) for t in xs], gumbel_exponential=case_gumbel(c).to(DEV))
            plog = model.last_policy_logits.detach().cpu()
        else:
            model.zero_grad()
            logits, sel, plog = model(xs.to(DEV)), None, None
        loss = F.cross_entropy(logits, target.to(DEV))
        if sel is not None and model.update_policy_net:
            from oracle import adamml_oracle as O
            loss = loss + O.policy_loss("blockdrop", sel, torch.ones(sel.shape[-1], device=DEV), torch.tensor(10.0, device=DEV), logits,
                                        target.to(DEV))
        loss.backward()
        if after_backward is not None:
            after_backward()
        torch.cuda.synchronize()
    finally:
        for n in nets_of(model):
            n.rt.capture = None
    pid = {id(p): k for k, p in model.named_parameters()}
    captured = {pid[i]: (torch.cat([t.cpu() for t in ys]) if len(ys) > 1 else ys[0].cpu()) for i, ys in cap.items()}
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return logits.detach().cpu(), (None if sel is None else sel.detach().cpu()), plog, grads, state, captured


def load_emu(name):
    """bf16-storage emulation of a full-size case by the ORACLE (tools/gen_golden_emu.py; same record layout as the golden)."""
    return load_golden(name + "_bf16emu")


def check_policy_logits(gold, emu, mode, plog, key):
    e_hf = rel_max(plog.numpy(), gold[mode + ".policy_logits"])
    e_he = rel_max(plog.numpy(), emu[mode + ".policy_logits"])
    e_ef = rel_max(emu[mode + ".policy_logits"], gold[mode + ".policy_logits"])
    emu_gate(key + ".plog", "plog", e_hf, e_he, e_ef, "policy logits, of scale")
    check(key + ".plog", e_hf, "policy logits vs fp32 reference, of scale", soft=True)


def check_forward_vs_golden(gold, emu, mode, logits, state, key, groups):
    """key: prefix of this case's entries in tests/parity_bounds.json (<key>.logits / .stats / .stats_p90).  Hard gates: relative to the
    oracle's bf16-storage emulation (tests/parity_bounds.emu_gate); the table entries are printed regression figures."""
    e = rel_max(logits.numpy(), gold[mode + ".logits"])
    emu_gate(key + ".logits", "logits", e, rel_max(logits.numpy(), emu[mode + ".logits"]), rel_max(emu[mode + ".logits"], gold[mode + ".logits"]),
             "logits, of scale")
    check(key + ".logits", e, "logits vs fp32 reference, of scale", soft=True)
    names = list(gold[mode + ".stats_full_names"])
    assert list(emu[mode + ".stats_full_names"]) == names
    flat, eflat, off, errs, errs_e, errs_ef = gold[mode + ".stats_full"], emu[mode + ".stats_full"], 0, {}, {}, {}
    for k in names:
        n = state[k].numel()
        g, q = flat[off:off + n].reshape(state[k].shape), eflat[off:off + n].reshape(state[k].shape)
        errs[k], errs_e[k], errs_ef[k] = rel_l2(state[k], g), rel_l2(state[k], q), rel_l2(q, g)
        off += n
    v, ve, vef = sorted(errs.values()), sorted(errs_e.values()), sorted(errs_ef.values())
    worst = max(errs, key=errs.get)
    i90 = int(0.9 * len(v))
    print("  [%s] %d running statistics vs fp32 reference: median %.2e p90 %.2e max %.2e (%s)" % (
        mode, len(v), v[len(v) // 2], v[i90], v[-1], worst))
    emu_gate(key + ".stats", "stats", v[-1], ve[-1], vef[-1], "running statistics, max rel L2 (%s)" % worst)
    emu_gate(key + ".stats_p90", "stats_p90", v[i90], ve[i90], vef[i90], "running statistics, p90")
    check(key + ".stats", v[-1], worst, soft=True)
    check(key + ".stats_p90", v[i90], soft=True)
    for k, t in state.items():
        if k.endswith("num_batches_tracked"):
            assert int(t) == groups, (k, int(t))                 # one momentum update per segment call (models/adamml.py:84-86)


def check_grads_vs_golden(gold, emu, mode, grads, key):
    """Head gradients in full (the worst one: gated relative to the emulation, <key>.head of the table printed); every other tensor by its
    norm (see the module docstring)."""
    worst = worst_e = worst_ef = 0.0
    for k, g in grads.items():
        if is_head(k) and (mode + ".grad." + k) in gold:
            e = rel_l2(g, gold[mode + ".grad." + k])
            print("  [%s] head gradient %-48s rel L2 vs fp32 reference %.2e" % (mode, k, e))
            worst = max(worst, e)
            worst_e = max(worst_e, rel_l2(g, emu[mode + ".grad." + k]))
            worst_ef = max(worst_ef, rel_l2(emu[mode + ".grad." + k], gold[mode + ".grad." + k]))
    cat = "head_policy" if mode == "train_policy" else "head"
    emu_gate(key + ".head", cat, worst, worst_e, worst_ef, "worst head gradient, rel L2")
    check(key + ".head", worst, "worst head gradient", soft=True)
    names = list(gold[mode + ".grad_names"])
    ref_l2 = dict(zip(names, gold[mode + ".grad_probe"][:, 1]))
    gmax = max(ref_l2.values())
    ratios = {k: float(g.double().norm()) / ref_l2[k] for k, g in grads.items() if ref_l2[k] >= 1e-4 * gmax}
    r = sorted(ratios.values())
    inside = sum(0.75 <= x <= 1.35 for x in r) / len(r)
    print("  [%s] gradient norms HIP / fp32 reference over %d tensors: min %.2f median %.2f max %.2f; %.0f %% within [0.75, 1.35]" % (
        mode, len(r), r[0], r[len(r) // 2], r[-1], 100 * inside))
    assert all(np.isfinite(x) for x in r)
    return inside


def check_replay(c, mode, logits, plog, grads, state, captured, top_prefixes, key):
    rep = oracle_case_forced(c, mode, captured)
    e = rel_max(logits.numpy(), rep["logits"].numpy())
    print("  [%s] forced-forward replay: logits %.2e" % (mode, e))
    assert e <= 2e-3, e
    if plog is not None:
        ep = rel_max(plog.numpy(), rep["policy_logits"].numpy())
        print("  [%s] forced-forward replay: policy logits %.2e" % (mode, ep))
        assert ep <= 2e-3, ep
    se = max(rel_l2(state[k], v) for k, v in rep["state"].items() if k.endswith(("running_mean", "running_var")))
    print("  [%s] forced-forward replay: running statistics max rel L2 %.2e" % (mode, se))
    # 1e-7 for every BatchNorm whose conv output is forced; the bottleneck conv3 of layers 1-2 is fused with its BatchNorm and
    # the residual add (runtime.conv_bn_add): its output is never stored, so it cannot be forced, and its statistics are those
    # of the UNROUNDED conv output (Gram matrix of the input) where the oracle takes them from the bf16-rounded one: <= 1e-3
    assert se <= 3e-3, se
    gmax = max(float(g.norm()) for g in rep["grads"].values())
    errs = {k: rel_l2(grads[k], g) for k, g in rep["grads"].items() if float(g.norm()) >= 1e-4 * gmax and k in grads}
    v = sorted(errs.values())
    worst = max(errs, key=errs.get)
    top = {k: e_ for k, e_ in errs.items() if k.startswith(top_prefixes)}
    print("  [%s] forced-forward replay: %d gradient tensors, rel L2 median %.3f p90 %.3f max %.3f (%s); %d tensors next to the "
          "heads: max %.3f" % (mode, len(v), v[len(v) // 2], v[int(0.9 * len(v))], v[-1], worst, len(top), max(top.values())))
    check(key + ".replay_top", max(top.values()), max(top, key=top.get))
    check(key + ".replay_p90", v[int(0.9 * len(v))])
    # every per-channel sum is order-fixed and exact across workgroups (csrc/common.h), so these are reproducible numbers: the bound holds
    # for EVERY tensor (round 3 exempted the two worst ones: with fp64 atomic statistics the deepest MobileNetV2 tensor read 0.12 .. 0.59
    # from run to run)
    check(key + ".replay_max", v[-1], worst)


def test_c1_resnet50_fullsize():
    c = CASES["resnet50_c1"]
    gold = load_golden("resnet50_c1")
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    assert list(model.state_dict().keys()) == list(sd.keys())
    print("resnet50_c1 (B=4, 8 frames, 224^2)")
    logits, _, _, grads, state, captured = hip_train_step(model, c, "train", sd)
    # bounds = 1.3 x measured (reproducible: order-fixed sums): logits 8.4e-3, statistics max 2.77e-3 / p90 1.65e-3, fc gradient 8.7e-3,
    # replay gradients p90 0.099 / max 0.180 (bn1.bias, ~100 bf16 gradient roundings below the loss) / next to the head 0.015
    emu = load_emu("resnet50_c1")
    check_forward_vs_golden(gold, emu, "train", logits, state, "c1.train", groups=1)
    inside = check_grads_vs_golden(gold, emu, "train", grads, "c1.train")
    assert inside >= 0.9
    check_replay(c, "train", logits, None, grads, state, captured, top_prefixes=("layer4.", "fc."), key="c1.train")
    # inference on calibrated running statistics (BatchNorm = fixed affine map)
    xs, _ = case_inputs(c)
    model.load_state_dict(calibrated_state(c, sd, xs))
    model.eval()
    with torch.no_grad():
        y = model(xs.to(DEV))
    e = rel_max(y.cpu().numpy(), gold["eval_cal.logits"])
    check("c1.eval.logits", e, "inference on calibrated statistics")


@pytest.mark.parametrize("mode", ["train_main", "train_policy"])
def test_c2_adamml_fullsize(mode):
    c = CASES["adamml_c2"]
    gold = load_golden("adamml_c2")
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    assert list(model.state_dict().keys()) == list(sd.keys())
    print("adamml_c2 (B=4, S=5, 224^2 / 256^2), golden min decision margin %.3f" % float(gold["min_decision_margin"]))
    logits, sel, plog, grads, state, captured = hip_train_step(model, c, mode, sd)
    assert np.array_equal(np.round(sel.numpy()), np.round(gold[mode + ".decisions"])), "decisions differ from the reference"
    # forward figures: gated relative to the oracle's bf16-storage emulation (tests/parity_bounds.emu_gate: the emulation itself sits at
    # logits 4.04e-2 / policy logits 7.4e-2 / statistics 1.7e-2 from the fp32 golden); the table figures (round 5: policy logits 4.7e-2,
    # logits 3.99e-2, statistics max 1.88e-2 / p90 4.7e-3) are printed beside them
    emu = load_emu("adamml_c2")
    check_policy_logits(gold, emu, mode, plog, "c2." + mode)
    check_forward_vs_golden(gold, emu, mode, logits, state, "c2." + mode, groups=c["S"])
    # main stage: the heads sit on ResNet / MobileNetV2 features (4.8e-2 worst, the sound classifier); policy stage: the head
    # gradients are driven by d(loss)/d(decisions), a difference of class logits of the gated main nets (0.30 worst, fcs.1)
    inside = check_grads_vs_golden(gold, emu, mode, grads, "c2." + mode)
    assert inside >= 0.9
    top = ("main_net.nets.0.layer4.", "main_net.nets.0.fc.", "main_net.nets.1.features.17.", "main_net.nets.1.features.18.",
           "main_net.nets.1.classifier.", "main_net.lf_weights") if mode == "train_main" else \
          ("policy_net.fcs.", "policy_net.lstm.", "policy_net.joint_net.joint.")
    # replay gradients measured: main stage p90 0.089 / max 0.110 / heads 0.030; policy stage p90 0.055 / max 0.185 (policy rgb stem BatchNorm)
    check_replay(c, mode, logits, plog, grads, state, captured, top_prefixes=top, key="c2." + mode)


# BASELINE.json configs[3] / configs[4] at full size (B = 2 videos, S = 5, 224^2 / 256^2).  What these add to C2: the policy / main
# modality-order quirk (models/adamml.py:143-146,85-86 -- decisions index (rgb, [sound,] rgbdiff) while the main nets take
# (rgb, [sound,] flow)), the 10-channel ResNet stem (models/resnet.py:138) and the 15-channel policy stem (models/policy_net.py:195-200),
# two ResNet-50 main nets side by side.  Bounds: tests/parity_bounds.json (c4.* / c5.*).
SHORT = {"adamml_c4": "c4", "adamml_c5": "c5"}


@pytest.mark.parametrize("name,mode", [("adamml_c4", "train_main"), ("adamml_c5", "train_main"), ("adamml_c5", "train_policy")])
def test_c4_c5_adamml_fullsize(name, mode):
    c = CASES[name]
    gold = load_golden(name)
    key = "%s.%s" % (SHORT[name], mode)
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    assert list(model.state_dict().keys()) == list(sd.keys())
    print("%s (%s, B=%d, S=%d, 224^2), golden min decision margin %.3f" % (name, "+".join(c["modality"]), c["B"], c["S"],
                                                                          float(gold["min_decision_margin"])))
    logits, sel, plog, grads, state, _ = hip_train_step(model, c, mode, sd)
    # decisions [B, S, M'] over the POLICY modalities (rgbdiff stands in for flow): the same hard decisions as the reference
    assert sel.shape == gold[mode + ".decisions"].shape, (sel.shape, gold[mode + ".decisions"].shape)
    assert np.array_equal(np.round(sel.numpy()), np.round(gold[mode + ".decisions"])), "decisions differ from the reference"
    emu = load_emu(name)
    check_policy_logits(gold, emu, mode, plog, key)
    check_forward_vs_golden(gold, emu, mode, logits, state, key, groups=c["S"])
    inside = check_grads_vs_golden(gold, emu, mode, grads, key)
    assert inside >= 0.9


@pytest.mark.parametrize("name", ["adamml_c4", "adamml_c5"])
def test_c4_c5_adamml_fullsize_inference(name):
    c = CASES[name]
    gold = load_golden(name)
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    xs, _ = case_inputs(c)
    model.load_state_dict(calibrated_state(c, sd, xs))
    model.to(DEV).eval()
    with torch.no_grad():
        logits, sel = model([t.to(DEV) for t in xs], gumbel_exponential=case_gumbel(c).to(DEV))
    assert np.array_equal(np.round(sel.cpu().numpy()), np.round(gold["eval_cal.decisions"])), "decisions differ from the reference"
    ep = rel_max(model.last_policy_logits.cpu().numpy(), gold["eval_cal.policy_logits"])
    e = rel_max(logits.cpu().numpy(), gold["eval_cal.logits"])
    print("%s [eval_cal, decision-driven skipping on] policy logits %.4f, logits %.4f of scale vs fp32 reference" % (name, ep, e))
    check(SHORT[name] + ".eval.plog", ep)
    check(SHORT[name] + ".eval.logits", e)


def test_c2_adamml_fullsize_inference():
    c = CASES["adamml_c2"]
    gold = load_golden("adamml_c2")
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    xs, _ = case_inputs(c)
    model.load_state_dict(calibrated_state(c, sd, xs))
    model.to(DEV).eval()
    with torch.no_grad():
        logits, sel = model([t.to(DEV) for t in xs], gumbel_exponential=case_gumbel(c).to(DEV))
    assert np.array_equal(np.round(sel.cpu().numpy()), np.round(gold["eval_cal.decisions"])), "decisions differ from the reference"
    ep = rel_max(model.last_policy_logits.cpu().numpy(), gold["eval_cal.policy_logits"])
    e = rel_max(logits.cpu().numpy(), gold["eval_cal.logits"])
    print("adamml_c2 [eval_cal, decision-driven skipping on] policy logits %.4f, logits %.4f of scale vs fp32 reference" % (ep, e))
    check("c2.eval.plog", ep)
    check("c2.eval.logits", e)


def _snapshot(logits, grads, state):
    return {"logits": logits, **{"g." + k: v for k, v in grads.items()}, **{"s." + k: v for k, v in state.items()}}


@pytest.mark.parametrize("name,mode", [("resnet50_c1", "train"), ("adamml_c2", "train_main"), ("adamml_c2", "train_policy")])
def test_two_runs_are_bit_identical_and_correct(name, mode):
    """Every per-channel statistic / BatchNorm-backward sum is order-fixed inside a workgroup and accumulated in exact integer bins
    across workgroups (csrc/common.h: the only mode).  Two runs of the same full-size step must therefore agree BIT FOR BIT in logits,
    every gradient and every running statistic (with the fp64 atomics of round 3 the order of arrival changed the last bits of the sums
    and bf16 rounding amplified that: run to run, ResNet-50 gradients 0.7 % median, the MobileNetV2 stacks up to 40 %); the same
    forced-forward replay bounds hold."""
    from adamml_amd import hip
    c = CASES[name]
    model = build(c)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    assert hip.deterministic()
    logits, sel, plog, grads, state, captured = hip_train_step(model, c, mode, sd)
    a = _snapshot(logits, grads, state)
    l2, _, _, g2, s2, _ = hip_train_step(model, c, mode, sd)
    b = _snapshot(l2, g2, s2)
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    print("%s [%s] two runs: %d tensors compared, %d differ between two runs" % (name, mode, len(a), len(diff)))
    assert not diff, diff[:8]
    if name == "resnet50_c1":
        check_replay(c, mode, logits, None, grads, state, captured, top_prefixes=("layer4.", "fc."), key="c1.train")
    else:
        top = ("main_net.nets.0.layer4.", "main_net.nets.0.fc.", "main_net.nets.1.features.17.", "main_net.nets.1.features.18.",
               "main_net.nets.1.classifier.", "main_net.lf_weights") if mode == "train_main" else \
              ("policy_net.fcs.", "policy_net.lstm.", "policy_net.joint_net.joint.")
        check_replay(c, mode, logits, plog, grads, state, captured, top_prefixes=top, key="c2." + mode)      # (the SAME table entries as test_c2_adamml_fullsize)
