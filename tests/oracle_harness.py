"""Runs a golden case through the ORACLE (CPU restatement) and returns the same
record layout tools/gen_golden.py stores for the real reference."""
import gzip
import json
import os
import numpy as np
import torch
import torch.nn.functional as F

from adamml_amd import synth
from oracle import adamml_oracle as O
from tests.golden_cases import CASES, CH, grad_probe, stat_probe, is_head

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def arch_key(c):
    return "%s:%s:%s" % (c["kind"], "+".join(c["modality"]), c.get("causality", "lstm"))


def manifest(c):
    with gzip.open(os.path.join(GOLDEN_DIR, "state_manifest.json.gz"), "rt") as f:
        man = json.load(f)[arch_key(c)]
    return {k: torch.empty(shp, dtype=getattr(torch, dt.split(".")[1])) for k, (shp, dt) in man.items()}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def case_inputs(c):
    B, S = c["B"], c.get("S", 1)
    if c["kind"] == "adamml":
        xs = synth.synth_inputs(c["modality"], B, S, c["groups"], c["size"], c["sound_size"], seed=42)
    elif c["kind"] == "resnet":
        xs = synth.synth_inputs(c["modality"], B, 1, c["groups"], c["size"], seed=42)[0]
    else:
        xs = synth.synth_inputs(["sound"], B, 1, sound_size=c["sound_size"], seed=42)[0]
    return xs, synth.synth_labels(B, 31, seed=42)


def case_gumbel(c):
    """Exponential(1) draws [S, M*B, 2] of an AdaMML case; the seed was chosen by tools/gen_golden.py (robust margins)."""
    name = [k for k, v in CASES.items() if v is c][0]
    seed = int(load_golden(name)["gumbel_seed"])
    return synth.synth_gumbel_exponential(c["S"], num_policy_modality(c), c["B"], seed=seed)


def num_policy_modality(c):
    mod = c["modality"]
    return len(mod) - 1 if ("rgbdiff" in mod and "flow" in mod) else len(mod)


def round_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def oracle_case(c, emulate_bf16=False, modes=None, device="cpu", keep_grads=False):
    """emulate_bf16=True runs the oracle with its bf16-STORAGE emulation hook (oracle.QUANT): every tensor the
    HIP path keeps in bf16 (dense-conv operands, conv outputs, residual / pool outputs) is rounded with a
    straight-through estimator, all arithmetic stays fp32.  The HIP path must match that run tightly; the distance
    between that run and the exact fp32 run is the intrinsic bf16 tolerance of the network."""
    kind = c["kind"]
    sd0 = synth.synth_state_dict(manifest(c), seed=1234)
    xs, target = case_inputs(c)
    sd0 = {k: v.to(device) for k, v in sd0.items()}
    xs = [t.to(device) for t in xs] if isinstance(xs, list) else xs.to(device)
    target = target.to(device)
    old_q = O.QUANT
    O.QUANT = O.bf16_straight_through if emulate_bf16 else None
    try:
        return _oracle_case(c, kind, sd0, xs, target, modes, keep_grads, device)
    finally:
        O.QUANT = old_q


def _oracle_case(c, kind, sd0, xs, target, modes, keep_grads, device):
    B, S = c["B"], c.get("S", 1)
    out = {}
    for mode in (modes or c["modes"]):
        training = mode not in ("eval", "eval_cal")
        if kind == "adamml":
            pref = {"eval": (), "eval_cal": (), "train": ("main_net.", "policy_net."), "train_main": ("main_net.",),
                    "train_policy": ("policy_net.",)}[mode]
        else:
            pref = ("",) if training else ()
        sd = O.make_leaf_state(sd0 if mode != "eval_cal" else calibrated_state(c, sd0, xs, device), pref)
        ctx = torch.enable_grad() if training else torch.no_grad()
        with ctx:
            if kind == "resnet":
                logits = O.resnet_forward(sd, "", xs, c["groups"], 50, c.get("pooling", "max"), False, 0.0, training)
            elif kind == "sound":
                logits = O.sound_mbv2_forward(sd, "", xs, 0.0, training)
            else:
                expo = case_gumbel(c).to(device)
                logits, sel, plog = O.adamml_forward(sd, xs, c["modality"], S, c["groups"], 50, c.get("tau", 5.0), expo,
                                                     c.get("causality", "lstm"), c.get("pooling", "max"), False, 0.0,
                                                     training)
                out[mode + ".decisions"] = sel.detach().cpu().numpy()
                cw = torch.tensor(c.get("cost_weights", [1.0] * sel.shape[-1]), device=device)
                gam = torch.tensor(10.0, device=device)
                pl_b = O.policy_loss("blockdrop", sel, cw, gam, logits, target)
                pl_m = O.policy_loss("mean", sel, cw, gam, logits, target)
                out[mode + ".policy_loss_blockdrop"] = pl_b.detach().cpu().numpy()
                out[mode + ".policy_loss_mean"] = pl_m.detach().cpu().numpy()
                out[mode + ".policy_logits"] = plog.detach().cpu().numpy()
            out[mode + ".logits"] = logits.detach().cpu().numpy()
            ce = F.cross_entropy(logits, target)
            out[mode + ".ce"] = ce.detach().cpu().numpy()
            if training:
                loss = ce
                if kind == "adamml" and mode in ("train", "train_policy"):
                    loss = loss + pl_b
                loss.backward()
                gp = {k: grad_probe(k, v.grad) for k, v in sd.items() if v.requires_grad and v.grad is not None}
                out[mode + ".grad_names"] = np.array(sorted(gp.keys()))
                out[mode + ".grad_probe"] = np.stack([gp[k] for k in sorted(gp.keys())])
                st = {k: stat_probe(v) for k, v in sd.items()
                      if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
                out[mode + ".stat_names"] = np.array(sorted(st.keys()))
                out[mode + ".stat_probe"] = np.stack([st[k] for k in sorted(st.keys())])
                if c.get("full"):
                    keys = [k for k in sorted(sd.keys()) if k.endswith(("running_mean", "running_var"))]
                    out[mode + ".stats_full_names"] = np.array(keys)
                    out[mode + ".stats_full"] = np.concatenate([sd[k].detach().cpu().numpy().reshape(-1) for k in keys]).astype(np.float32)
                    for k, v in sd.items():
                        if v.requires_grad and v.grad is not None and is_head(k):
                            out[mode + ".grad." + k] = v.grad.detach().cpu().numpy().astype(np.float32)
                if keep_grads:
                    out[mode + ".grads"] = {k: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None}
                    out[mode + ".state"] = {k: v.detach() for k, v in sd.items()}
    if kind == "adamml":
        both = "rgbdiff" in c["modality"] and "flow" in c["modality"]
        p_mod = [m for m in c["modality"] if not (both and m == "flow")]
        m_mod = [m for m in c["modality"] if not (both and m == "rgbdiff")]
        p_x, m_x = O.data_layer(xs, c["modality"], p_mod, m_mod, S, c["groups"])
        out["eval.p_x_probe"] = np.stack([stat_probe(t) for t in p_x])
        out["eval.m_x_probe"] = np.stack([stat_probe(t) for t in m_x])
    return out


def calibrated_state(c, sd0, xs, device="cpu"):
    """"eval_cal" state: running statistics := batch statistics of the case's input, by one train-mode pass with BatchNorm
    momentum 1 (what tools/gen_golden.py does with the reference's modules); num_batches_tracked back to 0."""
    cal = {k: v.clone() for k, v in sd0.items()}
    old = O.BN_MOMENTUM
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            if c["kind"] == "resnet":
                O.resnet_forward(cal, "", xs, c["groups"], 50, c.get("pooling", "max"), False, 0.0, True)
            elif c["kind"] == "sound":
                O.sound_mbv2_forward(cal, "", xs, 0.0, True)
            else:
                O.adamml_forward(cal, xs, c["modality"], c["S"], c["groups"], 50, c.get("tau", 5.0), case_gumbel(c).to(device),
                                 c.get("causality", "lstm"), c.get("pooling", "max"), False, 0.0, True)
    finally:
        O.BN_MOMENTUM = old
    for k in cal:
        if k.endswith("num_batches_tracked"):
            cal[k].zero_()
    return cal


def compare_records(got, ref, rtol=1e-4, atol=1e-5, grad_rtol=2e-3, skip=()):
    """Assert every golden entry is reproduced.  Gradient probes use a norm-relative bound."""
    for k, v in ref.items():
        if k in skip or k in ("n_state", "gumbel_seed", "min_decision_margin"):
            continue
        if k.endswith("policy_logits") and k not in got:
            continue
        assert k in got, "missing " + k
        g = got[k]
        if k.endswith("_names"):
            assert list(g) == list(v), k
        elif k.endswith("grad_probe"):
            # columns: sum, l2, 4 samples -> compare l2 relatively, others against l2 scale
            l2 = np.maximum(v[:, 1:2], 1e-5 * v[:, 1].max())   # analytically-zero grads are rounding noise
            err = np.abs(g - v) / l2
            # the 'sum' column cancels over many elements: allow 5x the bound there
            err[:, 0] /= 5.0
            assert err.max() < grad_rtol, "%s max rel err %g (row %d)" % (k, err.max(), int(err.max(1).argmax()))
        elif ".grad." in k or k.endswith("stats_full"):
            err = np.linalg.norm((g - v).ravel().astype(np.float64)) / (np.linalg.norm(v.ravel().astype(np.float64)) + 1e-30)
            assert err < grad_rtol, "%s rel L2 %g" % (k, err)
        elif k.endswith("decisions"):
            assert np.array_equal(np.round(g), np.round(v)), k
            np.testing.assert_allclose(g, v, atol=1e-5, err_msg=k)
        else:
            np.testing.assert_allclose(g, v, rtol=rtol, atol=atol, err_msg=k)


def oracle_case_forced(c, mode, captured, device="cpu"):
    """FORCED-FORWARD REPLAY: the fp32 oracle runs the case's train step with every conv output REPLACED (value only, the
    autograd graph stays) by the bf16 tensor the HIP forward stored for that conv -- `captured`: parameter name -> NHWC
    tensor [G*N, OH, OW, C] (G segment groups, group-major).  Both pipelines then take identical ReLU / ReLU6 / max-pool
    decisions, so their backward passes differ only by arithmetic (fp32 here, bf16-stored gradients there) and can be
    compared tightly; without forcing, any two bf16 pipelines decorrelate through rounding-boundary flips.
    Returns {"logits", "policy_logits", "grads", "state"}."""
    kind = c["kind"]
    sd0 = synth.synth_state_dict(manifest(c), seed=1234)
    xs, target = case_inputs(c)
    if kind == "adamml":
        pref = {"train": ("main_net.", "policy_net."), "train_main": ("main_net.",), "train_policy": ("policy_net.",)}[mode]
    else:
        pref = ("",)
    sd = O.make_leaf_state(sd0, pref)
    names = {id(v): k for k, v in sd.items()}
    used = {}

    def hook(w, y):
        k = names[id(w)]
        if k not in captured:                # the HIP path never stored this conv's output (conv3 + BatchNorm + add fused in one kernel)
            return y
        i = used.get(k, 0)
        used[k] = i + 1
        n = y.shape[0]
        t = captured[k][i * n:(i + 1) * n].permute(0, 3, 1, 2).float()
        assert t.shape == y.shape, (k, tuple(t.shape), tuple(y.shape))
        return y + (t - y).detach()
    O.CONV_HOOK = hook
    plog = None
    try:
        if kind == "resnet":
            logits = O.resnet_forward(sd, "", xs, c["groups"], 50, c.get("pooling", "max"), False, 0.0, True)
        elif kind == "sound":
            logits = O.sound_mbv2_forward(sd, "", xs, 0.0, True)
        else:
            logits, sel, plog = O.adamml_forward(sd, xs, c["modality"], c["S"], c["groups"], 50, c.get("tau", 5.0), case_gumbel(c),
                                                 c.get("causality", "lstm"), c.get("pooling", "max"), False, 0.0, True)
        loss = F.cross_entropy(logits, target)
        if kind == "adamml" and mode in ("train", "train_policy"):
            loss = loss + O.policy_loss("blockdrop", sel, torch.ones(sel.shape[-1]), torch.tensor(10.0), logits, target)
        loss.backward()
    finally:
        O.CONV_HOOK = None
    return {"logits": logits.detach(), "policy_logits": None if plog is None else plog.detach(),
            "grads": {k: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None},
            "state": {k: v.detach() for k, v in sd.items()}}
