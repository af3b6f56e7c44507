#!/usr/bin/env python3
"""Whole-network parity study at FULL size (GPU box): where do HIP, the oracle's bf16-storage emulation and the fp32
oracle sit relative to each other, and how large is the run-to-run spread of the HIP path itself?

    python tools/parity_study.py resnet 4 224          # C1: unimodal ResNet-50, B videos, size
    python tools/parity_study.py adamml 4 224 5        # C2: RGB+Audio AdaMML, B videos, size, S segments

Prints logits / running-statistic / per-tensor gradient distances.  Test infrastructure (imports the oracle)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from adamml_amd import synth  # noqa: E402
from oracle import adamml_oracle as O  # noqa: E402


def rl2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return F.cosine_similarity(a, b, dim=0).item()


def summarize(tag, errs):
    v = sorted(errs.values())
    n = len(v)
    print("  %-34s n=%d median %.4f p90 %.4f max %.4f | <=5e-2: %.0f%%  <=1e-1: %.0f%%" % (
        tag, n, v[n // 2], v[int(n * 0.9)], v[-1], 100.0 * sum(e <= 5e-2 for e in v) / n, 100.0 * sum(e <= 1e-1 for e in v) / n))


def main():
    kind, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    stage = sys.argv[5] if len(sys.argv) > 5 else "main"
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dev = "cuda"
    if kind == "resnet":
        from adamml_amd.resnet import resnet
        model = resnet(depth=50, num_classes=31, without_t_stride=False, groups=8, dropout=0.0, pooling_method="max",
                       input_channels=3, imagenet_pretrained=False)
        mod = ["rgb"]
    else:
        from adamml_amd import adamml
        mod = ["rgb", "sound"]
        model = adamml(groups=8, modality=mod, input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
                       causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0, pooling_method="max",
                       fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    sd0 = synth.synth_state_dict(model.state_dict(), seed=1234)
    xs = synth.synth_inputs(mod, B, S, 8, size, 256 if size == 224 else size, seed=42)
    tgt = synth.synth_labels(B, 31, seed=42)
    expo = synth.synth_gumbel_exponential(S, 2, B, seed=7) if kind == "adamml" else None
    pref = ("",) if kind == "resnet" else (("main_net.",) if stage == "main" else ("policy_net.",))

    def run_oracle(q):
        O.QUANT = O.bf16_straight_through if q else None
        sd = O.make_leaf_state(sd0, pref)
        t0 = time.time()
        try:
            if kind == "resnet":
                y = O.resnet_forward(sd, "", xs[0], 8, 50, "max", False, 0.0, True)
                pl = None
            else:
                y, sel, pl = O.adamml_forward(sd, xs, mod, S, 8, 50, 5.0, expo, "lstm", "max", False, 0.0, True)
            loss = F.cross_entropy(y, tgt)
            if kind == "adamml" and stage != "main":
                loss = loss + O.policy_loss("blockdrop", sel, torch.ones(2), torch.tensor(10.0), y, tgt)
            loss.backward()
        finally:
            O.QUANT = None
        print("  oracle(%s) %.1f s" % ("bf16 emulation" if q else "fp32", time.time() - t0))
        return y.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}, {k: v.detach() for k, v in sd.items()}, pl

    def run_oracle_forced(cap):
        """fp32 oracle whose conv outputs are REPLACED by the bf16 tensors the HIP forward stored (same ReLU / max-pool decisions)."""
        sd = O.make_leaf_state(sd0, pref)
        names = {id(v): k for k, v in sd.items()}
        used = {}

        def hook(w, y):
            k = names[id(w)]
            i = used.get(k, 0)
            used[k] = i + 1
            ys = cap[k]                                  # NHWC bf16 [G*N, OH, OW, C] (groups = segments, group-major)
            n = y.shape[0]
            t = ys[i * n:(i + 1) * n].permute(0, 3, 1, 2).float()
            assert t.shape == y.shape, (k, t.shape, y.shape)
            return y + (t - y).detach()
        O.CONV_HOOK = hook
        try:
            if kind == "resnet":
                y = O.resnet_forward(sd, "", xs[0], 8, 50, "max", False, 0.0, True)
                pl = None
            else:
                y, sel, pl = O.adamml_forward(sd, xs, mod, S, 8, 50, 5.0, expo, "lstm", "max", False, 0.0, True)
            loss = F.cross_entropy(y, tgt)
            if kind == "adamml" and stage != "main":
                loss = loss + O.policy_loss("blockdrop", sel, torch.ones(2), torch.tensor(10.0), y, tgt)
            loss.backward()
        finally:
            O.CONV_HOOK = None
        return y.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}, {k: v.detach() for k, v in sd.items()}, pl

    def run_hip(capture=None):
        if capture is not None:
            nets = model.backbones() if hasattr(model, "backbones") else [model]
            for n_ in nets:
                n_.rt.capture = capture
        model.load_state_dict(sd0)
        model.to(dev)
        if kind == "adamml":
            model.unfreeze_policy_net()
            model.unfreeze_main_net()
            (model.freeze_policy_net if stage == "main" else model.freeze_main_net)()
        model.train()
        model.zero_grad()
        if kind == "resnet":
            y = model(xs[0].to(dev))
            pl = None
        else:
            y, sel = model([t.to(dev) for t in xs], gumbel_exponential=expo.to(dev))
            pl = model.last_policy_logits.detach().cpu()
        loss = F.cross_entropy(y, tgt.to(dev))
        if kind == "adamml" and stage != "main":
            loss = loss + O.policy_loss("blockdrop", sel, torch.ones(2, device=dev), torch.tensor(10.0, device=dev), y, tgt.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None and p.requires_grad}
        st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        return y.detach().cpu(), g, st, pl

    cap = {}
    y_h, g_h, s_h, pl_h = run_hip(cap)
    pid = {id(p): k for k, p in model.named_parameters()}
    cap_named = {pid[i]: torch.cat([t.cpu() for t in ys]) if len(ys) > 1 else ys[0].cpu() for i, ys in cap.items()}
    cap.clear()
    for n_ in (model.backbones() if hasattr(model, "backbones") else [model]):
        n_.rt.capture = None
    y_h2, g_h2, s_h2, _ = run_hip()
    y_r, g_r, s_r, pl_r = run_oracle_forced(cap_named)
    y_f, g_f, s_f, pl_f = run_oracle(False)
    y_e, g_e, s_e, pl_e = run_oracle(True)
    sc = y_f.abs().max().item()
    print("%s B=%d size=%d S=%d stage=%s" % (kind, B, size, S, stage))
    print("  logits scale %.3f: |HIP-fp32| %.4f  |HIP-emu| %.4f  |emu-fp32| %.4f  |HIP-HIP'| %.2e" % (
        sc, (y_h - y_f).abs().max().item() / sc, (y_h - y_e).abs().max().item() / sc, (y_e - y_f).abs().max().item() / sc,
        (y_h - y_h2).abs().max().item() / sc))
    if pl_h is not None:
        ps = pl_f.abs().max().item()
        print("  policy logits scale %.3f: |HIP-fp32| %.4f |HIP-emu| %.4f |emu-fp32| %.4f" % (
            ps, (pl_h - pl_f.detach()).abs().max().item() / ps, (pl_h - pl_e.detach()).abs().max().item() / ps,
            (pl_e.detach() - pl_f.detach()).abs().max().item() / ps))
    stat_keys = [k for k in s_f if k.endswith(("running_mean", "running_var"))]
    summarize("running stats relL2(HIP,fp32)", {k: rl2(s_h[k], s_f[k]) for k in stat_keys})
    summarize("running stats relL2(HIP,emu)", {k: rl2(s_h[k], s_e[k]) for k in stat_keys})
    summarize("running stats relL2(emu,fp32)", {k: rl2(s_e[k], s_f[k]) for k in stat_keys})
    gmax = max(g.norm().item() for g in g_f.values())
    keys = [k for k in g_f if g_f[k].norm().item() >= 1e-4 * gmax and k in g_h]
    summarize("grads relL2(HIP,fp32)", {k: rl2(g_h[k], g_f[k]) for k in keys})
    summarize("grads relL2(HIP,emu)", {k: rl2(g_h[k], g_e[k]) for k in keys})
    summarize("grads relL2(emu,fp32)", {k: rl2(g_e[k], g_f[k]) for k in keys})
    summarize("grads relL2(HIP,HIP')", {k: rl2(g_h[k], g_h2[k]) for k in keys})
    print("  forced-forward replay: logits |HIP-replay| %.2e" % ((y_h - y_r).abs().max().item() / sc))
    summarize("running stats relL2(HIP,replay)", {k: rl2(s_h[k], s_r[k]) for k in stat_keys})
    summarize("grads relL2(HIP,replay)", {k: rl2(g_h[k], g_r[k]) for k in keys})
    summarize("grads 1-cos(HIP,fp32)", {k: 1 - cos(g_h[k], g_f[k]) for k in keys})
    order = [k for k in sd0 if k in keys]
    print("  per tensor (reverse network order): relL2 HIP-fp32 | HIP-emu | emu-fp32 | HIP-HIP'")
    step = max(1, len(order) // 60)
    for k in order[::-1][::step]:
        print("    %-58s %.4f %.4f %.4f %.2e | replay %.4f" % (k, rl2(g_h[k], g_f[k]), rl2(g_h[k], g_e[k]), rl2(g_e[k], g_f[k]), rl2(g_h[k], g_h2[k]),
                                                          rl2(g_h[k], g_r[k])))


if __name__ == "__main__":
    main()
