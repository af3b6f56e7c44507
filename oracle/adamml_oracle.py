"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement of the AdaMML hot path as
plain functions over a flat ``state_dict`` (reference parameter names), executed
with torch CPU operators.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
``adamml_amd`` never does (it fails loudly when the HIP library is missing).

Parity pinning: every function here is checked against golden vectors captured
from the real reference (IBM/AdaMML @ /root/reference, torch 2.10.0 CPU fp32) by
``tools/gen_golden.py``; fixtures live in ``tests/golden/`` and
``tests/test_oracle_golden.py`` asserts them (rtol 1e-4 / atol 1e-5 on logits).
The reference itself holds no tests or golden vectors, so
the pin is "reference code run here", not a reference-owned fixture.

Each function cites the reference file:line whose behaviour it restates.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default used everywhere in the reference
BN_MOMENTUM = 0.1

RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}  # models/resnet.py:124-129
# (expand t, channels c, repeats n, stride s)  -- models/sound_mobilenet_v2.py:101-110, models/policy_net.py:102-111
MBV2_CFG = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


# ----------------------------------------------------------------------------- bf16-storage emulation hook
# QUANT is None for the exact fp32 restatement (the one pinned against the reference goldens).  The parity tests
# set it to a straight-through bf16 rounding to emulate WHERE the HIP path stores bf16 (dense-conv operands, every
# conv output, residual / pooling outputs) while keeping all arithmetic fp32: the HIP path must match that emulation
# tightly, and the emulation's own distance from the fp32 result is the stated, intrinsic bf16 tolerance.
QUANT = None
# CONV_HOOK(w, y) -> y' (or None): lets a test substitute the value of a conv output while keeping its autograd graph.  The
# forced-forward replay test (tests/test_parity_fullsize_gpu.py) injects the conv outputs the HIP path actually stored, so
# that both pipelines take identical ReLU / max-pool decisions and their backward passes can be compared tightly.
CONV_HOOK = None


def bf16_straight_through(x):
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def _q(x):
    return x if QUANT is None else QUANT(x)


def conv(x, w, stride=1, padding=0, groups=1):
    """nn.Conv2d(bias=False).  Emulation: dense convs read bf16 activations and bf16 weights; depthwise convs
    read the fp32-evaluated activation and fp32 weights -- and so does the one-channel spectrogram stem of the MobileNetV2s
    (round 3: the HIP stem reads the caller's fp32 tensor and the fp32 weights, adamml_conv_stem1_fwd); every conv output
    is stored in bf16."""
    if groups == 1 and x.shape[1] > 1:
        y = F.conv2d(_q(x), _q(w), stride=stride, padding=padding)
    else:
        y = F.conv2d(x, w, stride=stride, padding=padding, groups=groups)
    if CONV_HOOK is not None:
        y = CONV_HOOK(w, y)
    return _q(y)


# ----------------------------------------------------------------------------- primitives
def batchnorm(sd, p, x, training):
    """nn.BatchNorm2d forward incl. running-stat side effects (train mode updates
    running_mean/var with momentum 0.1 and unbiased variance; num_batches_tracked += 1)."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, w, b, training, BN_MOMENTUM, BN_EPS)


def temporal_pool(x, frames, mode):
    """models/common.py:4-33: 3-tap stride-2 pool over the frame axis, pad 1
    (max: -inf pad; avg: zeros counted, always /3).  x: [N*T, C, H, W] -> [N*T_out, C, H, W]."""
    nt, c, h, w = x.shape
    n = nt // frames
    v = x.reshape(n, frames, c, h, w)
    t_out = (frames + 2 - 3) // 2 + 1
    outs = []
    for t in range(t_out):
        taps = [v[:, i] for i in (2 * t - 1, 2 * t, 2 * t + 1) if 0 <= i < frames]
        if mode == "max":
            # nn.MaxPool3d keeps the FIRST maximum of the window (strict > in scan order) and routes the whole gradient
            # to it; torch.max(dim) returns the first maximal index too (torch.maximum would split the gradient on ties,
            # which do occur once activations are stored in bf16)
            r = torch.stack(taps, 0).max(0)[0]
        elif mode == "avg":
            r = taps[0]
            for q in taps[1:]:
                r = r + q
            r = r / 3.0
        else:
            raise ValueError("only support avg or max")
        outs.append(r)
    return torch.stack(outs, dim=1).reshape(n * t_out, c, h, w)


def dropout(x, p, training, mask=None):
    if mask is not None:          # explicit keep-mask (already scaled by 1/(1-p)) for parity runs
        return x * mask
    return F.dropout(x, p, training)


# ----------------------------------------------------------------------------- ResNet (A5)
def _bottleneck(sd, p, x, stride, has_down, training):
    """models/resnet.py:94-113."""
    out = conv(x, sd[p + ".conv1.weight"])
    out = F.relu(batchnorm(sd, p + ".bn1", out, training))
    out = conv(out, sd[p + ".conv2.weight"], stride=stride, padding=1)
    out = F.relu(batchnorm(sd, p + ".bn2", out, training))
    out = conv(out, sd[p + ".conv3.weight"])
    out = batchnorm(sd, p + ".bn3", out, training)
    if has_down:
        idt = conv(x, sd[p + ".downsample.0.weight"], stride=stride)
        idt = batchnorm(sd, p + ".downsample.1", idt, training)
    else:
        idt = x
    return _q(F.relu(out + idt))


def resnet_features(sd, p, x, num_frames, depth=50, pooling_method="max", without_t_stride=False, training=False):
    """models/resnet.py:195-210 (stem, maxpool, layer1-4 with temporal pooling between)."""
    n, c_t, h, w = x.shape
    if c_t != 1:
        x = x.reshape(n * num_frames, c_t // num_frames, h, w)
    x = conv(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = F.relu(batchnorm(sd, p + "bn1", x, training))
    x = _q(F.max_pool2d(x, 3, 2, 1))
    frames = num_frames
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), RESNET_BLOCKS[depth])):
        stride = 1 if li == 0 else 2
        for bi in range(nblk):
            s = stride if bi == 0 else 1
            has_down = bi == 0 and (s != 1 or inplanes != planes * 4)
            x = _bottleneck(sd, "%slayer%d.%d" % (p, li + 1, bi), x, s, has_down, training)
            inplanes = planes * 4
        if li < 3 and not without_t_stride:
            x = _q(temporal_pool(x, frames, pooling_method.lower()))   # models/resnet.py:145-153
            frames = max(1, frames // 2)
    return x


def resnet_forward(sd, p, x, num_frames, depth=50, pooling_method="max", without_t_stride=False,
                   dropout_p=0.5, training=False, dropout_mask=None):
    """models/resnet.py:195-223: features -> GAP -> dropout -> FC -> mean over remaining frames."""
    n = x.shape[0]
    f = resnet_features(sd, p, x, num_frames, depth, pooling_method, without_t_stride, training)
    f = f.mean(dim=(2, 3))
    f = dropout(f, dropout_p, training, dropout_mask)
    y = F.linear(f, sd[p + "fc.weight"], sd[p + "fc.bias"])
    return y.reshape(n, -1, y.shape[-1]).mean(dim=1)


# ----------------------------------------------------------------------------- Sound MobileNetV2 (A6)
def sound_mbv2_features(sd, p, x, training):
    """models/sound_mobilenet_v2.py:120-131,152-155 (torchvision-style naming)."""
    def cbr(pp, x, stride=1, groups=1, k=3):
        x = conv(x, sd[pp + ".0.weight"], stride=stride, padding=(k - 1) // 2, groups=groups)
        return F.relu6(batchnorm(sd, pp + ".1", x, training))

    x = cbr(p + "features.0", x, stride=2)
    inp = 32
    idx = 1
    for t, c, nrep, s in MBV2_CFG:
        for i in range(nrep):
            stride = s if i == 0 else 1
            hid = int(round(inp * t))
            pp = "%sfeatures.%d.conv" % (p, idx)
            y = x
            j = 0
            if t != 1:
                y = cbr("%s.%d" % (pp, j), y, k=1)
                j += 1
            y = cbr("%s.%d" % (pp, j), y, stride=stride, groups=hid)
            j += 1
            y = conv(y, sd["%s.%d.weight" % (pp, j)])
            y = batchnorm(sd, "%s.%d" % (pp, j + 1), y, training)
            x = _q(x + y) if (stride == 1 and inp == c) else y
            inp = c
            idx += 1
    return cbr("%sfeatures.%d" % (p, idx), x, k=1)


def sound_mbv2_forward(sd, p, x, dropout_p=0.5, training=False, dropout_mask=None):
    """models/sound_mobilenet_v2.py:152-159."""
    f = sound_mbv2_features(sd, p, x, training).mean(dim=(2, 3))
    f = dropout(f, dropout_p, training, dropout_mask)
    return F.linear(f, sd[p + "classifier.1.weight"], sd[p + "classifier.1.bias"])


# ----------------------------------------------------------------------------- Policy MobileNetV2 (A2)
def policy_mbv2_features(sd, p, x, num_frames, training):
    """models/policy_net.py:142-149 with the block layout of :54-95 and the temporal
    max-pool placement of :120-130 (first block of the c=64 and c=160 stages)."""
    n, c_t, h, w = x.shape
    x = x.reshape(n * num_frames, c_t // num_frames, h, w)
    x = conv(x, sd[p + "features.0.0.weight"], stride=2, padding=1)
    x = F.relu6(batchnorm(sd, p + "features.0.1", x, training))
    inp, idx, cur = 32, 1, num_frames
    for t, c, nrep, s in MBV2_CFG:
        has_tp = c in (64, 160)
        for i in range(nrep):
            stride = s if i == 0 else 1
            hid = round(inp * t)
            pp = "%sfeatures.%d.conv" % (p, idx)
            if i == 0 and has_tp and cur not in (0, 1):
                x = _q(temporal_pool(x, cur, "max"))
            y = x
            j = 0
            if t != 1:
                y = conv(y, sd["%s.0.weight" % pp])
                y = F.relu6(batchnorm(sd, "%s.1" % pp, y, training))
                j = 3
            y = conv(y, sd["%s.%d.weight" % (pp, j)], stride=stride, padding=1, groups=hid)
            y = F.relu6(batchnorm(sd, "%s.%d" % (pp, j + 1), y, training))
            y = conv(y, sd["%s.%d.weight" % (pp, j + 3)])
            y = batchnorm(sd, "%s.%d" % (pp, j + 4), y, training)
            x = _q(x + y) if (stride == 1 and inp == c) else y
            inp = c
            idx += 1
        if has_tp:
            cur //= 2
    x = conv(x, sd[p + "conv.0.weight"])
    x = F.relu6(batchnorm(sd, p + "conv.1", x, training))
    return x.mean(dim=(2, 3))


def gumbel_hard_select(logits, tau, expo):
    """F.gumbel_softmax(logits, tau, hard=True)[:, -1] (models/policy_net.py:283-290) with the
    Exponential(1) draw supplied by the caller: g = -log(E); y = softmax((l+g)/tau);
    ret = onehot(argmax y) - y.detach() + y."""
    g = -torch.log(expo)
    y = F.softmax((logits + g) / tau, dim=-1)
    idx = y.argmax(dim=-1, keepdim=True)
    hard = torch.zeros_like(y).scatter_(-1, idx, 1.0)
    return (hard - y.detach() + y)[:, -1]


def policy_forward(sd, p, p_x, modality, num_frames, temperature, expo, causality="lstm", training=False):
    """models/policy_net.py:312-373.  p_x: list over modality of [S, B, F*C, H, W].
    expo: [S, M*B, 2] Exponential(1) samples.  Returns decisions [S,M,B], logits [S,M,B,2]."""
    M = len(modality)
    S = p_x[0].shape[0]
    feats = []
    for s in range(S):
        fs = []
        for mi, m in enumerate(modality):
            nf = 1 if m == "sound" else num_frames
            fs.append(policy_mbv2_features(sd, "%sjoint_net.nets.%d." % (p, mi), p_x[mi][s], nf, training))
        f = torch.cat(fs, dim=1)
        f = F.relu(F.linear(f, sd[p + "joint_net.joint.0.weight"], sd[p + "joint_net.joint.0.bias"]))
        f = F.relu(F.linear(f, sd[p + "joint_net.joint.2.weight"], sd[p + "joint_net.joint.2.bias"]))
        feats.append(f)
    return policy_head(sd, p, feats, M, temperature, expo, causality)


def policy_head(sd, p, feats, M, temperature, expo, causality="lstm"):
    """Causality head of PolicyNet.forward (models/policy_net.py:329-373) on the per-segment joint features
    (list of S tensors [B, 2048]): FC heads or LSTMCell with the previous logits fed back, then the hard Gumbel gate."""
    S = len(feats)
    B = feats[0].shape[0]
    all_logits, decisions = [], []
    if causality is None:
        o = torch.stack(feats, 0).reshape(S * B, -1)
        lg = torch.cat([F.linear(o, sd["%sfcs.%d.weight" % (p, mi)], sd["%sfcs.%d.bias" % (p, mi)]) for mi in range(M)], 0)
        # reference draws one noise tensor for the [M*S*B, 2] logits (policy_net.py:335-336)
        e = expo.reshape(M * S * B, 2)
        d = gumbel_hard_select(lg, temperature, e)
        return d.reshape(M, S, B).transpose(0, 1), lg.reshape(M, S, B, 2).transpose(0, 1)
    h = c = None
    logits = None
    for s in range(S):
        if s == 0:
            prev = torch.zeros(B, 2 * M, dtype=feats[0].dtype, device=feats[0].device)
            h = torch.zeros(B, 256, dtype=feats[0].dtype, device=feats[0].device)
            c = torch.zeros(B, 256, dtype=feats[0].dtype, device=feats[0].device)
        else:
            prev = logits.reshape(M, B, 2).permute(1, 0, 2).reshape(B, 2 * M)   # :353
        xin = torch.cat((feats[s], prev), dim=-1)
        gates = F.linear(xin, sd[p + "lstm.weight_ih"], sd[p + "lstm.bias_ih"]) + \
            F.linear(h, sd[p + "lstm.weight_hh"], sd[p + "lstm.bias_hh"])
        gi, gf, gg, go = gates.chunk(4, dim=1)                                  # torch LSTMCell gate order i,f,g,o
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        logits = torch.cat([F.linear(h, sd["%sfcs.%d.weight" % (p, mi)], sd["%sfcs.%d.bias" % (p, mi)])
                            for mi in range(M)], dim=0)                         # [M*B, 2]
        all_logits.append(logits.reshape(M, B, 2))
        decisions.append(gumbel_hard_select(logits, temperature, expo[s]))
    return torch.stack(decisions, 0).reshape(S, M, B), torch.stack(all_logits, 0)


# ----------------------------------------------------------------------------- data layer (A1)
def data_layer(x, modality, p_modality, m_modality, num_segments, frames_per_segment, p_size=(160, 160)):
    """models/adamml.py:42-67.  Returns p_x, m_x as lists of [S, B, F*C, H, W]."""
    p_x, m_x = [], []
    for x_, m in zip(x, modality):
        b = x_.shape[0]
        if m == "sound":
            if x_.shape[-1] != x_.shape[-2]:
                t = torch.stack(x_.chunk(num_segments, dim=-1), dim=0).contiguous()
            else:
                t = x_.reshape(b, num_segments, -1, *x_.shape[-2:]).transpose(0, 1).contiguous()
            p_x.append(t)
            m_x.append(t)
            continue
        if m in p_modality:
            t = F.interpolate(x_, size=p_size, mode="bilinear")      # align_corners=False, no antialias
            t = t.reshape(b, num_segments, frames_per_segment, -1, *p_size)[:, :, 0::2]
            p_x.append(t.reshape(b, num_segments, -1, *p_size).transpose(0, 1).contiguous())
        if m in m_modality:
            m_x.append(x_.reshape(b, num_segments, -1, *x_.shape[-2:]).transpose(0, 1).contiguous())
    return p_x, m_x


# ----------------------------------------------------------------------------- main net + top level (A8, A9)
def main_net_forward(sd, p, xs, m_modality, decisions, num_frames, depth=50, pooling_method="max",
                     without_t_stride=False, dropout_p=0.5, training=False, learnable_lf_weights=True):
    """models/joint_resnet_mobilenetv2.py:84-128 (fusion_point='logits')."""
    outs = []
    for i, (x, m) in enumerate(zip(xs, m_modality)):
        pp = "%snets.%d." % (p, i)
        if m == "sound":
            y = sound_mbv2_forward(sd, pp, x, dropout_p, training)
        else:
            y = resnet_forward(sd, pp, x, num_frames, depth, pooling_method, without_t_stride, dropout_p, training)
        if decisions is not None:
            y = y * decisions[i].reshape(-1, 1)
        outs.append(y)
    out = torch.stack(outs, 0)
    if learnable_lf_weights and (p + "lf_weights") in sd:
        lw = sd[p + "lf_weights"]
        wts = torch.cat((lw, 1.0 - lw.sum(0, keepdim=True)), 0).reshape(-1, 1, 1)
        return (out * wts).sum(0)
    return out.mean(0)


def adamml_forward(sd, x, modality, num_segments, groups=8, depth=50, temperature=5.0, expo=None,
                   causality="lstm", pooling_method="max", without_t_stride=False, dropout_p=0.5,
                   training=False, learnable_lf_weights=True, decisions=None):
    """models/adamml.py:69-91.  Returns (logits [B,C], decisions [B,S,M], policy logits [S,M,B,2])."""
    both = "rgbdiff" in modality and "flow" in modality
    p_mod = [m for m in modality if not (both and m == "flow")]
    m_mod = [m for m in modality if not (both and m == "rgbdiff")]
    p_x, m_x = data_layer(x, modality, p_mod, m_mod, num_segments, groups)
    p_logits = None
    if decisions is None:
        decisions, p_logits = policy_forward(sd, "policy_net.", p_x, p_mod, max(1, groups // 2), temperature,
                                             expo, causality, training)
    all_logits = []
    for s in range(num_segments):
        all_logits.append(main_net_forward(sd, "main_net.", [t[s] for t in m_x], m_mod, decisions[s], groups, depth,
                                           pooling_method, without_t_stride, dropout_p, training, learnable_lf_weights))
    final = torch.stack(all_logits, 1).mean(1)
    return final, decisions.permute(2, 0, 1), p_logits


# ----------------------------------------------------------------------------- losses (A10)
def policy_loss(penalty_type, selection, cost_weights, gammas, cls_logits, cls_targets):
    """utils/utils.py:166-184."""
    M = selection.shape[-1]
    loss = selection.new_zeros(())
    if penalty_type == "mean":
        for mi in range(M):
            loss = loss + cost_weights[mi] * selection[..., mi].mean()
    elif penalty_type == "blockdrop":
        correct = (cls_logits.detach().argmax(-1) == cls_targets).to(cls_logits.dtype)
        usage = selection.mean(dim=1) ** 2
        for mi in range(M):
            # reference: `correctness * pl` with correctness [N] and pl = chunk(...) of shape [N,1] BROADCASTS to [N,N]
            # (utils/utils.py:177-181), i.e. mean(correct) * mean(usage), not the per-video product -- mirrored as is
            # (pinned by tests/golden/policy_loss_cases.npz)
            loss = loss + cost_weights[mi] * (correct.unsqueeze(0) * usage[:, mi:mi + 1]).mean()
        loss = loss + ((1.0 - correct) * gammas).mean()
    return loss


def make_leaf_state(sd, trainable_prefixes=("main_net.",)):
    """Clone a state_dict into leaf tensors; entries under trainable_prefixes require grad."""
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if t.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")) and \
                any(k.startswith(pf) for pf in trainable_prefixes):
            t.requires_grad_(True)
        out[k] = t
    return out
