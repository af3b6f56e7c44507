"""Policy subnet of AdaMML on libadamml_hip: per-modality MobileNetV2 (d-li14 variant with temporal max-pooling),
joint FC, LSTM causality head and hard Gumbel-softmax gate.
Mirrors models/policy_net.py:54-387 (MobileNetV2, JointMobileNetV2, PolicyNet, p_joint_mobilenet; same state_dict)."""
import math
import torch
import torch.nn as nn

from . import imagenet_init
from .backbone import HipBackbone
from .common import MeanStdMixin
from .functional import hip_linear, policy_head, gumbel_gate
from .mobilenet_common import BlockPlan, run_blocks
from .runtime import Lazy, conv_bn, conv_stem1_bn, stem1_supported, gap, ACT_NONE, ACT_RELU, ACT_RELU6

_CFGS = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]


class InvertedResidual(nn.Module):
    """Parameter container with the Sequential indices of models/policy_net.py:63-86."""

    def __init__(self, inp, oup, stride, expand_ratio, num_frames=None):
        super().__init__()
        assert stride in [1, 2]
        hidden_dim = round(inp * expand_ratio)
        self.identity = stride == 1 and inp == oup
        self.tpool_frames = num_frames if num_frames else None
        self.expand = expand_ratio != 1
        if expand_ratio == 1:
            self.conv = nn.Sequential(
                nn.Conv2d(hidden_dim, hidden_dim, 3, stride, 1, groups=hidden_dim, bias=False), nn.BatchNorm2d(hidden_dim),
                nn.ReLU6(inplace=True),
                nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup))
        else:
            self.conv = nn.Sequential(
                nn.Conv2d(inp, hidden_dim, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden_dim), nn.ReLU6(inplace=True),
                nn.Conv2d(hidden_dim, hidden_dim, 3, stride, 1, groups=hidden_dim, bias=False), nn.BatchNorm2d(hidden_dim),
                nn.ReLU6(inplace=True),
                nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup))


class MobileNetV2(HipBackbone, MeanStdMixin):
    """models/policy_net.py:98-203 (feature extractor part; the classifier is dropped by JointMobileNetV2)."""

    def __init__(self, num_classes=1000, num_frames=4, input_channels=3, width_mult=1.):
        super().__init__()
        if width_mult != 1.:
            raise ValueError("adamml_amd policy MobileNetV2: width_mult must be 1.0")
        self.input_channels = input_channels
        self.num_frames = num_frames
        self.orig_num_frames = num_frames
        input_channel = 32
        self.out_frames = num_frames         # frames per clip left after the temporal max-pools
        layers = [nn.Sequential(nn.Conv2d(input_channels, input_channel, 3, 2, 1, bias=False), nn.BatchNorm2d(input_channel),
                                nn.ReLU6(inplace=True))]
        for t, c, n, s in _CFGS:
            has_tp = c == 64 or c == 160
            for i in range(n):
                nf = self.num_frames if i == 0 and has_tp and self.num_frames != 1 else None
                if nf:
                    self.out_frames = (self.out_frames - 1) // 2 + 1
                layers.append(InvertedResidual(input_channel, c, s if i == 0 else 1, t, num_frames=nf))
                input_channel = c
            if has_tp:
                self.num_frames //= 2
        self.features = nn.Sequential(*layers)
        self.last_channel = 1280
        self.conv = nn.Sequential(nn.Conv2d(input_channel, 1280, 1, 1, 0, bias=False), nn.BatchNorm2d(1280), nn.ReLU6(inplace=True))
        self.classifier = nn.Linear(1280, num_classes)
        self._initialize_weights()

        f0 = self.features[0]
        self._stem = (self._register_conv(f0[0]), f0[1])
        # one-channel input (spectrogram): the stem can read the fp32 tensor directly (runtime.conv_stem1_bn)
        self._stem1 = self._register_conv(f0[0], depthwise=True) if input_channels == 1 else None
        self._plans = []
        for blk in self.features[1:]:
            seq = blk.conv
            if blk.expand:
                pw = (self._register_conv(seq[0]), seq[1])
                dw = (self._register_conv(seq[3], depthwise=True), seq[4])
                pwl = (self._register_conv(seq[6]), seq[7])
            else:
                pw = None
                dw = (self._register_conv(seq[0], depthwise=True), seq[1])
                pwl = (self._register_conv(seq[3]), seq[4])
            self._plans.append(BlockPlan(pw, dw, pwl, blk.identity, blk.tpool_frames))
        self._last = (self._register_conv(self.conv[0]), self.conv[1])

    def _initialize_weights(self):                       # models/policy_net.py:166-178
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def _run(self, x, groups, need_grad):
        """x: [G*B*T, H, W, pad8(C)] bf16.  Returns pooled features fp32 [G*B*T', 1280] (feature_extraction, :142-149)."""
        rt = self.rt
        tape = rt.begin_forward(x.device, self.training, need_grad, groups)
        self._repack(need_grad)
        self._mark_grads_ready_after(tape, [self])        # one gradient bucket: exchanged as soon as this net's backward is enqueued
        if x.dtype == torch.float32:
            h = conv_stem1_bn(rt, x, self._stem1, self._stem[1], ACT_RELU6)
        else:
            h = conv_bn(rt, Lazy(x, requires_grad=False), self._stem[0], self._stem[1], ACT_RELU6)
        h = run_blocks(rt, h, self._plans)
        h = conv_bn(rt, h, self._last[0], self._last[1], ACT_RELU6)
        feat, push = gap(rt, h)
        rt.end_forward()
        if need_grad:
            tape.record(lambda: push(tape.grad_out))
        return feat, tape

    def out_shape(self, x_shape, groups):
        n = x_shape[0] * x_shape[1] if (len(x_shape) == 4 and x_shape[1] == groups and x_shape[-1] != 8 and self._stem1 is not None) else x_shape[0]
        return (n // self.orig_num_frames * self.out_frames, self.last_channel)

    def accepts_f32(self, x):
        """Can the stem read this [B, G, H, W] one-channel-per-group fp32 tensor directly?"""
        return self._stem1 is not None and x.dim() == 4 and stem1_supported(self._stem1, x)

    def feature_extraction(self, frames_nhwc, groups=1):
        return self.call(frames_nhwc, groups)

    @property
    def network_name(self):
        return 'mobilenet_v2'


class JointMobileNetV2(nn.Module):
    """models/policy_net.py:206-258."""

    def __init__(self, num_frames, modality, num_classes=1000, dropout=0.5, input_channels=None):
        super().__init__()
        self.num_frames = num_frames
        self.modality = modality
        self.nets = nn.ModuleList()
        chans = []
        for i, m in enumerate(modality):
            net = MobileNetV2(num_classes, num_frames=1 if m == 'sound' else num_frames, input_channels=input_channels[i])
            del net.classifier
            chans.append(net.last_channel)
            # models/policy_net.py:221 `net.load_imagenet_model()`, unconditional in the reference -- from a local file here
            imagenet_init.init_mobilenet_v2(net, input_channels[i], True, what="policy mobilenet_v2", arch="mobilenetv2_160x160")
            self.nets.append(net)
        self.last_channels = 2048
        self.joint = nn.Sequential(nn.Linear(sum(chans), 2048), nn.ReLU(True), nn.Linear(2048, 2048), nn.ReLU(True))

    def features(self, multi_modalities, groups=1):
        """multi_modalities: list of NHWC bf16 frame tensors (`groups` segments stacked along dim 0).  -> [G*B, 2048]."""
        return self.joint_features([net.feature_extraction(x, groups) for net, x in zip(self.nets, multi_modalities)])

    def joint_features(self, feats):
        """cat over modalities -> Linear -> ReLU -> Linear -> ReLU (models/policy_net.py:243-245) on the backbones' pooled features."""
        out = torch.cat(list(feats), dim=1)
        out = hip_linear(out, self.joint[0].weight, self.joint[0].bias, ACT_RELU)
        return hip_linear(out, self.joint[2].weight, self.joint[2].bias, ACT_RELU)


class PolicyNet(nn.Module):
    """models/policy_net.py:261-379."""

    def __init__(self, joint_net, modality, causality_modeling='lstm'):
        super().__init__()
        self.joint_net = joint_net
        self.modality = modality
        self.causality_modeling = causality_modeling
        self.num_modality = len(modality)
        self.temperature = 5.0
        feature_dim = self.joint_net.last_channels
        if causality_modeling is not None:
            embedded_dim = 256
            self.lstm = nn.LSTMCell(feature_dim + 2 * self.num_modality, embedded_dim)
            self.fcs = nn.ModuleList([nn.Linear(embedded_dim, 2) for _ in range(self.num_modality)])
        else:
            self.fcs = nn.ModuleList([nn.Linear(feature_dim, 2) for _ in range(self.num_modality)])

    def wrapper_gumbel_softmax(self, logits, expo=None):
        if expo is None:
            expo = torch.empty_like(logits).exponential_()
        return gumbel_gate(logits, expo, self.temperature)

    def set_temperature(self, temperature):
        self.temperature = temperature

    def decay_temperature(self, decay_ratio=None):
        if decay_ratio:
            self.temperature *= decay_ratio
        print("Current temperature: {}".format(self.temperature), flush=True)

    def all_segment_features(self, x):
        """Joint features of ALL segments in one batched pass (per-segment BatchNorm statistics are kept by the
        `groups` mechanism of the backbones; the joint FCs have no batch statistics).  -> list of S tensors [B, 2048]."""
        S = x[0].shape[0]
        out = self.joint_net.features([(x[m_i] if x[m_i].dtype == torch.float32 else x[m_i].flatten(0, 1)) for m_i in range(self.num_modality)], groups=S)
        return list(out.view(S, -1, out.shape[-1]).unbind(0))

    def forward(self, x, gumbel_exponential=None):
        """x: list over modality of [S, B*Fk, H, W, C] NHWC bf16 frames.  Returns decisions [S,M,B], logits [S,M,B,2]."""
        return self.decide(self.all_segment_features(x), gumbel_exponential)

    def decide(self, outs, gumbel_exponential=None):
        """Causality head + hard Gumbel-softmax over the per-segment features (models/policy_net.py:329-373) on the HIP
        head kernels: `lstm` = one adamml_policy_head_fwd launch for all S segments (the per-video recurrence runs inside
        the kernel); None = FC heads on adamml_gemm_f32 + adamml_gumbel_gate_fwd.  gumbel_exponential: optional
        Exponential(1) draw, [S, M*B, 2] (lstm) / [M*S*B, 2] (None) -- the reference's row order -- for parity runs."""
        M = self.num_modality
        feats = torch.stack(list(outs), 0) if not torch.is_tensor(outs) else outs        # [S, B, F]
        S, B = feats.shape[0], feats.shape[1]
        if gumbel_exponential is None:
            gumbel_exponential = torch.empty(S * M * B, 2, dtype=torch.float32, device=feats.device).exponential_()
        if self.causality_modeling is None:
            o = feats.reshape(S * B, -1)
            logits = torch.cat([hip_linear(o, fc.weight, fc.bias) for fc in self.fcs], dim=0)      # (MSB) x 2
            decisions = gumbel_gate(logits, gumbel_exponential.reshape(M * S * B, 2), self.temperature)
            return decisions.view(M, S, -1).transpose(0, 1), logits.view(M, S, -1, 2).transpose(0, 1)
        if self.causality_modeling != 'lstm':
            raise ValueError("unknown mode")
        return policy_head(feats, self.lstm, self.fcs, self.temperature, gumbel_exponential.reshape(S, M, B, 2))

    @property
    def network_name(self):
        return 'j_mobilenet_v2{}'.format('-' + self.causality_modeling if self.causality_modeling else '')


def p_joint_mobilenet(num_frames, modality, input_channels, causality_modeling):
    """models/policy_net.py:382-387.  (The reference downloads ImageNet weights for every policy MobileNetV2 here; on the target
    systems they are read from a local file when one is configured: imagenet_init.py.)"""
    joint_net = JointMobileNetV2(num_frames=num_frames, modality=modality, input_channels=input_channels)
    return PolicyNet(joint_net, modality, causality_modeling=causality_modeling)
