"""torchvision-style MobileNetV2 for 1x256x256 log-spectrograms on libadamml_hip.
Mirrors models/sound_mobilenet_v2.py:33-198 (class MobileNetV2, factory sound_mobilenet_v2; same state_dict)."""
import torch
import torch.nn as nn

from . import hip
from .backbone import HipBackbone, FlatBuffers, StockDDPAware, NotifyingSequential
from .common import MeanStdMixin
from .mobilenet_common import BlockPlan, run_blocks
from .runtime import Lazy, conv_bn, conv_stem1_bn, stem1_supported, head, clip_to_nhwc, ACT_RELU6

__all__ = ['MobileNetV2', 'sound_mobilenet_v2']

_CFG = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvBNReLU(nn.Sequential):
    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1):
        padding = (kernel_size - 1) // 2
        super().__init__(nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, groups=groups, bias=False),
                         nn.BatchNorm2d(out_planes), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    """Parameter container (models/sound_mobilenet_v2.py:43-63 naming)."""

    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        self.stride = stride
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = self.stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(inp, hidden_dim, kernel_size=1))
        layers.extend([ConvBNReLU(hidden_dim, hidden_dim, stride=stride, groups=hidden_dim),
                       nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)])
        self.conv = nn.Sequential(*layers)
        self.expand = expand_ratio != 1


class MobileNetV2(HipBackbone, MeanStdMixin, StockDDPAware):

    def __init__(self, num_classes=1000, width_mult=1.0, inverted_residual_setting=None, round_nearest=8, block=None,
                 input_channels=3, dropout=0.5):
        super().__init__()
        self._install_ddp_probe()
        if width_mult != 1.0 or inverted_residual_setting is not None or block is not None:
            raise ValueError("adamml_amd sound MobileNetV2: only the default width/setting of the AdaMML hot path is built")
        input_channel = _make_divisible(32 * width_mult, round_nearest)
        self.last_channel = _make_divisible(1280 * max(1.0, width_mult), round_nearest)
        self.input_channels = input_channels
        features = [ConvBNReLU(input_channels, input_channel, stride=2)]
        for t, c, n, s in _CFG:
            output_channel = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                features.append(InvertedResidual(input_channel, output_channel, s if i == 0 else 1, expand_ratio=t))
                input_channel = output_channel
        features.append(ConvBNReLU(input_channel, self.last_channel, kernel_size=1))
        self.features = nn.Sequential(*features)
        self.dropout_p = dropout
        # (a Sequential that reports `net.classifier[1] = nn.Linear(..)` to this backbone: the head is what callers replace)
        self.classifier = NotifyingSequential(nn.Dropout(dropout), nn.Linear(self.last_channel, num_classes)).bind_owner(self)
        for m in self.modules():                      # models/sound_mobilenet_v2.py:140-150
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

        f0 = self.features[0]
        self._stem = (self._register_conv(f0[0]), f0[1])
        # one-channel input: the stem can read the fp32 spectrogram directly (runtime.conv_stem1_bn; tap-major fp32 weight pack)
        self._stem1 = self._register_conv(f0[0], depthwise=True) if input_channels == 1 else None
        self._plans = []
        for blk in self.features[1:-1]:
            seq = list(blk.conv)
            j = 0
            pw = None
            if blk.expand:
                pw = (self._register_conv(seq[0][0]), seq[0][1])
                j = 1
            dw = (self._register_conv(seq[j][0], depthwise=True), seq[j][1])
            pwl = (self._register_conv(seq[j + 1]), seq[j + 2])
            self._plans.append(BlockPlan(pw, dw, pwl, blk.use_res_connect))
        fl = self.features[-1]
        self._last = (self._register_conv(fl[0]), fl[1])
        self.flat_owner = FlatBuffers(self)

    def _run(self, x, groups, need_grad):
        """x: [G*B, H, W, 8] bf16 (1 real channel), or the fp32 spectrograms [B, G, H, W] themselves.  Returns fp32 logits [G*B, num_classes]."""
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
        # GAP -> dropout -> classifier (models/sound_mobilenet_v2.py:155-158): one fused launch per direction
        out, head_backward = head(rt, h, self.classifier[1], 1, self.dropout_p if self.training else 0.0,
                                  getattr(self, "_dropout_keep_mask", None))
        rt.end_forward()
        if need_grad:
            tape.record(lambda: head_backward(tape.grad_out))
        return out, tape

    def forward(self, x):
        """x [B, 1, H, W] fp32 -> logits (models/sound_mobilenet_v2.py:152-162)."""
        hip.require_gpu(x)
        self.flat_owner.ensure(x.device)
        if self.training and torch.is_grad_enabled():
            self.flat_owner.ensure_grads()
        if self.accepts_f32(x):
            return self.call(x.float().contiguous(), 1)           # [B, 1, H, W]: one BatchNorm group
        xs = clip_to_nhwc(x, 1, 1, x.shape[1])[0]
        return self.call(xs)

    def accepts_f32(self, x):
        """Can the stem read this [B, G, H, W] one-channel-per-group fp32 tensor directly?"""
        return self._stem1 is not None and x.dim() == 4 and stem1_supported(self._stem1, x.float() if x.dtype != torch.float32 else x)

    def forward_nhwc(self, frames_nhwc, groups=1):
        return self.call(frames_nhwc, groups)

    def out_shape(self, x_shape, groups):
        return (x_shape[0] * x_shape[1] if len(x_shape) == 4 and x_shape[1] == groups and x_shape[-1] != 8 else x_shape[0], self.classifier[1].out_features)


def sound_mobilenet_v2(num_classes, input_channels, dropout, imagenet_pretrained=True, **kwargs):
    """Factory with the signature of models/sound_mobilenet_v2.py:177-198.  imagenet_pretrained: as for resnet() -- torchvision's
    mobilenet_v2 file read from disk (imagenet_init.py), classifier dropped, a non-RGB stem = mean over RGB expanded to input_channels."""
    from . import imagenet_init
    model = MobileNetV2(num_classes=num_classes, input_channels=input_channels, dropout=dropout)
    return imagenet_init.init_mobilenet_v2(model, input_channels, imagenet_pretrained, what="sound_mobilenet_v2")
