"""TSN-style 2-D ResNet-50/101/152 over N*T frames with temporal pooling between stages, executed by
libadamml_hip (bf16 MFMA implicit-GEMM convs with fused BatchNorm statistics).

Mirrors the interface and state_dict of models/resnet.py:116-259 (class ResNet, factory resnet()).
"""
import os

import torch
import torch.nn as nn

from . import hip
from .backbone import HipBackbone, FlatBuffers, StockDDPAware
from .common import MeanStdMixin
from .runtime import (Lazy, conv_bn, conv_bn_add, conv_bn_add_supported, conv_bn_add_tpool_supported, add_act, maxpool3x3s2, temporal_pool, head, clip_to_nhwc, pad8,
                      ACT_NONE, ACT_RELU)

__all__ = ['ResNet', 'resnet']

_BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class _Bottleneck(nn.Module):
    """Parameter container with the names of models/resnet.py:77-92."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


STEM_PAD4 = os.environ.get("ADAMML_STEM_PAD4", "1") != "0"      # 4-channel input pixels for the 7x7 stem kernels (A/B aid)


class ResNet(HipBackbone, MeanStdMixin, StockDDPAware):

    def __init__(self, depth, num_frames, num_classes=1000, dropout=0.5, zero_init_residual=False,
                 without_t_stride=False, pooling_method='max', input_channels=3):
        super().__init__()
        self._install_ddp_probe()
        if depth not in _BLOCKS:
            raise ValueError("adamml_amd.ResNet: the HIP path implements the Bottleneck depths 50/101/152 "
                             "(the AdaMML hot path uses 50); got depth=%r" % (depth,))
        layers = _BLOCKS[depth]
        self.pooling_method = pooling_method.lower()
        self.depth = depth
        self.num_frames = num_frames
        self.orig_num_frames = num_frames
        self.num_classes = num_classes
        self.without_t_stride = without_t_stride
        self.input_channels = input_channels

        self.inplanes = 64
        self.conv1 = nn.Conv2d(input_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.dropout_p = dropout
        self.fc = nn.Linear(2048, num_classes)

        self._stem = self._register_conv(self.conv1)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in layer:
                b._cs1 = self._register_conv(b.conv1)
                b._cs2 = self._register_conv(b.conv2)
                b._cs3 = self._register_conv(b.conv3)
                b._csd = self._register_conv(b.downsample[0]) if b.downsample is not None else None
        self.flat_owner = FlatBuffers(self)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, 1, None))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------------------------------
    def _run(self, x, groups, need_grad):
        """x: [G*N*T, H, W, pad8(C)] bf16 frames (G groups of N clips).  Returns fp32 logits [G*N, num_classes]."""
        rt = self.rt
        tape = rt.begin_forward(x.device, self.training, need_grad, groups)
        self._repack(need_grad)
        frames = self.orig_num_frames
        nt = x.shape[0]
        n = nt // frames                              # clips over all groups
        h = Lazy(x, requires_grad=False)
        # gradient buckets for the data-parallel exchange (parameter order == flat-buffer order): [stem, layer1, layer2],
        # [layer3], [layer4, fc]; each marker fires when its bucket's last weight gradient has been enqueued
        self._mark_grads_ready_after(tape, [self.conv1, self.bn1, self.layer1, self.layer2])
        h = conv_bn(rt, h, self._stem, self.bn1, ACT_RELU)
        h = maxpool3x3s2(rt, h, sole_consumer=True)
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            if li == 2:
                self._mark_grads_ready_after(tape, [self.layer3])
            elif li == 3:
                self._mark_grads_ready_after(tape, [self.layer4, self.fc])
            for bi, b in enumerate(layer):
                # the downsample branch is issued FIRST so that its data gradient runs LAST in the reversed tape: it then
                # accumulates into the gradient conv1 already wrote, and a stride-2 1x1 only touches a quarter of the pixels
                idn = conv_bn(rt, h, b._csd, b.downsample[1], ACT_NONE) if b._csd is not None else h
                # without a downsample branch conv1 is the last consumer of h in the reversed tape (the identity path of
                # the add is reversed first): its data-gradient epilogue finishes the previous block's residual backward
                o = conv_bn(rt, h, b._cs1, b.bn1, ACT_RELU, last_consumer=b._csd is None)
                o = conv_bn(rt, o, b._cs2, b.bn2, ACT_RELU, sole_consumer=True)
                pooled = False
                if conv_bn_add_supported(rt, o, b._cs3, need_grad, idn):
                    # conv3 + bn3 + residual add + ReLU in one kernel (statistics from the Gram matrix of conv3's input in train mode);
                    # behind the last block of a stage the temporal max-pool runs in that kernel's epilogue as well
                    pooled = (b is layer[-1] and li < 3 and not self.without_t_stride and
                              conv_bn_add_tpool_supported(rt, o, b._cs3, idn, ACT_RELU, frames, self.pooling_method))
                    # (within a stage the next block's conv1 is the only conv of the block output: it can run inside this kernel)
                    nxt = layer[bi + 1]._cs1 if (bi + 1 < len(layer) and not pooled) else None
                    h = conv_bn_add(rt, o, b._cs3, b.bn3, idn, ACT_RELU, idn_sole=b._csd is not None, tpool=frames if pooled else 0, next_cs=nxt)
                else:
                    o = conv_bn(rt, o, b._cs3, b.bn3, ACT_NONE, sole_consumer=True)
                    h = add_act(rt, o, idn, ACT_RELU, idn_sole=b._csd is not None)
            if li < 3 and not self.without_t_stride:
                if not pooled:
                    h = temporal_pool(rt, h, frames, self.pooling_method, sole_consumer=True)
                frames = max(1, frames // 2)
        # GAP -> dropout -> fc -> mean over the remaining frames (models/resnet.py:212-221): one fused launch per direction
        out, head_backward = head(rt, h, self.fc, frames, self.dropout_p if self.training else 0.0, getattr(self, "_dropout_keep_mask", None))
        rt.end_forward()
        if need_grad:
            tape.record(lambda: head_backward(tape.grad_out))
        return out, tape

    def forward(self, x):
        """models/resnet.py:195-223 contract: x [N, T*C, H, W] fp32 -> logits [N, num_classes]."""
        hip.require_gpu(x)
        self.flat_owner.ensure(x.device)
        if self.training and torch.is_grad_enabled():
            self.flat_owner.ensure_grads()
        n, c_t, hh, ww = x.shape
        frames = self.orig_num_frames if c_t != 1 else 1
        xs = clip_to_nhwc(x, 1, frames, c_t // frames, cpad=self.input_cpad(hh, ww))[0]
        return self.call(xs)

    def input_cpad(self, h, w):
        """Channel padding of the NHWC input this net wants for h x w frames: 4 when the 7x7 stem kernel (csrc/conv_stem.hip, <= 4
        input channels) serves the shape -- it reads 8 bytes per pixel, and 4-channel pixels halve the bytes of the re-layout store
        and of the stem's forward / weight-gradient loads -- else the generic 8-channel multiple."""
        cs = self._stem
        if cs.stem and STEM_PAD4:
            from ctypes import byref
            d = cs.desc((1, h, w, 4), 0, 1, 0)
            if hip.load().adamml_conv_stem_supported(byref(d)):
                return 4
        return pad8(cs.cin_true)

    def forward_nhwc(self, frames_nhwc, groups=1):
        return self.call(frames_nhwc, groups)

    def out_shape(self, x_shape, groups):
        return (x_shape[0] // self.orig_num_frames, self.fc.out_features)


def resnet(depth, num_classes, without_t_stride, groups, dropout, pooling_method,
           input_channels, imagenet_pretrained=True, **kwargs):
    """Factory with the signature of models/resnet.py:244-259.  imagenet_pretrained: the reference downloads torchvision's ImageNet
    weights (models/resnet.py:251-257); here they come from a LOCAL torchvision-format file (imagenet_init.py: --imagenet_weights /
    ADAMML_IMAGENET_DIR / a path passed as the flag itself) with the reference's conversion (fc dropped, a non-RGB 7x7 stem = mean over RGB
    expanded to input_channels).  True with no file configured warns once and leaves the initialisation as it is."""
    from . import imagenet_init
    model = ResNet(depth, num_frames=groups, num_classes=num_classes, without_t_stride=without_t_stride,
                   dropout=dropout, pooling_method=pooling_method, input_channels=input_channels)
    return imagenet_init.init_resnet(model, depth, input_channels, imagenet_pretrained)
