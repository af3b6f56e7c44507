"""Main net of AdaMML: one backbone per modality (ResNet for visual streams, MobileNetV2 for sound), per-sample
decision masking of the logits and learnable late fusion.
Mirrors models/joint_resnet_mobilenetv2.py:11-157 (fusion_point='logits', the only path the policy can drive)."""
import torch
import torch.nn as nn

from .common import MeanStdMixin
from .functional import fuse_segments
from .resnet import ResNet
from .sound_mobilenet_v2 import MobileNetV2

__all__ = ['joint_resnet_mobilenetv2']


class JointResNetMobileNetV2(nn.Module, MeanStdMixin):

    def __init__(self, depth, num_frames, modality, num_classes=1000, dropout=0.5, zero_init_residual=False,
                 without_t_stride=False, pooling_method='max', input_channels=None, fusion_point='logits',
                 learnable_lf_weights=False):
        super().__init__()
        if fusion_point != 'logits':
            # models/joint_resnet_mobilenetv2.py:96 -- the decision mask only exists for 'logits'; AdaMML never uses 'fc2'
            raise ValueError("only support logits mode")
        self.depth = depth
        self.num_frames = num_frames
        self.without_t_stride = without_t_stride
        self.pooling_method = pooling_method
        self.fusion_point = fusion_point
        self.modality = modality
        self.learnable_lf_weights = learnable_lf_weights
        self.nets = nn.ModuleList()
        self.last_channels = []
        for i, m in enumerate(modality):
            if m != 'sound':
                net = ResNet(depth, num_frames, num_classes, dropout, zero_init_residual, without_t_stride, pooling_method,
                             input_channels[i])
                self.last_channels.append(2048)
            else:
                net = MobileNetV2(num_classes, dropout=dropout, input_channels=input_channels[i])
                self.last_channels.append(net.last_channel)
            net.flat_owner = None           # parameters are flattened once for the whole main net
            self.nets.append(net)
        self.lf_weights = None
        if learnable_lf_weights:
            init_prob = 1.0 / len(self.modality)
            self.lf_weights = nn.Parameter(torch.tensor([init_prob] * (len(self.modality) - 1)))

    @property
    def network_name(self):
        name = 'joint_resnet-{}_mobilenet_v2-{}'.format(self.depth, self.fusion_point)
        if self.lf_weights is not None:
            name += "-llf" if self.learnable_lf_weights else '-llfc'
        if not self.without_t_stride:
            name += "-ts-{}".format(self.pooling_method)
        return name

    def backbone_logits(self, multi_modalities, side_stream=None, groups=1):
        """Per-modality logits of `groups` segments stacked along dim 0 (each segment = one reference module call with
        its own BatchNorm statistics).  With a side stream the MobileNetV2 (sound) backbone is enqueued there so that its
        many small launches overlap the ResNet's HBM-bound kernels; the caller joins the streams."""
        out = []
        for i, x in enumerate(multi_modalities):
            net = self.nets[i]
            if side_stream is not None and self.modality[i] == 'sound':
                with torch.cuda.stream(side_stream):
                    out.append(net.forward_nhwc(x, groups))
            else:
                out.append(net.forward_nhwc(x, groups))          # [G*B, classes] fp32
        return out

    def fuse_segments(self, logits, decisions, num_segments):
        """logits: list over modality of [S*B, classes] (segment-major); decisions [S,M,B] or None -> [B, classes]:
        decision mask (:94), late fusion with cat(lf_weights, 1 - sum) or the plain mean (:112-127) and the mean over the
        segments (models/adamml.py:88), one adamml_fusion_fwd launch."""
        return fuse_segments(list(logits), decisions, self.lf_weights, num_segments)

    def fuse(self, logits, decisions=None):
        """One segment: logits list of [B, classes], decisions [M,B] or None."""
        return self.fuse_segments(logits, decisions.reshape(1, len(logits), -1) if decisions is not None else None, 1)

    def forward(self, multi_modalities, decisions=None):
        """multi_modalities: list of NHWC bf16 frame tensors of ONE segment; decisions [M, B] or None."""
        return self.fuse(self.backbone_logits(multi_modalities), decisions)


def _load_unimodal_checkpoints(model, paths):
    """Contract of models/joint_resnet_mobilenetv2.py:141-155 (--unimodality_pretrained): one train_unimodal.py checkpoint per backbone, in
    backbone order, each a dict whose 'state_dict' carries DataParallel's `module.` prefix; loaded strictly.  A wrong count is a ValueError
    with the reference's message.  (fusion_point is always 'logits' here, so no head is dropped.)"""
    if not paths:
        return
    if len(paths) != len(model.nets):
        raise ValueError("the number of pretrained models is incorrect.")
    for net, path in zip(model.nets, paths):
        print("Loading unimodality pretrained model from: {}".format(path))
        weights = torch.load(path, map_location='cpu')['state_dict']
        net.load_state_dict({name.replace("module.", ""): t for name, t in weights.items()}, strict=True)
        if hasattr(net, "mark_weights_dirty"):
            net.mark_weights_dirty()


def joint_resnet_mobilenetv2(depth, num_classes, without_t_stride, groups, dropout, pooling_method, input_channels,
                             fusion_point, modality, unimodality_pretrained, learnable_lf_weights, **kwargs):
    model = JointResNetMobileNetV2(depth, num_frames=groups, num_classes=num_classes, without_t_stride=without_t_stride,
                                   dropout=dropout, pooling_method=pooling_method, input_channels=input_channels,
                                   fusion_point=fusion_point, modality=modality, learnable_lf_weights=learnable_lf_weights)
    _load_unimodal_checkpoints(model, list(unimodality_pretrained))
    return model
