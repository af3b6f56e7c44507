"""Top-level AdaMML module on libadamml_hip: input re-layout, policy net, gated main net over segments.
Mirrors models/adamml.py:12-171 (class AdaMML, factory adamml(); same state_dict, same caller-visible surface)."""
import os

import torch
import torch.nn as nn

from . import hip, interleave, plan
from .backbone import FlatBuffers, StockDDPAware
from .common import MeanStdMixin
from .joint_resnet_mobilenetv2 import joint_resnet_mobilenetv2
from .policy_net import p_joint_mobilenet
from .runtime import clip_to_nhwc, clip_u8_to_nhwc, clip_u8_rgbdiff_to_nhwc, SyncCtx

__all__ = ['adamml']


def _frames(t):
    """Input of one backbone call over all segments: [S, B*F, H, W, C] bf16 -> [S*B*F, H, W, C] (group-major); the fp32 one-channel
    form [B, S, H, W] (data_layer) is passed through as it is."""
    return t if t.dtype == torch.float32 else t.flatten(0, 1)


class AdaMML(nn.Module, MeanStdMixin, StockDDPAware):

    def __init__(self, policy_net, main_net, num_frames, num_segments, modality, rng_policy, rng_threshold, num_classes,
                 input_channels=None):
        super().__init__()
        self._install_ddp_probe()
        self.rng_policy = rng_policy
        self.policy_net = policy_net
        self.main_net = main_net
        self.num_segments = num_segments
        self.num_frames = num_frames * num_segments
        self.num_frames_per_segment = num_frames
        self.modality = modality
        self.input_channels = input_channels
        if 'rgbdiff' in modality and 'flow' in modality:
            self.num_modality = len(modality) - 1
        else:
            self.num_modality = len(modality)
        self.p_data_idx = [self.modality.index(x) for x in self.policy_net.modality]
        self.m_data_idx = [self.modality.index(x) for x in self.main_net.modality]
        self.rng_threshold = rng_threshold
        self.decay_ratio = 0.965
        self.update_policy_net = True
        self.update_main_net = True
        self.use_side_stream = True
        self.use_wgrad_stream = os.environ.get("ADAMML_WGRAD_STREAM", "all")     # "0" | "resnet" | "all" (A/B aid)
        # inference only: run the main nets on the (segment, video) pairs the policy selected, instead of computing every
        # backbone call and multiplying the skipped ones by zero (models/adamml.py:81-86)
        self.skip_unselected = True
        self.last_skip_stats = None
        self._side = None
        self._flat_policy = FlatBuffers(self.policy_net)
        self._flat_main = FlatBuffers(self.main_net)
        if self.rng_policy:
            self.freeze_policy_net()
            del self.policy_net.fcs

    # -- data layer (models/adamml.py:42-67) as one re-layout launch per modality and consumer ----------------
    def data_layer(self, x, num_segments, p_rgb_size=(160, 160)):
        p_x, m_x = [], []
        f = self.num_frames_per_segment
        for idx, (x_, m) in enumerate(zip(x, self.modality)):
            hip.require_gpu(x_)
            if m == 'sound':
                if x_.size(-1) != x_.size(-2):
                    # legacy "consecutive segments stacked along the last dim" layout (:49-51)
                    x_ = torch.stack(x_.chunk(num_segments, dim=-1), dim=1).reshape(x_.size(0), -1, x_.size(-2),
                                                                                    x_.size(-1) // num_segments)
                c = x_.size(1) // num_segments
                if c == 1 and self._sound_f32_ok(idx, x_):
                    # one-channel spectrograms go to the MobileNetV2 stems AS THEY ARE: [B, S, H, W] fp32, segment = BatchNorm group
                    # (runtime.conv_stem1_bn) -- no re-layout pass, and the -5 +- 3 log-power range is not rounded to bf16
                    t = x_.float().contiguous()
                else:
                    t = clip_to_nhwc(x_, num_segments, 1, c)
                p_x.append(t)
                m_x.append(t)
                continue
            if x_.dtype == torch.uint8:
                # MI355X extension: decoded frames [N, H, W, S*F*C] uint8 straight from `Stack` (video_transforms.py:302-318);
                # ToTorchFormatTensor + GroupNormalize (mean / std of models/adamml.py:93-99) run inside the re-layout launch
                c = x_.size(3) // (num_segments * f)
                if m == 'rgbdiff' and c == 18:
                    # decoded RGB frames, 6 consecutive ones per frame group: the 5 difference images of the reference's loader
                    # (utils/video_dataset.py:32-38,75-84) are formed inside the re-layout launch
                    if idx in self.p_data_idx:
                        p_x.append(clip_u8_rgbdiff_to_nhwc(x_, num_segments, f, self.mean(m), self.std(m), out_hw=p_rgb_size, frame_step=2))
                    if idx in self.m_data_idx:
                        m_x.append(clip_u8_rgbdiff_to_nhwc(x_, num_segments, f, self.mean(m), self.std(m)))
                    continue
                if idx in self.p_data_idx:
                    p_x.append(clip_u8_to_nhwc(x_, num_segments, f, c, self.mean(m), self.std(m), out_hw=p_rgb_size, frame_step=2))
                if idx in self.m_data_idx:
                    m_x.append(clip_u8_to_nhwc(x_, num_segments, f, c, self.mean(m), self.std(m), cpad=self._main_cpad(m, x_.size(1), x_.size(2))))
                continue
            c = x_.size(1) // (num_segments * f)
            if idx in self.p_data_idx:
                p_x.append(clip_to_nhwc(x_, num_segments, f, c, out_hw=p_rgb_size, frame_step=2))
            if idx in self.m_data_idx:
                m_x.append(clip_to_nhwc(x_, num_segments, f, c, cpad=self._main_cpad(m, x_.size(-2), x_.size(-1))))
        return p_x, m_x, num_segments

    def _sound_f32_ok(self, idx, x_):
        """Every consumer of modality `idx` (its main net, its policy backbone) reads a [B, S, H, W] fp32 tensor directly."""
        nets = []
        if idx in self.m_data_idx:
            nets.append(self.main_net.nets[self.m_data_idx.index(idx)])
        if idx in self.p_data_idx and hasattr(self.policy_net, "joint_net"):
            nets.append(self.policy_net.joint_net.nets[self.p_data_idx.index(idx)])
        return bool(nets) and all(hasattr(n, "accepts_f32") and n.accepts_f32(x_) for n in nets)

    def forward(self, x, num_segments=None, gumbel_exponential=None):
        """x: list over modality of [N, S*F*C, H, W] fp32 GPU tensors.  Returns (logits [N, classes], decisions [N,S,M]).
        gumbel_exponential (optional, [S, M*N, 2]) replaces the device-side Exponential(1) draw for parity runs."""
        num_segments = num_segments if num_segments else self.num_segments
        dev = x[0].device
        self._flat_policy.ensure(dev)
        self._flat_main.ensure(dev)
        if self.training and torch.is_grad_enabled():
            self._flat_policy.ensure_grads()
            self._flat_main.ensure_grads()
        p_x, m_x, num_segments = self.data_layer(x, num_segments)
        # p_x / m_x are allocated on the caller's stream but read on the side streams, in backward too (the stem weight
        # gradients read them last).  They stay referenced here until the NEXT forward: by then the end-of-backward join has
        # ordered the caller's stream behind every side stream, so the caching allocator cannot hand their blocks to
        # main-stream work while a side-stream kernel still reads them (no Tensor.record_stream: see runtime._on_wgrad_stream).
        self._live_inputs = (p_x, m_x)
        if self.skip_unselected and not self.training and not torch.is_grad_enabled():
            return self._forward_skipping(x, p_x, m_x, num_segments, gumbel_exponential)
        # Two HIP streams: the ResNet(s) stay on the caller's stream; the MobileNetV2 policy nets and the sound main net
        # (hundreds of small launches) are enqueued on a side stream and overlap them.  The main nets never depend on the
        # decisions before the logit mask (models/adamml.py:81-86), so the policy runs concurrently with segment 0..S-1.
        main = torch.cuda.current_stream()
        side = self._side_stream(dev) if self.use_side_stream else None
        pside = self._side_stream(dev, 1) if self.use_side_stream else None      # policy nets: their own stream
        if side is not None:
            side.wait_stream(main)
            pside.wait_stream(main)
        # The S per-segment module calls of the reference (:151-160) run as ONE launch sequence per backbone with S
        # BatchNorm groups (per-segment batch statistics, running statistics updated segment by segment): 5x fewer
        # launches and 5x larger grids.  Host issue order: the main nets first (the ResNet is the long pole and must have
        # work queued from the first microseconds of the step), then the policy nets on their own stream.
        S = num_segments
        B = x[0].size(0)
        for net in self.main_net.nets:
            # ResNets: weight gradients on their own stream, concurrent with the data-gradient chain (runtime._on_wgrad_stream)
            if self.use_side_stream and (self.use_wgrad_stream == "all" or
                                         (self.use_wgrad_stream == "resnet" and not hasattr(net, "classifier"))):
                if net.rt.wgrad_stream is None:
                    net.rt.wgrad_stream = torch.cuda.Stream(device=dev)
            else:
                net.rt.wgrad_stream = None
        if side is not None and interleave.ENABLED and any(n.rt.sync.enabled for n in self.backbones()):
            # SyncBatchNorm on one communicator: issue the backbones round-robin, one statistics exchange per turn, so that the
            # side-stream nets' exchanges do not queue behind all of the ResNet's (interleave.py)
            jobs, calls = [], []
            for m_i in range(self.num_modality):
                net, xin = self.main_net.nets[m_i], _frames(m_x[m_i])
                st = side if self.main_net.modality[m_i] == 'sound' else main
                jobs.append(((lambda net=net, xin=xin: net.run_raw(xin, S)), st))
                calls.append((net, xin, st))
            if not self.rng_policy:
                # one job per policy backbone (each on its own stream): a round then carries the statistics of ALL MobileNetV2s of a
                # BatchNorm depth in one collective (the ResNet's travel in their own, alternating with it: interleave.GROUPS)
                pstreams = [pside] + [self._side_stream(dev, 1 + k) for k in range(1, len(self.policy_net.joint_net.nets))]
                for ps in pstreams[1:]:
                    ps.wait_stream(main)
                for k, net in enumerate(self.policy_net.joint_net.nets):
                    xin = _frames(p_x[k])
                    jobs.append(((lambda net=net, xin=xin: net.run_raw(xin, S)), pstreams[k]))
                    calls.append((net, xin, pstreams[k]))
            interleave.clips_hint[0] = B * S
            raw = interleave.run_interleaved(jobs, dev, phase="fwd")
            # the launch sequences are issued; attach each to autograd (adamml::backbone_call with the precomputed result) on its stream
            res = []
            for (net, xin, st), pre in zip(calls, raw):
                with torch.cuda.stream(st):
                    res.append(net.call(xin, S, precomputed=pre))
            stacked = res[:self.num_modality]
            if not self.rng_policy:
                with torch.cuda.stream(pside):
                    feats = res[self.num_modality:]
                    for ps, f in zip(pstreams[1:], feats[1:]):
                        pside.wait_stream(ps)
                        f.record_stream(pside)
                    joint = self.policy_net.joint_net.joint_features(feats)
                    decisions, decision_logits = self.policy_net.decide(list(joint.view(S, -1, joint.shape[-1]).unbind(0)), gumbel_exponential)
                self.last_policy_logits = decision_logits
            else:
                decisions = (torch.rand((num_segments, self.num_modality, x[0].size(0)), dtype=torch.float32, device=dev)
                             > self.rng_threshold).float()
            main.wait_stream(side)
            main.wait_stream(pside)
            for t in [decisions] + list(stacked):
                t.record_stream(main)
            final_logits = self.main_net.fuse_segments(stacked, decisions, num_segments)
            return final_logits, decisions.permute((2, 0, 1))
        stacked = self.main_net.backbone_logits([_frames(m_x[m_i]) for m_i in range(self.num_modality)], side, groups=S)
        if not self.rng_policy:
            if side is not None:
                with torch.cuda.stream(pside):
                    decisions, decision_logits = self.policy_net.decide(self.policy_net.all_segment_features(p_x),
                                                                        gumbel_exponential)
            else:
                decisions, decision_logits = self.policy_net.decide(self.policy_net.all_segment_features(p_x), gumbel_exponential)
            self.last_policy_logits = decision_logits
        else:
            decisions = (torch.rand((num_segments, self.num_modality, x[0].size(0)), dtype=torch.float32, device=dev)
                         > self.rng_threshold).float()
        if side is not None:
            main.wait_stream(side)
            main.wait_stream(pside)
            for t in [decisions] + list(stacked):
                t.record_stream(main)
        final_logits = self.main_net.fuse_segments(stacked, decisions, num_segments)
        return final_logits, decisions.permute((2, 0, 1))

    def _forward_skipping(self, x, p_x, m_x, num_segments, gumbel_exponential):
        """Eval-mode forward with decision-driven compaction: the decisions are taken first, then each main net only sees
        the (segment, video) clips whose decision is 1 -- the compute saving AdaMML is about.  Exact: eval-mode BatchNorm is
        a per-sample affine map, so a clip's logits do not depend on which other clips share the launch, and the skipped
        clips' logits are multiplied by 0 in the reference (joint_resnet_mobilenetv2.py:94)."""
        S, B, dev = num_segments, x[0].size(0), x[0].device
        if not self.rng_policy:
            decisions, decision_logits = self.policy_net.decide(self.policy_net.all_segment_features(p_x), gumbel_exponential)
            self.last_policy_logits = decision_logits
        else:
            decisions = (torch.rand((S, self.num_modality, B), dtype=torch.float32, device=dev) > self.rng_threshold).float()
        stacked, ran = [], []
        for m_i in range(self.num_modality):
            net = self.main_net.nets[m_i]
            raw32 = m_x[m_i].dtype == torch.float32                           # one-channel fp32 input [B, S, H, W] (data_layer)
            frames = _frames(m_x[m_i])                                        # [S*B*F, H, W, C], clips contiguous
            fpc = 1 if raw32 else frames.shape[0] // (S * B)
            idx = (decisions[:, m_i, :].reshape(-1) > 0.5).nonzero().flatten()      # host sync: the launch sizes depend on it
            ncls = net.fc.out_features if hasattr(net, "fc") else net.classifier[1].out_features
            out = torch.zeros(S * B, ncls, dtype=torch.float32, device=dev)
            if idx.numel() == S * B:
                out = net.forward_nhwc(frames, S if raw32 else 1)             # (eval BatchNorm: the grouping is immaterial)
            nsel = int(idx.numel())
            if 0 < nsel < S * B:
                if plan.ENABLED and plan.EVAL_ENABLED and idx.numel() % 8:
                    # launch plans are per call shape: pad the selected clips to a multiple of 8 (the first one repeated; eval BatchNorm is
                    # per sample, the duplicates' logits overwrite identical values) so that few distinct shapes occur
                    idx = torch.cat([idx, idx[:1].expand(8 - idx.numel() % 8)])
                if raw32:
                    sel = frames.transpose(0, 1).reshape(S * B, 1, *frames.shape[2:]).index_select(0, idx)      # [n, 1, H, W]: one group
                else:
                    sel = frames.view(S * B, fpc, *frames.shape[1:]).index_select(0, idx).flatten(0, 1)
                out.index_copy_(0, idx, net.forward_nhwc(sel, 1))
            stacked.append(out)
            ran.append(nsel)
        self.last_skip_stats = {"clips": S * B, "executed_per_modality": ran}
        return self.main_net.fuse_segments(stacked, decisions, S), decisions.permute((2, 0, 1))

    def _main_cpad(self, modality, h, w):
        """Channel padding the main net of `modality` wants for its NHWC input (ResNet.input_cpad: 4-channel pixels for the 7x7 stem
        kernel), None = the generic multiple of 8."""
        net = self.main_net.nets[self.main_net.modality.index(modality)]
        return net.input_cpad(h, w) if hasattr(net, "input_cpad") else None

    def _side_stream(self, dev, idx=0):
        if self._side is None or self._side[0].device != dev:
            self._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        while len(self._side) <= idx:
            self._side.append(torch.cuda.Stream(device=dev))
        return self._side[idx]

    @property
    def network_name(self):
        name = 'adamml'
        if self.rng_policy:
            name += '-rng-{:.1f}'.format(self.rng_threshold)
        else:
            name += '-{}'.format(self.policy_net.network_name)
        name += '-{}'.format(self.main_net.network_name)
        return name

    def decay_temperature(self, decay_ratio=None):
        self.policy_net.decay_temperature(decay_ratio if decay_ratio else self.decay_ratio)

    def freeze_policy_net(self):
        self.update_policy_net = False
        for param in self.policy_net.parameters():
            param.requires_grad = False

    def unfreeze_policy_net(self):
        self.update_policy_net = True
        for param in self.policy_net.parameters():
            param.requires_grad = True

    def freeze_main_net(self):
        self.update_main_net = False
        for param in self.main_net.parameters():
            param.requires_grad = False

    def unfreeze_main_net(self):
        self.update_main_net = True
        for param in self.main_net.parameters():
            param.requires_grad = True

    # -- MI355X extensions (not in the reference surface) -------------------------------------------------------
    def backbones(self):
        nets = list(self.main_net.nets)
        if hasattr(self.policy_net, "joint_net"):
            nets += list(self.policy_net.joint_net.nets)
        return nets

    def enable_sync_bn(self, group=None, force=False):
        """SyncBatchNorm (train_adamml.py:126-127): BN statistic sums are all-reduced over RCCL.
        By default every backbone uses the caller's process group, i.e. ONE RCCL communicator: torch serialises all its
        collectives on one internal stream, so the side-stream backbones' exchanges queue behind the ResNet's.
        ADAMML_SYNCBN_GROUPS=per_net (opt-in; every rank must set it) gives each backbone its own communicator
        (torch.distributed.new_group over the same ranks): the nets' exchanges then proceed independently.  All ranks issue
        the collectives of all communicators in the same host order (the step is deterministic), which is what RCCL
        requires of concurrently used communicators -- but a device-synchronising runtime call on one rank (a hipMalloc of the
        caching allocator during the first steps) while two communicators have kernels in flight can still deadlock, so it
        stays opt-in until it has been run on an 8-GPU node.  With 2 gloo ranks on one MI355X (B = 8) it halves the step
        (333 -> 164 ms), which is the size of the serialisation it removes."""
        import torch.distributed as dist
        per_net = os.environ.get("ADAMML_SYNCBN_GROUPS", "") == "per_net" and dist.is_available() and dist.is_initialized()
        ranks = list(range(dist.get_world_size(group))) if per_net else None
        if per_net and group is not None:
            ranks = dist.get_process_group_ranks(group)
        for net in self.backbones():
            net.rt.sync = SyncCtx(dist.new_group(ranks) if per_net else group, True, force)

    def flat_grad_buffers(self):
        """Flat fp32 gradient buffers of the TRAINABLE sub-networks (one RCCL all-reduce each)."""
        out = []
        for fb, on in ((self._flat_policy, self.update_policy_net), (self._flat_main, self.update_main_net)):
            if on and fb.flat_grad is not None:
                out.append(fb.flat_grad)
        return out


def adamml(groups, modality, input_channels, num_segments, rng_policy, rng_threshold, causality_modeling,
           num_classes, depth, without_t_stride, dropout, pooling_method, fusion_point,
           unimodality_pretrained, learnable_lf_weights, **kwargs):
    """Factory with the signature of models/adamml.py:134-171 (tolerates the whole argparse namespace as kwargs)."""
    if 'rgbdiff' in modality and 'flow' in modality:
        p_modality = [x for x in modality if x != 'flow']
        m_modality = [x for x in modality if x != 'rgbdiff']
        p_input_channels = [x for x, m in zip(input_channels, modality) if m != 'flow']
        m_input_channels = [x for x, m in zip(input_channels, modality) if m != 'rgbdiff']
    else:
        p_modality, m_modality = modality, modality
        p_input_channels, m_input_channels = input_channels, input_channels
    policy_net = p_joint_mobilenet(num_frames=max(1, groups // 2), modality=p_modality,
                                   input_channels=p_input_channels, causality_modeling=causality_modeling)
    main_net = joint_resnet_mobilenetv2(depth=depth, num_classes=num_classes, without_t_stride=without_t_stride,
                                        groups=groups, dropout=dropout, pooling_method=pooling_method,
                                        input_channels=m_input_channels, fusion_point=fusion_point, modality=m_modality,
                                        unimodality_pretrained=unimodality_pretrained,
                                        learnable_lf_weights=learnable_lf_weights)
    return AdaMML(policy_net, main_net, num_frames=groups, num_segments=num_segments, modality=modality,
                  rng_policy=rng_policy, rng_threshold=rng_threshold, num_classes=num_classes,
                  input_channels=input_channels)
