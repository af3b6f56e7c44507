"""`torch.library` registration of the HIP hot path: namespace `adamml` (SURVEY.md section 8b "who calls it").

  adamml::backbone_call   one backbone invocation (ResNet-50 / Sound-MobileNetV2 / policy MobileNetV2: the whole fused launch
                          sequence of csrc/*.hip) as ONE operator with a registered autograd formula (the recorded tape in
                          reverse) and a fake (meta) implementation -- what nn.Module.forward of every backbone dispatches to.
  adamml::clip_to_nhwc    AdaMML.data_layer re-layout / bilinear resize (models/adamml.py:53-65)
  adamml::gemm_f32        fp32 GEMM (+bias, +ReLU) behind the joint FCs / LSTM gates / classifier heads
  adamml::conv_fwd        conv (+lazy BatchNorm/activation of the producer in the loader) + per-channel statistics epilogue
  adamml::temporal_pool   models/common.py:4-33 on NHWC frames

Every operator has `register_fake`, so shapes / dtypes propagate under FakeTensorMode (and torch.compile tracing) without a
GPU; the real implementations call the C ABI (adamml_amd/hip.py) and raise when the library or the GPU is missing."""
import weakref
from typing import List, Optional

import torch
from torch.library import custom_op

from . import hip
from .hip import ConvDesc, call, ptr, STAT_SLOTS
from ctypes import byref

_NETS = weakref.WeakValueDictionary()        # handle -> HipBackbone
_next_handle = [1]


def register_net(net):
    h = _next_handle[0]
    _next_handle[0] += 1
    _NETS[h] = net
    return h


def _net(handle):
    net = _NETS.get(handle)
    if net is None:
        raise RuntimeError("adamml::backbone_call: unknown backbone handle %d" % handle)
    return net


# ---------------------------------------------------------------------------------------------------- backbone call
@custom_op("adamml::backbone_call", mutates_args=())
def backbone_call(anchor: torch.Tensor, x: torch.Tensor, params: List[torch.Tensor], handle: int, groups: int,
                  need_grad: bool) -> torch.Tensor:
    """x: NHWC bf16 frames of `groups` stacked module calls; params: the backbone's trainable parameters when their gradients
    are to be delivered THROUGH autograd (stock DistributedDataParallel / torch.optim), else empty (the weight-gradient kernels
    then accumulate straight into the pre-attached flat .grad views).  BatchNorm running statistics of the backbone are
    updated in place as a side effect, exactly like nn.BatchNorm2d in train mode."""
    net = _net(handle)
    pre, net._precomputed = net._precomputed, None
    if pre is not None:            # the launch sequence already ran as a coroutine of interleave.run_interleaved (HipBackbone.run_raw)
        out, tape = pre
        out = out.clone()          # (an operator may not return its argument; [clips, classes] fp32)
    else:
        out, tape = net.run_planned(x, groups, need_grad)
        if tape.__class__.__name__ == "PlanTape":
            out = out.clone()      # (the plan's output buffer is rewritten by the next replay)
    net._pending_tape = tape
    return out


@backbone_call.register_fake
def _(anchor, x, params, handle, groups, need_grad):
    return x.new_empty(_net(handle).out_shape(tuple(x.shape), groups), dtype=torch.float32)


def _backbone_setup(ctx, inputs, output):
    anchor, x, params, handle, groups, need_grad = inputs
    net = _net(handle)
    ctx.net, ctx.tape, ctx.nparams = net, net._pending_tape, len(params)
    ctx.params = list(params)
    net._pending_tape = None


def _backbone_backward(ctx, g):
    from . import backbone
    grads = backbone.run_backward(ctx.net, ctx.tape, g, ctx.params)
    return None, None, grads, None, None, None


backbone_call.register_autograd(_backbone_backward, setup_context=_backbone_setup)


# ---------------------------------------------------------------------------------------------------- leaf operators
def _pad8(c):
    return (c + 7) // 8 * 8


@custom_op("adamml::clip_to_nhwc", mutates_args=())
def clip_to_nhwc(x: torch.Tensor, num_segments: int, frames: int, channels: int, out_h: int, out_w: int, frame_step: int) -> torch.Tensor:
    """[B, S*F*C, H, W] fp32 -> [S, B*Fk, out_h, out_w, pad8(C)] bf16 (bilinear, align_corners=False, when the size changes)."""
    hip.require_gpu(x)
    b, sfc, h, w = x.shape
    if sfc != num_segments * frames * channels:
        raise RuntimeError("clip_to_nhwc: channel dim %d != S*F*C = %d*%d*%d" % (sfc, num_segments, frames, channels))
    fk = (frames + frame_step - 1) // frame_step
    x = x.contiguous().float()
    y = torch.empty(num_segments, b * fk, out_h, out_w, _pad8(channels), dtype=torch.bfloat16, device=x.device)
    call("adamml_clip_to_nhwc", ptr(x), ptr(y), b, num_segments, frames, channels, h, w, out_h, out_w, frame_step, _pad8(channels))
    return y


@clip_to_nhwc.register_fake
def _(x, num_segments, frames, channels, out_h, out_w, frame_step):
    fk = (frames + frame_step - 1) // frame_step
    return x.new_empty((num_segments, x.shape[0] * fk, out_h, out_w, _pad8(channels)), dtype=torch.bfloat16)


@custom_op("adamml::gemm_f32", mutates_args=())
def gemm_f32(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor], act: int, trans_a: bool, trans_b: bool) -> torch.Tensor:
    """act(op(a) @ op(b) + bias): trans_a False: a [M,K] / True: [K,M]; trans_b True: b [N,K] / False: [K,N]."""
    from .runtime import gemm_f32 as _g
    hip.require_gpu(a)
    return _g(a, b, bias=bias, act=act, trans_a=trans_a, trans_b=trans_b)


@gemm_f32.register_fake
def _(a, b, bias, act, trans_a, trans_b):
    m = a.shape[1] if trans_a else a.shape[0]
    n = b.shape[0] if trans_b else b.shape[1]
    return a.new_empty((m, n), dtype=torch.float32)


@custom_op("adamml::conv_fwd", mutates_args=("stats",))
def conv_fwd(x: torch.Tensor, w_packed: torch.Tensor, in_scale: Optional[torch.Tensor], in_shift: Optional[torch.Tensor],
             stats: Optional[torch.Tensor], kh: int, kw: int, stride: int, pad: int, in_act: int, groups: int) -> torch.Tensor:
    """x [G*N,H,W,Cin] bf16 (value = in_act(in_scale*x + in_shift) when in_scale is given), w_packed [Cout, kh*kw*Cin] bf16 ->
    raw conv output [G*N,OH,OW,Cout] bf16; stats [G, ADAMML_STAT_SLOTS, 2*Cout] fp64 (zeroed by the caller) receives the
    per-channel sum / sum of squares of the stored output (BatchNorm batch statistics of the following layer)."""
    hip.require_gpu(x)
    n, h, w, cin = x.shape
    cout = w_packed.shape[0]
    oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    in_gstride = in_scale.stride(0) if (in_scale is not None and in_scale.dim() > 1) else 0     # [G, Cin] vectors: one pair per group
    d = ConvDesc(n // groups, h, w, cin, oh, ow, cout, kh, kw, stride, pad, 1, in_act, 0, groups, in_gstride)
    y = torch.empty(n, oh, ow, cout, dtype=torch.bfloat16, device=x.device)
    call("adamml_conv_fwd", byref(d), ptr(x.contiguous()), ptr(w_packed), ptr(in_scale), ptr(in_shift), ptr(y), ptr(stats))
    return y


@conv_fwd.register_fake
def _(x, w_packed, in_scale, in_shift, stats, kh, kw, stride, pad, in_act, groups):
    n, h, w, _ = x.shape
    return x.new_empty((n, (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1, w_packed.shape[0]), dtype=torch.bfloat16)


@custom_op("adamml::temporal_pool", mutates_args=())
def temporal_pool(x: torch.Tensor, frames: int, mode: int, groups: int) -> torch.Tensor:
    """[G*N*T,H,W,C] bf16 -> [G*N*T',H,W,C], T' = (T-1)//2+1; mode 0 max / 1 avg (3-tap, stride 2, pad 1 over the frame axis)."""
    hip.require_gpu(x)
    nt, h, w, c = x.shape
    nb, to = nt // frames, (frames - 1) // 2 + 1
    y = torch.empty(nb * to, h, w, c, dtype=torch.bfloat16, device=x.device)
    call("adamml_temporal_pool_fwd", ptr(x.contiguous()), None, None, 0, 0, ptr(y), nb // groups, frames, h * w * c, c, mode, groups)
    return y


@temporal_pool.register_fake
def _(x, frames, mode, groups):
    nt, h, w, c = x.shape
    return x.new_empty((nt // frames * ((frames - 1) // 2 + 1), h, w, c))


def stat_buffer(groups, channels, device):
    """Zeroed accumulator for adamml::conv_fwd's statistics epilogue."""
    return torch.zeros(groups, STAT_SLOTS, 2 * channels, dtype=torch.float64, device=device)
