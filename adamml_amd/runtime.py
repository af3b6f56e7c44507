"""Host-side executor of the HIP hot path: lazy-BatchNorm activation tensors, a reverse tape,
and one Python call per fused device op (each a single C-ABI launch into libadamml_hip.so).

Data layout in HBM: activations NHWC bf16 (channels padded to x8); every conv writes its
RAW output once and accumulates the per-channel sum / sum-of-squares in its epilogue; the consumer applies
the producer's BatchNorm scale/shift (+ReLU/ReLU6) while staging its input, so normalised activations are
never materialised except at residual adds.  BatchNorm statistics, scale/shift and master weights are fp32.
"""
import collections
import math
import os
import torch
import torch.distributed as dist

from . import hip, interleave
from .hip import ConvDesc, ACT_NONE, ACT_RELU, ACT_RELU6, STAT_SLOTS, call, ptr
from ctypes import byref

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# roles of the MFMA launches in bench.py's per-role roofline (4th element of hip.next_meta): each role has ONE bounding roof
R_1X1 = "conv1x1_streaming"            # plain 1x1 forward / data gradient (conv_gemm_kernel MODE 0): HBM-bound
R_FUSED = "conv1x1_fused_streaming"    # 1x1 with a fused BatchNorm / residual / algebraic epilogue or loader (FADD, RES, DUAL, CAT, stream)
R_KXK = "convKxK_mfma"                 # 3x3 / 7x7 / strided data gradients (MODE 1-3, conv3x3_c64, conv_stem): MFMA-bound
R_WGRAD = "weight_gradient"            # contractions over the pixel axis: weight gradients, g'^T a products, Gram matrices


def pad8(c):
    return (c + 7) // 8 * 8


class Lazy:
    """Activation tensor [G*N,H,W,C] bf16 (G BatchNorm groups, group-major) whose value is act(scale*data + shift)
    (scale None -> data); group g uses scale + g*gs, shift + g*gs (gs == 0: one pair for all groups)."""
    __slots__ = ("data", "scale", "shift", "gs", "act", "grad", "requires_grad", "vec", "src", "pre_sums", "res", "res_done", "pool_grad", "alg", "sums_partial",
                 "recompute", "_shape", "alg_in", "prod", "next_pre")

    def __init__(self, data, scale=None, shift=None, act=ACT_NONE, requires_grad=True, gs=0):
        self.data, self.scale, self.shift, self.act, self.gs = data, scale, shift, act, gs
        self.grad = None            # gradient w.r.t. the ACTIVATED value, bf16, same shape
        self.requires_grad = requires_grad
        self.vec = None             # train-mode BatchNorm vectors [G,4,C] (scale, shift, mean, invstd) of a lazy tensor
        self.src = None             # lazy tensor this plain tensor is the materialisation of
        self.pre_sums = None        # BatchNorm-backward sums already accumulated by the producer of .grad
        self.pool_grad = None       # (g_y, idx, OH, OW): .grad = routed max-pool gradient, recomputed by the BatchNorm backward
        self.res = None             # (z, idn, act, idn_sole) of the residual add that produced this tensor
        self.alg = False            # BatchNorm backward of this conv output is algebraic: producers of .grad need not read .data
        self.sums_partial = False   # .pre_sums holds sum(g') only; sum(g' zhat) is derived from g'^T a (adamml_alg_sumfix)
        self.res_done = False       # .grad is already act-masked and the add's BatchNorm-backward sums are in place
        self.recompute = None       # data is None: the raw tensor was never written (conv_bn_add); recompute() materialises it
        self.alg_in = None          # (lazy input of the conv that produced this tensor, its descriptor): algebraic backward only
        self.prod = None            # P = g'^T a [G, C, Cin] already accumulated by the producer of .grad (adamml_conv_bwd_data_res_prod)
        self.next_pre = None        # (ConvState, raw output, statistics) of a conv of THIS tensor that its producer already ran (conv_bn_add next_cs)
        self._shape = None

    @property
    def shape(self):
        return self.data.shape if self.data is not None else self._shape

    def ensure_data(self):
        """The raw tensor, materialising it first if the forward never stored it (rare fallback paths of conv_bn_add)."""
        if self.data is None:
            self.data = self.recompute()
        return self.data


class Tape:
    """Reverse-mode tape for one backbone call; closures run in reverse order."""

    def __init__(self, need_grad):
        self.need_grad = need_grad
        self.fns = []

    def record(self, fn):
        if self.need_grad:
            self.fns.append(fn)

    def backward(self):
        fns, self.fns = self.fns, []
        for fn in reversed(fns):
            fn()


class Arena:
    """Bump allocator over one zero-initialised fp64 buffer (BatchNorm statistic accumulators):
    one memset per backbone pass instead of one per layer."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0

    def reset(self, device):
        need = max(self.high, 1)
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.zeros(need, dtype=torch.float64, device=device)
            if hip.recorder is not None:
                # a launch plan must replay the memset of the FINAL buffer: an arena that is first sized (or grown) inside the recorded
                # call -- warm-up forwards that had no backward, another call shape raising `high` in between -- would be accumulated
                # into without ever being re-zeroed at replay.  This call stays eager; with its size known now the arena no longer moves,
                # so the key is recorded again after the usual warm-up (`retry`).
                hip.recorder.failed = "statistics arena (re)allocated while recording"
                hip.recorder.retry = True       # (the arena has its size now: record again after the usual warm-up)
        else:
            self.buf[:need].zero_()
            if hip.recorder is not None:                 # (a launch plan repeats the memset; the buffer has its final size after warm-up)
                hip.recorder.zero(self.buf[:need], hip._stream())
        self.off = 0

    def take(self, n):
        if self.buf is None or self.off + n > self.buf.numel():
            # first pass (size unknown yet): grow by individual zeroed allocations, remember the total
            t = torch.zeros(n, dtype=torch.float64, device=self.buf.device if self.buf is not None else "cuda")
            if hip.recorder is not None:
                hip.recorder.failed = "statistics arena overflowed while recording"
                hip.recorder.retry = True
        else:
            t = self.buf[self.off:self.off + n]
        self.off += n
        self.high = max(self.high, self.off)
        return t


class SyncCtx:
    """SyncBatchNorm plumbing (train_adamml.py:126-127): statistic sums are all-reduced over RCCL.
    force=True keeps the exchange on for a ONE-rank group too (the collectives are identities then): how the communication-stream
    / event ordering of the SyncBN path is exercised against a stream-asynchronous backend on a single GPU (tests/test_rccl_gpu.py)."""

    def __init__(self, group=None, enabled=False, force=False):
        self.group = group
        ok = enabled and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ok else 1
        self.enabled = bool(ok and (self.world > 1 or force))

    def reduce(self, t, C, groups):
        """t: [groups][STAT_SLOTS][2C] slot-interleaved sums.  Returns (sums, nslots) for the finalize kernel: under
        SyncBN the slots are collapsed first so that only groups*2C doubles cross xGMI, in ONE all-reduce -- coalesced with the
        exchanges the other backbones have pending in this round when the backbones are issued in lock-step (interleave.py):
        one collective per BatchNorm DEPTH instead of one per BatchNorm."""
        if not self.enabled:
            return t, STAT_SLOTS
        return interleave.exchange_stats(t, C, groups, self.group), 1


class NetRT:
    """Per-backbone runtime state: arenas, sync context, mode flags."""

    def __init__(self):
        self.fwd_arena = Arena()
        self.bwd_arena = Arena()
        self.sync = SyncCtx()
        self.training = False
        self.groups = 1
        self.tape = Tape(False)
        self.wgrad_stream = None     # optional side stream for the weight-gradient kernels (_on_wgrad_stream)
        self.wgrad_pending = collections.deque()      # (event, operand tensors) of weight-gradient launches still in flight
        self.capture = None          # test aid: dict id(conv weight Parameter) -> list of raw conv outputs of this forward
        self.pre_pending = 0         # next-conv results a fused kernel computed ahead (Lazy.next_pre) that no conv_bn has claimed yet
        self.pre_dropped = 0         # ... and were never claimed by the end of a forward (a silent fallback: tests assert 0)
        self.state_gen = 0           # bumped whenever BatchNorm tensors change behind torch's back (raw-pointer writes)

    def begin_forward(self, device, training, need_grad, groups=1):
        """groups = number of independent BatchNorm groups batched in this call: the S per-segment module calls of the
        reference (models/adamml.py:151-160) become one launch sequence with per-group statistics."""
        self.training = training
        self.groups = groups
        self.tape = Tape(need_grad)
        self.touched_bns = []
        self.pre_pending = 0
        if training:
            self.fwd_arena.reset(device)
        return self.tape

    def end_forward(self):
        """nn.BatchNorm2d bookkeeping: num_batches_tracked += 1 per group for every BN evaluated in train mode
        (one multi-tensor launch per backbone call instead of one per layer)."""
        self.pre_dropped += self.pre_pending          # (a precomputed next conv nobody consumed: its launch and its statistics slot were wasted)
        self.pre_pending = 0
        if self.training and self.touched_bns:
            torch._foreach_add_([b.num_batches_tracked for b in self.touched_bns], self.groups)
            self.state_gen += 1      # adamml_bn_finalize rewrote running_mean / running_var through raw pointers
            if hip.recorder is not None:
                nbt, grp = [b.num_batches_tracked for b in self.touched_bns], self.groups

                def bump():
                    torch._foreach_add_(nbt, grp)
                    self.state_gen += 1
                hip.recorder.post_fwd.append(bump)
        self.touched_bns = []


# ------------------------------------------------------------------------------------------ conv state
class ConvState:
    """Device-side operands derived from one fp32 OIHW master weight: bf16 GEMM packs."""

    def __init__(self, weight, stride, pad, depthwise=False):
        self.weight = weight
        self.cout, self.cin_true, self.kh, self.kw = weight.shape
        self.stride, self.pad = stride, pad
        self.depthwise = depthwise
        if depthwise:
            self.cin_true = self.cout
        self.cin = pad8(self.cin_true)
        self.w_fwd = None
        self.w_dgrad = None
        self.w_stem = None
        # ResNet stem (7x7/2 of an RGB-like image): LDS-patch kernel with its own weight pack (csrc/conv_stem.hip)
        self.stem = (not depthwise and self.kh == 7 and self.kw == 7 and stride == 2 and pad == 3 and self.cout == 64
                     and self.cin_true <= 4)
        self.version = -1

    def pack_rows(self, need_dgrad):
        """(w, out, cout, cin_true, cin_pad, kh, kw, mode, n_out_elements) of every pack adamml_pack_conv_weights_batched
        has to refresh for this conv (allocates the pack buffers on first use); the stem's own pack is not in the table."""
        w = self.weight
        rows = []
        if self.depthwise:
            if self.w_fwd is None:
                self.w_fwd = torch.empty(self.kh * self.kw, self.cout, dtype=torch.float32, device=w.device)
            return [(w, self.w_fwd, self.cout, 1, 1, self.kh, self.kw, 2, self.kh * self.kw * self.cout)]
        n = self.cout * self.kh * self.kw * self.cin
        if self.w_fwd is None:
            self.w_fwd = torch.empty(self.cout, self.kh * self.kw * self.cin, dtype=torch.bfloat16, device=w.device)
        rows.append((w, self.w_fwd, self.cout, self.cin_true, self.cin, self.kh, self.kw, 0, n))
        if need_dgrad:
            if self.w_dgrad is None:
                self.w_dgrad = torch.empty(self.cin, self.kh * self.kw * self.cout, dtype=torch.bfloat16, device=w.device)
            rows.append((w, self.w_dgrad, self.cout, self.cin_true, self.cin, self.kh, self.kw, 1, n))
        return rows

    def repack_stem(self):
        if self.stem:
            w = self.weight
            if self.w_stem is None:
                self.w_stem = torch.empty(self.cout, 7 * 8 * 4, dtype=torch.bfloat16, device=w.device)
            call("adamml_pack_stem_weight", ptr(w), ptr(self.w_stem), self.cout, self.cin_true)

    def repack(self, need_dgrad):
        w = self.weight
        if self.depthwise:
            if self.w_fwd is None:
                self.w_fwd = torch.empty(self.kh * self.kw, self.cout, dtype=torch.float32, device=w.device)
            call("adamml_pack_conv_weight", ptr(w), ptr(self.w_fwd), self.cout, 1, 1, self.kh, self.kw, 2)
            return
        if self.w_fwd is None:
            self.w_fwd = torch.empty(self.cout, self.kh * self.kw * self.cin, dtype=torch.bfloat16, device=w.device)
        call("adamml_pack_conv_weight", ptr(w), ptr(self.w_fwd), self.cout, self.cin_true, self.cin, self.kh, self.kw, 0)
        if self.stem:
            if self.w_stem is None:
                self.w_stem = torch.empty(self.cout, 7 * 8 * 4, dtype=torch.bfloat16, device=w.device)
            call("adamml_pack_stem_weight", ptr(w), ptr(self.w_stem), self.cout, self.cin_true)
        if need_dgrad:
            if self.w_dgrad is None:
                self.w_dgrad = torch.empty(self.cin, self.kh * self.kw * self.cout, dtype=torch.bfloat16, device=w.device)
            call("adamml_pack_conv_weight", ptr(w), ptr(self.w_dgrad), self.cout, self.cin_true, self.cin, self.kh, self.kw, 1)

    def desc(self, x_shape, act, groups=1, in_gstride=0):
        n, h, w, c = x_shape
        oh = (h + 2 * self.pad - self.kh) // self.stride + 1
        ow = (w + 2 * self.pad - self.kw) // self.stride + 1
        return ConvDesc(n // groups, h, w, c, oh, ow, self.cout, self.kh, self.kw, self.stride, self.pad, 1, act, 0, groups,
                        in_gstride)


def _bn_vectors(rt, bn, stats, count, C, device):
    G = rt.groups
    vec = torch.empty(G, 4, C, dtype=torch.float32, device=device)
    stats, nslots = rt.sync.reduce(stats, C, G)
    call("adamml_bn_finalize", ptr(stats), nslots, G, float(count * rt.sync.world), ptr(bn.weight), ptr(bn.bias),
         ptr(bn.running_mean), ptr(bn.running_var), BN_MOMENTUM, BN_EPS, ptr(vec), C)
    return vec


def _bn_eval_vectors(rt, bn, C, device):
    """Eval-mode BatchNorm as a per-channel affine (scale, shift).  Cached on the module while its four tensors are
    unchanged (inference runs the same weights over and over: 157 tiny launches per forward otherwise).  The HIP kernels
    write gamma / beta (fused optimizers) and the running statistics (adamml_bn_finalize) through raw pointers, which
    torch's `_version` counters do not see: `rt.state_gen` counts those writes (NetRT.end_forward, mark_weights_dirty)."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr(), str(device), rt.state_gen)
    cached = getattr(bn, "_hip_eval_vec", None)
    if cached is not None and cached[0] == key and not hip.profiler and hip.recorder is None:
        return cached[1]
    vec = torch.empty(2, C, dtype=torch.float32, device=device)
    call("adamml_bn_eval_affine", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), BN_EPS,
         ptr(vec[0]), ptr(vec[1]), C)
    if hip.recorder is None:           # (a launch plan owns the vectors it recomputes at every replay: not shared through the module)
        bn._hip_eval_vec = (key, vec)
    return vec


def _bn_backward(rt, out, y, vec, bn, act, count, defer_apply=False):
    """Turns out.grad (w.r.t. the activated value) into dz (w.r.t. the raw conv output); accumulates dgamma/dbeta.
    defer_apply=True stops after the finalize step and returns (g, coef, aff): the caller's data-gradient kernel applies
    dz = k0 (g - k1 - zhat k2) in its loader (adamml_conv_bwd_data_dual) instead of a separate pass over g and z."""
    g = out.grad
    out.grad = None
    n, oh, ow, C = y.shape
    G = rt.groups
    P = n // G * oh * ow                    # pixels per group
    if out.pool_grad is not None:
        # the only consumer was a 3x3/2 max-pool: its routed gradient is recomputed from the pooled gradient and the
        # arg-max indices inside both BatchNorm-backward passes instead of being written and re-read twice
        gy, idx, zsel, poh, pow_ = out.pool_grad
        out.pool_grad = None
        sums = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS)
        if zsel is not None:                # sums over the windows: each pooled gradient lands on exactly one input pixel
            call("adamml_bn_bwd_reduce", ptr(gy), ptr(zsel), ptr(vec), act, ptr(sums), n // G * poh * pow_, C, G)
        else:
            call("adamml_maxpool2d_bwd_bn_reduce", ptr(gy), ptr(idx), ptr(y), ptr(vec), act, ptr(sums), n // G, oh, ow, C, poh, pow_, G)
        sums, nslots = rt.sync.reduce(sums, C, G)
        coef = torch.empty(G, 3, C, dtype=torch.float32, device=y.device)
        train_bn = bn.weight.requires_grad
        call("adamml_bn_bwd_finalize", ptr(sums), nslots, G, float(count * rt.sync.world), ptr(bn.weight), ptr(vec),
             ptr(bn.weight.grad) if train_bn else None, ptr(bn.bias.grad) if train_bn else None, ptr(coef), C, 1.0 / rt.sync.world)
        dz = torch.empty_like(y)
        call("adamml_maxpool2d_bwd_bn_apply", ptr(gy), ptr(idx), ptr(y), ptr(vec), act, ptr(coef), ptr(dz), n // G, oh, ow, C, poh, pow_, G)
        return dz
    if out.pre_sums is not None:            # reduction fused into the kernel that produced g (already activation-masked)
        sums, out.pre_sums = out.pre_sums, None
    else:
        sums = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS)
        call("adamml_bn_bwd_reduce", ptr(g), ptr(y), ptr(vec), act, ptr(sums), P, C, G)
    sums, nslots = rt.sync.reduce(sums, C, G)
    coef = torch.empty(G, 3, C, dtype=torch.float32, device=y.device)
    train_bn = bn.weight.requires_grad
    if defer_apply:
        # the caller's data-gradient loader wants dz = A g + B z + C: finalize and the affine form in one launch
        aff = torch.empty(G, 3, C, dtype=torch.float32, device=y.device)
        call("adamml_bn_bwd_finalize_affine", ptr(sums), nslots, G, float(count * rt.sync.world), ptr(bn.weight), ptr(vec),
             ptr(bn.weight.grad) if train_bn else None, ptr(bn.bias.grad) if train_bn else None, ptr(coef), ptr(aff), C, 1.0 / rt.sync.world)
        return g, coef, aff
    call("adamml_bn_bwd_finalize", ptr(sums), nslots, G, float(count * rt.sync.world), ptr(bn.weight), ptr(vec),
         ptr(bn.weight.grad) if train_bn else None, ptr(bn.bias.grad) if train_bn else None, ptr(coef), C, 1.0 / rt.sync.world)
    dz = torch.empty_like(y)
    call("adamml_bn_bwd_apply", ptr(g), ptr(y), ptr(vec), act, ptr(coef), ptr(dz), P, C, G)
    return dz


def materialize(rt, x):
    """Evaluate a lazy tensor once into a plain bf16 tensor.  Used in front of KxK dense convs: their implicit-GEMM
    loader would otherwise re-apply the producer's BatchNorm+ReLU once per tap (9x for 3x3) on the VALU."""
    n, h, w, C = x.shape
    out_t = torch.empty_like(x.data)
    call("adamml_bn_act_add", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, None, None, None, 0, ptr(out_t),
         n // rt.groups * h * w, C, rt.groups)
    a = Lazy(out_t, requires_grad=x.requires_grad)
    a.src = x
    if rt.tape.need_grad:
        def bwd():
            g, a.grad = a.grad, None
            if g is not None:
                _accum_grad(x, g)          # d/d(activated value) is the same quantity on both sides
        rt.tape.record(bwd)
    return a


def conv_bn(rt, x, cs, bn, act, sole_consumer=False, last_consumer=False):
    """conv (dense or depthwise) + train/eval BatchNorm + activation, as one lazy tensor.
    sole_consumer=True promises that this conv is the ONLY consumer of x: its data-gradient epilogue may then apply
    x's activation mask and accumulate x's BatchNorm-backward sums (no separate reduction pass over g and z).
    last_consumer=True promises that every other consumer of x has already contributed to x.grad when this conv's
    backward runs (conv1 of a bottleneck: the identity path of the residual add is recorded later, hence reversed
    earlier).  If x is the output of a residual add, the data-gradient epilogue then finishes that add's backward too:
    it adds the identity-path gradient, applies the add's activation mask and accumulates the BatchNorm-backward sums
    of the add's operands (adamml_conv_bwd_data_res) -- the gradient of the block output is written once, already
    masked, instead of being written, re-read, masked and written again by adamml_residual_bwd."""
    G = rt.groups
    if x.scale is not None and not cs.depthwise and cs.kh * cs.kw > 1 and \
            not hip.load().adamml_conv_fused_input_supported(byref(cs.desc(x.shape, x.act, G, x.gs))):
        x = materialize(rt, x)
    d = cs.desc(x.shape, x.act, G, x.gs)
    if x.shape[3] != cs.cin and not (cs.stem and x.shape[3] == 4):        # (4-channel pixels: the 7x7 stem kernels only, checked below)
        raise RuntimeError("conv_bn: input has %d channels, weight pack expects %d" % (x.shape[3], cs.cin))
    if x.shape[0] % G:
        raise RuntimeError("conv_bn: %d images do not split into %d BatchNorm groups" % (x.shape[0], G))
    dev = x.data.device
    # forward already run by the producer of x (conv_bn_add next_cs).  Only the conv the result was computed FOR consumes it; another
    # conv of x issued first (a downsample branch, a reordering) leaves it in place (round-5 advisor finding: it used to be cleared by
    # whichever conv came first, and the fused result was then silently recomputed) -- NetRT.end_forward counts the ones nobody claimed
    pre = x.next_pre if (x.next_pre is not None and x.next_pre[0] is cs and rt.training) else None
    if pre is not None:
        x.next_pre = None
        rt.pre_pending -= 1
    y = pre[1] if pre is not None else torch.empty(G * d.N, d.OH, d.OW, d.Cout, dtype=torch.bfloat16, device=dev)
    C = d.Cout
    count = d.N * d.OH * d.OW               # elements per channel per group
    fwd = "adamml_dwconv_fwd" if cs.depthwise else "adamml_conv_fwd"
    stem = cs.stem and x.scale is None and hip.load().adamml_conv_stem_supported(byref(d))
    if x.shape[3] != cs.cin and not stem:
        raise RuntimeError("conv_bn: 4-channel pixels need the 7x7 stem kernel, which does not support this shape")
    # algorithmic work of this layer (true input channels, each tensor touched once), for the roofline report
    macs = float(count) * G * C * cs.kh * cs.kw * (1 if cs.depthwise else cs.cin_true)
    in_b, out_b = 2.0 * G * d.N * d.H * d.W * cs.cin_true, 2.0 * G * count * C
    w_b = 2.0 * C * cs.kh * cs.kw * (1 if cs.depthwise else cs.cin_true)
    # device kernel behind adamml_conv_fwd / adamml_conv_bwd_data[_bn] for this layer (bench.py groups launches by it)
    kern = kern_f = kern_dual = kern_acc = kern_bn = None
    if not cs.depthwise:
        lib = hip.load()
        kern = "conv3x3_c64_kernel" if lib.adamml_conv_fused_input_supported(byref(d)) else "conv_gemm_kernel"
        # the narrow 1x1 layers of the MobileNetV2s run on the streaming kernels of csrc/conv1x1_narrow.hip
        nar = [bool(lib.adamml_conv1x1_narrow_supported(byref(d), k)) for k in range(5)] if cs.kh * cs.kw == 1 else [False] * 5
        kern_f = "conv1x1_narrow_fwd_kernel" if nar[0] else kern
        kern_dual = "conv1x1_narrow_dgrad_kernel" if nar[2] else kern
        kern_acc = ("conv1x1_narrow_dgrad_kernel" if nar[4] else kern, "conv1x1_narrow_fwd_kernel" if nar[3] else kern)      # [accumulating?]
        kern_bn = "conv1x1_narrow_dgrad_kernel" if nar[4] else kern
        if cs.kh * cs.kw == 1 and d.Cin >= 256:
            # the wide 1x1 layers of ResNet layers 3-4 run on the activation-stationary streaming kernels of csrc/conv1x1_wide.hip
            wide = [bool(lib.adamml_conv1x1_wide_supported(byref(d), k)) for k in (0, 3, 4)]
            if wide[0]:
                kern_f = "wide_all_kernel"
            kern_acc = ("wide_all_kernel" if wide[2] else kern_acc[0], "wide_all_kernel" if wide[1] else kern_acc[1])
    role_f = None if cs.depthwise else (R_KXK if cs.kh * cs.kw > 1 else R_1X1)
    role_b = None if cs.depthwise else (R_KXK if (cs.kh * cs.kw > 1 or cs.stride > 1) else R_1X1)
    role_bf = role_b if role_b != R_1X1 else R_FUSED          # data gradient with a fused BatchNorm-backward / residual epilogue
    hip.next_meta = (2 * macs, in_b + out_b + w_b, "conv_stem_kernel" if stem else kern_f, role_f)
    if pre is not None:
        hip.next_meta = (0.0, 0.0)
        vec = _bn_vectors(rt, bn, pre[2], count, C, dev)
        rt.touched_bns.append(bn)
    elif rt.training:
        stats = rt.fwd_arena.take(G * 2 * C * STAT_SLOTS)
        if stem:
            call("adamml_conv_stem_fwd", byref(d), ptr(x.data), ptr(cs.w_stem), ptr(y), ptr(stats))
        else:
            call(fwd, byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(y), ptr(stats))
        vec = _bn_vectors(rt, bn, stats, count, C, dev)
        rt.touched_bns.append(bn)
    else:
        if stem:
            call("adamml_conv_stem_fwd", byref(d), ptr(x.data), ptr(cs.w_stem), ptr(y), None)
        else:
            call(fwd, byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(y), None)
        vec = _bn_eval_vectors(rt, bn, C, dev)
    if rt.capture is not None:
        rt.capture.setdefault(id(cs.weight), []).append(y)
    if rt.training:
        out = Lazy(y, vec[0, 0], vec[0, 1], act, gs=4 * C)
        out.vec = vec
        out.alg = bool(ALG_BN and rt.tape.need_grad and act == ACT_NONE and not cs.depthwise and not stem and x.requires_grad
                       and cs.weight.requires_grad and _alg_supported(cs, d))
    else:
        out = Lazy(y, vec[0], vec[1], act)
    if rt.tape.need_grad:
        def bwd():
            if out.grad is None and out.pool_grad is None:
                return
            if out.alg and out.pool_grad is None:
                _conv1x1_backward_alg(rt, out, x, y, vec, bn, cs, d, count, sole_consumer, macs, in_b, out_b, w_b, kern)
                return
            if DUAL_DGRAD and act == ACT_NONE and not cs.depthwise and not stem and out.pool_grad is None and x.requires_grad \
                    and rt.training and hip.load().adamml_conv_bwd_data_dual_supported(byref(d)):
                _conv1x1_backward_dual(rt, out, x, y, vec, bn, cs, d, count, sole_consumer, macs, in_b, out_b, w_b, kern_dual)
                return
            if DUAL_DGRAD and act != ACT_NONE and out.pre_sums is not None and not cs.depthwise and not stem and out.pool_grad is None \
                    and x.requires_grad and rt.training and cs.kh * cs.kw == 1 and nar[2]:
                # expansion conv of an inverted residual (round 6): its gradient arrives ALREADY masked by its ReLU6 (the depthwise conv's fused
                # backward applied the mask and accumulated the sums), so the BatchNorm-backward apply is the same affine A g' + B z + C as for
                # a linear BatchNorm and folds into the narrow streaming data gradient's loader the same way
                _conv1x1_backward_dual(rt, out, x, y, vec, bn, cs, d, count, sole_consumer, macs, in_b, out_b, w_b, kern_dual)
                return
            if cs.depthwise and DW_FUSED and out.pool_grad is None and out.pre_sums is not None and rt.training and sole_consumer \
                    and cs.weight.requires_grad and x.requires_grad and x.grad is None and x.src is None and x.vec is not None \
                    and x.pre_sums is None and x.scale is not None and x.scale.data_ptr() == x.vec.data_ptr() \
                    and hip.load().adamml_dwconv_bwd_fused_supported(byref(d)):
                # g is already masked by this conv's activation (out.pre_sums: the projection's data gradient did that), so
                # dz = A g + B z + C in the loader; apply + weight gradient + data gradient (mask / sums of the expansion) in one pass
                g, coef, aff = _bn_backward(rt, out, y, vec, bn, act, count, defer_apply=True)
                x.grad = torch.empty_like(x.data)
                sums = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS)
                ws = hip.scratch(hip.load().adamml_dwconv_bwd_fused_workspace(byref(d)), dev)
                hip.next_meta = (4 * macs, 2 * in_b + 2 * out_b + 2 * w_b, "dwconv_bwd_fused_kernel", None)
                call("adamml_dwconv_bwd_fused", byref(d), ptr(g), ptr(y), ptr(aff), ptr(cs.w_fwd), ptr(x.data), ptr(x.vec), x.act, ptr(x.grad),
                     ptr(sums), ptr(cs.weight.grad), ptr(ws), ws.numel() * 4)
                x.pre_sums = sums
                return
            dz = _bn_backward(rt, out, y, vec, bn, act, count)
            if cs.weight.requires_grad:
                with _on_wgrad_stream(rt, (dz, x.data, x.scale)):
                    hip.next_meta = (2 * macs, in_b + out_b + 2 * w_b, None, None if cs.depthwise else R_WGRAD)
                    if cs.depthwise:
                        ws = hip.wgrad_workspace(d, 0, dz.device, depthwise=True)
                        call("adamml_dwconv_bwd_weight", byref(d), ptr(dz), ptr(x.data), ptr(x.scale), ptr(x.shift), ptr(cs.weight.grad),
                             ptr(ws), ws.numel() * 4)
                    elif stem:
                        ws = hip.wgrad_workspace(d, cs.cin_true, dz.device, stem=True)
                        call("adamml_conv_stem_bwd_weight", byref(d), ptr(dz), ptr(x.data), ptr(cs.weight.grad), cs.cin_true, ptr(ws),
                             ws.numel() * 4)
                    else:
                        ws = hip.wgrad_workspace(d, cs.cin_true, dz.device)
                        call("adamml_conv_bwd_weight", byref(d), ptr(dz), ptr(x.data), ptr(x.scale), ptr(x.shift),
                             ptr(cs.weight.grad), cs.cin_true, ptr(ws), ws.numel() * 4)
            if x.requires_grad:
                acc = 1
                if x.grad is None:
                    x.grad = torch.empty_like(x.data)
                    acc = 0
                hip.next_meta = (2 * macs, in_b * (1 + acc) + out_b + w_b, kern if cs.depthwise else kern_acc[0 if acc else 1], role_b)
                tgt = x.src if x.src is not None else x
                if cs.depthwise and DW_BNZ and sole_consumer and acc == 0 and tgt.vec is not None and tgt.pre_sums is None \
                        and hip.load().adamml_dwconv_bwd_data_bn_supported(byref(d)):
                    # the expansion's BatchNorm-backward sums come out of this data gradient (mask applied here): no reduction pass
                    sums = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS)
                    call("adamml_dwconv_bwd_data_bn", byref(d), ptr(dz), ptr(cs.w_fwd), ptr(x.grad), ptr(tgt.data), ptr(tgt.vec), tgt.act, ptr(sums))
                    tgt.pre_sums = sums
                elif cs.depthwise:
                    call("adamml_dwconv_bwd_data", byref(d), ptr(dz), ptr(cs.w_fwd), ptr(x.grad), acc)
                elif last_consumer and _residual_fusable(x, d):
                    z, idn, ract, idn_sole, rmask = x.res
                    fb = idn is not None and idn_sole and idn.requires_grad and idn.vec is not None and idn.grad is None
                    sa = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS)
                    sb = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS) if fb else None
                    fb_alg = fb and idn.alg               # the downsample BatchNorm shares sum(g'); its second moment comes from g'^T a too
                    hip.next_meta = (2 * macs, in_b * ((1.0625 if rmask is not None else 2) + (0 if z.alg else 1) + acc
                                                       + (1 if (fb and not (fb and idn.alg)) else 0)) + out_b + w_b, kern, R_FUSED)
                    fbk = fb and not fb_alg
                    ain = z.alg_in
                    if (RES_PROD and z.alg and ain is not None and (not fb or fb_alg) and acc == 1 and rmask is not None and ain[0].data is not None
                            and hip.load().adamml_conv_bwd_data_res_prod_supported(byref(d), ain[1].Cin)):
                        # the product g'^T a of the algebraic backward of the conv that produced z (its input a = ain[0]) is accumulated
                        # from the gradient tile inside this kernel: no separate pass over g' and a
                        xa = ain[0]
                        z.prod = torch.empty(G, d.Cin, ain[1].Cin, dtype=torch.float32, device=dz.device)
                        need = hip.load().adamml_conv_bwd_data_res_prod_workspace(byref(d))
                        wsp = hip.scratch(need, dz.device)
                        # (layer 1: the barrier-free streaming kernel of csrc/res_prod_stream.hip; its workspace is sized for it)
                        kern_rp = "res_prod_stream_kernel" if hip.load().adamml_conv_bwd_data_res_prod_streams(byref(d), ain[1].Cin) else kern
                        hip.next_meta = (2 * macs + 2.0 * G * d.N * d.H * d.W * d.Cin * ain[1].Cin, in_b * 2.0625 + out_b + w_b + 2.0 * G * d.N * d.H * d.W * ain[1].Cin, kern_rp, R_FUSED)
                        call("adamml_conv_bwd_data_res_prod", byref(d), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), ptr(rmask), ract, ptr(sa), ptr(xa.data),
                             ptr(xa.scale), ptr(xa.shift), xa.act, xa.gs, ain[1].Cin, ptr(z.prod), ptr(wsp), wsp.numel() * 4)
                    else:
                        if acc == 1 and rmask is not None and z.alg and hip.load().adamml_conv_bwd_data_res_streams(byref(d)):
                            hip.next_meta = hip.next_meta[:2] + ("res_prod_stream_kernel", R_FUSED)       # (layer 2: csrc/res_prod_stream.hip)
                        call("adamml_conv_bwd_data_res", byref(d), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), acc, ptr(x.data), ptr(rmask), ract,
                             None if z.alg else ptr(z.data), ptr(z.vec), ptr(sa), ptr(idn.data) if fbk else None, ptr(idn.vec) if fbk else None,
                             ptr(sb) if fbk else None)
                    z.pre_sums = sa
                    z.sums_partial = z.alg
                    if fb:
                        if fb_alg:
                            call("adamml_copy2d", ptr(sb), 2 * d.Cin * 8, ptr(sa), 2 * d.Cin * 8, d.Cin * 8, G * STAT_SLOTS)       # the sum(g') columns
                            idn.sums_partial = True
                        idn.pre_sums = sb
                    x.res_done = True
                elif sole_consumer and acc == 0 and tgt.vec is not None and tgt.pre_sums is None:
                    sums = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS)
                    hip.next_meta = (2 * macs, 2 * in_b + out_b + w_b, kern_bn, role_bf)
                    call("adamml_conv_bwd_data_bn", byref(d), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), ptr(tgt.data), ptr(tgt.vec),
                         tgt.act, ptr(sums))
                    tgt.pre_sums = sums
                else:
                    call("adamml_conv_bwd_data", byref(d), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), acc)
        rt.tape.record(bwd)
    return out


STEM1_F32 = os.environ.get("ADAMML_STEM1_F32", "1") != "0"     # fp32 spectrogram straight into the MobileNetV2 stems (A/B aid)


def stem1_supported(cs, x):
    """x: the fp32 one-channel input [B, G, H, W] as the caller supplies it (models/adamml.py:49-53: G = segments).  True when the
    3x3 / stride-2 stem described by cs can read it directly (adamml_conv_stem1_fwd)."""
    if not STEM1_F32 or x.dtype != torch.float32 or x.dim() != 4 or cs.kh != 3 or cs.stride != 2 or cs.pad != 1 or cs.weight.shape[1] != 1:
        return False
    d = ConvDesc(x.shape[0], x.shape[2], x.shape[3], 8, (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1, cs.cout, 3, 3, 2, 1, 1, 0, 0,
                 x.shape[1], 0)
    return bool(hip.load().adamml_conv_stem1_supported(byref(d)))


def conv_stem1_bn(rt, x1, cs, bn, act):
    """First conv of a MobileNetV2 on a one-channel fp32 image + train/eval BatchNorm + activation, as one lazy tensor: the kernel reads
    the caller's [B, G, H, W] fp32 tensor (group = dim 1) -- no re-layout pass, no bf16 rounding of the spectrogram or of the 3x3
    weights.  cs: a ConvState registered with depthwise=True (tap-major fp32 pack [9][Cout]).  The network input takes no gradient."""
    B, G, H, W = x1.shape
    if G != rt.groups:
        raise RuntimeError("conv_stem1_bn: input has %d groups (dim 1), the call %d" % (G, rt.groups))
    if not x1.is_contiguous():
        if hip.recorder is not None:
            hip.recorder.failed = "non-contiguous input (its copy's address would be baked into the plan)"
        x1 = x1.contiguous()
    C = cs.cout
    d = ConvDesc(B, H, W, 8, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, 3, 3, 2, 1, 1, 0, 0, G, 0)
    dev = x1.device
    y = torch.empty(G * B, d.OH, d.OW, C, dtype=torch.bfloat16, device=dev)
    count = B * d.OH * d.OW
    istr, gstr = G * H * W, H * W
    hip.next_meta = (2.0 * count * G * C * 9, 4.0 * G * B * H * W + 2.0 * G * count * C)
    if rt.training:
        stats = rt.fwd_arena.take(G * 2 * C * STAT_SLOTS)
        call("adamml_conv_stem1_fwd", byref(d), ptr(x1), istr, gstr, ptr(cs.w_fwd), ptr(y), ptr(stats))
        vec = _bn_vectors(rt, bn, stats, count, C, dev)
        rt.touched_bns.append(bn)
        out = Lazy(y, vec[0, 0], vec[0, 1], act, gs=4 * C)
        out.vec = vec
    else:
        call("adamml_conv_stem1_fwd", byref(d), ptr(x1), istr, gstr, ptr(cs.w_fwd), ptr(y), None)
        vec = _bn_eval_vectors(rt, bn, C, dev)
        out = Lazy(y, vec[0], vec[1], act)
    if rt.capture is not None:
        rt.capture.setdefault(id(cs.weight), []).append(y)
    if rt.tape.need_grad:
        def bwd():
            if out.grad is None and out.pool_grad is None:
                return
            dz = _bn_backward(rt, out, y, vec, bn, act, count)
            if cs.weight.requires_grad:
                with _on_wgrad_stream(rt, (dz, x1)):
                    need = hip.load().adamml_conv_stem1_bwd_weight_workspace(byref(d))
                    ws = hip.scratch(need, dev)
                    call("adamml_conv_stem1_bwd_weight", byref(d), ptr(dz), ptr(x1), istr, gstr, ptr(cs.weight.grad), ptr(ws), ws.numel() * 4)
        rt.tape.record(bwd)
    return out


class _on_wgrad_stream:
    """Weight gradients are leaves of the backward dataflow: nothing downstream of a conv's backward needs dW, only dz.
    With rt.wgrad_stream set they are enqueued on that stream (after the kernels that produced their operands) and run
    concurrently with the data-gradient chain, which on layers 2-4 is a sequence of latency-bound kernels that leave
    most of the machine idle.  The tensors they read are kept alive by rt.wgrad_pending until an event recorded behind
    the kernel has completed (or the stream has been joined) -- NOT by Tensor.record_stream: with ~70 multi-GB blocks per
    step parked in the caching allocator's cross-stream list the step time degraded from 152 ms to over a second within
    eight steps (measured).  The backbone joins the stream before its gradients are consumed (backbone.run_tape,
    HipDDP bucket hooks)."""

    def __init__(self, rt, tensors):
        self.rt = rt
        self.ws = rt.wgrad_stream
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if self.ws is not None:
            cur = torch.cuda.current_stream()
            self.ws.wait_stream(cur)
            if hip.recorder is not None:
                hip.recorder.wait(self.ws.cuda_stream, cur.cuda_stream)
            self.ctx = torch.cuda.stream(self.ws)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            ev = torch.cuda.Event()
            ev.record(self.ws)
            self.ctx.__exit__(*exc)
            pend = self.rt.wgrad_pending
            pend.append((ev, self.tensors))
            while pend and pend[0][0].query():          # drop the references of the launches that have finished
                pend.popleft()
        return False


def _gram_colsum(rt, x, d):
    """G = a^T a [groups, Cin, Cin] and s = sum a [groups, Cin] over the pixels of each group, a = the lazily normalised conv
    input exactly as the conv's loader stages it (bf16).  Feeds the algebraic BatchNorm backward and, computed in the forward
    pass, the train-mode statistics of conv_bn_add (adamml_gram_stats)."""
    G, Cin, dev = rt.groups, d.Cin, x.data.device
    n, h, w_, _ = x.shape
    Gm = torch.empty(G, Cin, Cin, dtype=torch.float32, device=dev)
    sv = torch.empty(G, Cin, dtype=torch.float32, device=dev)
    if GRAM_KERNEL and hip.load().adamml_gram_colsum_supported(Cin):
        # one streaming pass (csrc/gram.hip) instead of the generic weight-gradient kernel with dz = x plus a column-sum pass
        P = n // G * h * w_
        wsg = hip.scratch(hip.load().adamml_gram_colsum_workspace(P, Cin, G), dev)
        hip.next_meta = (2.0 * G * P * Cin * Cin, 2.0 * G * P * Cin, None, R_WGRAD)
        call("adamml_gram_colsum", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(Gm), ptr(sv), P, Cin, G, ptr(wsg), wsg.numel() * 4)
        return Gm, sv
    dg = ConvDesc(d.N, d.H, d.W, Cin, d.H, d.W, Cin, 1, 1, 1, 0, 1, d.act, 0, G, d.in_gstride)
    wsg = hip.wgrad_workspace(dg, Cin, dev)
    hip.next_meta = (2.0 * G * d.N * d.H * d.W * Cin * Cin, 2.0 * G * d.N * d.H * d.W * Cin, None, R_WGRAD)
    call("adamml_conv_bwd_weight_grouped", byref(dg), ptr(x.data), ptr(x.scale), ptr(x.shift), x.act, x.gs, ptr(x.data), ptr(x.scale),
         ptr(x.shift), ptr(Gm), Cin, ptr(wsg), wsg.numel() * 4)
    call("adamml_lazy_colsum", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(sv), n // G * h * w_, Cin, G)
    return Gm, sv


TPOOL_PROD = os.environ.get("ADAMML_TPOOL_BWD_PROD", "1") != "0"   # temporal-pool backward + the product g'^T a in one pass (A/B aid)
RES_PROD = os.environ.get("ADAMML_RES_PROD", "1") != "0"     # g'^T a accumulated inside the residual-backward data gradient (A/B aid)
POOL_ZSEL = os.environ.get("ADAMML_POOL_ZSEL", "1") != "0"   # stem BatchNorm-backward sums over the pool windows (g_y, z_sel) (A/B aid)
DW_FUSED = os.environ.get("ADAMML_DW_BWD_FUSED", "1") != "0"    # whole stride-1 depthwise backward in one pass (csrc/dwconv_bwd_fused.hip; A/B aid)
DW_BNZ = os.environ.get("ADAMML_DW_BNZ", "1") != "0"         # BatchNorm-backward sums of the expansion inside the depthwise data gradient (A/B aid)
GRAM_KERNEL = os.environ.get("ADAMML_GRAM_KERNEL", "1") != "0"     # dedicated Gram + column-sum kernel (A/B aid)
ALG_BN = os.environ.get("ADAMML_ALG_BN", "1") != "0"     # algebraic BatchNorm backward through expanding 1x1 convs (A/B aid)
ALG_MAX_COUT = int(os.environ.get("ADAMML_ALG_MAX_COUT", "512"))     # measured: at Cout = 1024 (layer 3) the small per-group products cost more than the saved passes (146.1 vs 144.6 ms)
ALG_GEMM_CIN = 256     # from this input width on, the small per-group matrix products go through adamml_gemm_f32


def _alg_supported(cs, d):
    """Expanding 1x1 / stride-1 conv (bottleneck conv3, stride-1 downsample) whose output is large: Cin <= Cout / 2 so that the
    Gram matrix of the input and the extra K columns are cheap, Cout <= 512 (beyond, the tensors are small and the per-group
    weight products dominate), no channel padding."""
    return (cs.kh == 1 and cs.kw == 1 and cs.stride == 1 and cs.pad == 0 and cs.cin_true == d.Cin and d.Cin in (64, 128, 256)
            and d.Cout % 32 == 0 and 2 * d.Cin <= d.Cout <= ALG_MAX_COUT)


def _conv1x1_backward_alg(rt, out, x, y, vec, bn, cs, d, count, sole_consumer, macs, in_b, out_b, w_b, kern, Gm=None, sv=None):
    """Backward of z = W a (1x1) followed by a linear train-mode BatchNorm WITHOUT touching z or dz (include/adamml_hip.h,
    "algebraic BatchNorm backward"): with dz = A g' + B z + C per channel,
        dx = (W^T diag(A)) g' + (W^T diag(B) W) a + W^T C,      dW = A (.) (g'^T a) + B (.) (W G) + C (x) s,  G = a^T a, s = sum a.
    The products over pixels (g'^T a, a^T a, sum a) run on the weight-gradient stream; the data gradient is one GEMM over the
    concatenated input [g' | a] with per-group weights."""
    G = rt.groups
    Cout, Cin = d.Cout, d.Cin
    dev = y.device
    w2 = cs.weight                                              # fp32 master [Cout, Cin, 1, 1], contiguous
    P = out.prod if out.prod is not None else torch.empty(G, Cout, Cin, dtype=torch.float32, device=dev)
    g0 = out.grad

    def products():
        ws = hip.wgrad_workspace(d, Cin, dev)
        hip.next_meta = (2 * macs, in_b + out_b + 2 * w_b, None, R_WGRAD)
        call("adamml_conv_bwd_weight_grouped", byref(d), ptr(g0), None, None, 0, 0, ptr(x.data), ptr(x.scale), ptr(x.shift), ptr(P), Cin,
             ptr(ws), ws.numel() * 4)
    if out.sums_partial:
        # the producer of g' did not read z: sum(g' zhat) = invstd (sum_j W (.) P - mean sum g') needs P BEFORE the finalize step,
        # so P runs on this stream (behind the weight-gradient stream's backlog it would stall the whole data-gradient chain)
        out.sums_partial = False
        if out.prod is None:
            products()
        call("adamml_alg_sumfix", ptr(w2), ptr(P), ptr(vec), ptr(out.pre_sums), Cout, Cin, G)
    elif out.prod is not None:
        pass
    else:
        with _on_wgrad_stream(rt, (g0, x.data, x.scale)):
            products()
    g, coef, aff = _bn_backward(rt, out, y, vec, bn, ACT_NONE, count, defer_apply=True)
    # ---- data gradient (main stream)
    w_alg = torch.empty(G, Cin, Cout + Cin, dtype=torch.bfloat16, device=dev)
    cadd = torch.empty(G, Cin, dtype=torch.float32, device=dev)
    m_pre = None
    w2d = w2.view(Cout, Cin)
    if Cin >= ALG_GEMM_CIN:
        # W^T diag(B_g) W for all groups as ONE fp32 GEMM: [G*Cin, Cout] x [Cout, Cin]
        wb = (w2d.unsqueeze(1) * aff[:, 1].t().unsqueeze(2)).reshape(Cout, G * Cin)
        m_pre = gemm_f32(wb, w2d, trans_a=True, trans_b=False)
    call("adamml_alg_pack", ptr(w2), ptr(aff), ptr(m_pre), ptr(w_alg), ptr(cadd), Cout, Cin, G)
    acc = 1
    if x.grad is None:
        x.grad = torch.empty_like(x.data)
        acc = 0
    tgt = x.src if x.src is not None else x
    if (Cout, Cin) == (256, 64):
        kern = "alg_stream_kernel"            # csrc/conv1x1_stream.hip serves this shape (bench.py groups launches by device kernel)
    if sole_consumer and acc == 0 and tgt.vec is not None and tgt.pre_sums is None:
        sums = rt.bwd_arena.take(G * 2 * Cin * STAT_SLOTS)
        hip.next_meta = (2 * macs, 3 * in_b + out_b + w_b, kern, R_FUSED)
        call("adamml_conv_bwd_data_alg", byref(d), ptr(g), ptr(x.data), ptr(x.scale), ptr(x.shift), ptr(w_alg), ptr(cadd), ptr(x.grad), 0,
             ptr(tgt.data), ptr(tgt.vec), tgt.act, ptr(sums))
        tgt.pre_sums = sums
    else:
        hip.next_meta = (2 * macs, in_b * (2 + acc) + out_b + w_b, kern, R_FUSED)
        call("adamml_conv_bwd_data_alg", byref(d), ptr(g), ptr(x.data), ptr(x.scale), ptr(x.shift), ptr(w_alg), ptr(cadd), ptr(x.grad), acc,
             None, None, 0, None)
    # ---- weight gradient (weight-gradient stream): products over the pixels, then the per-group combination
    with _on_wgrad_stream(rt, (g, x.data, x.scale, aff, P)):
        if Gm is None:            # (conv_bn_add computed the Gram matrix and the column sums in the forward pass: its statistics)
            Gm, sv = _gram_colsum(rt, x, d)
        wg_pre = None
        if Cin >= ALG_GEMM_CIN:
            # W G_g for all groups as one GEMM: [Cout, Cin] x [Cin, G*Cin]
            wg_pre = gemm_f32(w2d, Gm.permute(1, 0, 2).reshape(Cin, G * Cin), trans_b=False)
        call("adamml_alg_wgrad_combine", ptr(w2), ptr(aff), ptr(P), ptr(Gm), ptr(wg_pre), ptr(sv), ptr(cs.weight.grad), Cout, Cin, G)


DUAL_DGRAD = True     # 1x1 / linear-BatchNorm layers: BatchNorm-backward apply folded into the data-gradient loader


def _conv1x1_backward_dual(rt, out, x, y, vec, bn, cs, d, count, sole_consumer, macs, in_b, out_b, w_b, kern):
    """Backward of a 1x1 / stride-1 conv followed by a linear (no activation) train-mode BatchNorm -- bn3 and the stride-1
    downsample BN of a bottleneck, the projection conv of an inverted residual.  After the BatchNorm-backward sums are
    finalised, the data-gradient kernel reads (g, z) directly and forms dz = A g + B z + C in its loader; dz is written
    once, as a side output, for the weight-gradient kernel.  Saves one full pass over the layer's largest tensor."""
    G = rt.groups
    C = d.Cout
    g, coef, aff = _bn_backward(rt, out, y, vec, bn, ACT_NONE, count, defer_apply=True)
    need_w = cs.weight.requires_grad
    dz = torch.empty_like(y) if need_w else None
    acc = 1
    if x.grad is None:
        x.grad = torch.empty_like(x.data)
        acc = 0
    tgt = x.src if x.src is not None else x
    if sole_consumer and acc == 0 and tgt.vec is not None and tgt.pre_sums is None:
        sums = rt.bwd_arena.take(G * 2 * d.Cin * STAT_SLOTS)
        hip.next_meta = (2 * macs, 2 * in_b + (3 if need_w else 2) * out_b + w_b, kern, R_FUSED)
        call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(y), ptr(aff), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), 0, ptr(tgt.data),
             ptr(tgt.vec), tgt.act, ptr(sums))
        tgt.pre_sums = sums
    else:
        hip.next_meta = (2 * macs, in_b * (1 + acc) + (3 if need_w else 2) * out_b + w_b, kern, R_FUSED)
        call("adamml_conv_bwd_data_dual", byref(d), ptr(g), ptr(y), ptr(aff), ptr(dz), ptr(cs.w_dgrad), ptr(x.grad), acc, None, None, 0,
             None)
    if need_w:
        with _on_wgrad_stream(rt, (dz, x.data, x.scale)):
            hip.next_meta = (2 * macs, in_b + out_b + 2 * w_b, None, R_WGRAD)
            ws = hip.wgrad_workspace(d, cs.cin_true, dz.device)
            call("adamml_conv_bwd_weight", byref(d), ptr(dz), ptr(x.data), ptr(x.scale), ptr(x.shift), ptr(cs.weight.grad), cs.cin_true,
                 ptr(ws), ws.numel() * 4)


def _residual_fusable(x, d):
    """x is the output of a residual add whose backward (mask + BatchNorm-backward sums of its BatchNorm'd operand) can
    be finished by the data-gradient epilogue of the conv described by d."""
    if x.res is None or x.res_done or x.src is not None:
        return False
    z = x.res[0]
    if not (z.requires_grad and z.vec is not None and z.grad is None and z.pre_sums is None):
        return False
    return bool(hip.load().adamml_conv_bwd_data_res_supported(byref(d)))


def _accum_grad(t, g):
    if not t.requires_grad:
        return
    if t.grad is None:
        t.grad = g
    else:
        n = g.numel() // g.shape[-1]
        call("adamml_bn_act_add", ptr(t.grad), None, None, 0, ACT_NONE, ptr(g), None, None, 0, ptr(t.grad), n, g.shape[-1], 1)


def add_act(rt, z, idn, act, idn_sole=False):
    """out = act(value(z) + value(idn)); idn may be None (pure materialisation).  z must have no other consumer;
    idn_sole=True promises the same for a lazily normalised idn (the downsample branch of a bottleneck)."""
    n, h, w, C = z.shape
    if z.act != ACT_NONE or (idn is not None and idn.act != ACT_NONE):
        raise RuntimeError("add_act: operands must be linear (no pending activation)")
    out_t = torch.empty_like(z.data)
    G = rt.groups
    P = n // G * h * w
    # 1 bit per element of act'(out): what the fused residual backward needs of `out` (None when nothing will be masked)
    mask_t = torch.empty(n, h, w, C // 8, dtype=torch.uint8, device=out_t.device) if (rt.tape.need_grad and act != ACT_NONE) else None
    call("adamml_bn_act_add_mask", ptr(z.data), ptr(z.scale), ptr(z.shift), z.gs, act, ptr(idn.data) if idn is not None else None,
         ptr(idn.scale) if idn is not None else None, ptr(idn.shift) if idn is not None else None,
         idn.gs if idn is not None else 0, ptr(out_t), ptr(mask_t), P, C, G)
    out = Lazy(out_t)
    out.res = (z, idn, act, idn_sole, mask_t)
    if rt.tape.need_grad:
        rt.tape.record(lambda: _add_backward(rt, out, out_t, z, idn, act, idn_sole, P, C))
    return out


def _add_backward(rt, out, out_t, z, idn, act, idn_sole, P, C):
    """Backward of out = act(value(z) + value(idn)) (add_act / conv_bn_add)."""
    G = rt.groups
    g = out.grad
    out.grad = None
    if g is None:
        return
    fa = z.requires_grad and z.vec is not None and z.grad is None
    fb = idn is not None and idn_sole and idn.requires_grad and idn.vec is not None and idn.grad is None
    if out.res_done:
        # the last consumer's data-gradient epilogue already masked g and accumulated the sums (conv_bn)
        out.res_done = False
        _accum_grad(z, g)
        if idn is not None:
            _accum_grad(idn, g)
        return
    g2 = torch.empty_like(g) if act != ACT_NONE else g
    if fa or fb:
        sa = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS) if fa else None
        sb = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS) if fb else None
        call("adamml_residual_bwd", ptr(g), ptr(out_t), act, ptr(g2), ptr(z.ensure_data()) if fa else None,
             ptr(z.vec) if fa else None, ptr(sa), ptr(idn.data) if fb else None, ptr(idn.vec) if fb else None, ptr(sb),
             P, C, G)
        if fa:
            z.pre_sums = sa
        if fb:
            idn.pre_sums = sb
    elif act != ACT_NONE:
        call("adamml_act_bwd_from_output", ptr(g), ptr(out_t), act, ptr(g2), g.numel())
    _accum_grad(z, g2)
    if idn is not None:
        _accum_grad(idn, g2)


FUSE_ADD = os.environ.get("ADAMML_FUSE_ADD", "1") != "0"     # conv3 + BatchNorm + residual add in one kernel (A/B aid)
FUSE_TPOOL = os.environ.get("ADAMML_FUSE_TPOOL", "1") != "0"     # ... and the temporal max-pool behind the last block of a stage (A/B aid)


def conv_bn_add_supported(rt, x, cs, need_grad, idn=None):
    """Can `conv (1x1) -> BatchNorm -> (+ identity) -> activation` run as conv_bn_add?  Eval mode: every 1x1 / stride-1 conv (the
    BatchNorm is a known affine map).  Train mode: the statistics must come from the Gram matrix of the conv INPUT, which only
    pays for expanding convs (bottleneck conv3 of layers 1-2: same shapes as the algebraic BatchNorm backward, whose products it
    shares), and the backward must be the algebraic one (it never reads the raw conv output)."""
    if not FUSE_ADD or cs.depthwise or cs.stem or x.shape[3] != cs.cin:
        return False
    if idn is not None and idn.act != ACT_NONE:
        return False             # the epilogue applies the identity's scale / shift only: a pending activation needs add_act's check
    d = cs.desc(x.shape, x.act, rt.groups, x.gs)
    if not hip.load().adamml_conv_fwd_bn_add_supported(byref(d)):
        return False
    if not rt.training:
        return not need_grad
    if not (ALG_BN and _alg_supported(cs, d)):
        return False
    return (not need_grad) or (x.requires_grad and cs.weight.requires_grad)


def conv_bn_add_tpool_supported(rt, x, cs, idn, act, frames, mode):
    """Can conv_bn_add additionally run the temporal max-pool that is the block output's ONLY consumer in its epilogue
    (adamml_conv_fwd_bn_add_tpool)?  The identity must be a plain tensor (the last block of a stage has no downsample branch)."""
    if not FUSE_TPOOL or mode != "max" or idn is None or idn.scale is not None or frames not in (2, 4, 8) or x.shape[0] % (rt.groups * frames):
        return False
    d = cs.desc(x.shape, x.act, rt.groups, x.gs)
    return bool(hip.load().adamml_conv_fwd_bn_add_tpool_supported(byref(d), frames, act, 1 if x.scale is not None else 0))


FADD_NEXT = os.environ.get("ADAMML_FADD_NEXT", "1") != "0"     # conv3 + bn3 + add + ReLU and the NEXT block's conv1 in one streaming kernel (A/B aid)


def conv_bn_add(rt, x, cs, bn, idn, act, idn_sole=False, tpool=0, next_cs=None):
    """out = act(BatchNorm(conv1x1(x)) + value(idn)) in ONE kernel whose epilogue normalises, adds and activates
    (adamml_conv_fwd_bn_add): the raw conv output is never written to HBM nor re-read by a separate add pass.
    tpool = T > 0 (caller checked conv_bn_add_tpool_supported): the block output feeds only TemporalPooling(max) over the T frames of a
    clip (models/resnet.py:205-209); the epilogue pools too and the POOLED tensor is returned -- the full-rate block output, its mask and
    the pool's pass never reach HBM; the backward routes the pooled gradient with the 2-bit codes the forward stored.
    Train mode: BatchNorm needs the batch statistics BEFORE that epilogue runs; for z = W a they follow from the Gram matrix
    G = a^T a and the column sums s of the (4x narrower) conv input: sum z = W s, sum z^2 = diag(W G W^T) (adamml_gram_stats).
    G and s are exactly the products the algebraic BatchNorm backward needs (_conv1x1_backward_alg), so computing them here
    only moves work from backward to forward.  Caller checks conv_bn_add_supported()."""
    G = rt.groups
    if idn is not None and idn.act != ACT_NONE:
        raise RuntimeError("conv_bn_add: the identity operand must be linear (no pending activation)")
    d = cs.desc(x.shape, x.act, G, x.gs)
    dev = x.data.device
    C, Cin = d.Cout, d.Cin
    count = d.N * d.OH * d.OW
    need_grad = rt.tape.need_grad
    macs = float(count) * G * C * cs.cin_true
    in_b, out_b, w_b = 2.0 * G * count * cs.cin_true, 2.0 * G * count * C, 2.0 * C * cs.cin_true
    kern = "conv_gemm_kernel"
    Gm = sv = None
    if rt.training:
        Gm, sv = _gram_colsum(rt, x, d)
        sums = torch.empty(G * 2 * C, dtype=torch.float64, device=dev)
        call("adamml_gram_stats", ptr(cs.w_fwd), ptr(Gm), ptr(sv), ptr(sums), C, Cin, G)
        if rt.sync.enabled:                  # SyncBatchNorm: the (already collapsed) sums are all-reduced, as SyncCtx.reduce does
            sums = interleave.exchange(sums, rt.sync.group)
        vec = torch.empty(G, 4, C, dtype=torch.float32, device=dev)
        call("adamml_bn_finalize", ptr(sums), 1, G, float(count * rt.sync.world), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
             ptr(bn.running_var), BN_MOMENTUM, BN_EPS, ptr(vec), C)
        rt.touched_bns.append(bn)
    else:
        # [G][4][C] with (scale, shift) of the eval-mode affine in rows 0 / 1 of every group (the epilogue reads nothing else): stream-ordered
        # C-ABI copies, so that a launch plan records them
        ev = _bn_eval_vectors(rt, bn, C, dev)
        c4 = getattr(bn, "_hip_eval_vec4", None)
        if c4 is not None and c4[0] is ev and c4[1].shape[0] == G and hip.recorder is None:
            vec = c4[1]                      # (built from this very eval-affine tensor: valid as long as that cache entry is)
        else:
            vec = torch.empty(G, 4, C, dtype=torch.float32, device=dev)
            for gi in range(G):
                call("adamml_copy2d", ptr(vec[gi]), 2 * C * 4, ptr(ev), 2 * C * 4, 2 * C * 4, 1)
            if hip.recorder is None:
                bn._hip_eval_vec4 = (ev, vec)
    if tpool:
        to = tpool // 2
        out_t = torch.empty(G * d.N // tpool * to, d.OH, d.OW, C, dtype=torch.bfloat16, device=dev)        # POOLED block output
        code_t = torch.empty(G * d.N // tpool * to, d.OH, d.OW, C // 8, dtype=torch.int16, device=dev) if need_grad else None
        mask_t = None
        kern_tp = (kern, "conv1x1_fadd_tpool_kernel", "conv1x1_fadd_tpool_stream_kernel")[hip.load().adamml_conv_fwd_bn_add_tpool_streams(byref(d), tpool)]
        hip.next_meta = (2 * macs, in_b + 1.5 * out_b + w_b + (out_b / 16 if code_t is not None else 0), kern_tp, R_FUSED)
        call("adamml_conv_fwd_bn_add_tpool", byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(vec), ptr(idn.data), None, None, 0,
             act, tpool, ptr(out_t), ptr(code_t))
        full_shape = (G * d.N, d.OH, d.OW, C)
    else:
        out_t = torch.empty(G * d.N, d.OH, d.OW, C, dtype=torch.bfloat16, device=dev)
        mask_t = torch.empty(G * d.N, d.OH, d.OW, C // 8, dtype=torch.uint8, device=dev) if (need_grad and act != ACT_NONE) else None
        hip.next_meta = (2 * macs, in_b + (2 if idn is not None else 1) * out_b + w_b + (out_b / 16 if mask_t is not None else 0), kern, R_FUSED)
        nxt = None
        if (FADD_NEXT and next_cs is not None and rt.training and next_cs.kh * next_cs.kw == 1 and next_cs.stride == 1
                and not next_cs.depthwise and next_cs.cin == C and hip.load().adamml_conv_fwd_bn_add_next_supported(byref(d), next_cs.weight.shape[0])):
            # the NEXT block's conv1 consumes the block-output tile while it is still in LDS (csrc/conv1x1_fadd_next.hip): conv_bn(out, next_cs)
            # finds its raw output and statistics here and launches nothing
            Cn = next_cs.weight.shape[0]
            y_n = torch.empty(G * d.N, d.OH, d.OW, Cn, dtype=torch.bfloat16, device=dev)
            st_n = rt.fwd_arena.take(G * 2 * Cn * STAT_SLOTS)
            hip.next_meta = (2 * macs + 2.0 * count * G * C * Cn, in_b + (2 if idn is not None else 1) * out_b + w_b + 2.0 * G * count * Cn + 2.0 * C * Cn
                             + (out_b / 16 if mask_t is not None else 0), "conv1x1_fadd_next_kernel", R_FUSED)
            call("adamml_conv_fwd_bn_add_next", byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(vec),
                 ptr(idn.data) if idn is not None else None, ptr(idn.scale) if idn is not None else None,
                 ptr(idn.shift) if idn is not None else None, idn.gs if idn is not None else 0, act, ptr(out_t), ptr(mask_t),
                 ptr(next_cs.w_fwd), ptr(y_n), ptr(st_n))
            nxt = (next_cs, y_n, st_n)
        else:
            if idn is not None and hip.load().adamml_conv_fwd_bn_add_streams(byref(d)):
                hip.next_meta = hip.next_meta[:2] + ("conv1x1_fadd_stream_kernel", R_FUSED)       # (the layer-2 shape: csrc/conv1x1_fadd_stream.hip)
            call("adamml_conv_fwd_bn_add", byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(vec),
                 ptr(idn.data) if idn is not None else None, ptr(idn.scale) if idn is not None else None,
                 ptr(idn.shift) if idn is not None else None, idn.gs if idn is not None else 0, act, ptr(out_t), ptr(mask_t))
        full_shape = tuple(out_t.shape)
    out = Lazy(out_t)
    if not tpool:
        out.next_pre = nxt
        if nxt is not None:
            rt.pre_pending += 1
    if not need_grad:
        return out
    # the raw conv output as a (never materialised) lazy tensor: the generic residual machinery only needs its BatchNorm vectors
    z = Lazy(None, vec[0, 0], vec[0, 1], ACT_NONE, gs=4 * C)
    z._shape = torch.Size(full_shape)
    z.vec = vec
    z.alg = True

    def recompute():
        y = torch.empty(full_shape, dtype=torch.bfloat16, device=dev)
        call("adamml_conv_fwd", byref(d), ptr(x.data), ptr(cs.w_fwd), ptr(x.scale), ptr(x.shift), ptr(y), None)
        return y
    z.recompute = recompute
    z.alg_in = (x, d)
    # `blk` = the full-rate block output as the residual machinery sees it; with the pool fused it is never materialised (data None):
    # its gradient arrives from the pool's backward below, already masked, with the sum(g') of bn3 accumulated (res_done)
    blk = out if not tpool else Lazy(None)
    if tpool:
        blk._shape = torch.Size(full_shape)
    blk.res = (z, idn, act, idn_sole, mask_t)

    def conv_bwd():
        if z.grad is None:
            return
        if z.pre_sums is None:
            z.ensure_data()              # the producer of g' could not fuse the BatchNorm-backward sums: the reduction pass reads z
        y = z.data if z.data is not None else _Meta(full_shape, dev)
        _conv1x1_backward_alg(rt, z, x, y, vec, bn, cs, d, count, True, macs, in_b, out_b, w_b, kern, Gm=Gm, sv=sv)
    rt.tape.record(conv_bwd)
    rt.tape.record(lambda: _add_backward(rt, blk, out_t if not tpool else None, z, idn, act, idn_sole, count, C))
    if tpool:
        def pool_bwd():
            g, out.grad = out.grad, None
            if g is None:
                return
            gx = torch.empty(full_shape, dtype=torch.bfloat16, device=dev)
            sa = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS)
            lib = hip.load()
            if TPOOL_PROD and z.alg and x.requires_grad and cs.weight.requires_grad and _alg_supported(cs, d) \
                    and lib.adamml_temporal_pool_bwd_code_prod_supported(tpool, C, d.Cin):
                # the expanded gradient AND the product g'^T a the algebraic backward of conv3 needs first, from one pass (the product
                # kernel read the 4.6 GB of g' back)
                P = torch.empty(G, C, d.Cin, dtype=torch.float32, device=dev)
                ws = hip.scratch(lib.adamml_temporal_pool_bwd_code_prod_workspace(d.N // tpool, tpool, d.OH * d.OW, C, d.Cin, G), dev)
                hip.next_meta = (2 * macs, in_b + 1.5 * out_b + out_b / 16, "tpool_bwd_prod_kernel", R_WGRAD)
                call("adamml_temporal_pool_bwd_code_prod", ptr(g), ptr(code_t), ptr(gx), ptr(sa), ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act,
                     ptr(P), ptr(ws), ws.numel() * 4, d.N // tpool, tpool, d.OH * d.OW, C, d.Cin, G)
                z.prod = P
            else:
                call("adamml_temporal_pool_bwd_code", ptr(g), ptr(code_t), ptr(gx), ptr(sa), d.N // tpool, tpool, d.OH * d.OW, C, G)
            z.pre_sums, z.sums_partial = sa, True
            blk.res_done = True
            blk.grad = gx
        rt.tape.record(pool_bwd)
    return out


class _Meta:
    """Shape / device stand-in for a tensor that was never materialised (only its metadata is consulted)."""

    def __init__(self, shape, device):
        self.shape, self.device = shape, device


def maxpool3x3s2(rt, x, sole_consumer=False):
    """nn.MaxPool2d(3, 2, 1) on a lazy input.  sole_consumer=True promises the pool is the only consumer of x: for a
    train-mode BatchNorm'd x the backward then hands (g_y, idx) to x's BatchNorm backward instead of materialising x.grad."""
    n, h, w, C = x.shape
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = torch.empty(n, oh, ow, C, dtype=torch.bfloat16, device=x.data.device)
    idx = torch.empty(n, oh, ow, C, dtype=torch.uint8, device=x.data.device)
    G = rt.groups
    fuse_bn = sole_consumer and rt.tape.need_grad and x.requires_grad and x.vec is not None
    zsel = torch.empty_like(y) if fuse_bn and POOL_ZSEL else None
    call("adamml_maxpool2d_fwd", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(y), ptr(idx), ptr(zsel), n // G, h, w, C, oh, ow, G)
    out = Lazy(y)
    if rt.tape.need_grad:
        def bwd():
            g = out.grad
            out.grad = None
            if g is None or not x.requires_grad:
                return
            if sole_consumer and x.grad is None and x.vec is not None and x.pre_sums is None:
                x.pool_grad = (g, idx, zsel, oh, ow)
                return
            acc = 1
            if x.grad is None:
                x.grad = torch.empty_like(x.data)
                acc = 0
            call("adamml_maxpool2d_bwd", ptr(g), ptr(idx), ptr(x.grad), n, h, w, C, oh, ow, acc)
        rt.tape.record(bwd)
    return out


def temporal_pool(rt, x, frames, mode, sole_consumer=False):
    """models/common.py:4-33 on [G*N*T,H,W,C]; mode 'max' | 'avg'.  sole_consumer=True promises that the pool is the only
    consumer of x: when x is the output of a residual add, the pool's backward then finishes that add's backward in the
    same pass (adamml_temporal_pool_bwd_res)."""
    nt, h, w, C = x.shape
    G = rt.groups
    nb = nt // frames                       # clips over all groups
    to = (frames - 1) // 2 + 1
    m = {"max": 0, "avg": 1}.get(mode)
    if m is None:
        raise ValueError("only support avg or max")
    y = torch.empty(nb * to, h, w, C, dtype=torch.bfloat16, device=x.data.device)
    call("adamml_temporal_pool_fwd", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(y), nb // G, frames, h * w * C, C, m, G)
    out = Lazy(y)
    if rt.tape.need_grad:
        def bwd():
            g = out.grad
            out.grad = None
            if g is None or not x.requires_grad:
                return
            gx = torch.empty_like(x.data)
            if sole_consumer and x.grad is None and x.scale is None and x.res is not None and not x.res_done and \
                    hip.load().adamml_temporal_pool_bwd_res_supported(frames, C, m):
                z, idn, ract, idn_sole, _ = x.res
                fb = idn is not None and idn_sole and idn.requires_grad and idn.vec is not None and idn.grad is None
                if z.requires_grad and z.vec is not None and z.grad is None and z.pre_sums is None and not fb:
                    sa = rt.bwd_arena.take(G * 2 * C * STAT_SLOTS)
                    call("adamml_temporal_pool_bwd_res", ptr(g), ptr(x.data), ract, ptr(gx), None if z.alg else ptr(z.data), ptr(z.vec), ptr(sa),
                         nb // G, frames, h * w, C, G)
                    z.pre_sums = sa
                    z.sums_partial = z.alg
                    x.res_done = True
                    x.grad = gx
                    return
            call("adamml_temporal_pool_bwd", ptr(g), ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(gx), nb // G, frames,
                 h * w * C, C, m, G)
            _accum_grad(x, gx)
        rt.tape.record(bwd)
    return out


def gap(rt, x):
    """AdaptiveAvgPool2d(1): lazy [N,H,W,C] -> fp32 [N,C]; returns (tensor, grad_setter)."""
    n, h, w, C = x.shape
    G = rt.groups
    f = torch.empty(n, C, dtype=torch.float32, device=x.data.device)
    call("adamml_gap_fwd", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(f), n // G, h * w, C, G)

    def push_grad(gf):
        if not x.requires_grad:
            return
        gx = torch.empty_like(x.data)
        call("adamml_gap_bwd", ptr(gf), ptr(gx), n, h * w, C)
        _accum_grad(x, gx)
    return f, push_grad


def head(rt, x, fc, frames, dropout_p, keep_mask=None):
    """Classifier head of a backbone (models/resnet.py:212-221, models/sound_mobilenet_v2.py:155-158) as one launch per direction:
    lazy x [G*N*T', H, W, C] -> AdaptiveAvgPool2d(1) -> Dropout(p) -> fc -> mean over the T' = `frames` remaining frames of a
    clip -> fp32 logits [G*N, classes].  keep_mask: optional bool [G*N*T', C] replacing the Bernoulli(1-p) draw (parity runs).
    Returns (logits, backward) with backward(g [G*N, classes]) accumulating fc.weight.grad / fc.bias.grad and pushing x's gradient."""
    nt, h, w, C = x.shape
    clips = nt // frames
    K = fc.out_features
    dev = x.data.device
    inv_keep = 1.0
    if keep_mask is None and rt.training and dropout_p > 0:
        keep_mask = torch.rand(nt, C, device=dev) < (1.0 - dropout_p)
        if hip.recorder is not None:        # a launch plan re-draws the mask into this very buffer before every replay
            km = keep_mask
            hip.recorder.pre_fwd.append(lambda: km.copy_(torch.rand(nt, C, device=dev) < (1.0 - dropout_p)))
    elif keep_mask is not None and hip.recorder is not None:
        hip.recorder.failed = "caller-supplied dropout mask"
    if keep_mask is not None:
        inv_keep = 1.0 / (1.0 - dropout_p)
        keep_mask = keep_mask.contiguous()
        keep_mask = keep_mask.view(torch.uint8) if keep_mask.dtype == torch.bool else keep_mask.ne(0).view(torch.uint8)
    feat = torch.empty(nt, C, dtype=torch.float32, device=dev)
    logits = torch.empty(clips, K, dtype=torch.float32, device=dev)
    call("adamml_head_fwd", ptr(x.data), ptr(x.scale), ptr(x.shift), x.gs, x.act, ptr(keep_mask), inv_keep, ptr(fc.weight), ptr(fc.bias),
         ptr(feat), ptr(logits), clips, frames, h * w, C, K, rt.groups)

    def backward(g):
        g = g.contiguous()
        need_w = fc.weight.requires_grad
        gx = torch.empty_like(x.data) if x.requires_grad else None
        g_rows = torch.empty(nt, K, dtype=torch.float32, device=dev) if (need_w and frames > 1) else None
        if gx is not None or g_rows is not None:
            if gx is None:                   # (frozen trunk below a trainable head cannot happen in AdaMML; keep the kernel's contract)
                gx = torch.empty_like(x.data)
            call("adamml_head_bwd", ptr(g), ptr(keep_mask), inv_keep, ptr(fc.weight), ptr(gx), ptr(g_rows), clips, frames, h * w, C, K)
        if need_w:
            gemm_f32(g_rows if g_rows is not None else g, feat, out=fc.weight.grad, trans_a=True, trans_b=False, accumulate=True)   # dW += gy^T feat
            call("adamml_colsum_f32", ptr(g), ptr(fc.bias.grad), clips, K, 1)                                                     # db += sum_n g
        if x.requires_grad:
            _accum_grad(x, gx)
    return logits, backward


def gemm_f32(a, b, out=None, bias=None, act=ACT_NONE, trans_a=False, trans_b=True, accumulate=False):
    """out[M,N] (+)= act(op(a) @ op(b)^T-ish + bias) on 2-D fp32 tensors via adamml_gemm_f32.
    trans_a=False: a is [M,K]; True: a is [K,M].  trans_b=True: b is [N,K]; False: b is [K,N]."""
    if trans_a:
        K, M = a.shape
        a_sm, a_sk = a.stride(1), a.stride(0)
    else:
        M, K = a.shape
        a_sm, a_sk = a.stride(0), a.stride(1)
    if trans_b:
        N, K2 = b.shape
        b_sn, b_sk = b.stride(0), b.stride(1)
    else:
        K2, N = b.shape
        b_sn, b_sk = b.stride(1), b.stride(0)
    if K != K2:
        raise RuntimeError("gemm_f32: inner dimensions differ (%d vs %d)" % (K, K2))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    call("adamml_gemm_f32", ptr(a), a_sm, a_sk, ptr(b), b_sn, b_sk, ptr(out), out.stride(0), out.stride(1), ptr(bias), act,
         1 if accumulate else 0, M, N, K)
    return out


def clip_to_nhwc(x, num_segments, frames, channels, out_hw=None, frame_step=1, cpad=None):
    """AdaMML.data_layer re-layout (models/adamml.py:53-65): [B, S*F*C, H, W] fp32 -> [S, B*Fk, OH, OW, pad8(C)] bf16.
    cpad = 4: 4-channel pixels for a consumer that is the 7x7 stem kernel (ResNet.input_cpad)."""
    hip.require_gpu(x)
    b, sfc, h, w = x.shape
    if sfc != num_segments * frames * channels:
        raise RuntimeError("clip_to_nhwc: channel dim %d != S*F*C = %d*%d*%d" % (sfc, num_segments, frames, channels))
    oh, ow = out_hw if out_hw else (h, w)
    fk = (frames + frame_step - 1) // frame_step
    cp = cpad or pad8(channels)
    x = x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    y = torch.empty(num_segments, b * fk, oh, ow, cp, dtype=torch.bfloat16, device=x.device)
    call("adamml_clip_to_nhwc", ptr(x), ptr(y), b, num_segments, frames, channels, h, w, oh, ow, frame_step, cp)
    return y


def clip_u8_to_nhwc(x, num_segments, frames, channels, mean, std, out_hw=None, frame_step=1, div255=True, cpad=None):
    """Decoded uint8 frames [B, H, W, S*F*C] (the HW(FC) array of utils/video_transforms.py:302-318 `Stack`, batched) ->
    [S, B*Fk, OH, OW, pad8(C)] bf16, normalised as ToTorchFormatTensor + GroupNormalize do (video_transforms.py:62-84,321-343)."""
    hip.require_gpu(x)
    if x.dtype != torch.uint8:
        raise RuntimeError("clip_u8_to_nhwc: expected uint8 frames, got %s" % x.dtype)
    b, h, w, sfc = x.shape
    if sfc != num_segments * frames * channels:
        raise RuntimeError("clip_u8_to_nhwc: last dim %d != S*F*C = %d*%d*%d" % (sfc, num_segments, frames, channels))
    oh, ow = out_hw if out_hw else (h, w)
    fk = (frames + frame_step - 1) // frame_step
    cp = cpad or pad8(channels)
    x = x.contiguous()
    y = torch.empty(num_segments, b * fk, oh, ow, cp, dtype=torch.bfloat16, device=x.device)
    import ctypes
    mean, std = [float(v) for v in mean], [float(v) for v in std]
    call("adamml_clip_u8_to_nhwc", ptr(x), ptr(y), b, num_segments, frames, channels, h, w, oh, ow, frame_step, cp,
         (ctypes.c_float * len(mean))(*mean), (ctypes.c_float * len(std))(*std), len(mean), 1 if div255 else 0)
    return y


def clip_u8_rgbdiff_to_nhwc(x, num_segments, frames, mean, std, out_hw=None, frame_step=1, diffs=5):
    """RGB-diff input computed on the GPU (utils/video_dataset.py:32-38,75-84): decoded RGB frames [B, H, W, S*F*(diffs+1)*3]
    uint8 -- diffs+1 consecutive frames per frame group -> [S, B*Fk, OH, OW, pad8(3*diffs)] bf16 difference channels,
    quantised to uint8 as the reference's loader does, then normalised / resized like every other decoded-frame input."""
    hip.require_gpu(x)
    if x.dtype != torch.uint8:
        raise RuntimeError("clip_u8_rgbdiff_to_nhwc: expected uint8 frames, got %s" % x.dtype)
    b, h, w, last = x.shape
    if last != num_segments * frames * (diffs + 1) * 3:
        raise RuntimeError("clip_u8_rgbdiff_to_nhwc: last dim %d != S*F*(D+1)*3 = %d*%d*%d*3" % (last, num_segments, frames, diffs + 1))
    oh, ow = out_hw if out_hw else (h, w)
    fk = (frames + frame_step - 1) // frame_step
    cp = pad8(3 * diffs)
    y = torch.empty(num_segments, b * fk, oh, ow, cp, dtype=torch.bfloat16, device=x.device)
    import ctypes
    mean, std = [float(v) for v in mean], [float(v) for v in std]
    call("adamml_clip_u8_rgbdiff_to_nhwc", ptr(x.contiguous()), ptr(y), b, num_segments, frames, diffs, h, w, oh, ow, frame_step, cp,
         (ctypes.c_float * len(mean))(*mean), (ctypes.c_float * len(std))(*std), len(mean))
    return y
