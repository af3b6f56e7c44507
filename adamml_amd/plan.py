"""Launch plans: record the static launch sequence of one backbone call once, replay it with one C call per segment.

A backbone call (HipBackbone._run and the reverse tape it returns) issues several hundred C-ABI launches from Python: ~13 us of
interpreter, ctypes marshalling and tensor bookkeeping per launch.  At the benchmark batch (72 videos per GPU) that is hidden behind
120 ms of device work; at the per-GPU share of the reference's own recipe (global batch 72 over 8 GPUs = 9 videos per GPU,
train_adamml.py:122) the device needs ~20 ms and the step is bound by the host.  The shapes of a call are static per (input shape, mode),
so the sequence of entry points, descriptors and pointers is identical every step.  A `Recorder` listens to one ordinary (eager) call --
hip.call() reports every launch, hip.ptr() every tensor whose address a launch received (the recorder keeps them alive: the plan OWNS
the activations of its call, they are never returned to the caching allocator) -- and produces a `Plan`: segments of fixed-size records
(include/adamml_hip.h: adamml_plan_op_t) replayed by adamml_plan_run, separated by the few points where Python has to act:
a SyncBatchNorm exchange (interleave.exchange*), a gradient-bucket hook of the data-parallel wrapper.  What changes from step to step
goes through pointer slots (slot 0: the call's input tensor, slot 1: the incoming output gradient); the dropout keep-mask is re-drawn
into its fixed buffer by a pre-replay hook, `num_batches_tracked` is advanced by a post-replay hook.

Opt-in (`ADAMML_LAUNCH_PLAN=1`, `plan.ENABLED = True`, `bench.py --launch-plan`): a plan pins every intermediate tensor of its call, i.e.
the SUM of the call's allocations instead of their peak -- right for small per-GPU batches, wrong for B = 72.  Inference calls can be
planned too (EVAL_ENABLED, serving-sized inputs up to MAX_EVAL_ELEMENTS: their eval-mode BatchNorm affines are recomputed inside the plan,
policy-gated inference pads its selected-clip count to a multiple of 8 so that few distinct shapes occur) -- correct, but measured to gain
nothing: see EVAL_ENABLED."""
import ctypes
import os
import struct
import weakref

import torch

from . import hip

ENABLED = os.environ.get("ADAMML_LAUNCH_PLAN", "0") not in ("", "0")
# inference calls as well (serving-sized inputs only).  Off by default: measured on MI355X, policy-gated inference at B = 1 / 4 / 8 videos
# takes 5.5 / 5.9 / 6.7 ms per batch eagerly and 5.2 / 6.0 / 7.6 ms with plans -- it is bound by the DEVICE-side chain of ~330 dependent tiny
# launches of the two policy backbones and by the host sync on the decisions, not by the host's issue rate
EVAL_ENABLED = os.environ.get("ADAMML_LAUNCH_PLAN_EVAL", "0") not in ("", "0")
WARMUP_CALLS = 2               # eager calls of a key before it is recorded (arenas and scratch buffers have reached their sizes)
MAX_PLANS_PER_NET = 4          # training plans kept per backbone (each pins the activations of one call shape / mode)
MAX_EVAL_PLANS_PER_NET = 12    # inference plans kept per backbone (policy-gated inference: one per padded clip count)
MAX_EVAL_ELEMENTS = 40 * 8 * 224 * 224 * 4     # largest inference input planned (40 clips of 8 RGB frames): serving-sized calls
MAX_ARGS = 21
KIND_CALL, KIND_WAIT, KIND_ZERO = 0, 1, 2
stats = {"recorded": 0, "replayed_segments": 0, "replayed_ops": 0}


class PlanOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("fn", ctypes.c_int32), ("nargs", ctypes.c_int32), ("stream", ctypes.c_int32),
                ("slot_mask", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("a", ctypes.c_uint64 * MAX_ARGS)]


_FN_IDS = None


def fn_id(name):
    global _FN_IDS
    if _FN_IDS is None:
        _FN_IDS = {n: i for i, n in enumerate(sorted(hip.SIGNATURES))}
    return _FN_IDS[name]


def _f64_bits(v):
    return struct.unpack("<Q", struct.pack("<d", float(v)))[0]


class Segment:
    __slots__ = ("ops", "n", "boundary")

    def __init__(self, ops, boundary):
        self.n = len(ops)
        self.ops = (PlanOp * max(self.n, 1))(*ops)
        self.boundary = boundary              # callable run after the segment (may park the coroutine: SyncBN exchange), or None


class Recorder:
    """Listens to one eager backbone call (forward, later its backward) and builds the Plan."""

    def __init__(self, x):
        self.keep = [x]                       # tensors / host structures whose addresses the records hold
        self.streams = []                     # raw hipStream_t handles, in slot order (slot 0: the stream the call was issued on)
        self.slots = {x.data_ptr(): 0}        # pointer value -> slot index
        self.n_events = 0
        self.cur = []
        self.fwd, self.bwd = [], None
        self.pre_fwd, self.post_fwd, self.pre_bwd = [], [], []
        self.failed = None
        self.retry = False                    # failed for a reason that will not recur (arena sized inside the recording): try again

    # ---- called by hip.call / hip.ptr / the runtime while recording ----------------------------------------------------
    def stream_slot(self, handle):
        if handle not in self.streams:
            self.streams.append(handle)
        return self.streams.index(handle)

    def call(self, name, args, stream):
        kinds = hip.SIGNATURES[name][:-1]
        if len(args) != len(kinds) or len(args) > MAX_ARGS:
            self.failed = "%s: %d arguments" % (name, len(args))
            return
        op = PlanOp()
        op.kind, op.fn, op.nargs, op.stream = KIND_CALL, fn_id(name), len(args), self.stream_slot(stream)
        mask = 0
        for j, (v, k) in enumerate(zip(args, kinds)):
            if k is hip._DESC:
                d = hip.ConvDesc.from_buffer_copy(v._obj)
                self.keep.append(d)
                op.a[j] = ctypes.addressof(d)
            elif k is hip._P:
                if v is None:
                    op.a[j] = 0
                elif isinstance(v, int):
                    if v in self.slots:
                        mask |= 1 << j
                        op.a[j] = self.slots[v]
                    else:
                        op.a[j] = v
                else:                         # a host array handed to the entry point (ctypes): keep it, pass its address
                    self.keep.append(v)
                    op.a[j] = ctypes.addressof(v)
            elif k in (hip._F, hip._D):
                op.a[j] = _f64_bits(v)
            else:
                op.a[j] = int(v) & 0xFFFFFFFFFFFFFFFF
        op.slot_mask = mask
        self.cur.append(op)

    def wait(self, waiting_stream, on_stream):
        """`waiting_stream` waits for everything enqueued so far on `on_stream` (raw handles)."""
        op = PlanOp()
        op.kind, op.stream, op.nargs = KIND_WAIT, self.stream_slot(waiting_stream), 2
        op.a[0], op.a[1] = self.stream_slot(on_stream), self.n_events
        self.n_events += 1
        self.cur.append(op)

    def zero(self, tensor, stream):
        op = PlanOp()
        op.kind, op.stream, op.nargs = KIND_ZERO, self.stream_slot(stream), 2
        op.a[0], op.a[1] = tensor.data_ptr(), tensor.numel() * tensor.element_size()
        self.keep.append(tensor)
        self.cur.append(op)

    def boundary(self, fn):
        """Ends the current segment; fn() runs between it and the next one at replay."""
        (self.fwd if self.bwd is None else self.bwd).append(Segment(self.cur, fn))
        self.cur = []

    # ---- phases ------------------------------------------------------------------------------------------------------------
    def end_forward(self):
        self.fwd.append(Segment(self.cur, None))
        self.cur = []

    def begin_backward(self, g):
        self.bwd = []
        self.slots[g.data_ptr()] = 1

    def end_backward(self):
        self.bwd.append(Segment(self.cur, None))
        self.cur = []


class Plan:
    def __init__(self, rec, out):
        self.rec = rec                        # (keeps every recorded tensor alive)
        self.out = out
        self.streams = (ctypes.c_void_p * len(rec.streams))(*rec.streams)
        self.n_streams = len(rec.streams)
        self.events = (ctypes.c_void_p * max(rec.n_events, 1))()
        self.n_events = rec.n_events
        if rec.n_events:
            rc = hip.load().adamml_plan_events_create(self.events, rec.n_events)
            if rc:
                raise RuntimeError("adamml_plan_events_create failed: %s" % hip.load().adamml_last_error_string().decode())
        self.slot_arr = (ctypes.c_uint64 * 2)()
        self._live_tape = None                # weakref to the PlanTape of a replayed forward whose backward has not run yet

    def busy(self):
        """True while a replayed forward of this plan still waits for its backward: the plan owns ONE set of activation buffers
        (saved activations, dropout mask, statistic arenas), so a second forward before that backward would overwrite what the first
        one's backward reads (two model calls summed into one loss, gradient accumulation with a deferred backward).  The caller
        (HipBackbone.run_planned) then runs that call eagerly.  A forward whose graph was dropped without a backward frees the plan
        when its tape is collected."""
        t = self._live_tape() if self._live_tape is not None else None
        return t is not None and not t.done

    def __del__(self):
        try:
            if self.n_events:
                hip.load().adamml_plan_events_destroy(self.events, self.n_events)
        except Exception:
            pass

    def _run(self, segments, x_ptr, g_ptr):
        lib = hip.load()
        self.slot_arr[0], self.slot_arr[1] = x_ptr, g_ptr
        # the call may be issued on another stream than the recorded one (slot 0 is "the caller's stream")
        self.streams[0] = hip._stream()
        for seg in segments:
            if seg.n:
                rc = lib.adamml_plan_run(seg.ops, seg.n, self.streams, self.n_streams, self.events, self.n_events, self.slot_arr, 2)
                if rc != 0:
                    raise RuntimeError("adamml_plan_run failed (%d): %s" % (rc, lib.adamml_last_error_string().decode()))
                stats["replayed_segments"] += 1
                stats["replayed_ops"] += seg.n
            if seg.boundary is not None:
                seg.boundary()

    def forward(self, x):
        for h in self.rec.pre_fwd:
            h()
        self._run(self.rec.fwd, x.data_ptr(), 0)
        for h in self.rec.post_fwd:
            h()
        tape = PlanTape(self, x)
        if self.rec.bwd is not None:          # (a plan recorded without a backward owns nothing a later call could clobber)
            self._live_tape = weakref.ref(tape)
        return self.out, tape

    def backward(self, x, g):
        for h in self.rec.pre_bwd:
            h()
        self._run(self.rec.bwd, x.data_ptr(), g.data_ptr())


class PlanTape:
    """Stands in for runtime.Tape on a replayed call: backward() replays the recorded reverse tape."""

    def __init__(self, plan, x):
        self.plan, self.x = plan, x
        self.need_grad = True
        self.grad_out = None
        self.recorder = None
        self.done = False

    def backward(self):
        if self.done:
            raise RuntimeError("launch plan: the backward of this call has already run (retain_graph is not supported by a replayed "
                               "call: its activation buffers belong to the plan; set ADAMML_LAUNCH_PLAN=0)")
        if self.plan._live_tape is None or self.plan._live_tape() is not self:
            raise RuntimeError("launch plan: the activations of this call were overwritten by a later replay of the same plan")
        g = self.grad_out
        if not g.is_contiguous():
            g = g.contiguous()
        self._g = g                           # (alive until the replayed kernels have been enqueued; stream-ordered afterwards)
        self.plan.backward(self.x, g)
        self.done = True
