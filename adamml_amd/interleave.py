"""Lock-step issue of independent launch sequences whose statistic exchanges are COALESCED into one collective per round.

Under SyncBatchNorm every BatchNorm of every backbone all-reduces a small per-channel statistic vector (<= 16 KB), and the
next layer of that backbone cannot start before the result is back: 209 exchanges in forward and 105 in backward per step
(RGB+Audio), each latency-bound.  Issued backbone after backbone on one communicator they would also serialise the backbones
(torch orders a process group's collectives on one internal stream in host issue order).

Here the host runs the backbones as ROUNDS: every job (one backbone's forward or backward, on its own HIP stream) is a
COROUTINE (greenlet) of the calling thread; it runs until it needs an exchange (`exchange()` / `exchange_stats()`), where it
parks.  When every unfinished job is parked, the scheduler gives each parked vector a slice of the round's persistent flat
buffer (the statistics are collapsed straight into their slice: no concatenation), issues ONE all-reduce for the round on a
communication stream that waits for the jobs' streams, hands each job its slice and resumes them.  The sequence of collectives
is a pure function of the program (identical on every rank, which is all RCCL needs), there is one communicator, and the
count per step drops from the SUM of the backbones' BatchNorm layers (314) to their MAXIMUM per direction and exchange group:
53 + 53 with one group, 2 x (53 + 52) with the default two (GROUPS below: the ResNet-50 and the MobileNetV2s alternate, so that
one group's kernels run while the other's exchange is in flight) -- each carrying all vectors of its group at that depth (a few
tens of KB: still one latency-bound message on xGMI).

Round 3: coroutines instead of threads.  The first form parked every job in its own Python thread behind semaphores; at the
per-GPU share of the reference recipe (B = 9 of a global 72, train_adamml.py:122) the 848 thread hand-overs, the per-round
torch.cat / record_stream / wait_stream bookkeeping and the GIL traffic made the step HOST-bound: 77 ms per step against 21 ms
without SyncBatchNorm (profiles/r03_bench_b9_*.json).  A greenlet switch costs ~1 us, needs no lock, and a job that raises
simply ends -- nothing stays parked.  torch's current stream and grad mode are thread-local, i.e. shared by the coroutines of
one thread: the scheduler swaps them at every switch.  A coroutine must not park INSIDE a torch dispatcher call (the
dispatcher's key guards are thread-local too): the jobs therefore call the backbones' launch sequences directly
(HipBackbone.run_raw), and the results are attached to autograd afterwards (HipBackbone.call(..., precomputed=...)).
"""
import os

import torch
import torch.distributed as dist

try:
    import greenlet                                # requirements.txt: the only dependency besides torch / numpy
except ImportError:                                # pragma: no cover
    greenlet = None

from . import hip

ENABLED = os.environ.get("ADAMML_INTERLEAVE", "1") != "0"      # A/B aid; only ever used when SyncBatchNorm is on
# Exchange groups.  1: every job parks in the same round and ONE collective carries all their vectors (fewest collectives; but while that
# collective's latency chain runs -- collapse kernel, event hops into and out of the process group's stream, the RCCL kernel: ~110 us --
# no stream has anything to run: B = 72 with the configs[2] choreography forced on one rank 127.0 ms against 115.2 without).  2 (default):
# the first job (the ResNet-50 of the main net) and the other jobs (the MobileNetV2s) exchange in ALTERNATING collectives A0 B0 A1 B1 ...:
# while one group's exchange is in flight the other group's kernels run.  Twice the collectives -- still one communicator, still a sequence
# that is a pure function of the program, identical on every rank -- and measured on one rank at B = 72 / 36 / 18 / 9 (launch plans):
# 127.0 / 71.5 / 43.8 / 29.3 -> 122.2 / 66.3 / 39.2 / 26.0 ms per step.
# Every collective costs ~65 us of host time; an EAGER step at the per-GPU share of the reference recipe (9 videos) is bound by the host
# (37.3 ms with two groups against 30.4 with one), so "auto" alternates only when the step is not: launch plans on (26.2 against 29.3 ms),
# or at least SMALL_CLIPS clips per rank (the forward call leaves its clip count in `clips_hint` for the backward call).
GROUPS = os.environ.get("ADAMML_SYNC_GROUPS", "auto")            # "1" | "2" | "auto"
SMALL_CLIPS = 80
clips_hint = [0]
_resolved = [None]
DIRECT = os.environ.get("ADAMML_SYNC_DIRECT", "1") != "0"      # single-job groups exchange on the job's own stream (A/B aid)
stats = {"collectives": 0, "coalesced_vectors": 0}            # counters (tests, design notes)
_current = [None]                                             # the job whose coroutine is running (None: plain code)


def active():
    """True while the caller runs inside a job of run_interleaved()."""
    return _current[0] is not None


def yield_point():
    """Hand the turn to the next job without an exchange; a no-op outside run_interleaved()."""
    job = _current[0]
    if job is not None:
        job.park()


def exchange(t, group=None):
    """All-reduce(sum) of the statistic vector t over `group`; returns the reduced tensor (t itself, or a slice of the round's
    flat buffer).  Inside run_interleaved() the call parks the job until the round's coalesced collective has been issued."""
    job = _current[0]
    rec = hip.recorder
    if rec is not None:
        # a launch plan is being recorded: the exchange ends a segment and is repeated, with these very tensors, at every replay
        if job is None:
            rec.failed = "SyncBatchNorm exchange outside the lock-step scheduler (no persistent round buffer)"
        rec.keep.append(t)
        hip.recorder = None
        rec.boundary(lambda: exchange(t, group))
    if job is None:
        dist.all_reduce(t, group=group)
        stats["collectives"] += 1
        stats["coalesced_vectors"] += 1
        out = t
    else:
        job.pending = ("vec", t, group, 0, 0)
        job.park()
        out, job.reduced = job.reduced, None
    if rec is not None:
        rec.keep.append(out)
        hip.recorder = rec
    return out


def exchange_stats(acc, C, groups, group=None):
    """acc: [groups][STAT_SLOTS][2C] slot-interleaved fp64 accumulators of one BatchNorm.  Returns the all-reduced collapsed sums
    [groups * 2C] (fp64; the finalize kernels read them with nslots = 1).  Inside run_interleaved() the collapse kernel writes
    straight into this job's slice of the round's flat buffer."""
    job = _current[0]
    rec = hip.recorder
    if rec is not None:
        if job is None:
            rec.failed = "SyncBatchNorm exchange outside the lock-step scheduler (no persistent round buffer)"
        rec.keep.append(acc)
        hip.recorder = None
        rec.boundary(lambda: exchange_stats(acc, C, groups, group))
    if job is None:
        out = torch.empty(groups * 2 * C, dtype=torch.float64, device=acc.device)
        hip.call("adamml_stats_collapse", hip.ptr(acc), hip.ptr(out), C, groups)
        dist.all_reduce(out, group=group)
        stats["collectives"] += 1
        stats["coalesced_vectors"] += 1
    else:
        job.pending = ("stats", acc, group, C, groups)
        job.park()
        out, job.reduced = job.reduced, None
    if rec is not None:
        rec.keep.append(out)
        hip.recorder = rec
    return out


class _Job:
    def __init__(self, fn, stream, grad_enabled, sched):
        self.fn, self.stream, self.grad, self.sched = fn, stream, grad_enabled, sched
        self.done = False
        self.result = None
        self.error = None
        self.pending = None                 # ("vec" | "stats", tensor, group, C, groups) parked at an exchange
        self.reduced = None
        self.recorder = None                # hip.recorder of this coroutine while it is parked
        self.g = greenlet.greenlet(self._main, parent=sched)

    def _main(self):
        try:
            self.result = self.fn()
        except BaseException as e:          # re-raised by the scheduler
            self.error = e
        self.done = True                    # returning switches to the parent (the scheduler)

    def park(self):
        self.sched.switch()                 # back to the scheduler; returns when the scheduler resumes this job


class _Round:
    """Persistent per-round state: the flat exchange buffer and the events of its stream choreography (reused every step:
    the sizes are a function of the model, and a round's buffer is consumed -- stream-ordered -- long before the same round of
    the next step is produced)."""
    __slots__ = ("flat", "done", "ready")

    def __init__(self):
        self.flat = None
        self.done = torch.cuda.Event()
        self.ready = []


_rounds = {}
_comm = {}


def _comm_stream(device):
    s = _comm.get(device)
    if s is None:
        s = _comm[device] = torch.cuda.Stream(device=device)
    return s


def _vec_dtype(p):
    return torch.float64 if p[0] == "stats" else p[1].dtype


def _coalesced_all_reduce(parked, device, phase, ridx):
    """ONE collective for the vectors parked in this round (all on `device`, one process group)."""
    group = parked[0].pending[2]
    on_gpu = parked[0].pending[1].is_cuda
    dtype = _vec_dtype(parked[0].pending)
    if any(j.pending[2] is not group or _vec_dtype(j.pending) != dtype for j in parked):
        for j in parked:                    # mixed groups / dtypes: no coalescing (not used by the hot path)
            kind, t, grp, C, G = j.pending
            prev = torch.cuda.current_stream(device) if on_gpu else None
            if on_gpu:
                torch.cuda.set_stream(j.stream)
            try:
                if kind == "stats":
                    out = torch.empty(G * 2 * C, dtype=torch.float64, device=t.device)
                    hip.call("adamml_stats_collapse", hip.ptr(t), hip.ptr(out), C, G)
                    t = out
                dist.all_reduce(t, group=grp)
            finally:
                if on_gpu:
                    torch.cuda.set_stream(prev)
            stats["collectives"] += 1
            stats["coalesced_vectors"] += 1
            j.reduced, j.pending = t, None
        return
    if not on_gpu:                          # host tensors (the gloo tests of the scheduling logic): concatenate
        tensors = [j.pending[1] for j in parked]
        flat = torch.cat([t.reshape(-1) for t in tensors]) if len(tensors) > 1 else tensors[0].reshape(-1)
        dist.all_reduce(flat, group=group)
        stats["collectives"] += 1
        stats["coalesced_vectors"] += len(tensors)
        off = 0
        for j, t in zip(parked, tensors):
            j.reduced, j.pending = flat[off:off + t.numel()].view(t.shape), None
            off += t.numel()
        return
    sizes = [(j.pending[1].numel() if j.pending[0] == "vec" else j.pending[4] * 2 * j.pending[3]) for j in parked]
    total = sum(sizes)
    rd = _rounds.get((device, phase, ridx, dtype))
    if rd is None:
        rd = _rounds[(device, phase, ridx, dtype)] = _Round()
    if rd.flat is None or rd.flat.numel() < total:
        rd.flat = torch.empty(max(total, 1024), dtype=dtype, device=device)
    while len(rd.ready) < len(parked):
        rd.ready.append(torch.cuda.Event())
    flat = rd.flat[:total]
    prev = torch.cuda.current_stream(device)
    if len(parked) == 1 and DIRECT:
        # a group of ONE job (the ResNet-50 under alternating groups -- the long pole of the step): the collective is issued on the job's
        # own stream.  No communication stream, no events of ours: the round's latency chain loses two of its four cross-stream hops
        # (job -> comm -> process group's stream -> comm -> job becomes job -> process group's stream -> job)
        j = parked[0]
        kind, t, _, C, G = j.pending
        try:
            torch.cuda.set_stream(j.stream)
            if kind == "stats":
                hip.call("adamml_stats_collapse", hip.ptr(t), hip.ptr(flat), C, G)
            else:
                flat.copy_(t.reshape(-1))
            dist.all_reduce(flat, group=group)
        finally:
            torch.cuda.set_stream(prev)
        stats["collectives"] += 1
        stats["coalesced_vectors"] += 1
        j.reduced = flat if kind == "stats" else flat.view(t.shape)
        j.pending = None
        return
    comm = _comm_stream(device)
    try:
        # every job's vector is collapsed / copied into its slice ON THE JOB'S OWN STREAM (the collapse kernels of the round then run
        # concurrently, not one after the other in front of the collective); the communication stream waits for all of them
        off = 0
        for j, n, ev in zip(parked, sizes, rd.ready):
            kind, t, _, C, G = j.pending
            torch.cuda.set_stream(j.stream)
            if kind == "stats":
                hip.call("adamml_stats_collapse", hip.ptr(t), hip.ptr(flat[off:off + n]), C, G)
            else:
                flat[off:off + n].copy_(t.reshape(-1))
            off += n
            ev.record(j.stream)
            comm.wait_event(ev)
        torch.cuda.set_stream(comm)
        dist.all_reduce(flat, group=group)
        rd.done.record(comm)
    finally:
        torch.cuda.set_stream(prev)
    stats["collectives"] += 1
    stats["coalesced_vectors"] += len(parked)
    off = 0
    for j, n in zip(parked, sizes):
        t = j.pending[1]
        j.reduced = flat[off:off + n] if j.pending[0] == "stats" else flat[off:off + n].view(t.shape)
        off += n
        j.stream.wait_event(rd.done)        # the job's next launch reads its slice of the reduced buffer
        j.pending = None


def run_interleaved(jobs, device, phase="fwd", groups=None):
    """jobs: list of (callable, stream or None).  Runs them to completion in rounds: every unfinished job gets one turn per round,
    in fixed order, and runs until it parks at an exchange / yield_point() or finishes; the exchanges parked by the jobs of one
    EXCHANGE GROUP in a round are all-reduced as ONE collective (GROUPS above: all jobs, or {first job} / {the others} alternating).
    Returns the list of results; the first exception is re-raised (the other jobs' coroutines are dropped: nothing stays parked).
    `phase` names the persistent round buffers ("fwd" / "bwd"); `groups`: 1 or 2, default by GROUPS."""
    if _current[0] is not None:
        raise RuntimeError("run_interleaved: nested call from inside a job")
    if greenlet is None:
        raise RuntimeError("SyncBatchNorm over several backbones issues them as coroutines and needs the `greenlet` package "
                           "(requirements.txt; pip install greenlet), or ADAMML_INTERLEAVE=0 for sequential issue (one exchange per "
                           "BatchNorm instead of one per BatchNorm depth)")
    sched = greenlet.getcurrent()
    ge = torch.is_grad_enabled()
    on_gpu = device is not None and torch.device(device).type == "cuda"
    dev = torch.device(device) if on_gpu else None
    home = torch.cuda.current_stream(dev) if on_gpu else None
    js = [_Job(fn, (stream if stream is not None else home), ge, sched) for fn, stream in jobs]
    if groups is None:
        if _resolved[0] is None:
            # decided ONCE per process, at the first SyncBatchNorm step: every rank must take the same decision for the whole run (the
            # sequence of collectives depends on it), and the inputs -- launch plans on / off, the per-rank batch of the first step --
            # are configuration, equal on all ranks of a job (DistributedSampler hands every rank equally sized batches)
            from . import plan
            _resolved[0] = 1 if GROUPS == "1" else 2 if (GROUPS == "2" or plan.ENABLED or clips_hint[0] >= SMALL_CLIPS) else 1
        groups = _resolved[0]
    two = len(js) > 1 and groups == 2
    groups = [js[:1], js[1:]] if two else [js]
    active_jobs = list(js)
    ridx = 0
    try:
        while active_jobs:
            for gi, members in enumerate(groups):
                for j in members:
                    if j not in active_jobs:
                        continue
                    _current[0] = j
                    if on_gpu:
                        torch.cuda.set_stream(j.stream)
                    torch.set_grad_enabled(j.grad)
                    hip.recorder = j.recorder                  # (a job may be recording a launch plan: per coroutine, like the stream)
                    try:
                        j.g.switch()
                    finally:
                        j.recorder, hip.recorder = hip.recorder, None
                        j.grad = torch.is_grad_enabled()
                        _current[0] = None
                        if on_gpu:
                            torch.cuda.set_stream(home)
                        torch.set_grad_enabled(ge)
                    if j.done:
                        active_jobs.remove(j)
                        if j.error is not None:
                            raise j.error
                parked = [j for j in members if j in active_jobs and j.pending is not None]
                if parked:
                    _coalesced_all_reduce(parked, dev, "%s%d" % (phase, gi) if two else phase, ridx)
            ridx += 1
    finally:
        _current[0] = None
    return [j.result for j in js]
