"""Lock-step issue of independent launch sequences whose statistic exchanges are COALESCED into one collective per round.

Under SyncBatchNorm every BatchNorm of every backbone all-reduces a small per-channel statistic vector (<= 16 KB), and the
next layer of that backbone cannot start before the result is back: 209 exchanges in forward and 105 in backward per step
(RGB+Audio), each latency-bound.  Issued backbone after backbone on one communicator they would also serialise the backbones
(torch orders a process group's collectives on one internal stream in host issue order).

Here the host runs the backbones as ROUNDS: every job (one backbone's forward or backward, on its own HIP stream) runs in
its own thread -- torch's current stream and grad mode are thread-local -- but only ONE job runs at any time, and it runs
until it needs an exchange (`exchange()`), where it parks.  When every unfinished job is parked, the scheduler concatenates
the parked vectors into one flat buffer, issues ONE all-reduce for the round on a communication stream that waits for the
jobs' streams, hands each job its slice of the result and resumes them.  The sequence of collectives is a pure function of
the program (identical on every rank, which is all RCCL needs), there is one communicator, and the count per step drops from
the SUM of the backbones' BatchNorm layers to their MAXIMUM per direction: 53 + 53 for ResNet-50 + MobileNetV2s instead of
314 -- each carrying all backbones' vectors of that depth (a few tens of KB: still one latency-bound message on xGMI).
"""
import os
import threading

import torch
import torch.distributed as dist

_local = threading.local()
ENABLED = os.environ.get("ADAMML_INTERLEAVE", "1") != "0"      # A/B aid; only ever used when SyncBatchNorm is on
stats = {"collectives": 0, "coalesced_vectors": 0}            # counters (tests, design notes)


def yield_point():
    """Hand the turn to the next job without an exchange; a no-op outside run_interleaved()."""
    job = getattr(_local, "job", None)
    if job is not None:
        job.handoff()


def exchange(t, group=None):
    """All-reduce(sum) of the statistic vector t over `group`; returns the reduced tensor (t itself, or a slice of the round's
    flat buffer).  Inside run_interleaved() the call parks the job until the round's coalesced collective has been issued."""
    job = getattr(_local, "job", None)
    if job is None:
        dist.all_reduce(t, group=group)
        stats["collectives"] += 1
        stats["coalesced_vectors"] += 1
        return t
    job.pending = (t, group)
    job.handoff()
    out, job.reduced = job.reduced, None
    return out


class _Job:
    def __init__(self, fn, stream, device, grad_enabled, back):
        self.fn, self.stream, self.device, self.grad_enabled, self.back = fn, stream, device, grad_enabled, back
        self.go = threading.Semaphore(0)
        self.done = False
        self.result = None
        self.error = None
        self.pending = None                 # (tensor, group) parked at exchange()
        self.reduced = None
        self.thread = threading.Thread(target=self._main, daemon=True)

    def _main(self):
        self.go.acquire()
        _local.job = self
        try:
            if self.device is not None and torch.device(self.device).type == "cuda":
                torch.cuda.set_device(self.device)
            with torch.set_grad_enabled(self.grad_enabled):
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        self.result = self.fn()
                else:
                    self.result = self.fn()
        except BaseException as e:          # re-raised in the caller's thread
            self.error = e
        finally:
            _local.job = None
            self.done = True
            self.back.release()

    def handoff(self):
        self.back.release()                 # give the turn back to the scheduler ...
        self.go.acquire()                   # ... and wait for the next one


def _coalesced_all_reduce(parked, device):
    """ONE collective for the vectors parked in this round (all on `device`, one process group)."""
    group = parked[0].pending[1]
    tensors = [j.pending[0] for j in parked]
    on_gpu = tensors[0].is_cuda
    if any(j.pending[1] is not group for j in parked) or len({t.dtype for t in tensors}) > 1:
        for j in parked:                    # mixed groups / dtypes: no coalescing (not used by the hot path)
            dist.all_reduce(j.pending[0], group=j.pending[1])
            stats["collectives"] += 1
            stats["coalesced_vectors"] += 1
            j.reduced, j.pending = j.pending[0], None
        return
    if on_gpu:
        comm = _comm_stream(tensors[0].device)
        for j in parked:
            comm.wait_stream(j.stream if j.stream is not None else torch.cuda.current_stream(tensors[0].device))
        ctx = torch.cuda.stream(comm)
    else:
        comm, ctx = None, _Null()
    with ctx:
        flat = torch.cat([t.reshape(-1) for t in tensors]) if len(tensors) > 1 else tensors[0].reshape(-1)
        dist.all_reduce(flat, group=group)
    stats["collectives"] += 1
    stats["coalesced_vectors"] += len(tensors)
    off = 0
    for j, t in zip(parked, tensors):
        n = t.numel()
        j.reduced = flat[off:off + n].view(t.shape)
        off += n
        if on_gpu:
            s = j.stream if j.stream is not None else torch.cuda.current_stream(t.device)
            s.wait_stream(comm)             # the job's next launch reads its slice of the reduced buffer
            flat.record_stream(s)           # (a few KB: the cross-stream bookkeeping of the caching allocator is harmless here)
        j.pending = None


_comm = {}


def _comm_stream(device):
    s = _comm.get(device)
    if s is None:
        s = _comm[device] = torch.cuda.Stream(device=device)
    return s


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def run_interleaved(jobs, device):
    """jobs: list of (callable, stream or None).  Runs them to completion in rounds: every unfinished job gets one turn per round,
    in fixed order, and runs until it parks at exchange() / yield_point() or finishes; the exchanges parked in a round are
    all-reduced as ONE collective.  Returns the list of results; the first exception is re-raised."""
    back = threading.Semaphore(0)
    ge = torch.is_grad_enabled()
    js = [_Job(fn, stream, device, ge, back) for fn, stream in jobs]
    for j in js:
        j.thread.start()
    active = list(js)
    while active:
        for j in list(active):
            j.go.release()
            back.acquire()
            if j.done:
                active.remove(j)
                if j.error is not None:
                    raise j.error           # (the other jobs' daemon threads stay parked; the step is lost anyway)
        parked = [j for j in active if j.pending is not None]
        if parked:
            _coalesced_all_reduce(parked, device)
    for j in js:
        j.thread.join()
    return [j.result for j in js]
