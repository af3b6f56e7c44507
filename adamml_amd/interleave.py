"""Deterministic interleaving of independent launch sequences that all issue collectives on ONE communicator.

Under SyncBatchNorm every BatchNorm of every backbone all-reduces its statistics, and torch serialises the collectives of a
process group on one internal stream in host issue order.  If the host issues backbone A completely and then backbone B,
B's first exchange queues behind A's last one: the side-stream backbones no longer overlap the ResNet, they run after it
(and in backward the ResNet runs after the sound net).  Here the host issues the backbones ROUND-ROBIN, one collective per
turn: each job runs in its own thread (torch's current stream and grad mode are thread-local, so a job keeps the stream it
was given), but only ONE job runs at any time and the hand-over points are the collectives themselves
(`SyncCtx.reduce` calls `yield_point()`), so the sequence of collectives is a pure function of the program -- identical
on every rank, which is all RCCL needs.  In the communicator's queue the exchanges of the backbones now alternate, each
waiting only for its own backbone's previous layer: the small nets advance one layer per ResNet layer and finish with it.
"""
import os
import threading

import torch

_local = threading.local()
ENABLED = os.environ.get("ADAMML_INTERLEAVE", "1") != "0"      # A/B aid; only ever used when SyncBatchNorm is on


def yield_point():
    """Called at every collective of a job; a no-op outside run_interleaved()."""
    job = getattr(_local, "job", None)
    if job is not None:
        job.handoff()


class _Job:
    def __init__(self, fn, stream, device, grad_enabled, back):
        self.fn, self.stream, self.device, self.grad_enabled, self.back = fn, stream, device, grad_enabled, back
        self.go = threading.Semaphore(0)
        self.done = False
        self.result = None
        self.error = None
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


def run_interleaved(jobs, device):
    """jobs: list of (callable, stream or None).  Runs them to completion, one turn (= up to and including one collective)
    at a time in fixed round-robin order.  Returns the list of results; the first exception is re-raised."""
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
    for j in js:
        j.thread.join()
    return [j.result for j in js]
