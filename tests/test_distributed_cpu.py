"""world_size-2 gloo tests of the data-parallel host logic (runs on CPU): batch sharding, flat-buffer gradient averaging
in HipDDP, and the SyncBatchNorm statistic exchange (sum / sum-of-squares all-reduce == statistics of the concatenated
batch), using the oracle as the arithmetic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _grad_average(rank, world):
    from adamml_amd.backbone import FlatBuffers
    from adamml_amd.distributed import HipDDP, shard_batch

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 3)
            self.b = torch.nn.Linear(3, 2)
            self.fb = FlatBuffers(self)

        def flat_grad_buffers(self):
            return [self.fb.flat_grad]

        def forward(self, x):
            return self.b(torch.tanh(self.a(x)))

    torch.manual_seed(0)
    m = Tiny()
    m.fb.ensure(torch.device("cpu"))
    m.fb.ensure_grads()
    x = torch.arange(32, dtype=torch.float32).reshape(8, 4) / 10
    (xs,) = shard_batch([x], rank, world)
    assert xs.shape[0] == 4 and torch.equal(xs[0], x[rank])
    ddp = HipDDP(m)
    ddp.broadcast_parameters()
    ddp(xs).pow(2).mean().backward()
    # a straggler gradient that does NOT live in the flat buffer must be averaged as well
    m.extra = torch.nn.Parameter(torch.ones(3))
    m.extra.grad = torch.full((3,), float(rank + 1))
    ddp.reduce_gradients()
    # views stay attached to the flat buffer
    assert m.a.weight.grad.data_ptr() == m.fb.flat_grad.data_ptr()
    return m.fb.flat_grad.clone(), m.extra.grad.clone()


def test_flat_gradient_allreduce_matches_full_batch():
    out = _run(_grad_average)
    g0, e0 = out[0]
    g1, e1 = out[1]
    assert torch.allclose(g0, g1)
    assert torch.allclose(e0, torch.full((3,), 1.5)) and torch.allclose(e1, e0)
    # single-process reference on the concatenated batch
    torch.manual_seed(0)
    a, b = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2)
    x = torch.arange(32, dtype=torch.float32).reshape(8, 4) / 10
    b(torch.tanh(a(x))).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in list(a.parameters()) + list(b.parameters())])
    assert torch.allclose(g0, ref, atol=1e-6)


def _bucketed_overlap(rank, world):
    """HipDDP's overlapped exchange: a backbone announces a contiguous slice of the flat buffer from inside backward
    (async all-reduce), reduce_gradients() finishes the gaps; the result must equal the one-shot average."""
    from adamml_amd.backbone import FlatBuffers
    from adamml_amd.distributed import HipDDP, shard_batch

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 3)
            self.b = torch.nn.Linear(3, 5)
            self.c = torch.nn.Linear(5, 2)
            self.fb = FlatBuffers(self)
            self.grad_hook = None

        def backbones(self):
            return [self]

        def flat_grad_buffers(self):
            return [self.fb.flat_grad]

        def forward(self, x):
            return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))

    torch.manual_seed(0)
    m = Net()
    m.fb.ensure(torch.device("cpu"))
    m.fb.ensure_grads()
    x = torch.arange(32, dtype=torch.float32).reshape(8, 4) / 10
    (xs,) = shard_batch([x], rank, world)
    ddp = HipDDP(m)
    assert m.grad_hook is not None
    ddp.broadcast_parameters()
    ddp(xs).pow(2).mean().backward()
    m.grad_hook(list(m.b.parameters()))                        # middle slice first (as layer4 would be), then the tail
    m.grad_hook(list(m.c.parameters()))
    m.grad_hook([m.a.weight, m.c.bias])                        # not contiguous: must be left to reduce_gradients()
    assert len(ddp._pending) == 2
    ddp.reduce_gradients()
    assert not ddp._pending
    return m.fb.flat_grad.clone()


def test_bucketed_async_allreduce_matches_full_batch():
    g0, g1 = _run(_bucketed_overlap)
    assert torch.allclose(g0, g1)
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 3), torch.nn.Linear(3, 5), torch.nn.Linear(5, 2)
    x = torch.arange(32, dtype=torch.float32).reshape(8, 4) / 10
    c(torch.tanh(b(torch.tanh(a(x))))).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for m in (a, b, c) for p in m.parameters()])
    assert torch.allclose(g0, ref, atol=1e-6)


def _syncbn_stats(rank, world):
    """What SyncCtx.reduce does with the per-rank [sum, sumsq] vectors, on gloo."""
    torch.manual_seed(1)
    full = torch.randn(8, 6, 5, 5)
    mine = full[rank::world]
    C = 6
    stats = torch.cat([mine.double().sum((0, 2, 3)), (mine.double() ** 2).sum((0, 2, 3))])
    dist.all_reduce(stats)
    count = float(mine.numel() // C * world)
    mean = stats[:C] / count
    var = stats[C:] / count - mean * mean
    return mean, var


def test_syncbn_sum_exchange_equals_global_batch_statistics():
    (m0, v0), (m1, v1) = _run(_syncbn_stats)
    torch.manual_seed(1)
    full = torch.randn(8, 6, 5, 5).double()
    assert torch.allclose(m0, full.mean((0, 2, 3))) and torch.allclose(m1, m0)
    assert torch.allclose(v0, full.var((0, 2, 3), unbiased=False)) and torch.allclose(v1, v0)
