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


def _gather(rank, world):
    from adamml_amd.train import concat_all_gather
    out = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank          # per-rank `output [B, classes]`
    tgt = torch.tensor([rank, rank + 10, rank + 20])                                 # `target [B]` (int64)
    sel = torch.full((3, 2, 2), float(rank))                                         # `selection [B, S, M]`
    return concat_all_gather(out), concat_all_gather(tgt), concat_all_gather(sel)


def test_validation_all_gather_concatenates_in_rank_order():
    """utils/utils.py:484-490,539-550 (`-e` / per-epoch validation): outputs, targets and selections of all ranks are gathered
    and concatenated along the batch axis in rank order before the global top-k is taken."""
    r0, r1 = _run(_gather)
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                                                     # every rank holds the same global tensors
    out, tgt, sel = r0
    assert out.shape == (6, 2) and torch.equal(out[3:], out[:3] + 100)
    assert tgt.tolist() == [0, 10, 20, 1, 11, 21] and tgt.dtype == torch.int64
    assert sel.shape == (6, 2, 2) and float(sel[:3].sum()) == 0.0 and float(sel[3:].mean()) == 1.0


def _coalesced(rank, world):
    """Three "backbones" with 3 / 1 / 2 statistic exchanges each, issued in lock-step rounds (adamml_amd.interleave)."""
    from adamml_amd import interleave
    interleave.stats["collectives"] = interleave.stats["coalesced_vectors"] = 0
    log = []

    def job(name, n, width):
        def f():
            got = []
            for i in range(n):
                t = torch.full((width,), float((rank + 1) * (i + 1)), dtype=torch.float64) + ord(name)
                log.append((name, i))
                got.append(interleave.exchange(t).clone())
            return got
        return f
    res = interleave.run_interleaved([(job("a", 3, 4), None), (job("b", 1, 2), None), (job("c", 2, 6), None)], None, groups=1)
    lone = interleave.exchange(torch.tensor([float(rank)], dtype=torch.float64))         # outside run_interleaved: a plain all-reduce
    st = dict(interleave.stats)
    # the same jobs with ALTERNATING exchange groups ({first job} / {the others}: the default, interleave.GROUPS)
    interleave.stats["collectives"] = interleave.stats["coalesced_vectors"] = 0
    del log[:]
    res2 = interleave.run_interleaved([(job("a", 3, 4), None), (job("b", 1, 2), None), (job("c", 2, 6), None)], None, groups=2)
    return res, list(log), st, lone, res2, dict(interleave.stats)


def test_statistic_exchanges_of_a_round_are_one_collective():
    """SyncBatchNorm exchange (train_adamml.py:126-127): the vectors the backbones have pending at the same depth travel in ONE
    all-reduce; every job still receives exactly the sum over the ranks of ITS vector; the issue order is fixed."""
    (r0, log0, st0, lone0, r0b, st0b), (r1, log1, st1, lone1, r1b, st1b) = _run(_coalesced)
    assert log0 == log1 == [("a", 0), ("b", 0), ("c", 0), ("a", 1), ("c", 1), ("a", 2)]
    # alternating groups: the same values for every job, on both ranks; the first job's 3 exchanges travel alone, the others' in 2 rounds
    for a, b, c in zip(r0, r0b, r1b):
        assert all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(a, b, c))
    assert st0b["collectives"] == 3 + 2 and st0b["coalesced_vectors"] == 6 and st0b == st1b
    for name, n, width, got in (("a", 3, 4, r0[0]), ("b", 1, 2, r0[1]), ("c", 2, 6, r0[2])):
        for i in range(n):
            want = sum(float((rk + 1) * (i + 1)) + ord(name) for rk in range(2))
            assert got[i].shape == (width,) and torch.all(got[i] == want), (name, i, got[i], want)
    for a, b in zip(r0, r1):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert st0["collectives"] == 3 + 1 and st0["coalesced_vectors"] == 6 + 1        # 3 rounds for 6 vectors (+ the lone exchange)
    assert float(lone0) == float(lone1) == 1.0
