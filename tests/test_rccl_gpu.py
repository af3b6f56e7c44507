"""configs[2] as far as ONE MI355X allows: the data-parallel step (train_adamml.py:83,122-129 -- SyncBatchNorm + gradient
all-reduce) on a ONE-rank RCCL communicator (torch.distributed backend "nccl" == RCCL on ROCm).

With a single rank every collective is an identity, so the forced-distributed step must reproduce the plain step BIT FOR BIT
-- but it runs everything the N > 1 path runs against the real, stream-asynchronous backend: the backbones
issued in lock-step rounds on their own HIP streams with ONE coalesced statistics all-reduce per BatchNorm depth on a
communication stream (interleave.py: 53 + 53 rounds, one collective per round and exchange group), the backward tapes deferred to the end-of-backward callback, the
bucketed ASYNCHRONOUS gradient all-reduce started from inside backward (distributed.py), and the waits between RCCL's internal
stream and ours.  gloo (tests/test_syncbn_gpu.py) blocks the host in every collective and hides ordering bugs; RCCL does not.
A missing stream wait shows up here as a bit difference (or garbage) in a gradient or a running statistic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES  # noqa: E402
from tests.oracle_harness import manifest  # noqa: E402
from tests.test_parity_fullsize_gpu import build, hip_train_step, _snapshot  # noqa: E402


@pytest.fixture()
def rccl_one_rank():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["train_main", "train_policy"])
def test_one_rank_rccl_step_equals_the_plain_step_bit_for_bit(rccl_one_rank, mode):
    from adamml_amd import hip, interleave
    from adamml_amd.distributed import HipDDP
    c = CASES["adamml_c2"]                                   # B = 4 videos, 5 segments, 224^2 / 256^2: the configs[1] workload
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    probe = torch.ones(1, device="cuda")
    dist.all_reduce(probe)                                   # the communicator is RCCL and alive
    assert dist.get_backend() == "nccl" and float(probe.item()) == 1.0
    plain = build(c)
    logits, _, _, grads, state, _ = hip_train_step(plain, c, mode, sd)
    a = _snapshot(logits, grads, state)
    del plain
    model = build(c)
    ddp = HipDDP(model, sync_bn=True, force_collectives=True)
    assert ddp.active and all(n.rt.sync.enabled for n in model.backbones())
    for rep in range(2):                                 # twice: the second step reuses arenas / streams / pending lists
        before = dict(interleave.stats)
        l2, _, _, g2, s2, _ = hip_train_step(model, c, mode, sd, after_backward=ddp.reduce_gradients)
        rounds = interleave.stats["collectives"] - before["collectives"]
        vectors = interleave.stats["coalesced_vectors"] - before["coalesced_vectors"]
        b = _snapshot(l2, g2, s2)
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        print("adamml_c2 [%s] rep %d: one-rank RCCL step vs plain step: %d tensors compared, %d differ; %d statistic rounds carrying "
              "%d vectors; %d bucket + %d gap gradient all-reduces" % (mode, rep, len(a), len(diff), rounds, vectors,
                                                                       ddp.stats["bucket_all_reduces"], ddp.stats["gap_all_reduces"]))
        assert set(a) == set(b)
        assert not diff, diff[:8]
        # forward: one round per BatchNorm depth of the deepest backbone (53 for ResNet-50 / MobileNetV2); backward: the same for
        # the sub-networks that train in this stage
        assert vectors > rounds >= 53, (rounds, vectors)
    assert ddp.stats["bucket_all_reduces"] >= 2          # the asynchronous buckets really went out from inside backward
