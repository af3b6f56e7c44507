"""The reference's OWN wrapping and iteration, unchanged, on the module build_model returns (SURVEY.md section 8b "who calls
it"): nn.SyncBatchNorm.convert_sync_batchnorm + torch's DistributedDataParallel(find_unused_parameters=True)
(train_adamml.py:126-129), two torch.optim optimizers on model.module.{policy,main}_net.parameters() (:250-257), freeze /
unfreeze through model.module (:344-345), and the iteration of utils/utils.py:359-400 in its order.  Two gloo ranks share one
MI355X (RCCL needs one device per rank); rank r holds videos r::2.

What has to hold: DistributedDataParallel SEES the gradients (they are delivered through autograd once DDP has wrapped the
module), so after the step both ranks hold identical parameters, and the update equals the single-process full-batch step
(same convention as tests/test_syncbn_gpu.py: tight on the classifier heads, direction deep in the net)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from tests.mp_plain import manager, plain, tensors  # noqa: E402

pytestmark = pytest.mark.gpu
S, B = 2, 4
LR, P_LR, WD = 0.05, 0.01, 1e-4


def Args():
    """The argparse namespace the reference hands to build_model (opts.py + the fields train_adamml.py:71-95 adds)."""
    import argparse
    return argparse.Namespace(
        backbone_net="adamml", depth=50, groups=8, num_segments=S, frames_per_group=1, modality=["rgb", "sound"], input_channels=[3, 1],
        num_classes=31, rng_policy=False, rng_threshold=0.5, causality_modeling="lstm", without_t_stride=False, dropout=0.0,
        pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True, dataset="kinetics-sounds",
        dense_sampling=False, lr_scheduler="cosine", sync_bn=True, batch_size=B, prefix="", epochs=1, lr=LR, gpu=0, distributed=True)


def _build():
    from adamml_amd import build_model, synth
    model, arch = build_model(Args())
    assert arch.startswith("kinetics-sounds-rgb-sound-adamml-j_mobilenet_v2-lstm-") and arch.endswith("-f8-cosine-syncbn-bs4-e1"), arch
    sd = synth.synth_state_dict(model.state_dict(), seed=1234)
    for m in range(2):
        # the loop calls model(images) without supplying Gumbel noise: make every decision "use" whatever the draw (+-50 logit
        # bias), so that the 2-rank and the single-process runs gate the main nets identically
        sd["policy_net.fcs.%d.bias" % m] = torch.tensor([-50.0, 50.0])
    model.load_state_dict(sd)
    return model.cuda()


def _batch(rank, world):
    from adamml_amd import synth
    xs = [t[rank::world].cuda() for t in synth.synth_inputs(["rgb", "sound"], B, S, 8, 64, seed=5)]
    return xs, synth.synth_labels(B, 31, seed=5)[rank::world].cuda()


def _reference_iteration(model, images, target, optimizer, p_optimizer, world):
    """utils/utils.py:359-400, line for line in meaning (model.module.* accesses included)."""
    from adamml_amd.train import compute_policy_loss, accuracy
    output, selection = model(images)
    policy_loss = compute_policy_loss("blockdrop", selection, torch.tensor([1.0, 0.05], device="cuda"), torch.tensor(10.0, device="cuda"),
                                      output, target)
    loss = F.cross_entropy(output, target)
    prec1, prec5 = accuracy(output, target)
    if world > 1:
        dist.all_reduce(prec1)
        dist.all_reduce(prec5)
    if model.module.update_policy_net:
        loss = loss + policy_loss
    p_optimizer.zero_grad()
    optimizer.zero_grad()
    loss.backward()
    if model.module.update_policy_net:
        p_optimizer.step()
    if model.module.update_main_net:
        optimizer.step()
    return loss.detach()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                          # same Gumbel draws per rank are not required: the policy is frozen below
        model = _build()
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                                   # train_adamml.py:126-127
        model = nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)    # :129
        p_optimizer = torch.optim.Adam(model.module.policy_net.parameters(), P_LR, weight_decay=WD)       # :250-252
        optimizer = torch.optim.SGD(model.module.main_net.parameters(), LR, momentum=0.9, weight_decay=WD)   # :253-257
        model.module.freeze_policy_net()                                                          # :344-345
        model.module.unfreeze_main_net()
        model.train()
        images, target = _batch(rank, world)
        loss = _reference_iteration(model, images, target, optimizer, p_optimizer, world)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()}
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: float(v.double().sum()) for k, v in sd.items() if v.dtype.is_floating_point})
        if rank == 0:
            ret["sd"], ret["sums"], ret["loss"] = plain(sd), gathered, float(loss)
            ret["sync"] = [n.rt.sync.enabled for n in model.module.backbones()]
            ret["expose"] = [n.expose_param_grads for n in model.module.backbones()]
    finally:
        dist.destroy_process_group()


def test_reference_wrapping_and_iteration_work_unchanged():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = manager()                                         # (tests/mp_plain.py: spawned server, numpy payloads)
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert all(ret["sync"]) and all(ret["expose"])         # SyncBatchNorm containers adopted, gradients delivered through autograd
    # (1) both ranks hold the same parameters after the step: DistributedDataParallel averaged the gradients it was handed
    a, b = ret["sums"]
    diff = [k for k in a if a[k] != b[k]]
    assert not diff, diff[:5]
    # (2) the step equals the single-process step on the concatenated batch (fused flat SGD of this package, same hyper-parameters)
    from adamml_amd.optim import FlatSGD
    one = _build()
    sd0 = {k: v.detach().cpu().clone() for k, v in one.state_dict().items()}
    one.freeze_policy_net()
    one.train()
    images, target = _batch(0, 1)
    out, sel = one(images)
    F.cross_entropy(out, target).backward()
    FlatSGD(one._flat_main, lr=LR, momentum=0.9, weight_decay=WD).step()
    torch.cuda.synchronize()
    sd1 = {k: v.detach().cpu() for k, v in one.state_dict().items()}
    two = tensors(ret["sd"])
    rel = lambda x, y: ((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)).item()
    changed = [k for k in sd0 if k.startswith("main_net.") and k.endswith("weight") and not torch.equal(two[k], sd0[k])]
    assert len(changed) > 100                               # the main nets were updated ...
    assert all(torch.equal(two[k], sd0[k]) for k in sd0 if k.startswith("policy_net.") and k.endswith(("weight", "bias")))   # ... the frozen policy was not
    # bounds = 1.3 x measured (reproducible: order-fixed sums on both sides): head updates 8.3e-3 / 3.3e-4 / 1.8e-2 / 8.3e-3; all main-net
    # weight updates as one vector: rel L2 printed below, cosine 0.793 -- the distance at the bottom of a randomly initialised network is
    # the bf16 amplification of the 1e-7 difference between the two partitions' partial sums, not a gradient error
    head_tol = {"main_net.nets.0.fc.weight": 1.1e-2, "main_net.nets.0.fc.bias": 4.4e-4, "main_net.nets.1.classifier.1.weight": 2.4e-2,
                "main_net.lf_weights": 1.1e-2}
    for k, tol in head_tol.items():
        e = rel(two[k] - sd0[k], sd1[k] - sd0[k])
        print("  update of %-44s DDP(2 ranks, torch.optim.SGD) vs single process (flat SGD): rel L2 %.2e" % (k, e))
        assert e <= tol, (k, e)
    upd2 = torch.cat([(two[k] - sd0[k]).flatten() for k in changed]).double()
    upd1 = torch.cat([(sd1[k] - sd0[k]).flatten() for k in changed]).double()
    cos = F.cosine_similarity(upd2, upd1, dim=0).item()
    e_all = ((upd2 - upd1).norm() / upd1.norm()).item()
    print("  all main-net weight updates: rel L2 %.3f, cosine %.3f; loss %.5f" % (e_all, cos, ret["loss"]))
    assert cos >= 0.75 and e_all <= 0.83, (cos, e_all)            # measured 0.793 / 0.639
    for k in ("main_net.nets.0.bn1.running_mean", "main_net.nets.1.features.0.1.running_var"):
        assert rel(two[k], sd1[k]) <= 1e-4, k              # SyncBatchNorm statistics == full-batch statistics
