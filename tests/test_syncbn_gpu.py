"""SURVEY.md section 8(e) parity check on the real backbones: an N-rank data-parallel step with SyncBatchNorm (videos r::N per rank)
must reproduce the single-process step on the concatenated batch -- BatchNorm batch statistics (through the running statistics they
update), loss and the averaged gradient -- "within fp32 reduction-order tolerance".  Every per-channel sum is order-fixed and exact
across workgroups (csrc/common.h: reproducible reductions), so both sides are reproducible numbers and the bounds below are
1.3 x ONE measured value, not a noise band: what remains between the two sides is the fp32 rounding of per-workgroup partial sums over
differently composed tiles (1e-7 relative on a first-layer statistic), amplified by the bf16 storage of ~100 layers on the way down.

  * two gloo ranks sharing one MI355X (runs on every GPU box);
  * min(8, device_count) RCCL ranks, one GPU each, over xGMI -- skipped (not failed) on a one-GPU box, so the first multi-GPU node the
    suite meets exercises SyncBatchNorm + the bucketed gradient all-reduce with real peers (train_adamml.py:122-129)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from tests.mp_plain import manager, plain, tensors  # noqa: E402

pytestmark = pytest.mark.gpu
S, B = 2, 4            # B = videos of the GLOBAL batch of the two-rank test (the RCCL test uses 2 per rank)
# BatchNorm gamma / beta gradients right under the classifier heads (above the noisy part of the backward pass): torch's
# SyncBatchNorm keeps them LOCAL and DDP averages them, so the averaged value equals the single-process gradient
BN_GRAD_KEYS = ("main_net.nets.0.layer4.2.bn3.weight", "main_net.nets.0.layer4.2.bn3.bias",
                "main_net.nets.1.features.18.1.weight", "main_net.nets.1.features.18.1.bias")


def _build():
    from adamml_amd import adamml, synth
    m = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
               causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0, pooling_method="max",
               fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=1234))
    return m.to("cuda")


# gradient stages of the main nets, from the heads down (the 224^2 form asserts one rel-L2 figure per stage: the deeper the stage, the
# more bf16 layers with ReLU / max-pool decisions have amplified the 1e-7 difference of the first statistics)
STAGES = {
    "heads": ("main_net.nets.0.layer4.", "main_net.nets.0.fc.", "main_net.nets.1.features.17.", "main_net.nets.1.features.18.",
              "main_net.nets.1.classifier.", "main_net.lf_weights"),
    "layer3": ("main_net.nets.0.layer3.",),
    "layer12": ("main_net.nets.0.conv1.", "main_net.nets.0.bn1.", "main_net.nets.0.layer1.", "main_net.nets.0.layer2."),
    "mbv2": tuple("main_net.nets.1.features.%d." % i for i in range(17)),
}


def _step(model, ddp, rank, world, nvid=B, px=64, reduce=True):
    from adamml_amd import synth
    dev = torch.device("cuda", torch.cuda.current_device())
    xs = [t[rank::world].to(dev) for t in synth.synth_inputs(["rgb", "sound"], nvid, S, 8, px, seed=5)]
    tgt = synth.synth_labels(nvid, 31, seed=5)[rank::world].to(dev)
    expo = synth.synth_gumbel_exponential(S, 2, nvid, seed=11).view(S, 2, nvid, 2)[:, :, rank::world].reshape(S, -1, 2).to(dev)
    model.freeze_policy_net()
    model.train()
    model.zero_grad()
    out, sel = (ddp or model)(xs, gumbel_exponential=expo)
    # mean over the LOCAL batch then average over ranks == mean over the global batch (equal shards)
    loss = F.cross_entropy(out, tgt)
    loss.backward()
    if ddp is not None:
        if reduce:
            ddp.reduce_gradients()
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        loss = lt / world
    torch.cuda.synchronize()
    sd = model.state_dict()
    named = dict(model.named_parameters())
    stage = {st: torch.cat([named[k].grad.detach().flatten() for k in named if k.startswith(pre) and named[k].grad is not None]).cpu()
             for st, pre in STAGES.items()}
    return {"loss": float(loss.detach()), "sel": sel.detach().cpu(), "grad": model._flat_main.flat_grad.detach().cpu().clone(),
            "stage_grads": stage,
            "fc_grad": model.main_net.nets[0].fc.weight.grad.detach().cpu().clone(),
            "sound_fc_grad": model.main_net.nets[1].classifier[1].weight.grad.detach().cpu().clone(),
            "bn_grads": {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if k in BN_GRAD_KEYS},
            "stats": {k: v.detach().cpu().clone() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}}


def _exchange_linearity(model, ddp, rank, world, nvid, px, averaged):
    """The gradient exchange is LINEAR, so it has a tight statement that no ReLU / max-pool decision can blur (round-5 advisor finding:
    a 0.7-0.9 rel-L2 figure against the full-batch step cannot tell a wrong average from chaotic amplification).  The same step is taken
    again WITHOUT the exchange (same inputs, same weights, SyncBatchNorm still on: every sum is order-fixed, so each rank reproduces its
    local gradients bit for bit), the local flat gradients are summed by ONE plain all-reduce and divided by the world size, and the
    result must equal what the bucketed asynchronous path (slices all-reduced from inside backward + gaps + one 1/world scale) left in
    the parameters' .grad -- per stage, to fp32 summation order.  A missed bucket, a doubly reduced slice or a wrong factor reads >= 0.3."""
    ov, ddp.overlap = ddp.overlap, False
    try:
        _step(model, ddp, rank, world, nvid, px, reduce=False)
    finally:
        ddp.overlap = ov
    named = dict(model.named_parameters())
    out = {}
    for st, pre in STAGES.items():
        loc = torch.cat([named[k].grad.detach().flatten() for k in named if k.startswith(pre) and named[k].grad is not None]).clone()
        dist.all_reduce(loc)
        loc /= world
        out[st] = _rel(averaged["stage_grads"][st], loc.cpu())
    return out


def _worker(rank, world, port, ret, backend, nvid, px=64):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if backend == "nccl" else 0)          # RCCL: one GPU per rank; gloo: the ranks share GPU 0
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from adamml_amd.distributed import HipDDP
        from adamml_amd import hip, interleave
        assert hip.deterministic()
        one = torch.ones(1, device="cuda")
        dist.all_reduce(one)                                         # the communicator really spans `world` ranks
        model = _build()
        ddp = HipDDP(model, sync_bn=True)
        interleave.stats["collectives"] = interleave.stats["coalesced_vectors"] = 0
        r = _step(model, ddp, rank, world, nvid, px)
        exchange = dict(interleave.stats)                            # (the exchange counts of ONE step: the linearity probe below takes another)
        r["lin"] = _exchange_linearity(model, ddp, rank, world, nvid, px, r)
        if rank == 0:
            ret["r"] = plain(r)
            ret["exchange"] = exchange
            ret["communicator_ranks"] = int(one.item())
    finally:
        dist.destroy_process_group()


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


# Bounds = 1.3 x the value measured on MI355X (two gloo ranks, B = 4, S = 2, 64 px; reproducible: see the module docstring).
FIRST_LAYER_STATS = ("main_net.nets.0.bn1.", "main_net.nets.1.features.0.1.", "policy_net.joint_net.nets.0.features.0.1.",
                     "policy_net.joint_net.nets.1.features.0.1.")
TOL = {"first_layer_stats": 1e-6,      # measured 1.8e-7: fp32 rounding of per-workgroup partial sums over differently composed tiles
       "stats_max": 7.1e-3, "stats_p90": 2.8e-3,     # measured 5.46e-3 (policy rgb features.17, ~50 bf16 stages down) / 2.11e-3
       "loss": 1.7e-3,               # measured 1.29e-3 (3.571150 vs 3.566559)
       "fc_grad": 1.9e-2, "sound_fc_grad": 3.0e-2,   # measured 1.46e-2 / 2.29e-2
       "bn_grad": 0.13,                # measured 1.01e-1 (layer4.2.bn3.weight) .. 2.5e-2
       "grad_rel_l2": 0.94}          # measured 0.721 (cosine 0.743) over all 25.8 M main-net gradient elements, see _compare


def _check_linearity(two):
    for st, e in two["lin"].items():
        print("  gradient exchange of stage %-8s: bucketed asynchronous average vs one plain all-reduce of the local gradients: rel L2 %.2e" % (st, e))
        assert e <= 1e-5, (st, e)


def _compare(two, one, world, tol=TOL):
    assert torch.equal(two["sel"], one["sel"][0::world])                 # rank 0 holds videos 0, N, 2N.. and takes the same decisions
    _check_linearity(two)
    errs = {k: _rel(two["stats"][k], v) for k, v in one["stats"].items()}
    first = max(e for k, e in errs.items() if k.startswith(FIRST_LAYER_STATS))
    ranked = sorted(errs.values())
    p90, worst = ranked[int(0.9 * (len(ranked) - 1))], ranked[-1]
    wk = max(errs, key=errs.get)
    print("  %d running statistics, rel L2: first layers (no bf16 stage above them) max %.2e | p90 %.2e | max %.2e (%s)"
          % (len(errs), first, p90, worst, wk))
    g_all = _rel(two["grad"], one["grad"])
    cos = F.cosine_similarity(two["grad"].double(), one["grad"].double(), dim=0).item()
    e_fc, e_sfc = _rel(two["fc_grad"], one["fc_grad"]), _rel(two["sound_fc_grad"], one["sound_fc_grad"])
    print("  loss %.6f vs %.6f | head gradients rel L2: resnet fc %.2e, sound classifier %.2e | all main-net gradients: rel L2 %.3f, cosine %.3f"
          % (two["loss"], one["loss"], e_fc, e_sfc, g_all, cos))
    assert first <= tol["first_layer_stats"], first
    assert worst <= tol["stats_max"] and p90 <= tol["stats_p90"], (worst, p90, wk)
    assert abs(two["loss"] - one["loss"]) <= tol["loss"] * abs(one["loss"])
    # the classifier heads sit above the chaotic part of the backward pass: their averaged gradients must agree closely
    assert e_fc <= tol["fc_grad"] and e_sfc <= tol["sound_fc_grad"], (e_fc, e_sfc)
    for k in BN_GRAD_KEYS:
        e = _rel(two["bn_grads"][k], one["bn_grads"][k])
        print("  %-60s averaged gradient rel L2 %.2e" % (k, e))
        assert e <= tol["bn_grad"], (k, e)     # (a world-times-too-large SyncBN gamma / beta gradient would read 1.0)
    # every main-net gradient, as ONE vector: the deep layers of a randomly initialised train-mode-BatchNorm network amplify the 1e-7
    # difference of the first statistics through ReLU / max-pool decision flips (tools/conditioning_study.py: sqrt law)
    assert g_all <= tol["grad_rel_l2"], g_all


def _spawn(world, backend, nvid, px=64):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = manager()                                         # (tests/mp_plain.py: spawned server, numpy payloads)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, backend, nvid, px), nprocs=world, join=True)
    return tensors(ret["r"]), ret["exchange"], ret["communicator_ranks"]


def _loose(tol, k):
    return {n: v * k for n, v in tol.items()}


# Bounds per world size (2 videos per rank beyond world 2): 1.3 x the values measured with gloo ranks sharing one MI355X -- the arithmetic
# of an N-rank step does not depend on the transport (the statistic exchange sums fp64 vectors, the gradient all-reduce fp32 buckets), so
# these are also the RCCL test's bounds; a world size that was never measured gets 3 x the two-rank bounds.
TOL_BY_WORLD = {
    2: TOL,
    # 4 ranks, 8 videos: statistics 1.85e-7 / max 5.27e-3 / p90 1.41e-3, loss 9.0e-4, heads 1.14e-2 / 2.11e-2, bn3 gamma 9.1e-2, all gradients 0.652
    4: {"first_layer_stats": 1e-6, "stats_max": 6.9e-3, "stats_p90": 1.9e-3, "loss": 1.2e-3, "fc_grad": 1.5e-2, "sound_fc_grad": 2.8e-2,
        "bn_grad": 0.12, "grad_rel_l2": 0.85},
    # 8 ranks, 16 videos (ADAMML_TEST_GLOO_WORLDS=8): 1.45e-7 / 3.71e-3 / 9.8e-4, loss 4.6e-4, heads 1.17e-2 / 2.41e-2, bn3 gamma 8.5e-2, 0.647
    8: {"first_layer_stats": 1e-6, "stats_max": 4.9e-3, "stats_p90": 1.3e-3, "loss": 6.1e-4, "fc_grad": 1.6e-2, "sound_fc_grad": 3.2e-2,
        "bn_grad": 0.112, "grad_rel_l2": 0.85},
}


def _compare_224(two, one, world, key):
    """The well-conditioned form (224^2: >= 196 samples per BatchNorm channel even in layer 4): per-stage gradient figures from the
    heads down, each against its own table entry (tests/parity_bounds.json, `key`.*), instead of one number over 25.8 M elements.
    Asserted for every transport and world size: the LINEAR statement (_exchange_linearity, <= 1e-5 per stage) and a cosine >= 0.5 per
    stage against the full-batch step; the rel-L2 figures against the full-batch step have table entries for the ResNet-50 stages of a
    world size that was measured (key present) -- the random-weight Sound-MobileNetV2 trunk's figure (0.73 at two ranks: ~1.09x
    amplification per layer over 52 layers) is PRINTED only: a bound of 0.95 on a quantity that reads 1.0 for a zero gradient was no gate."""
    from tests.parity_bounds import check, table
    assert torch.equal(two["sel"], one["sel"][0::world])
    _check_linearity(two)
    have = (key + ".stats") in table() or os.environ.get("ADAMML_REBASE")
    base = key if have else "nrank2_224"         # (statistics / loss / heads: the two-rank entries hold for any world -- fp64 / fp32 sums, no transport factor)
    errs = {k: _rel(two["stats"][k], v) for k, v in one["stats"].items()}
    ranked = sorted(errs.values())
    check(base + ".first_stats", max(e for k, e in errs.items() if k.startswith(FIRST_LAYER_STATS)), "first-layer running statistics", cat="nrank_first_stats")
    check(base + ".stats", ranked[-1], max(errs, key=errs.get), cat="nrank_stats")
    check(base + ".stats_p90", ranked[int(0.9 * (len(ranked) - 1))], cat="nrank_stats")
    check(base + ".loss", abs(two["loss"] - one["loss"]) / abs(one["loss"]), "%.6f vs %.6f" % (two["loss"], one["loss"]), cat="nrank_loss")
    check(base + ".head", max(_rel(two["fc_grad"], one["fc_grad"]), _rel(two["sound_fc_grad"], one["sound_fc_grad"])),
          "classifier-head gradients (resnet fc, sound classifier)", cat="nrank_head")
    for k in BN_GRAD_KEYS:                               # (a world-times-too-large SyncBN gamma / beta gradient would read 1.0)
        e = _rel(two["bn_grads"][k], one["bn_grads"][k])
        print("  %-60s averaged gradient rel L2 %.2e" % (k, e))
        assert e <= 0.2, (k, e)
    for st in STAGES:
        a, b = two["stage_grads"][st], one["stage_grads"][st]
        e = _rel(a, b)
        cos = F.cosine_similarity(a.double(), b.double(), dim=0).item()
        assert cos >= 0.5, (st, cos)                     # (a zero or sign-flipped averaged gradient reads rel L2 1.0 / cosine <= 0)
        if have and st != "mbv2":
            check("%s.grad_%s" % (key, st), e, "%d elements, cosine %.3f" % (a.numel(), cos), cat="nrank_grad")
        else:                                            # unmeasured world size / the chaotic trunk: printed
            print("  gradients of stage %-8s vs the full-batch step: rel L2 %.3f cosine %.3f over %d elements (printed)" % (st, e, cos, a.numel()))


def _n_rank(world, backend, nvid, px=64):
    got, ex, ranks = _spawn(world, backend, nvid, px)
    assert ranks == world, "the communicator spans %d ranks, expected %d" % (ranks, world)
    # SyncBatchNorm exchanges of one step (S = 2 segments batched as groups): the 4 backbones have 53 / 52 / 52 / 52 BatchNorm
    # layers in forward and the two trainable ones 53 / 52 in backward = 314 statistic vectors; issued in rounds they travel in one
    # collective per BatchNorm depth and exchange group (the ResNet alone, the MobileNetV2s together, alternating: interleave.GROUPS):
    # <= (53 + 52) forward + (53 + 52) backward; <= 53 + 53 with ADAMML_SYNC_GROUPS=1
    print("  %d %s ranks: SyncBatchNorm exchange: %d statistic vectors in %d collectives" % (world, backend, ex["coalesced_vectors"], ex["collectives"]))
    assert ex["coalesced_vectors"] == 53 + 3 * 52 + 53 + 52 and ex["collectives"] <= 2 * (53 + 52)
    torch.cuda.set_device(0)
    one = _step(_build(), None, 0, 1, nvid, px)
    if px == 224:
        # keyed by the ACTUAL world size; a world size without table entries uses the two-rank entries for statistics, loss and head
        # gradients (the exchange sums fp64 vectors and fp32 buckets: no transport-dependent factor) and prints the per-stage figures
        # against the full-batch step -- the gradient exchange itself is asserted tightly for every transport and world (_exchange_linearity)
        _compare_224(got, one, world, "nrank%d_224" % world)
        return
    _compare(got, one, world, TOL_BY_WORLD.get(world) or _loose(TOL, 3.0))


def test_two_rank_syncbn_step_equals_single_process_full_batch():
    _n_rank(2, "gloo", B)


@pytest.mark.parametrize("world", [int(w) for w in os.environ.get("ADAMML_TEST_GLOO_WORLDS", "4").split(",")])
def test_n_rank_gloo_syncbn_step_equals_full_batch(world):
    """The N-rank form of the test above (2 videos per rank; ranks share the one GPU): the harness and the bounds the RCCL test below
    will use on a multi-GPU box, exercised wherever the suite runs.  ADAMML_TEST_GLOO_WORLDS=8 runs the 8-rank form (calibration)."""
    _n_rank(world, "gloo", 2 * world)


def test_two_rank_syncbn_step_at_224_per_stage_gradients():
    """The well-conditioned statement (round-4 review): B = 4 videos, S = 2 segments at 224^2 / 256^2 on two gloo ranks against the
    single-process full batch -- statistics, loss, head gradients and one gradient figure PER STAGE (heads + layer 4, layer 3, stem +
    layers 1-2, the Sound-MobileNetV2 trunk), each with its own entry in tests/parity_bounds.json.  The 64-pixel forms above stay as
    plumbing tests (4-16 samples per BatchNorm channel in the deep layers: ill-conditioned by construction)."""
    _n_rank(2, "gloo", B, px=224)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: RCCL ranks over xGMI, one GPU each")
def test_n_rank_rccl_step_equals_full_batch():
    """configs[2] with real peers: min(8, device_count) RCCL ranks, 2 videos each at 224^2, SyncBatchNorm + bucketed asynchronous
    gradient all-reduce, against the one-process step on the concatenated batch.  Asserted: `communicator_ranks`, the exchange counts,
    running statistics, loss and head gradients (the two-rank 224^2 table entries, no transport factor), the gradient exchange per
    stage to fp32 summation order (_exchange_linearity) and a cosine >= 0.5 per stage against the full-batch step."""
    world = min(8, torch.cuda.device_count())
    _n_rank(world, "nccl", 2 * world, px=224)
