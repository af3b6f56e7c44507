"""Launch plans (adamml_amd/plan.py, csrc/plan.hip): the static launch sequence of a backbone call recorded once and replayed by one C
call per segment must compute EXACTLY what the Python-issued sequence computes.  Deterministic mode, the same model stepped six times
eagerly and six times with plans enabled (calls 1-2 eager warm-up, call 3 recorded, calls 4-6 replayed): losses, logits, every
parameter, every BatchNorm buffer bit-identical after every step -- in the plain configuration and in the data-parallel one (one-rank
RCCL communicator with SyncBatchNorm and the bucketed gradient all-reduce forced on: the plan's segments are then cut at every
statistics exchange and at the gradient-bucket hooks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES, CH  # noqa: E402
from tests.oracle_harness import manifest, case_inputs  # noqa: E402

DEV = "cuda"


def _build(c, dropout):
    from adamml_amd import adamml
    mod = c["modality"]
    return adamml(groups=c["groups"], modality=mod, input_channels=[CH[m] for m in mod], num_segments=c["S"], rng_policy=False,
                  rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=dropout,
                  pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)


def _steps(c, n, planned, forced, stage, dropout=0.0):
    from adamml_amd import plan, hip
    from adamml_amd.distributed import HipDDP
    from adamml_amd.optim import FlatSGD, FlatAdam
    plan.ENABLED = planned
    try:
        model = _build(c, dropout)
        model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
        model.to(DEV)
        ddp = HipDDP(model, sync_bn=forced, force_collectives=forced)
        (model.freeze_policy_net if stage == "main" else model.freeze_main_net)()
        model.train()
        xs, target = case_inputs(c)
        xs, target = [t.to(DEV) for t in xs], target.to(DEV)
        expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=11).to(DEV)
        opt = None
        trace = []
        hold = []
        for it in range(n):
            # a real loader hands over a NEW tensor every batch: the plan's pointer slots (input, incoming gradient) must follow the moving
            # addresses (an allocation of changing size in between keeps the caching allocator from returning the same block)
            hold.append(torch.empty(4096 * (it + 1) + 17, device=DEV))
            xs_it = [t.clone() for t in xs]
            logits, sel = ddp(xs_it, gumbel_exponential=expo)
            loss = F.cross_entropy(logits, target)
            if stage == "policy":
                loss = loss + (sel.mean(dim=1) ** 2).mean()
            loss.backward()
            ddp.reduce_gradients()
            if opt is None:
                opt = FlatSGD(model._flat_main, lr=0.01, momentum=0.9, weight_decay=1e-4) if stage == "main" else \
                    FlatAdam(model._flat_policy, lr=1e-3, weight_decay=1e-4)
            opt.step()
            opt.zero_grad()
            torch.cuda.synchronize()
            trace.append((loss.detach().cpu().clone(), logits.detach().cpu().clone(),
                          {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
        return trace, dict(plan.stats)
    finally:
        plan.ENABLED = False


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


def _compare(eager, planned, what):
    for it, ((l0, y0, s0), (l1, y1, s1)) in enumerate(zip(eager, planned)):
        assert torch.equal(l0, l1), "%s: loss differs at step %d (%r vs %r)" % (what, it, float(l0), float(l1))
        assert torch.equal(y0, y1), "%s: logits differ at step %d" % (what, it)
        diff = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not diff, "%s: step %d: %d state tensors differ, e.g. %s" % (what, it, len(diff), diff[:4])


@pytest.mark.parametrize("stage", ["main", "policy"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(stage):
    from adamml_amd import hip, plan
    c = CASES["adamml_rgb_sound"]                    # B = 2, S = 3, 96^2: every kernel family of the headline workload, in seconds
    eager, _ = _steps(c, 6, False, False, stage)
    before = dict(plan.stats)
    planned, st = _steps(c, 6, True, False, stage)
    rec, ops = st["recorded"] - before["recorded"], st["replayed_ops"] - before["replayed_ops"]
    print("launch plan [%s stage]: %d plans recorded, %d launches replayed from C over 3 steps" % (stage, rec, ops))
    assert rec >= (2 if stage == "main" else 2) and ops > 500
    _compare(eager, planned, "plain")


def test_replayed_data_parallel_steps_equal_eager_steps(rccl_one_rank):
    from adamml_amd import hip, plan
    c = CASES["adamml_rgb_sound"]
    eager, _ = _steps(c, 6, False, True, "main")
    before = dict(plan.stats)
    planned, st = _steps(c, 6, True, True, "main")
    rec, seg = st["recorded"] - before["recorded"], st["replayed_segments"] - before["replayed_segments"]
    print("launch plan [one-rank RCCL, SyncBN + bucketed all-reduce]: %d plans recorded, %d segments replayed over 3 steps" % (rec, seg))
    assert rec >= 4 and seg > 300                # 4 backbones; a segment per statistics exchange / gradient-bucket hook
    _compare(eager, planned, "data-parallel")


def test_plans_with_dropout_redraw_the_mask_every_replay():
    """Dropout(0.5) in the heads: the keep mask lives in a fixed buffer of the plan and is re-drawn before every replay -- two replayed
    steps on identical inputs and weights (lr = 0) must differ in their logits, and training must stay finite."""
    from adamml_amd import plan
    from adamml_amd.optim import FlatSGD
    c = CASES["adamml_rgb_sound"]
    plan.ENABLED = True
    try:
        model = _build(c, 0.5)
        model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
        model.to(DEV)
        model.freeze_policy_net()
        model.train()
        xs, target = case_inputs(c)
        xs, target = [t.to(DEV) for t in xs], target.to(DEV)
        expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=11).to(DEV)
        outs = []
        for it in range(6):
            logits, _ = model(xs, gumbel_exponential=expo)
            F.cross_entropy(logits, target).backward()
            model._flat_main.flat_grad.zero_()
            outs.append(logits.detach().clone())
        assert all(torch.isfinite(o).all() for o in outs)
        assert not torch.equal(outs[4], outs[5]) and not torch.equal(outs[3], outs[4])
    finally:
        plan.ENABLED = False


def test_inference_plans_equal_eager_inference():
    """Policy-gated inference (models/adamml.py:81-86 with decision-driven skipping) at a serving-sized batch: with launch plans the
    eval-mode calls of the policy backbones and of the main nets (selected-clip count padded to a multiple of 8) are replayed from C; logits
    and decisions must equal the eager path's exactly, also after the weights changed (the plan recomputes its BatchNorm affines)."""
    from adamml_amd import plan
    c = CASES["adamml_rgb_sound"]
    xs, _ = case_inputs(c)
    xs = [t.to(DEV) for t in xs]
    expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=11).to(DEV)
    model = _build(c, 0.0)
    model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
    model.to(DEV).eval()

    def infer():
        with torch.no_grad():
            y, sel = model(xs, gumbel_exponential=expo)
        return y.clone(), sel.clone(), dict(model.last_skip_stats)

    ref, sel_ref, st_ref = infer()
    assert 0 < sum(st_ref["executed_per_modality"]) < 2 * st_ref["clips"]          # some clips skipped, some run: the compaction path
    before = dict(plan.stats)
    plan.ENABLED = plan.EVAL_ENABLED = True
    try:
        outs = [infer() for _ in range(5)]
        assert plan.stats["recorded"] - before["recorded"] >= 3 and plan.stats["replayed_ops"] - before["replayed_ops"] > 300
        for y, sel, st in outs:
            assert torch.equal(sel, sel_ref) and torch.equal(y, ref) and st == st_ref
        # new weights / statistics: eager reference from a fresh model, planned result from the replaying one
        sd2 = synth.synth_state_dict(manifest(c), seed=77)
        model.load_state_dict(sd2)
        for n in model.backbones():
            n.mark_weights_dirty()
        got, sel_got, _ = infer()
    finally:
        plan.ENABLED = plan.EVAL_ENABLED = False
    fresh = _build(c, 0.0)
    fresh.load_state_dict(sd2)
    fresh.to(DEV).eval()
    with torch.no_grad():
        want, sel_want = fresh(xs, gumbel_exponential=expo)
    assert torch.equal(sel_got, sel_want) and torch.equal(got, want)


def test_second_forward_before_the_first_backward_runs_eagerly():
    """A plan owns ONE set of activation buffers.  Two forwards of the same key whose backward passes run afterwards (two model calls
    summed into one loss): the second forward must not replay into the buffers the first backward still needs -- it runs eagerly
    (plan.Plan.busy) and the result equals the all-eager computation bit for bit."""
    from adamml_amd import plan
    c = CASES["adamml_rgb_sound"]

    def run(planned):
        plan.ENABLED = planned
        try:
            model = _build(c, 0.0)
            model.load_state_dict(synth.synth_state_dict(manifest(c), seed=1234))
            model.to(DEV)
            model.freeze_policy_net()
            model.train()
            xs, target = case_inputs(c)
            xs, target = [t.to(DEV) for t in xs], target.to(DEV)
            xs2 = [t * 0.5 for t in xs]
            expo = synth.synth_gumbel_exponential(c["S"], 2, c["B"], seed=11).to(DEV)
            outs = []
            for it in range(5):                      # two eager + one recorded + two replayed iterations
                model.zero_grad(set_to_none=False)
                la, _ = model(xs, gumbel_exponential=expo)
                lb, _ = model(xs2, gumbel_exponential=expo)
                (F.cross_entropy(la, target) + F.cross_entropy(lb, target)).backward()
                torch.cuda.synchronize()
                outs.append((la.detach().cpu().clone(), lb.detach().cpu().clone(),
                             {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}))
            return outs
        finally:
            plan.ENABLED = False
    before = plan.stats.get("busy_fallbacks", 0)
    eager = run(False)
    planned = run(True)
    assert plan.stats.get("busy_fallbacks", 0) > before, "the second forward never met a busy plan"
    for it, ((a0, b0, g0), (a1, b1, g1)) in enumerate(zip(eager, planned)):
        assert torch.equal(a0, a1) and torch.equal(b0, b1), "logits differ at iteration %d" % it
        diff = [k for k in g0 if not torch.equal(g0[k], g1[k])]
        assert not diff, "iteration %d: %d gradients differ, e.g. %s" % (it, len(diff), diff[:4])
