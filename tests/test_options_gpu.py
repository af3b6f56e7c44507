"""Options of the reference's factory that the golden cases do not exercise:
rng_policy (models/adamml.py:76-78), without_t_stride (models/resnet.py:205), plain-mean fusion
(learnable_lf_weights=False, joint_resnet_mobilenetv2.py:127), the num_segments override of forward
(models/adamml.py:69-72) and odd batch sizes.  Train-mode main-net stage on reduced inputs (64x64 frames); decisions are
taken from the HIP run and fed to the oracle, logits must match the oracle's bf16-storage emulation within
max(3e-2, 1.5 x |emulation - fp32|) (same criterion as tests/test_models_gpu.py), BatchNorm bookkeeping exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from adamml_amd import adamml, synth  # noqa: E402
from oracle import adamml_oracle as O  # noqa: E402

DEV = "cuda"


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def _run(B, S_model, S_call, rng_policy=False, without_t_stride=False, learnable=True, modality=("rgb", "sound"), noise_seed=11):
    mod = list(modality)
    ch = {"rgb": 3, "sound": 1, "flow": 10, "rgbdiff": 15}
    model = adamml(groups=8, modality=mod, input_channels=[ch[m] for m in mod], num_segments=S_model, rng_policy=rng_policy,
                   rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=without_t_stride,
                   dropout=0.0, pooling_method="max", fusion_point="logits", unimodality_pretrained=[],
                   learnable_lf_weights=learnable)
    sd = synth.synth_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict(sd)
    model.to(DEV)
    model.freeze_policy_net()
    model.train()
    xs = synth.synth_inputs(mod, B, S_call, 8, 64, seed=5)
    expo = synth.synth_gumbel_exponential(S_call, 2, B, seed=noise_seed)
    torch.manual_seed(3)
    logits, sel = model([t.to(DEV) for t in xs], num_segments=S_call if S_call != S_model else None,
                        gumbel_exponential=None if rng_policy else expo.to(DEV))
    assert logits.shape == (B, 31) and sel.shape == (B, S_call, len(mod))
    logits.sum().backward()                                      # backward must run for every option as well
    dec = sel.detach().permute(1, 2, 0).cpu()                     # [S, M, B] as the oracle takes it
    assert set(np.unique(np.round(dec.numpy()))) <= {0.0, 1.0}
    outs = []
    for quant in (O.bf16_straight_through, None):
        O.QUANT = quant
        try:
            with torch.no_grad():
                o, _, _ = O.adamml_forward({k: v.clone() for k, v in sd.items()}, xs, mod, S_call, 8, 50, 5.0, expo, "lstm", "max",
                                           without_t_stride, 0.0, True, learnable, decisions=dec)
        finally:
            O.QUANT = None
        outs.append(o.numpy())
    emu, ref = outs
    e, e_ef = rel_err(logits.detach().cpu().numpy(), emu), rel_err(emu, ref)
    print("  |HIP-emulation| %.4f  |emulation-fp32| %.4f" % (e, e_ef))
    assert e <= max(3e-2, 1.5 * e_ef), (e, e_ef)
    # main-net BatchNorms saw S_call train-mode calls
    nbt = model.state_dict()["main_net.nets.0.bn1.num_batches_tracked"]
    assert int(nbt) == int(sd["main_net.nets.0.bn1.num_batches_tracked"]) + S_call
    return model, sel


def test_rng_policy_decisions_and_logits():
    model, sel = _run(B=2, S_model=3, S_call=3, rng_policy=True)
    assert not hasattr(model.policy_net, "fcs")                 # models/adamml.py:38-40 deletes the heads


def test_without_t_stride():
    _run(B=2, S_model=2, S_call=2, without_t_stride=True)


def test_plain_mean_fusion():
    model, _ = _run(B=2, S_model=3, S_call=3, learnable=False)
    assert model.main_net.lf_weights is None


def test_num_segments_override_and_odd_batch():
    _run(B=3, S_model=5, S_call=2)


def test_single_video():
    for seed in range(11, 40):                                    # a Gumbel draw that selects at least one (segment, modality)
        _, sel = _run(B=1, S_model=2, S_call=2, noise_seed=seed)
        if float(sel.sum()) > 0:
            return
    raise AssertionError("no noise seed selected any modality")
