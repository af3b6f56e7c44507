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


def test_uint8_frame_input_path_matches_reference_pipeline():
    """Decoded uint8 frames -> (ToTorchFormatTensor(div) + GroupNormalize + data_layer) in one launch
    (adamml_clip_u8_to_nhwc) against the reference arithmetic: x = (u8 / 255 - mean[c]) / std[c] (utils/video_transforms.py:
    62-84,321-343), then models/adamml.py:42-67.  bf16 output: exact for the full-resolution main-net input, <= 1 bf16 ulp for
    the bilinear 160x160 policy input (fp32 rounding order of the four-tap blend)."""
    from adamml_amd.runtime import clip_u8_to_nhwc, clip_to_nhwc
    from adamml_amd.common import MeanStdMixin
    ms = MeanStdMixin()
    torch.manual_seed(1)
    B, S, Fr, H, W = 2, 3, 8, 64, 48
    for m, C in (("rgb", 3), ("rgbdiff", 15), ("flow", 10)):
        u8 = torch.randint(0, 256, (B, H, W, S * Fr * C), dtype=torch.uint8)
        mean, std = ms.mean(m), ms.std(m)
        x = u8.permute(0, 3, 1, 2).float().div(255)                          # ToTorchFormatTensor, per video
        rep = (S * Fr * C) // len(mean)
        mt = torch.tensor(mean * rep).view(1, -1, 1, 1)
        st = torch.tensor(std * rep).view(1, -1, 1, 1)
        x = (x - mt) / st                                                    # GroupNormalize
        for out_hw, step in ((None, 1), ((40, 40), 2)):
            got = clip_u8_to_nhwc(u8.to(DEV), S, Fr, C, mean, std, out_hw=out_hw, frame_step=step)
            ref = clip_to_nhwc(x.to(DEV), S, Fr, C, out_hw=out_hw, frame_step=step)      # validated against the oracle's data_layer
            assert got.shape == ref.shape
            if out_hw is None:
                assert torch.equal(got, ref), m
            else:
                d = (got.float() - ref.float()).abs()
                assert (d <= ref.float().abs() * 2 ** -7 + 1e-6).all(), (m, d.max().item())
                assert (got != ref).float().mean().item() < 0.02
            if C <= 4:                                                       # 4-channel pixels for the stem kernels
                got4 = clip_u8_to_nhwc(u8.to(DEV), S, Fr, C, mean, std, out_hw=out_hw, frame_step=step, cpad=4)
                assert got4.shape[-1] == 4 and torch.equal(got4[..., :C], got[..., :C]) and got4[..., C:].float().abs().max().item() == 0


def test_adamml_accepts_uint8_frames():
    """End to end: the model fed with uint8 frames returns the logits of the same model fed with the normalised fp32 tensor."""
    torch.manual_seed(2)
    B, S = 2, 2
    model = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
                   causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0, pooling_method="max",
                   fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=1234))
    model.to(DEV).eval()
    u8 = torch.randint(0, 256, (B, 64, 64, S * 8 * 3), dtype=torch.uint8)
    snd = torch.randn(B, S, 64, 64) * 3 - 5
    mean, std = model.mean("rgb"), model.std("rgb")
    x = u8.permute(0, 3, 1, 2).float().div(255)
    x = (x - torch.tensor(mean * (S * 8)).view(1, -1, 1, 1)) / torch.tensor(std * (S * 8)).view(1, -1, 1, 1)
    expo = synth.synth_gumbel_exponential(S, 2, B, seed=11).to(DEV)
    with torch.no_grad():
        a, da = model([u8.to(DEV), snd.to(DEV)], gumbel_exponential=expo)
        b, db = model([x.to(DEV), snd.to(DEV)], gumbel_exponential=expo)
    assert torch.equal(da, db)
    assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()) + 1e-6)


def test_rgbdiff_from_rgb_frames_on_the_gpu():
    """RGB-diff computed inside the input kernel from decoded RGB frames (adamml_clip_u8_rgbdiff_to_nhwc) == the reference's
    loader arithmetic (utils/video_dataset.py:32-38,75-84: for frame k of a group, uint8((img[k+1] - img[k] + 255) * 255 / 510),
    5 differences of 6 consecutive frames) followed by the uint8 pipeline already pinned above.  Bit-exact at full resolution
    (the difference images are integers); <= 1 bf16 ulp after the bilinear policy resize."""
    import numpy as np
    from adamml_amd.runtime import clip_u8_rgbdiff_to_nhwc, clip_u8_to_nhwc
    from adamml_amd.common import MeanStdMixin
    ms = MeanStdMixin()
    mean, std = ms.mean("rgbdiff"), ms.std("rgbdiff")
    torch.manual_seed(4)
    B, S, Fr, H, W, D = 2, 2, 8, 32, 24, 5
    rgb = torch.randint(0, 256, (B, H, W, S * Fr, D + 1, 3), dtype=torch.uint8)
    a = rgb.numpy().astype(np.float64)
    diff = a[:, :, :, :, 1:] - a[:, :, :, :, :-1]                              # compute_img_diff(tmp[k+1], tmp[k])
    diff += 255.0
    diff *= 255.0 / float(2 * 255.0)
    diff_u8 = torch.from_numpy(diff.astype(np.uint8)).reshape(B, H, W, S * Fr * D * 3)
    for out_hw, step in ((None, 1), ((20, 20), 2)):
        got = clip_u8_rgbdiff_to_nhwc(rgb.reshape(B, H, W, -1).to(DEV), S, Fr, mean, std, out_hw=out_hw, frame_step=step)
        ref = clip_u8_to_nhwc(diff_u8.to(DEV), S, Fr, 3 * D, mean, std, out_hw=out_hw, frame_step=step)
        assert got.shape == ref.shape == (S, B * (Fr // step), *(out_hw or (H, W)), 16)
        if out_hw is None:
            assert torch.equal(got, ref)
        else:       # the four-tap blend is contracted into FMAs differently in the two template instances: <= 1 bf16 ulp, rarely
            d = (got.float() - ref.float()).abs()
            assert (d <= ref.float().abs() * 2 ** -7 + 1e-6).all() and (got != ref).float().mean().item() < 0.02
    # the model takes the raw frames for its rgbdiff modality (6 RGB frames per group -> 15 difference channels)
    model = adamml(groups=8, modality=["rgb", "flow", "rgbdiff"], input_channels=[3, 10, 15], num_segments=S, rng_policy=False,
                   rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0,
                   pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=1234))
    model.to(DEV).eval()
    H = W = 64
    rgb_u8 = torch.randint(0, 256, (B, H, W, S * Fr * 3), dtype=torch.uint8)
    flow_u8 = torch.randint(0, 256, (B, H, W, S * Fr * 10), dtype=torch.uint8)
    raw = torch.randint(0, 256, (B, H, W, S * Fr, D + 1, 3), dtype=torch.uint8)
    d = raw.numpy().astype(np.float64)
    d_u8 = torch.from_numpy(((d[:, :, :, :, 1:] - d[:, :, :, :, :-1] + 255.0) * 0.5).astype(np.uint8)).reshape(B, H, W, -1)
    expo = synth.synth_gumbel_exponential(S, 2, B, seed=3).to(DEV)
    with torch.no_grad():
        a1, s1 = model([rgb_u8.to(DEV), flow_u8.to(DEV), raw.reshape(B, H, W, -1).to(DEV)], gumbel_exponential=expo)
        a2, s2 = model([rgb_u8.to(DEV), flow_u8.to(DEV), d_u8.to(DEV)], gumbel_exponential=expo)
    assert torch.equal(s1, s2) and torch.allclose(a1, a2, rtol=2e-2, atol=2e-2 * float(a2.abs().max()))
