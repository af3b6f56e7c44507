"""Golden-fixture case table shared by tools/gen_golden.py (reference side) and the
oracle / HIP parity tests.  Data only: sizes and flags of each case."""
import zlib
import numpy as np

CH = {"rgb": 3, "flow": 10, "rgbdiff": 15, "sound": 1}

CASES = {
    # unimodal RGB ResNet-50 (config 1 shape at full resolution, eval) and a reduced train step
    "resnet50_full": dict(kind="resnet", modality=["rgb"], groups=8, B=1, size=224, modes=["eval"]),
    "resnet50_train": dict(kind="resnet", modality=["rgb"], groups=8, B=2, size=96, modes=["eval", "train"]),
    "resnet50_avg": dict(kind="resnet", modality=["rgb"], groups=16, B=1, size=64, pooling="avg", modes=["train"]),
    "resnet50_flow": dict(kind="resnet", modality=["flow"], groups=8, B=1, size=64, modes=["train"]),
    "sound_mbv2": dict(kind="sound", modality=["sound"], B=2, sound_size=128, modes=["eval", "train"]),
    # AdaMML RGB+Audio (config 2 shape, reduced B/S/resolution)
    "adamml_rgb_sound": dict(kind="adamml", modality=["rgb", "sound"], groups=8, B=2, S=3, size=96, sound_size=96,
                             modes=["eval", "train_main", "train_policy"]),
    "adamml_rgb_sound_nolstm": dict(kind="adamml", modality=["rgb", "sound"], groups=8, B=2, S=2, size=64,
                                    sound_size=64, causality=None, modes=["eval", "train_policy"]),
    # config 4 / 5 modality sets
    "adamml_rgb_flow_rgbdiff": dict(kind="adamml", modality=["rgb", "flow", "rgbdiff"], groups=8, B=1, S=2, size=64,
                                    sound_size=64, modes=["eval", "train_main"]),
    "adamml_4mod": dict(kind="adamml", modality=["rgb", "sound", "flow", "rgbdiff"], groups=8, B=1, S=2, size=64,
                        sound_size=64, modes=["eval", "train_policy"]),
    # FULL-SIZE, well-conditioned cases (196+ samples per BatchNorm channel everywhere): BASELINE.json configs[0] (C1) and the
    # configs[1] workload (C2) at B = 4 videos.  full=True stores every running statistic and the head gradients in full;
    # "eval_cal" = eval mode on running statistics calibrated by one momentum-1 train-mode pass over the same input.
    "resnet50_c1": dict(kind="resnet", modality=["rgb"], groups=8, B=4, size=224, modes=["eval_cal", "train"], full=True),
    "adamml_c2": dict(kind="adamml", modality=["rgb", "sound"], groups=8, B=4, S=5, size=224, sound_size=256,
                      modes=["eval_cal", "train_main", "train_policy"], full=True, margin=0.08),
    # BASELINE.json configs[3] / configs[4] at full size (B = 2 videos, 5 segments, 224^2): the 3- and 4-modality workloads with the
    # policy / main modality-order quirk (models/adamml.py:143-146,85-86: decisions index (rgb, [sound,] rgbdiff), main nets take
    # (rgb, [sound,] flow)), the 10-channel ResNet stem (models/resnet.py:138) and the 15-channel policy stem (models/policy_net.py:195-200)
    "adamml_c4": dict(kind="adamml", modality=["rgb", "flow", "rgbdiff"], groups=8, B=2, S=5, size=224, sound_size=256,
                      modes=["eval_cal", "train_main"], full=True, margin=0.08),
    "adamml_c5": dict(kind="adamml", modality=["rgb", "sound", "flow", "rgbdiff"], groups=8, B=2, S=5, size=224, sound_size=256,
                      modes=["eval_cal", "train_main", "train_policy"], full=True, margin=0.08),
}


def _idx(name, n, k=4):
    r = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    return r.integers(0, n, size=(k,))


def grad_probe(name, g):
    """[sum, l2-norm, 4 sampled elements] of a gradient tensor (float64 accumulate)."""
    a = np.asarray(g.detach().cpu().double().numpy()).reshape(-1)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[_idx(name, a.size)]])


HEAD_GRADS = ("fc.weight", "fc.bias", "classifier.1.weight", "classifier.1.bias", "lf_weights", "lstm.bias_ih", "fcs.0.weight",
              "fcs.1.weight", "joint.2.bias")


def is_head(name):
    return name.endswith(HEAD_GRADS)


def stat_probe(t):
    """[mean, abs-mean, l2-norm, first, last] of a tensor."""
    a = np.asarray(t.detach().cpu().double().numpy()).reshape(-1)
    return np.array([a.mean(), np.abs(a).mean(), np.sqrt((a * a).sum()), a[0], a[-1]])
