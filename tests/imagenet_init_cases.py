"""Shared by tools/gen_imagenet_init_golden.py (runs the reference) and tests/test_host_cpu.py (runs adamml_amd.imagenet_init): the
synthetic torchvision-format files, the cases and the digest stored per state_dict entry."""
import numpy as np
import torch

from adamml_amd import synth

CASES = {
    "resnet50_rgb": {"kind": "resnet", "ch": 3, "stem": "conv1.weight"},
    "resnet50_flow": {"kind": "resnet", "ch": 10, "stem": "conv1.weight"},           # models/resnet.py:19-33: mean over RGB x 10
    "sound_mobilenet_v2": {"kind": "sound", "ch": 1, "stem": "features.0.0.weight"},
    "policy_rgb": {"kind": "policy", "ch": 3, "stem": "features.0.0.weight"},
    "policy_rgbdiff": {"kind": "policy", "ch": 15, "stem": "features.0.0.weight"},   # models/policy_net.py:195-200
    "policy_sound": {"kind": "policy", "ch": 1, "stem": "features.0.0.weight"},
}


def _tv_shapes(arch):
    """Names / shapes of torchvision's resnet50 / mobilenet_v2 state_dict (1000 classes, RGB stem), written out from the architecture:
    no torchvision on the build systems."""
    shp = {}

    def bn(p, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            shp[p + "." + n] = (c,)
        shp[p + ".num_batches_tracked"] = ()
    if arch == "resnet50":
        shp["conv1.weight"] = (64, 3, 7, 7)
        bn("bn1", 64)
        inp = 64
        for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), 1):
            for b in range(blocks):
                p = "layer%d.%d" % (li, b)
                shp[p + ".conv1.weight"] = (planes, inp, 1, 1)
                bn(p + ".bn1", planes)
                shp[p + ".conv2.weight"] = (planes, planes, 3, 3)
                bn(p + ".bn2", planes)
                shp[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                bn(p + ".bn3", planes * 4)
                if b == 0:
                    shp[p + ".downsample.0.weight"] = (planes * 4, inp, 1, 1)
                    bn(p + ".downsample.1", planes * 4)
                inp = planes * 4
        shp["fc.weight"], shp["fc.bias"] = (1000, 2048), (1000,)
        return shp
    shp["features.0.0.weight"] = (32, 3, 3, 3)
    bn("features.0.1", 32)
    inp, i = 32, 1
    for t, c, n, s in ([1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]):
        for _ in range(n):
            p, hid, j = "features.%d.conv" % i, inp * t, 0
            if t != 1:
                shp["%s.0.0.weight" % p] = (hid, inp, 1, 1)
                bn("%s.0.1" % p, hid)
                j = 1
            shp["%s.%d.0.weight" % (p, j)] = (hid, 1, 3, 3)
            bn("%s.%d.1" % (p, j), hid)
            shp["%s.%d.weight" % (p, j + 1)] = (c, hid, 1, 1)
            bn("%s.%d" % (p, j + 2), c)
            inp, i = c, i + 1
    shp["features.18.0.weight"] = (1280, 320, 1, 1)
    bn("features.18.1", 1280)
    shp["classifier.1.weight"], shp["classifier.1.bias"] = (1000, 1280), (1000,)
    return shp


def torchvision_like(arch, seed=4321, like=None):
    """A state_dict with the names and shapes of the published file of `arch` (torchvision's resnet50 / mobilenet_v2; for d-li14's
    mobilenetv2_160x160 the caller passes `like` = the state_dict of a 1000-class RGB policy MobileNetV2, whose names the file uses),
    values from the name-keyed generator."""
    if like is None:
        like = {k: torch.empty(s, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, s in _tv_shapes(arch).items()}
    sd = synth.synth_state_dict(like, seed=seed)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=torch.int64)
    return sd


def digest(sd, stem_key):
    """The stem kernel in full; (sum, sum |.|, four strided samples) of everything else, fp64."""
    out = {}
    for k, v in sd.items():
        a = v.detach().double().reshape(-1).numpy()
        if k == stem_key:
            out[k] = v.detach().float().numpy()
        else:
            n = a.size
            idx = [0, n // 3, (2 * n) // 3, n - 1] if n else []
            out[k] = np.array([a.sum(), np.abs(a).sum()] + [a[i] for i in idx] + [float(n)])
    return out
