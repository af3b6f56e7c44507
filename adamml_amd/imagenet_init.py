"""ImageNet initialisation from LOCAL files, with the reference's channel conversion.

The reference starts every recipe from torchvision's ImageNet weights: `resnet(..., imagenet_pretrained=True)` downloads
`resnet{depth}` (models/resnet.py:251-257), `sound_mobilenet_v2(..., imagenet_pretrained=True)` and -- unconditionally --
every policy MobileNetV2 download a MobileNetV2 file (models/sound_mobilenet_v2.py:186-196: torchvision's `mobilenet_v2`;
models/policy_net.py:13-15,193-203,221: d-li14's `mobilenetv2_160x160`, whose entries carry the policy net's own names).  The target
systems have no network, so the same `state_dict` files are read from disk instead:

    python train_adamml.py ... --imagenet_weights resnet50=/data/resnet50-19c8e357.pth mobilenet_v2=/data/mobilenet_v2-b0353104.pth \
                                                  mobilenetv2_160x160=/data/mobilenetv2_160x160-64dc7fa1.pth
    ADAMML_IMAGENET_DIR=/data            (looked up by the published file names: the first `<arch>*.pth` in that directory)
    imagenet_init.configure(resnet50=..., mobilenet_v2=...)      /      resnet(..., imagenet_pretrained="/data/resnet50.pth")

and converted exactly as the reference converts them:
  * ResNet: `fc.*` dropped; for input_channels != 3 every `conv1.weight` entry with 3 input channels and a 7x7 kernel -- the stem; the
    3x3 `layerN.M.conv1.weight` never matches -- becomes its mean over RGB expanded to input_channels (models/resnet.py:19-33);
  * MobileNetV2 (sound main net, policy nets): `features.0.0.weight` -> mean over RGB expanded to input_channels when != 3; the
    classifier entries are dropped (`classifier.1.*` of the torchvision file, `classifier.*` of d-li14's);
  * `load_state_dict(strict=False)`: everything else is taken by name.
With no file configured the factories behave as before this module existed (weights stay at their initialisation), and say so once."""
import glob
import os
import warnings

import torch

_PATHS = {}
_WARNED = set()
# torchvision's file names (models/resnet.py:9-15, models/policy_net.py:9-11): what ADAMML_IMAGENET_DIR is searched for
_FILE_STEMS = {"resnet18": "resnet18", "resnet34": "resnet34", "resnet50": "resnet50", "resnet101": "resnet101", "resnet152": "resnet152",
               "mobilenet_v2": "mobilenet_v2", "mobilenetv2_160x160": "mobilenetv2_160x160"}


def configure(**paths):
    """configure(resnet50="/path/resnet50-19c8e357.pth", mobilenet_v2="/path/mobilenet_v2-b0353104.pth"); None removes an entry."""
    for arch, p in paths.items():
        if arch not in _FILE_STEMS:
            raise ValueError("imagenet_init.configure: unknown architecture %r (one of %s)" % (arch, sorted(_FILE_STEMS)))
        if p is None:
            _PATHS.pop(arch, None)
        else:
            _PATHS[arch] = os.fspath(p)


def configure_from_args(items):
    """`--imagenet_weights resnet50=PATH mobilenet_v2=PATH` of the launcher."""
    kv = {}
    for it in items or []:
        if "=" not in it:
            raise ValueError("--imagenet_weights expects ARCH=PATH items, got %r" % it)
        a, p = it.split("=", 1)
        kv[a] = p
    configure(**kv)


def path_for(arch):
    """The configured file of `arch`, or the first `<arch>*.pth` in $ADAMML_IMAGENET_DIR, or None."""
    p = _PATHS.get(arch)
    if p:
        return p
    d = os.environ.get("ADAMML_IMAGENET_DIR")
    if d:
        hits = sorted(glob.glob(os.path.join(d, _FILE_STEMS[arch] + "*.pth")))
        if len(hits) > 1:
            warnings.warn("%s: %d files match %s*.pth in $ADAMML_IMAGENET_DIR; using %s (configure the path explicitly to choose another)"
                          % (arch, len(hits), _FILE_STEMS[arch], hits[0]))
        if hits:
            return hits[0]
    return None


def _read(path):
    # weights_only: the path comes from a command-line flag / an environment variable; a published checkpoint is tensors only
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def expand_rgb_kernel(w, input_channels):
    """[O, 3, kh, kw] -> [O, input_channels, kh, kw]: the mean over RGB repeated (models/resnet.py:29-30)."""
    o, _, kh, kw = w.shape
    return w.mean(dim=1, keepdim=True).expand(o, input_channels, kh, kw).contiguous()


def convert_resnet(state_dict, input_channels):
    """models/resnet.py:19-33 + :253-256 on a torchvision ResNet state_dict."""
    out = {}
    for k, v in state_dict.items():
        if k in ("fc.weight", "fc.bias"):
            continue
        if input_channels != 3 and "conv1.weight" in k and v.dim() == 4 and v.shape[1] == 3 and v.shape[2] == 7 and v.shape[3] == 7:
            v = expand_rgb_kernel(v, input_channels)
        out[k] = v
    return out


def convert_mobilenet_v2(state_dict, input_channels):
    """models/policy_net.py:195-202 / models/sound_mobilenet_v2.py:188-195 on a torchvision MobileNetV2 state_dict."""
    out = {k: v for k, v in state_dict.items() if not k.startswith("classifier.")}
    if input_channels != 3:
        out["features.0.0.weight"] = expand_rgb_kernel(out["features.0.0.weight"], input_channels)
    return out


def _load(model, sd, what):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if hasattr(model, "mark_weights_dirty"):
        model.mark_weights_dirty()
    return missing, unexpected


def _resolve(arch, flag, what):
    if isinstance(flag, (str, os.PathLike)):
        return os.fspath(flag)
    if not flag:
        return None
    p = path_for(arch)
    if p is None and (arch, what) not in _WARNED:
        _WARNED.add((arch, what))
        warnings.warn("%s: the reference initialises this network from downloaded ImageNet weights (%s); no local file is "
                      "configured (--imagenet_weights %s=PATH / ADAMML_IMAGENET_DIR / imagenet_init.configure), so the weights keep "
                      "their random initialisation" % (what, arch, arch))
    return p


def init_resnet(model, depth, input_channels, flag=True):
    """models/resnet.py:251-257.  flag: True (use the configured file), a path, or False."""
    p = _resolve("resnet%d" % depth, flag, "resnet")
    if p is not None:
        _load(model, convert_resnet(_read(p), input_channels), "resnet")
    return model


def init_mobilenet_v2(model, input_channels, flag=True, what="mobilenet_v2", arch="mobilenet_v2"):
    """models/sound_mobilenet_v2.py:186-196 (arch "mobilenet_v2": torchvision's file) and models/policy_net.py:193-203 (arch
    "mobilenetv2_160x160": d-li14's file, named like the policy MobileNetV2 itself)."""
    p = _resolve(arch, flag, what)
    if p is not None:
        _load(model, convert_mobilenet_v2(_read(p), input_channels), what)
    return model
