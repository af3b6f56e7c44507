"""Model registry: the drop-in boundary of the reference (models/model_builder.py:3-38).

Contract kept: `MODEL_TABLE` maps `--backbone_net` to a factory that is called with the WHOLE argparse namespace as
keyword arguments; `build_model(args, test_mode)` returns `(model, arch_name)` where `arch_name` names the log / snapshot
folder, so its spelling is part of the on-disk interface (train_adamml.py:314-318 derives the folder from it)."""
from .adamml import adamml
from .resnet import resnet
from .sound_mobilenet_v2 import sound_mobilenet_v2

MODEL_TABLE = {"adamml": adamml, "resnet": resnet, "sound_mobilenet_v2": sound_mobilenet_v2}


def _run_name(args, network_name, test_mode):
    """<dataset>-<modalities joined by '-'>-<network>-f<groups>[-s<frames_per_group>][-<sched>[-syncbn]-bs<b>[-<prefix>]-e<epochs>]"""
    modalities = args.modality if isinstance(args.modality, str) else "-".join(args.modality)
    parts = [args.dataset, modalities, network_name, "f%d" % args.groups]
    if args.dense_sampling:
        parts.append("s%d" % args.frames_per_group)
    if not test_mode:                                   # the training settings only appear in training runs
        parts.append(args.lr_scheduler)
        if args.sync_bn:
            parts.append("syncbn")
        parts.append("bs%d" % args.batch_size)
        if args.prefix:
            parts.append(args.prefix)
        parts.append("e%d" % args.epochs)
    return "-".join(str(p) for p in parts)


def build_model(args, test_mode=False):
    factory = MODEL_TABLE[args.backbone_net]
    model = factory(**vars(args))
    # the reference falls back to the flag when the module has no usable `network_name` (its ResNet / sound MobileNetV2
    # properties raise AttributeError, so hasattr() is False there: SURVEY.md section 8b)
    return model, _run_name(args, getattr(model, "network_name", args.backbone_net), test_mode)
