"""Model registry: the drop-in boundary of the reference (models/model_builder.py:3-38)."""
from .adamml import adamml
from .resnet import resnet
from .sound_mobilenet_v2 import sound_mobilenet_v2

MODEL_TABLE = {
    'adamml': adamml,
    'resnet': resnet,
    'sound_mobilenet_v2': sound_mobilenet_v2
}


def build_model(args, test_mode=False):
    """Same contract as models/model_builder.py:10-38: returns (model, arch_name)."""
    model = MODEL_TABLE[args.backbone_net](**vars(args))
    network_name = model.network_name if hasattr(model, 'network_name') else args.backbone_net
    if isinstance(args.modality, list):
        modality = '-'.join([x for x in args.modality])
    else:
        modality = args.modality
    arch_name = "{dataset}-{modality}-{arch_name}".format(dataset=args.dataset, modality=modality, arch_name=network_name)
    arch_name += "-f{}".format(args.groups)
    if args.dense_sampling:
        arch_name += "-s{}".format(args.frames_per_group)
    if not test_mode:
        arch_name += "-{}{}-bs{}{}-e{}".format(args.lr_scheduler, "-syncbn" if args.sync_bn else "", args.batch_size,
                                             '-' + args.prefix if args.prefix else "", args.epochs)
    return model, arch_name
