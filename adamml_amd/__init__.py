"""adamml_amd -- MI355X-native (gfx950) AdaMML hot path behind the reference's model-registry boundary.

    from adamml_amd import build_model, MODEL_TABLE        # == models.model_builder of IBM/AdaMML

All device math is hand-written HIP in adamml_amd/libadamml_hip.so (C ABI: include/adamml_hip.h)."""
from .adamml import adamml
from .resnet import resnet
from .sound_mobilenet_v2 import sound_mobilenet_v2
from .model_builder import build_model, MODEL_TABLE

__all__ = ['adamml', 'resnet', 'sound_mobilenet_v2', 'build_model', 'MODEL_TABLE']
