"""Mean / std contract shared by every model (models/adamml.py:93-99, models/resnet.py:225-231)."""
import numpy as np


class MeanStdMixin:
    def mean(self, modality='rgb'):
        return [0.485, 0.456, 0.406] if modality == 'rgb' or modality == 'rgbdiff' else [0.5]

    def std(self, modality='rgb'):
        return [0.229, 0.224, 0.225] if modality == 'rgb' or modality == 'rgbdiff' \
            else [np.mean([0.229, 0.224, 0.225])]
