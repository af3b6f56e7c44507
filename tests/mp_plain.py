"""Results of spawned ranks travel to the test process through a multiprocessing Manager.  Torch tensors in that channel are pickled by
torch.multiprocessing's storage reductions (file descriptors / shared memory handed around by the Manager's server process); the server
forked from a pytest process that holds a HIP context once died in its garbage collector while doing so (round 3).  So: the Manager runs
in a SPAWNED process (no inherited HIP state) and only plain numpy / Python values cross it."""
import multiprocessing

import numpy as np
import torch


def manager():
    return multiprocessing.get_context("spawn").Manager()


def plain(obj):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu()
        if t.dtype == torch.bfloat16:
            return ("__bf16__", t.float().numpy().copy())
        return t.numpy().copy()
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(plain(v) for v in obj)
    return obj


def tensors(obj):
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and obj[0] == "__bf16__":
        return torch.from_numpy(obj[1]).bfloat16()
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(obj)
    if isinstance(obj, dict):
        return {k: tensors(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tensors(v) for v in obj)
    return obj
