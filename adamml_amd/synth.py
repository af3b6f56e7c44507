"""Deterministic, torch-RNG-independent synthetic weights / inputs / noise.

Every tensor is generated from a seed derived from its *name* (crc32) so the
build container (where the reference is importable) and the GPU box (where it
is not) regenerate bit-identical fp32 values from the recipe alone; only
outputs have to be stored as golden fixtures.
"""
import re
import zlib
import numpy as np
import torch

__all__ = ["det_normal", "det_uniform", "synth_state_dict", "synth_inputs", "synth_gumbel_exponential"]


_RESIDUAL_BN = re.compile(r"(bn3\.weight|features\.\d+\.conv\.(3|7)\.weight)$")


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed & 0xFFFFFFFF]))


def det_normal(name, shape, seed=1234, mean=0.0, std=1.0):
    a = _rng(name, seed).standard_normal(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def det_uniform(name, shape, seed=1234, lo=0.0, hi=1.0):
    a = _rng(name, seed).random(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(hi - lo) + np.float32(lo))


def synth_state_dict(ref_sd, seed=1234):
    """Fill every entry of a state_dict (name -> tensor, only shapes/dtypes are
    used) with deterministic values appropriate to its role."""
    out = {}
    for k, v in ref_sd.items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shp, dtype=torch.int64)
        elif k.endswith("running_mean"):
            out[k] = det_normal(k, shp, seed, 0.0, 0.1)
        elif k.endswith("running_var"):
            out[k] = det_uniform(k, shp, seed, 0.5, 1.5)
        elif k.endswith("lf_weights"):
            out[k] = det_uniform(k, shp, seed, 0.3, 0.6)
        elif v.dim() == 4:  # conv OIHW: He-style on fan_in
            fan_in = shp[1] * shp[2] * shp[3]
            out[k] = det_normal(k, shp, seed, 0.0, (2.0 / fan_in) ** 0.5)
        elif v.dim() == 2:  # linear / lstm matrices
            out[k] = det_normal(k, shp, seed, 0.0, (1.0 / shp[1]) ** 0.5)
        elif v.dim() == 1:
            # BN weight vs (BN|linear|lstm) bias: BN weights are named '*.weight' with 1 dim
            if _RESIDUAL_BN.search(k):
                # last BN of a residual branch: small gain ("zero-init residual" regime).  With O(1) gains on
                # every branch a randomly initialised 16-block residual stack is chaotic: rounding only the
                # weights to bf16 moves fp32 logits by 20 %, so no reduced-precision parity could be stated.
                out[k] = det_uniform(k, shp, seed, 0.1, 0.3)
            elif k.endswith("weight"):
                out[k] = det_uniform(k, shp, seed, 0.5, 1.5)
            else:
                out[k] = det_normal(k, shp, seed, 0.0, 0.1)
        else:
            out[k] = det_normal(k, shp, seed, 0.0, 1.0)
        out[k] = out[k].to(v.dtype) if v.dtype.is_floating_point else out[k]
    return out


_IN_STATS = {"rgb": (0.0, 1.0), "flow": (0.0, 1.0), "rgbdiff": (0.0, 1.0), "sound": (-5.0, 3.0)}
_IN_CH = {"rgb": 3, "flow": 10, "rgbdiff": 15, "sound": 1}


def synth_inputs(modality, batch, num_segments, groups=8, size=224, sound_size=256, seed=42):
    """Synthetic clip batch in the reference input contract:
    visual [B, S*F*C, H, W]; sound [B, S, 256, 256]."""
    xs = []
    for m in modality:
        mu, sd = _IN_STATS[m]
        if m == "sound":
            shp = (batch, num_segments, sound_size, sound_size)
        else:
            shp = (batch, num_segments * groups * _IN_CH[m], size, size)
        xs.append(det_normal("input." + m, shp, seed, mu, sd))
    return xs


def synth_labels(batch, num_classes=31, seed=42):
    a = _rng("labels", seed).integers(0, num_classes, size=(batch,))
    return torch.from_numpy(a.astype(np.int64))


def synth_gumbel_exponential(num_segments, num_modality, batch, seed=7):
    """Exponential(1) samples E with the layout the policy head consumes:
    [S, M*B, 2]; gumbel noise is -log(E) (F.gumbel_softmax: -empty.exponential_().log())."""
    u = _rng("gumbel", seed).random((num_segments, num_modality * batch, 2), dtype=np.float64)
    e = -np.log1p(-u)  # u in [0,1) -> E in [0, inf)
    e = np.maximum(e, 1e-10)
    return torch.from_numpy(e.astype(np.float32))
