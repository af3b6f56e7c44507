"""Batch data-parallel wrapper over RCCL (torch.distributed backend "nccl" on ROCm), one process per GPU
(train_adamml.py:60,83,122-129).  The video batch is sharded across ranks; after backward the flat gradient buffers of
the TRAINABLE sub-networks are averaged with one all-reduce each (the reference's DDP reduces all 42 M parameters in
every stage; frozen sub-networks have no gradients here)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class HipDDP(nn.Module):
    """DistributedDataParallel-shaped wrapper: exposes `.module`, forwards calls, averages gradients on demand.
    Call `reduce_gradients()` after `loss.backward()` (the restated train loop does)."""

    def __init__(self, module, process_group=None, sync_bn=False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if sync_bn and self.world > 1 and hasattr(module, "enable_sync_bn"):
            module.enable_sync_bn(process_group)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def broadcast_parameters(self, src=0):
        if self.world == 1:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    def reduce_gradients(self):
        if self.world == 1:
            return
        flats = self.module.flat_grad_buffers() if hasattr(self.module, "flat_grad_buffers") else []
        covered = set()
        for fg in flats:
            dist.all_reduce(fg, group=self.group)
            fg.div_(self.world)
            lo, hi = fg.data_ptr(), fg.data_ptr() + fg.numel() * 4
            covered.add((lo, hi))
        for p in self.module.parameters():        # stragglers not living in a flat buffer
            if p.grad is None:
                continue
            if any(lo <= p.grad.data_ptr() < hi for lo, hi in covered):
                continue
            dist.all_reduce(p.grad, group=self.group)
            p.grad.div_(self.world)


def shard_batch(tensors, rank, world):
    """Rank r takes videos r::world of the global batch (DistributedSampler semantics, utils/utils.py:157)."""
    return [t[rank::world].contiguous() for t in tensors]
