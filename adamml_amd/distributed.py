"""Batch data-parallel wrapper over RCCL (torch.distributed backend "nccl" on ROCm), one process per GPU
(train_adamml.py:60,83,122-129).  The video batch is sharded across ranks; the flat gradient buffers of the TRAINABLE
sub-networks are averaged over the ranks (the reference's DDP reduces all 42 M parameters in every stage; frozen
sub-networks have no gradients here).  The backbones announce contiguous slices of the flat buffer as soon as the
backward pass has finished writing them (ResNet: layer4+fc, layer3, stem..layer2; a MobileNetV2 as a whole); each slice
is all-reduced asynchronously right then, so the exchange of the deep layers' gradients (most of the bytes) travels over
xGMI while the shallow layers' backward kernels still run.  reduce_gradients() waits for those and reduces what is left."""
import torch
import torch.distributed as dist
import torch.nn as nn


class HipDDP(nn.Module):
    """DistributedDataParallel-shaped wrapper: exposes `.module`, forwards calls, averages gradients on demand.
    Call `reduce_gradients()` after `loss.backward()` (the restated train loop does)."""

    def __init__(self, module, process_group=None, sync_bn=False, overlap=True, force_collectives=False):
        """force_collectives=True issues every collective of the N > 1 path on a ONE-rank group as well (each is an identity
        there): the bucketed asynchronous all-reduce from inside backward, the coalesced SyncBatchNorm exchange and their
        stream / event ordering then run against the real RCCL backend on a single GPU (tests/test_rccl_gpu.py)."""
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.overlap = overlap
        self._pending = []
        self.stats = {"bucket_all_reduces": 0, "gap_all_reduces": 0}
        if self.active and hasattr(module, "backbones"):
            for net in module.backbones():
                net.grad_hook = self._on_grads_ready
        if sync_bn and self.active and hasattr(module, "enable_sync_bn"):
            module.enable_sync_bn(process_group, force=force_collectives)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def broadcast_parameters(self, src=0):
        """Rank `src`'s parameters and buffers to every rank, as torch's DistributedDataParallel does at construction
        (train_adamml.py:129).  The writes go through .data: the backbones re-pack their bf16 operands afterwards."""
        if not self.active:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src, group=self.group)
        for m in self.module.modules():
            if hasattr(m, "mark_weights_dirty"):
                m.mark_weights_dirty()

    # -- overlapped bucket exchange --------------------------------------------------------------------------------
    def _flat_buffers(self):
        for name in ("_flat_main", "_flat_policy", "flat_owner"):
            fb = getattr(self.module, name, None)
            if fb is not None and getattr(fb, "detached", False):
                raise RuntimeError("HipDDP: the parameter gradients of this model are delivered through autograd (it went through a stock "
                                   "DistributedDataParallel wrap / enable_autograd_param_grads): there is no flat gradient buffer to reduce")
        return self.module.flat_grad_buffers() if hasattr(self.module, "flat_grad_buffers") else []

    def _on_grads_ready(self, params):
        """Called by a backbone from inside backward when the gradients of `params` are final (the kernels writing them
        are already enqueued on the current stream).  Starts an asynchronous all-reduce of the slice of the flat gradient
        buffer they occupy; non-contiguous or non-flat gradients are left to reduce_gradients()."""
        if not self.active or not self.overlap:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        lo = min(g.data_ptr() for g in grads)
        hi = max(g.data_ptr() + g.numel() * g.element_size() for g in grads)
        if hi - lo != sum(g.numel() * g.element_size() for g in grads):
            return                                          # not one contiguous run of the flat buffer
        for fg in self._flat_buffers():
            base = fg.data_ptr()
            if base <= lo and hi <= base + fg.numel() * 4:
                piece = fg[(lo - base) // 4:(hi - base) // 4]
                work = dist.all_reduce(piece, group=self.group, async_op=True)
                self.stats["bucket_all_reduces"] += 1
                self._pending.append((base, (lo - base) // 4, (hi - base) // 4, work))
                return

    def reduce_gradients(self):
        """Average the gradients over the ranks: waits for the slices already in flight, all-reduces the remaining gaps
        of each flat buffer (and any parameter living outside the flat buffers), then scales by 1/world once."""
        if not self.active:
            return
        pending, self._pending = self._pending, []
        for _, _, _, work in pending:
            work.wait()                                     # the current stream waits for the collective
        covered = set()
        for fg in self._flat_buffers():
            base, n = fg.data_ptr(), fg.numel()
            done = sorted((a, b) for (bs, a, b, _) in pending if bs == base)
            pos = 0
            for a, b in done + [(n, n)]:
                if a > pos:
                    dist.all_reduce(fg[pos:a], group=self.group)
                    self.stats["gap_all_reduces"] += 1
                pos = max(pos, b)
            if self.world > 1:
                fg.div_(self.world)
            covered.add((base, base + n * 4))
        for p in self.module.parameters():        # stragglers not living in a flat buffer
            if p.grad is None:
                continue
            if any(lo <= p.grad.data_ptr() < hi for lo, hi in covered):
                continue
            dist.all_reduce(p.grad, group=self.group)
            if self.world > 1:
                p.grad.div_(self.world)


def shard_batch(tensors, rank, world):
    """Rank r takes videos r::world of the global batch (DistributedSampler semantics, utils/utils.py:157)."""
    return [t[rank::world].contiguous() for t in tensors]
