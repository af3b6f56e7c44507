"""2 gloo ranks on one GPU: one SyncBN training step in both freeze stages; prints gradient / running-stat checksums so that
ADAMML_INTERLEAVE=0 and =1 (round-robin issue of the backbones) can be compared."""
import os, sys, torch
import torch.distributed as dist
import torch.nn.functional as F
sys.path.insert(0, ".")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from adamml_amd import adamml, synth
from adamml_amd.distributed import HipDDP
dev = torch.device("cuda", 0)
S, B = 3, 2
m = adamml(groups=8, modality=["rgb", "sound"], input_channels=[3, 1], num_segments=S, rng_policy=False, rng_threshold=0.5,
           causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.0, pooling_method="max",
           fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=1234)); m.to(dev)
ddp = HipDDP(m, sync_bn=True)
xs = [t[rank::world].to(dev) for t in synth.synth_inputs(["rgb", "sound"], B * world, S, 8, 64, seed=5)]
tgt = synth.synth_labels(B * world, 31, seed=5)[rank::world].to(dev)
expo = synth.synth_gumbel_exponential(S, 2, B * world, seed=11).view(S, 2, B * world, 2)[:, :, rank::world].reshape(S, 2 * B, 2).to(dev)
for stage in ("main", "policy"):
    m.unfreeze_main_net(); m.unfreeze_policy_net()
    (m.freeze_policy_net if stage == "main" else m.freeze_main_net)()
    m.train(); m.zero_grad()
    out, sel = ddp(xs, gumbel_exponential=expo)
    loss = F.cross_entropy(out, tgt) + (sel.mean(dim=1) ** 2).mean()
    loss.backward()
    ddp.reduce_gradients()
    torch.cuda.synchronize()
    fg = m._flat_main.flat_grad if stage == "main" else m._flat_policy.flat_grad
    rm = m.state_dict()["main_net.nets.0.layer3.2.bn3.running_mean"]
    if rank == 0:
        print("%s: loss %.6f grad_norm %.6e grad_abs_sum %.6e running_mean_sum %.6e sel %s" %
              (stage, loss.item(), fg.norm().item(), fg.abs().sum().item(), rm.double().sum().item(), sel.sum().item()))
dist.destroy_process_group()
