"""Debug: grouped call vs sequential calls vs sequential-again noise floor."""
import copy, sys, torch
sys.path.insert(0, ".")
from tests.test_models_gpu import _make_backbone, rel_l2, DEV
from adamml_amd.backbone import FlatBuffers
for kind in ("resnet", "sound", "policy"):
    net_a, x = _make_backbone(kind)
    nets = [net_a, copy.deepcopy(net_a), copy.deepcopy(net_a)]
    S = x.shape[0]
    outs = []
    for net, grouped in zip(nets, (True, False, False)):
        if net.flat_owner is None:
            net.flat_owner = FlatBuffers(net)
        net.flat_owner.ensure(x.device); net.flat_owner.ensure_grads()
        y = net.call(x.flatten(0, 1), S) if grouped else torch.cat([net.call(x[i].contiguous(), 1) for i in range(S)], 0)
        w = torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)
        (y * w).sum().backward(); torch.cuda.synchronize()
        outs.append(y.detach())
    print(kind, "grouped-vs-seq", rel_l2(outs[0], outs[1]), "seq-vs-seq", rel_l2(outs[1], outs[2]))
    for i in range(S):
        n = outs[0].shape[0] // S
        print("   seg", i, rel_l2(outs[0][i*n:(i+1)*n], outs[1][i*n:(i+1)*n]))
    pa, pb, pc = [dict(n.named_parameters()) for n in nets]
    for k in list(pa)[:6] + list(pa)[-4:]:
        print("   ", k, rel_l2(pa[k].grad, pb[k].grad), rel_l2(pb[k].grad, pc[k].grad))
