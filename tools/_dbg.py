import sys; sys.path.insert(0,'.')
import torch, numpy as np, torch.nn.functional as F
from adamml_amd import synth
from adamml_amd.resnet import resnet
from oracle import adamml_oracle as O
from tests.golden_cases import CASES
from tests.oracle_harness import manifest, case_inputs
name = sys.argv[1] if len(sys.argv) > 1 else 'resnet50_train'
c=CASES[name]
sd=synth.synth_state_dict(manifest(c),seed=1234)
m=resnet(depth=50,num_classes=31,without_t_stride=False,groups=c['groups'],dropout=0.0,pooling_method='max',input_channels=3,imagenet_pretrained=False)
m.load_state_dict(sd); m.cuda(); m.train()
x,t=case_inputs(c)
y=m(x.cuda()); F.cross_entropy(y,t.cuda()).backward()
s=O.make_leaf_state({k:v.cuda() for k,v in sd.items()},("",))
yo=O.resnet_forward(s,"",x.cuda(),c['groups'],50,'max',False,0.0,True)
F.cross_entropy(yo,t.cuda()).backward()
print('logits rel', ((y-yo).abs().max()/yo.abs().max()).item())
for k,p in m.named_parameters():
    g=p.grad; r=s[k].grad
    print("%-34s rel-l2 %.4f  cos %.5f  |ref| %.3e"%(k, ((g-r).norm()/(r.norm()+1e-20)).item(), (F.cosine_similarity(g.flatten(),r.flatten(),dim=0)).item(), r.norm().item()))
