"""Debug aid (GPU box): layer-by-layer comparison of the HIP ResNet pipeline against the oracle run in fp32 on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from adamml_amd import synth
from adamml_amd.resnet import resnet
from adamml_amd.runtime import Lazy, conv_bn, add_act, maxpool3x3s2, temporal_pool, gap, clip_to_nhwc, ACT_NONE, ACT_RELU
from adamml_amd.hip import call, ptr
from oracle import adamml_oracle as O
from tests.golden_cases import CASES
from tests.oracle_harness import manifest, case_inputs

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50_train"
training = len(sys.argv) > 2 and sys.argv[2] == "train"
c = CASES[name]
sd = synth.synth_state_dict(manifest(c), seed=1234)
model = resnet(depth=50, num_classes=31, without_t_stride=False, groups=c["groups"], dropout=0.0, pooling_method="max",
               input_channels=3, imagenet_pretrained=False)
model.load_state_dict(sd); model.cuda(); model.train(training)
x, _ = case_inputs(c); x = x.cuda()
sdg = {k: v.cuda() for k, v in sd.items()}

def mat(l):
    n, h, w, C = l.shape
    out = torch.empty_like(l.data)
    call("adamml_bn_act_add", ptr(l.data), ptr(l.scale), ptr(l.shift), l.gs, l.act, None, None, None, 0, ptr(out), n*h*w, C, 1)
    return out.float().permute(0, 3, 1, 2)

def cmp(tag, mine, ref):
    e = (mine - ref).abs().max().item(); s = ref.abs().max().item()
    print("%-28s max|err| %.4g  scale %.4g  rel %.4f  mean|ref| %.4g" % (tag, e, s, e / (s + 1e-12), ref.abs().mean().item()))

rt = model.rt
rt.begin_forward(x.device, training, False)
model._repack(False)
frames = c["groups"]
xs = clip_to_nhwc(x, 1, frames, 3)[0]
n = x.shape[0]
r = x.reshape(n * frames, 3, x.shape[2], x.shape[3])
r = F.conv2d(r, sdg["conv1.weight"], stride=2, padding=3)
cmp("stem raw", None if False else xs.new_zeros(1).float() * 0 + 0 if False else torch.zeros(1).cuda(), torch.zeros(1).cuda())
h = conv_bn(rt, Lazy(xs, requires_grad=False), model._stem, model.bn1, ACT_RELU)
cmp("stem conv raw", h.data.float().permute(0, 3, 1, 2), r)
r = F.relu(O.batchnorm(sdg, "bn1", r, training))
cmp("stem bn relu", mat(h), r)
h = maxpool3x3s2(rt, h); r = F.max_pool2d(r, 3, 2, 1)
cmp("maxpool", mat(h), r)
inpl = 64
for li, (planes, layer) in enumerate(zip((64, 128, 256, 512), (model.layer1, model.layer2, model.layer3, model.layer4))):
    for bi, b in enumerate(layer):
        o = conv_bn(rt, h, b._cs1, b.bn1, ACT_RELU)
        o = conv_bn(rt, o, b._cs2, b.bn2, ACT_RELU)
        o = conv_bn(rt, o, b._cs3, b.bn3, ACT_NONE)
        idn = conv_bn(rt, h, b._csd, b.downsample[1], ACT_NONE) if b._csd is not None else h
        h = add_act(rt, o, idn, ACT_RELU)
        r = O._bottleneck(sdg, "layer%d.%d" % (li + 1, bi), r, b.stride, b._csd is not None, training)
        cmp("layer%d.%d" % (li + 1, bi), mat(h), r)
    if li < 3:
        h = temporal_pool(rt, h, frames, "max"); r = O.temporal_pool(r, frames, "max"); frames = max(1, frames // 2)
        cmp("tpool%d" % (li + 1), mat(h), r)
f, _ = gap(rt, h)
cmp("gap", f, r.mean((2, 3)))
