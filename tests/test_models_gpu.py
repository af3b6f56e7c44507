"""Whole-network parity on the MI355X: HIP path vs golden vectors captured from the reference (tests/golden)
and vs the CPU oracle on the same synthetic inputs.  bf16 activations against an fp32 reference:
logits rtol/atol 3e-2 of the logit scale; gradient probes 8e-2 of each tensor's L2 norm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adamml_amd import synth  # noqa: E402
from tests.golden_cases import CASES, grad_probe, stat_probe  # noqa: E402
from tests.oracle_harness import manifest, load_golden, case_inputs  # noqa: E402

DEV = "cuda"


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def check_grad_probes(model, gold, mode, tol):
    names = list(gold[mode + ".grad_names"])
    ref = gold[mode + ".grad_probe"]
    params = dict(model.named_parameters())
    worst = (0.0, None)
    floor = 1e-3 * ref[:, 1].max()
    for i, k in enumerate(names):
        g = params[k].grad
        assert g is not None, "no grad for " + k
        got = grad_probe(k, g)
        l2 = max(ref[i, 1], floor)
        e = max(abs(got[1] - ref[i, 1]) / l2, np.abs(got[2:] - ref[i, 2:]).max() / l2)
        if e > worst[0]:
            worst = (e, k)
    print("worst grad probe error %.4f at %s" % worst)
    assert worst[0] < tol, worst


def check_stat_probes(model, gold, mode, tol=3e-2):
    names = list(gold[mode + ".stat_names"])
    ref = gold[mode + ".stat_probe"]
    sd = model.state_dict()
    for i, k in enumerate(names):
        got = stat_probe(sd[k])
        if k.endswith("num_batches_tracked"):
            continue
        # mean / abs-mean / l2 of running stats
        assert abs(got[1] - ref[i, 1]) <= tol * (abs(ref[i, 1]) + 1e-3), (k, got, ref[i])


@pytest.mark.parametrize("name", ["resnet50_train", "resnet50_full", "resnet50_avg", "resnet50_flow"])
def test_resnet50_vs_golden(name):
    from adamml_amd.resnet import resnet
    c = CASES[name]
    gold = load_golden(name)
    sd = synth.synth_state_dict(manifest(c), seed=1234)
    model = resnet(depth=50, num_classes=31, without_t_stride=False, groups=c["groups"], dropout=0.0,
                   pooling_method=c.get("pooling", "max"), input_channels={"rgb": 3, "flow": 10}[c["modality"][0]],
                   imagenet_pretrained=False)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.to(DEV)
    x, target = case_inputs(c)
    x, target = x.to(DEV), target.to(DEV)
    for mode in c["modes"]:
        model.load_state_dict(sd)
        if mode == "eval":
            model.eval()
            with torch.no_grad():
                y = model(x)
        else:
            model.train()
            model.zero_grad()
            y = model(x)
            loss = F.cross_entropy(y, target)
            loss.backward()
        e = rel_err(y.detach().cpu().numpy(), gold[mode + ".logits"])
        print(name, mode, "logit rel err %.4f" % e)
        assert e < 3e-2
        if mode != "eval":
            check_grad_probes(model, gold, mode, 8e-2)
            check_stat_probes(model, gold, mode)
