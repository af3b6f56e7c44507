"""Restated launcher (adamml_amd/train.py) on the GPU: the three-stage schedule runs end to end
on synthetic batches, writes reference-format checkpoints, resumes from them, and a reference-style checkpoint
(`module.`-prefixed state_dict) loads back bit-exactly."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ARGS = ["--backbone_net", "adamml", "-d", "50", "--groups", "8", "--num_segments", "2", "--modality", "rgb", "sound",
        "--causality_modeling", "lstm", "--learnable_lf_weights", "-b", "2", "--input_size", "64", "--epochs", "1",
        "--warmup_epochs", "1", "--finetune_epochs", "1", "--val_num_clips", "2", "--cost_weights", "1.0", "0.05",
        "--synthetic", "2", "--print-freq", "1", "--lr", "0.001", "--p_lr", "0.0001"]


def test_three_stage_schedule_checkpoints_and_resume(tmp_path):
    from adamml_amd import train
    lines = []
    res = train.main(ARGS + ["--logdir", str(tmp_path)], log=lines.append)
    stages = [h[0] for h in res["history"]]
    assert stages == ["warmup", "main", "policy", "finetune"], stages
    assert all(torch.isfinite(torch.tensor(h[2])) for h in res["history"])
    folder = res["log_folder"]
    # one decay after the alternating epoch -- unless that epoch produced a best model: the fine-tune stage then restores
    # the temperature saved WITH it, i.e. before the decay (train_adamml.py:503-516, 541-545: save, then decay; set_temperature on load)
    had_best = os.path.exists(os.path.join(folder, "model_best.pth.tar"))
    assert abs(res["temperature"] - (5.0 if had_best else 5.0 * 0.965)) < 1e-9
    for f in ("checkpoint.pth.tar", "checkpoint_warmup_01.pth.tar", "checkpoint_main_01.pth.tar", "checkpoint_finetune_01.pth.tar"):
        assert os.path.exists(os.path.join(folder, f)), f
    ck = torch.load(os.path.join(folder, "checkpoint.pth.tar"), map_location="cpu")
    assert ck["stage"] == "finetune" and ck["epoch"] == 1 and abs(ck["temperature"] - res["temperature"]) < 1e-9
    assert all(k.startswith("module.") for k in ck["state_dict"])           # the reference's DDP-prefixed names
    assert "module.policy_net.lstm.weight_ih" in ck["state_dict"] and "module.main_net.nets.0.layer4.2.conv3.weight" in ck["state_dict"]
    # interchange: load it into a fresh model through the reference-checkpoint path, names and values identical
    args = train.arg_parser().parse_args(ARGS)
    train.resolve_args(args, log=lambda *_: None)
    model, _ = train.build_model(args)
    train.load_reference_checkpoint(model, os.path.join(folder, "checkpoint.pth.tar"))
    sd = model.state_dict()
    assert set("module." + k for k in sd) == set(ck["state_dict"])
    for k, v in sd.items():
        assert torch.equal(v.cpu(), ck["state_dict"]["module." + k]), k
    assert model.policy_net.temperature == ck["temperature"]
    # resume: the saved stage is 'finetune' at epoch 1 of 1 -> nothing left to train, state restored
    res2 = train.main(ARGS + ["--logdir", str(tmp_path), "--auto_resume"], log=lines.append)
    assert res2["history"] == [] and abs(res2["temperature"] - res["temperature"]) < 1e-9


def test_multiprocessing_distributed_spawns_its_own_ranks(tmp_path, monkeypatch):
    """`--multiprocessing-distributed` (train_adamml.py:52-63): the launcher itself starts one process per GPU and joins them over
    --dist-url.  One MI355X here, so two ranks share it over gloo (ADAMML_SPAWN_RANKS / --dist-backend gloo are the test aids);
    each rank trains on its half of the global batch with SyncBatchNorm, rank 0 writes the reference-format checkpoints."""
    import socket
    from adamml_amd import train
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    monkeypatch.setenv("ADAMML_SPAWN_RANKS", "2")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    argv = ["--multiprocessing-distributed", "--backbone_net", "adamml", "-d", "50", "--groups", "8", "--num_segments", "2", "--modality", "rgb",
            "sound", "--causality_modeling", "lstm", "--learnable_lf_weights", "-b", "2", "-j", "4", "--input_size", "64", "--epochs", "1",
            "--warmup_epochs", "0", "--finetune_epochs", "0", "--val_num_clips", "2", "--cost_weights", "1.0", "0.05", "--synthetic", "1",
            "--sync-bn", "--dist-backend", "gloo", "--dist-url", "tcp://127.0.0.1:%d" % port, "--logdir", str(tmp_path)]
    assert train.main(argv) is None
    folder = os.path.join(str(tmp_path), os.listdir(str(tmp_path))[0])
    ck = torch.load(os.path.join(folder, "checkpoint_main_01.pth.tar"), map_location="cpu")
    assert ck["stage"] == "alternative_training" and ck["epoch"] == 1
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())


def test_eval_after_training_steps_sees_the_updated_weights_and_statistics():
    """train -> eval -> train -> eval in one process: the fused optimizers and adamml_bn_finalize write gamma / beta and the
    running statistics through raw pointers, which torch's `_version` counters do not see; the eval-mode BatchNorm affine
    cache must still be refreshed (validation after every epoch, utils/utils.py:436).  Reference = a FRESH model loaded with
    the trained state_dict (no cache to go stale): logits bit-identical."""
    import torch.nn.functional as F
    from adamml_amd import synth
    from adamml_amd.backbone import FlatBuffers
    from adamml_amd.optim import FlatSGD
    from adamml_amd.resnet import resnet

    def build():
        m = resnet(depth=50, num_classes=31, without_t_stride=False, groups=8, dropout=0.0, pooling_method="max",
                   input_channels=3, imagenet_pretrained=False)
        return m
    model = build()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=1234))
    model.to("cuda")
    x = synth.synth_inputs(["rgb"], 2, 1, 8, 64, seed=3)[0].to("cuda")
    y = synth.synth_labels(2, 31, seed=3).to("cuda")
    opt = None
    evals = []
    for rnd in range(2):
        model.eval()
        with torch.no_grad():
            evals.append(model(x).clone())                                   # populates the eval-mode cache
        model.train()
        F.cross_entropy(model(x), y).backward()
        if opt is None:
            opt = FlatSGD(model.flat_owner, lr=0.05, momentum=0.9)
        opt.step()
        opt.zero_grad()
    model.eval()
    with torch.no_grad():
        got = model(x)
    fresh = build()
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    fresh.to("cuda").eval()
    with torch.no_grad():
        want = fresh(x)
    assert not torch.equal(evals[0], evals[1]) and not torch.equal(evals[1], got)      # every round changed the function
    assert torch.equal(got, want), (got - want).abs().max().item()
