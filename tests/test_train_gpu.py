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
    assert abs(res["temperature"] - 5.0 * 0.965) < 1e-9                      # one decay after the alternating epoch
    folder = res["log_folder"]
    for f in ("checkpoint.pth.tar", "checkpoint_warmup_01.pth.tar", "checkpoint_main_01.pth.tar", "checkpoint_finetune_01.pth.tar"):
        assert os.path.exists(os.path.join(folder, f)), f
    ck = torch.load(os.path.join(folder, "checkpoint.pth.tar"), map_location="cpu")
    assert ck["stage"] == "finetune" and ck["epoch"] == 1 and abs(ck["temperature"] - res["temperature"]) < 1e-9
    assert all(k.startswith("module.") for k in ck["state_dict"])           # the reference's DDP-prefixed names
    assert "module.policy_net.lstm.weight_ih" in ck["state_dict"] and "module.main_net.nets.0.layer4.2.conv3.weight" in ck["state_dict"]
    # interchange: load it into a fresh model through the reference-checkpoint path, names and values identical
    args = train.arg_parser().parse_args(ARGS)
    args.input_channels = [3, 1]
    args.imagenet_pretrained = False
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
