"""Restated `train_adamml` launcher on the HIP hot path.

Every command-line flag of the reference (opts.py:5-149; the README command lines parse verbatim -- tests/test_host_cpu.py --
and `--multiprocessing-distributed` spawns one process per GPU as train_adamml.py:52-63 does), the same
three-stage schedule (train_adamml.py:340-626: warm-up of the main nets with the policy frozen -> alternating main / policy
epochs with Gumbel-temperature decay -> fine-tuning of the main nets from the best checkpoint), two optimizers
(SGD-momentum for the main nets, Adam for the policy: train_adamml.py:250-257) with the reference's learning-rate
schedules, and the reference's checkpoint dictionary (`state_dict` with the DDP `module.` prefix, `stage`, `temperature`,
`epoch`, `best_top1`, ...), so checkpoints (weights, optimizer state in torch.optim's
per-parameter layout, scheduler epoch, stage, temperature) interchange with the reference in both directions.

The dataset / decoding / augmentation pipeline of the reference is out of scope of this repository (SURVEY.md section 8:
CPU-side I/O): pass your own iterables of `(list_of_modal_tensors, target)` to `main(train_loader=..., val_loader=...)`, name a
factory with `--loader_factory MODULE:FUNCTION` (called in every rank with (args, rank, world, device); `--datadir`, `-j`, ... are in
`args`), or use `--synthetic N` (N synthetic batches per epoch, adamml_amd.synth) to exercise the whole schedule.

    python -m adamml_amd.train --backbone_net adamml -d 50 --groups 8 --num_segments 5 --modality rgb sound \\
        --causality_modeling lstm --learnable_lf_weights -b 72 --epochs 20 --warmup_epochs 5 --finetune_epochs 10 \\
        --cost_weights 1.0 0.05 --sync-bn --synthetic 100
"""
import argparse
import math
import os
import shutil
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import synth
from .distributed import HipDDP
from .model_builder import build_model
from .optim import FlatAdam, FlatSGD

CHANNELS = {"rgb": 3, "flow": 10, "rgbdiff": 15, "sound": 1}


# number of classes per dataset (utils/dataset_config.py:18-28 -- the reference ships this one entry; --num_classes overrides)
DATASET_NUM_CLASSES = {"kinetics-sounds": 31}

# opts.py flags that configure the reference's CPU-side data pipeline, cuDNN or its prediction-ensembling tools: accepted so that the
# README command lines run unchanged, reported once on rank 0 when set, without effect on the HIP hot path
INERT_FLAGS = ("frames_per_group", "workers", "threed_data", "disable_scaleup", "random_sampling", "dense_sampling", "augmentor_ver",
               "scale_range", "mean", "std", "skip_normalization", "fps", "audio_length", "resampling_rate", "num_crops", "num_clips",
               "pred_files", "pred_weights", "after_softmax", "cudnn_benchmark", "lazy_eval")


def arg_parser():
    """Every flag of the reference's parser (opts.py:5-149), same spelling, type and default -- except `--backbone_net` (the
    reference's default 's3d' is not among its own choices, opts.py:9-10: 'adamml' here), `-d` (18 there; the HIP path implements
    the Bottleneck depths, 50 here) and `--dataset` (the reference's default 'activitynet' is absent from its own table:
    'kinetics-sounds' here).  Three flags are additions: --synthetic, --loader_factory, --num_classes."""
    p = argparse.ArgumentParser(description="AdaMML training on MI355X (restated reference launcher)")
    # model definition (opts.py:8-35)
    p.add_argument("--backbone_net", default="adamml", type=str, choices=["adamml", "resnet", "sound_mobilenet_v2"])
    p.add_argument("-d", "--depth", default=50, type=int, choices=[18, 34, 50, 101, 152])
    p.add_argument("--dropout", default=0.5, type=float)
    p.add_argument("--groups", default=8, type=int, help="number of frames")
    p.add_argument("--num_segments", default=1, type=int)
    p.add_argument("--frames_per_group", default=1, type=int)
    p.add_argument("--without_t_stride", dest="without_t_stride", action="store_true")
    p.add_argument("--pooling_method", default="max", choices=["avg", "max"])
    p.add_argument("--fusion_point", default="logits", type=str, choices=["fc2", "logits"])
    p.add_argument("--prefix", default="", type=str)
    p.add_argument("--learnable_lf_weights", action="store_true")
    p.add_argument("--causality_modeling", default=None, type=str, choices=[None, "lstm"])
    p.add_argument("--cost_weights", default=None, type=float, nargs="+")
    p.add_argument("--rng_policy", action="store_true")
    p.add_argument("--rng_threshold", type=float, default=0.5)
    p.add_argument("--gammas", default=10.0, type=float)
    p.add_argument("--penalty_type", default="blockdrop", type=str, choices=["mean", "blockdrop"])
    # training setting (opts.py:37-78)
    p.add_argument("--gpu", default=None, help="GPU index of this process (set by the launcher under --multiprocessing-distributed)")
    p.add_argument("--gpu_id", default=None, help="comma separated list of GPU(s) to use (exported as HIP/CUDA_VISIBLE_DEVICES)")
    p.add_argument("--disable_cudnn_benchmark", dest="cudnn_benchmark", action="store_false")
    p.add_argument("-b", "--batch-size", default=72, type=int, help="GLOBAL batch, split over the ranks (train_adamml.py:122)")
    p.add_argument("--lr", "--learning-rate", default=0.01, type=float)
    p.add_argument("--p_lr", "--p_learning-rate", default=0.01, type=float)
    p.add_argument("--lr_scheduler", default="cosine", type=str, choices=["step", "multisteps", "cosine", "plateau"])
    p.add_argument("--lr_steps", default=[15, 30, 45], type=float, nargs="+")
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--nesterov", action="store_true")
    p.add_argument("--weight-decay", "--wd", default=1e-4, type=float)
    p.add_argument("--epochs", default=50, type=int)
    p.add_argument("--warmup_epochs", default=5, type=int)
    p.add_argument("--finetune_epochs", default=10, type=int)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--auto_resume", action="store_true")
    p.add_argument("--pretrained", dest="pretrained", type=str, default=None)
    p.add_argument("--unimodality_pretrained", type=str, nargs="+", default=[])
    p.add_argument("--start-epoch", default=0, type=int)
    p.add_argument("--clip_gradient", "--cg", default=None, type=float)
    p.add_argument("--curr_stage", type=str, default="warmup", choices=["warmup", "alternative_training", "finetune"])
    # data-related (opts.py:80-113)
    p.add_argument("-j", "--workers", default=18, type=int)
    p.add_argument("--datadir", metavar="DIR", nargs="+", type=str, default=None, help="one path per modality")
    p.add_argument("--dataset", default="kinetics-sounds", type=str)
    p.add_argument("--threed_data", action="store_true")
    p.add_argument("--input_size", default=224, type=int)
    p.add_argument("--disable_scaleup", action="store_true")
    p.add_argument("--random_sampling", action="store_true")
    p.add_argument("--dense_sampling", action="store_true")
    p.add_argument("--augmentor_ver", default="v2", type=str, choices=["v1", "v2"])
    p.add_argument("--scale_range", default=[256, 320], type=int, nargs="+")
    p.add_argument("--modality", default=["rgb"], type=str, nargs="+", help="rgb | flow | rgbdiff | sound (checked in resolve_args, so "
                   "that the README's template lines with their MODALITY1 MODALITY2 placeholders still parse)")
    p.add_argument("--mean", type=float, nargs="+")
    p.add_argument("--std", type=float, nargs="+")
    p.add_argument("--skip_normalization", action="store_true")
    p.add_argument("--fps", type=float, default=29.97)
    p.add_argument("--audio_length", type=float, default=1.28)
    p.add_argument("--resampling_rate", type=float, default=24000)
    # logging (opts.py:115-119)
    p.add_argument("--logdir", default="", type=str)
    p.add_argument("--print-freq", default=100, type=int)
    p.add_argument("--show_model", action="store_true")
    # testing and validation (opts.py:121-133)
    p.add_argument("-e", "--evaluate", dest="evaluate", action="store_true")
    p.add_argument("--num_crops", default=1, type=int, choices=[1, 3, 5, 10])
    p.add_argument("--num_clips", default=1, type=int)
    p.add_argument("--val_num_clips", default=10, type=int)
    p.add_argument("--pred_files", type=str, nargs="+")
    p.add_argument("--pred_weights", type=float, nargs="+")
    p.add_argument("--after_softmax", action="store_true")
    p.add_argument("--lazy_eval", action="store_true", help="validate every 10 epochs and in the last 10 % of a stage's epochs")
    # distributed (opts.py:135-148)
    p.add_argument("--sync-bn", action="store_true")
    p.add_argument("--world-size", default=1, type=int, help="number of NODES")
    p.add_argument("--rank", default=0, type=int, help="NODE rank")
    p.add_argument("--dist-url", default="tcp://127.0.0.1:23456", type=str)
    p.add_argument("--hostfile", default="", type=str)
    p.add_argument("--dist-backend", default="nccl", type=str, help="'nccl' is RCCL on ROCm")
    p.add_argument("--multiprocessing-distributed", action="store_true",
                   help="spawn one process per visible GPU on this node (train_adamml.py:52-63)")
    # additions
    p.add_argument("--synthetic", default=0, type=int, metavar="N", help="N synthetic batches per epoch instead of a dataset")
    p.add_argument("--loader_factory", default=None, type=str, metavar="MODULE:FUNCTION",
                   help="data pipeline hook: FUNCTION(args, rank, world, device) -> (train_loader, val_loader), imported in every rank")
    p.add_argument("--num_classes", default=None, type=int, help="overrides the class count of --dataset")
    p.add_argument("--imagenet_weights", default=[], nargs="+", metavar="ARCH=PATH",
                   help="local torchvision ImageNet state_dict files the reference would download (models/resnet.py:251-257, "
                        "models/policy_net.py:193-203,221): resnet50=PATH mobilenet_v2=PATH mobilenetv2_160x160=PATH; also $ADAMML_IMAGENET_DIR")
    return p


def resolve_args(args, log=print, rank=0):
    """The bookkeeping train_adamml.py:66-95 does on the parsed namespace before building the model."""
    if args.num_classes is None:
        if args.dataset not in DATASET_NUM_CLASSES:
            raise SystemExit("unknown --dataset %r (known: %s): pass --num_classes" % (args.dataset, ", ".join(DATASET_NUM_CLASSES)))
        args.num_classes = DATASET_NUM_CLASSES[args.dataset]                 # train_adamml.py:70-71
    bad = [m for m in args.modality if m not in CHANNELS]
    if bad:
        raise SystemExit("--modality: invalid choice(s) %s (choose from %s)" % (", ".join(map(repr, bad)), ", ".join(CHANNELS)))
    args.input_channels = [CHANNELS[m] for m in args.modality]              # train_adamml.py:85-95
    # the reference's factories default to imagenet_pretrained=True and download; here the same files are read from disk when the caller
    # names them (imagenet_init.py), with the reference's channel conversion -- without them the initialisation stays random, quietly
    from . import imagenet_init
    imagenet_init.configure_from_args(getattr(args, "imagenet_weights", None))
    args.imagenet_pretrained = any(imagenet_init.path_for(a) for a in ("resnet%d" % args.depth, "mobilenet_v2", "mobilenetv2_160x160"))
    if args.datadir is not None and len(args.datadir) not in (1, len(args.modality)):
        raise SystemExit("--datadir takes one path per modality (%d given for %d modalities)" % (len(args.datadir), len(args.modality)))
    if rank == 0:
        defaults = arg_parser().parse_args([])
        inert = [k for k in INERT_FLAGS if getattr(args, k) != getattr(defaults, k)]
        if inert:
            log("flags accepted for command-line compatibility, without effect on the HIP path: " + ", ".join("--" + k for k in inert))
    return args


# ------------------------------------------------------------------------------------------------ helpers
def compute_policy_loss(penalty_type, selection, cost_weights, gammas, cls_logits, cls_targets):
    """utils/utils.py:166-184."""
    num_modality = selection.shape[-1]
    loss = torch.zeros((), dtype=selection.dtype, device=selection.device)
    if penalty_type == "mean":
        for w, pl in zip(cost_weights, selection.chunk(num_modality, dim=-1)):
            loss = loss + w * pl.mean()
    elif penalty_type == "blockdrop":
        correct = (cls_logits.detach().argmax(-1) == cls_targets).type_as(cls_logits)
        sel = selection.mean(dim=1) ** 2
        for w, pl in zip(cost_weights, sel.chunk(num_modality, dim=-1)):
            # pl is [N,1], correct is [N]: the product broadcasts to [N,N] exactly as in the reference (utils/utils.py:179)
            loss = loss + w * (correct * pl).mean()
        loss = loss + ((1.0 - correct) * gammas).mean()
    return loss


def accuracy(output, target, topk=(1, 5)):
    maxk = min(max(topk), output.shape[1])
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:min(k, maxk)].reshape(-1).float().sum() * (100.0 / target.size(0)) for k in topk]


class Meter:
    def __init__(self):
        self.sum, self.n = 0.0, 0

    def update(self, v, n=1):
        self.sum += float(v) * n
        self.n += n

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


class LRSchedule:
    """StepLR / MultiStepLR / CosineAnnealingLR / ReduceLROnPlateau('min') (train_adamml.py:259-270) for the flat optimizers;
    step(epoch) sets the learning rate of epoch `epoch` in closed form, as the reference's schedulers do when stepped with an
    epoch; the plateau kind is stepped with the validation loss instead (train_adamml.py:460-462) and follows torch's
    defaults (factor 0.1, patience 10, relative threshold 1e-4, no cooldown)."""

    def __init__(self, opt, kind, base_lr, epochs, steps):
        self.opt, self.kind, self.base, self.epochs, self.steps, self.last = opt, kind, base_lr, epochs, [int(s) for s in steps], 0
        self.best, self.bad = float("inf"), 0

    def step(self, epoch, metric=None):
        if self.kind == "plateau":
            self.last += 1
            if metric is not None and metric < self.best * (1.0 - 1e-4):
                self.best, self.bad = float(metric), 0
            else:
                self.bad += 1
            if self.bad > 10:
                self.opt.lr = self.opt.lr * 0.1
                self.bad = 0
            return
        self.last = epoch
        if self.kind == "step":
            lr = self.base * 0.1 ** (epoch // self.steps[0])
        elif self.kind == "multisteps":
            lr = self.base * 0.1 ** sum(1 for s in self.steps if epoch >= s)
        else:
            lr = 0.5 * self.base * (1 + math.cos(math.pi * epoch / max(self.epochs, 1)))
        self.opt.lr = lr

    def state_dict(self):
        """The keys torch's StepLR / MultiStepLR / CosineAnnealingLR restore from (`__dict__.update`, train_adamml.py:300-301)."""
        if self.kind == "plateau":
            return {"last_epoch": self.last, "best": self.best, "num_bad_epochs": self.bad, "_last_lr": [self.opt.lr]}
        return {"last_epoch": self.last, "base_lrs": [self.base], "_last_lr": [self.opt.lr], "_step_count": self.last + 1}

    def load_state_dict(self, sd):
        if self.kind == "plateau":
            self.last, self.best, self.bad = sd.get("last_epoch", 0), sd.get("best", float("inf")), sd.get("num_bad_epochs", 0)
            if sd.get("_last_lr"):
                self.opt.lr = sd["_last_lr"][0]
            return
        self.step(sd.get("last_epoch", 0))


def strip_module_prefix(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_reference_checkpoint(model, path_or_dict, strict=True):
    """Load a reference checkpoint (`{'state_dict': {'module.<name>': tensor}}`, a bare state_dict, with or without the
    DDP prefix) into an adamml_amd model; returns the checkpoint dict.  (weights stay OIHW fp32
    at this boundary, the bf16 GEMM packs are rebuilt on the next forward.)"""
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    target = model.module if hasattr(model, "module") else model
    with torch.no_grad():
        target.load_state_dict(strip_module_prefix(sd), strict=strict)
    for m in target.modules():
        if hasattr(m, "mark_weights_dirty"):
            m.mark_weights_dirty()
    if isinstance(ck, dict) and "temperature" in ck and hasattr(target, "policy_net"):
        target.policy_net.set_temperature(ck["temperature"])
    return ck if isinstance(ck, dict) else {"state_dict": sd}


def reference_state_dict(model):
    """state_dict with the `module.` prefix the reference's DDP-wrapped checkpoints carry (train_adamml.py:386)."""
    target = model.module if hasattr(model, "module") else model
    return {"module." + k: v.detach().cpu() for k, v in target.state_dict().items()}


def save_checkpoint(state, is_best, filepath, epoch=None, suffix=""):
    """utils/utils.py:89-96."""
    cur = os.path.join(filepath, "checkpoint.pth.tar")
    torch.save(state, cur)
    if epoch:
        shutil.copyfile(cur, os.path.join(filepath, "checkpoint{}_{:02d}.pth.tar".format(suffix, epoch)))
    if is_best:
        shutil.copyfile(cur, os.path.join(filepath, "model_best.pth.tar"))


class SyntheticLoader:
    """N deterministic synthetic batches per epoch (per-rank batch), already on the device."""

    def __init__(self, args, n, batch, device, rank, segments=None):
        self.args, self.n, self.batch, self.device, self.rank = args, n, batch, device, rank
        self.segments = segments or args.num_segments

    def __len__(self):
        return self.n

    def __iter__(self):
        a = self.args
        for i in range(self.n):
            seed = 42 + i + 1000 * self.rank
            xs = synth.synth_inputs(a.modality, self.batch, self.segments, a.groups, a.input_size, seed=seed)
            y = synth.synth_labels(self.batch, a.num_classes, seed=seed)
            yield [x.to(self.device) for x in xs], y.to(self.device)


def concat_all_gather(tensor, group=None):
    """utils/utils.py:539-550: all-gather a per-rank tensor and concatenate along dim 0 in rank order (no gradient)."""
    lst = [torch.empty_like(tensor) for _ in range(dist.get_world_size(group))]
    dist.all_gather(lst, tensor.contiguous(), group=group)
    return torch.cat(lst, dim=0)


# ------------------------------------------------------------------------------------------------ one epoch
def train_epoch(loader, ddp, opt, p_opt, epoch, args, cost_weights, rank=0, log=print):
    """utils/utils.py:320-426."""
    model = ddp.module
    model.train()
    opt.zero_grad()
    p_opt.zero_grad()
    device = next(model.parameters()).device
    cw = torch.tensor(cost_weights if cost_weights is not None else [0.0] * len(args.modality), device=device)
    gammas = torch.tensor(args.gammas, device=device)
    meters = {k: Meter() for k in ("loss", "top1", "top5", "time")}
    sel_meter = {m: Meter() for m in args.modality}
    end = time.time()
    for i, (images, target) in enumerate(loader):
        images = [x.to(device, non_blocking=True) for x in images]
        target = target.to(device, non_blocking=True)
        output, selection = ddp(images)
        loss = F.cross_entropy(output, target)
        if model.update_policy_net:
            loss = loss + compute_policy_loss(args.penalty_type, selection, cw, gammas, output, target)
        prec1, prec5 = accuracy(output, target)
        ratio = selection.detach().mean(0).mean(0)
        if dist.is_initialized():
            # the reference's three metric all-reduces (utils/utils.py:371-376) as ONE packed [2 + M] vector
            packed = torch.cat([prec1.reshape(1), prec5.reshape(1), ratio.reshape(-1).to(prec1.dtype)])
            dist.all_reduce(packed)
            packed /= dist.get_world_size()
            prec1, prec5, ratio = packed[0], packed[1], packed[2:]
        loss.backward()
        ddp.reduce_gradients()
        if args.clip_gradient is not None:
            bufs = model.flat_grad_buffers()
            total = torch.sqrt(sum((b.float() ** 2).sum() for b in bufs))
            scale = (args.clip_gradient / (total + 1e-6)).clamp(max=1.0)
            for b in bufs:
                b.mul_(scale)
        if model.update_policy_net:
            p_opt.step()
            p_opt.zero_grad()
        if model.update_main_net:
            opt.step()
            opt.zero_grad()
        meters["loss"].update(loss.item(), target.size(0))
        meters["top1"].update(prec1.item(), target.size(0))
        meters["top5"].update(prec5.item(), target.size(0))
        for ii, m in enumerate(args.modality[:ratio.numel()]):
            sel_meter[m].update(ratio[ii].item())
        meters["time"].update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0 and rank == 0:
            log("Epoch: [{}][{}/{}]\tTime {:.3f}\tLoss {:.4f}\tPrec@1 {:.3f}\tPrec@5 {:.3f}\t{}".format(
                epoch, i, len(loader), meters["time"].avg, meters["loss"].avg, meters["top1"].avg, meters["top5"].avg,
                " ".join("{}:{:.2f}".format(k, v.avg * 100) for k, v in sel_meter.items())))
    return meters, sel_meter


@torch.no_grad()
def validate(loader, ddp, args, num_segments):
    """utils/utils.py:427-507 (top-k on the concatenated outputs; mAP needs torchnet in the reference and is not restated)."""
    model = ddp.module
    model.eval()
    device = next(model.parameters()).device
    outs, labels, sels = [], [], []
    loss_m = Meter()
    for images, target in loader:
        images = [x.to(device, non_blocking=True) for x in images]
        target = target.to(device, non_blocking=True)
        output, selection = model(images, num_segments)
        loss_m.update(F.cross_entropy(output, target).item(), target.size(0))
        outs.append(output)
        labels.append(target)
        sels.append(selection)
    output, target, selection = torch.cat(outs), torch.cat(labels), torch.cat(sels)
    if dist.is_initialized():
        output, target, selection = concat_all_gather(output), concat_all_gather(target), concat_all_gather(selection)
    top1, top5 = accuracy(output, target)
    return top1.item(), top5.item(), loss_m.avg, selection


# ------------------------------------------------------------------------------------------------ schedule
def main(argv=None, train_loader=None, val_loader=None, log=print):
    """train_adamml.py:32-63: parse, resolve the node topology, then either run this process as ONE rank (plain call, or a rank
    started by torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, with --multiprocessing-distributed,
    spawn one process per visible GPU and run `main_worker` in each."""
    args = arg_parser().parse_args(argv)
    if args.gpu_id:                                                          # train_adamml.py:37-38
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu_id
    if args.hostfile != "":                                                  # train_adamml.py:40-50
        import platform
        node = platform.node().split(".")[0]
        with open(args.hostfile) as f:
            nodes = [x.strip() for x in f.readlines() if x.strip() != ""]
        for idx, line in enumerate(nodes):
            if node in line:
                args.rank = idx
                break
        args.world_size = len(nodes)
        args.dist_url = "tcp://{}:10598".format(nodes[0].split(" ")[0])
    if not torch.cuda.is_available():
        raise SystemExit("adamml_amd.train needs an MI355X: the HIP hot path has no CPU fallback")
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.multiprocessing_distributed and not under_launcher:
        ngpus = int(os.environ.get("ADAMML_SPAWN_RANKS", 0)) or torch.cuda.device_count()     # (test aid: ranks sharing one GPU)
        if train_loader is not None or val_loader is not None:
            raise SystemExit("--multiprocessing-distributed starts fresh processes: loader objects cannot follow them; "
                             "name a factory with --loader_factory MODULE:FUNCTION (or use --synthetic N)")
        args.world_size = ngpus * args.world_size                            # train_adamml.py:55-57
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL needs it on this driver
        import torch.multiprocessing as mp
        mp.spawn(_spawned_worker, nprocs=ngpus, args=(ngpus, args))
        return None
    return main_worker(args.gpu, torch.cuda.device_count(), args, train_loader, val_loader, log)


def _spawned_worker(gpu, ngpus_per_node, args):
    main_worker(gpu, ngpus_per_node, args, None, None, print)


def _load_factory(spec):
    import importlib
    mod, _, fn = spec.partition(":")
    if not fn:
        raise SystemExit("--loader_factory expects MODULE:FUNCTION, got %r" % spec)
    return getattr(importlib.import_module(mod), fn)


def main_worker(gpu, ngpus_per_node, args, train_loader=None, val_loader=None, log=print):
    """train_adamml.py:66-629 for one rank."""
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:                  # started by torch.distributed.run
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        init = dict(rank=rank, world_size=world)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    elif args.multiprocessing_distributed:                                   # spawned above: global rank = node rank * GPUs + gpu
        world, rank, local_rank = args.world_size, args.rank * ngpus_per_node + int(gpu), int(gpu)
        init = dict(init_method=args.dist_url, rank=rank, world_size=world)
    else:                                                                    # one process (--gpu N or the first GPU); several nodes
        world, rank, local_rank = args.world_size, args.rank, int(gpu) if gpu is not None else 0      # with one process each also land here
        init = dict(init_method=args.dist_url, rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ADAMML_DIST_BACKEND", args.dist_backend)
    if backend != "nccl":
        local_rank = local_rank % max(ndev, 1)                               # (gloo test aid: several ranks on one GPU)
    elif local_rank >= ndev:
        raise SystemExit("rank %d wants GPU %d but only %d are visible (RCCL needs one device per rank)" % (rank, local_rank, ndev))
    args.gpu, args.rank, args.world_size = local_rank, rank, world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, **init)                             # train_adamml.py:83
    args.distributed = world > 1
    resolve_args(args, log, rank)
    per_rank_batch = max(1, args.batch_size // world)                       # train_adamml.py:122
    if "ADAMML_LAUNCH_PLAN" not in os.environ and per_rank_batch <= 16:
        # small per-GPU batches (the README recipe: 72 videos over 8 GPUs) are bound by the Python issue of ~1300 launches per step:
        # replay the static launch sequences from C (adamml_amd/plan.py; costs the sum instead of the peak of a step's activations)
        from . import plan as _plan
        _plan.ENABLED = True
    if args.backbone_net != "adamml":
        raise SystemExit("adamml_amd.train restates train_adamml.py; unimodal training is models.resnet / sound_mobilenet_v2 "
                         "behind the same registry (build_model) with a plain loop")

    model, arch_name = build_model(args)
    model = model.to(device)
    if args.show_model and rank == 0:                                        # train_adamml.py:107-109
        log(str(model))
        return 0
    if args.pretrained:
        load_reference_checkpoint(model, args.pretrained, strict=False)
    ddp = HipDDP(model, sync_bn=(args.sync_bn and world > 1))
    log_folder = os.path.join(args.logdir or "snapshots", arch_name)
    if rank == 0:
        os.makedirs(log_folder, exist_ok=True)

    if train_loader is None:
        if args.loader_factory:
            train_loader, val_loader = _load_factory(args.loader_factory)(args, rank, world, device)
        elif args.synthetic:
            train_loader = SyntheticLoader(args, args.synthetic, per_rank_batch, device, rank)
            val_loader = SyntheticLoader(args, max(1, args.synthetic // 4), per_rank_batch, device, rank + 7777, args.val_num_clips)
        else:
            raise SystemExit("no dataset pipeline in this repository (SURVEY.md section 8: CPU-side I/O is out of scope): pass "
                             "train_loader / val_loader to main(), name a factory with --loader_factory MODULE:FUNCTION, or use "
                             "--synthetic N" + ("; --datadir %s was parsed and is handed to the factory" % args.datadir if args.datadir else ""))

    def make_optimizers():
        o = FlatSGD(model._flat_main, args.lr, args.momentum, args.weight_decay, args.nesterov)
        po = FlatAdam(model._flat_policy, args.p_lr, weight_decay=args.weight_decay)
        return o, po, LRSchedule(o, args.lr_scheduler, args.lr, args.epochs, args.lr_steps), \
            LRSchedule(po, args.lr_scheduler, args.p_lr, args.epochs, args.lr_steps)

    # flat parameter buffers must exist before the optimizers look at them
    model._flat_policy.ensure(device)
    model._flat_main.ensure(device)
    opt, p_opt, sched, p_sched = make_optimizers()
    best_top1, stage = 0.0, args.curr_stage
    if args.auto_resume and os.path.exists(os.path.join(log_folder, "checkpoint.pth.tar")):
        args.resume = os.path.join(log_folder, "checkpoint.pth.tar")
    if args.resume:
        ck = load_reference_checkpoint(model, args.resume)
        args.start_epoch, best_top1, stage = ck.get("epoch", 0), float(ck.get("best_top1", 0.0)), ck.get("stage", stage)
        sched.load_state_dict(ck.get("scheduler", {}))
        p_sched.load_state_dict(ck.get("p_scheduler", {}))
        for o, key in ((opt, "optimizer"), (p_opt, "p_optimizer")):
            if isinstance(ck.get(key), dict) and "param_groups" in ck[key]:
                o.load_state_dict(ck[key])                                  # torch.optim layout: the reference's checkpoints too
            elif rank == 0:
                log("resume: no usable '%s' state in the checkpoint, optimizer state starts from zero" % key)
    # torch's DistributedDataParallel broadcasts rank 0's parameters and buffers at construction (train_adamml.py:129)
    ddp.broadcast_parameters()

    def snapshot(epoch, st, is_best, suffix):
        """Rank 0 writes; every rank leaves only when the files are complete (the reference brackets validation and saving
        with dist.barrier(), train_adamml.py:355,421,453,468) -- the fine-tune stage reads model_best.pth.tar on all ranks."""
        if rank == 0:
            save_checkpoint({"epoch": epoch, "arch": arch_name, "state_dict": reference_state_dict(model), "best_top1": best_top1,
                             "p_optimizer": p_opt.state_dict(), "optimizer": opt.state_dict(), "p_scheduler": p_sched.state_dict(),
                             "scheduler": sched.state_dict(), "temperature": model.policy_net.temperature, "stage": st},
                            is_best, log_folder, epoch, suffix)
        if dist.is_initialized():
            dist.barrier()

    zero_cost = [0.0] * len(args.modality)
    if args.evaluate:
        top1, top5, loss, _ = validate(val_loader, ddp, args, args.val_num_clips)
        log("Val: Loss {:.4f}\tTop@1 {:.4f}\tTop@5 {:.4f}".format(loss, top1, top5))
        return {"top1": top1, "top5": top5, "loss": loss}

    history = []
    if stage == "warmup":                                                   # train_adamml.py:340-392
        if args.warmup_epochs > 0 and rank == 0:
            log("Stage [Warming up]: Main network with {} epochs".format(args.warmup_epochs))
        model.freeze_policy_net()
        model.unfreeze_main_net()
        for epoch in range(args.start_epoch, args.warmup_epochs):
            m, _ = train_epoch(train_loader, ddp, opt, p_opt, epoch + 1, args, zero_cost, rank, log)
            history.append(("warmup", epoch + 1, m["loss"].avg))
            snapshot(epoch + 1, "warmup", False, "_warmup")
        stage, args.start_epoch = "alternative_training", 0
        opt, p_opt, sched, p_sched = make_optimizers()
    if stage == "alternative_training":                                     # train_adamml.py:394-519
        if rank == 0:
            log("Stage [Alternative training]: {} epochs".format(args.epochs))
        for epoch in range(args.start_epoch, args.epochs):
            model.freeze_policy_net()
            model.unfreeze_main_net()
            m, _ = train_epoch(train_loader, ddp, opt, p_opt, epoch + 1, args, zero_cost, rank, log)
            history.append(("main", epoch + 1, m["loss"].avg))
            model.unfreeze_policy_net()
            model.freeze_main_net()
            m, _ = train_epoch(train_loader, ddp, opt, p_opt, epoch + 1, args, args.cost_weights, rank, log)
            history.append(("policy", epoch + 1, m["loss"].avg))
            top1, top5, vloss, _ = validate(val_loader, ddp, args, args.val_num_clips)
            sched.step(epoch + 1, vloss)
            p_sched.step(epoch + 1, vloss)
            is_best = top1 > best_top1
            best_top1 = max(top1, best_top1)
            if rank == 0:
                log("Val: [{:03d}/{:03d}]\tLoss {:.4f}\tTop@1 {:.4f}\tTop@5 {:.4f}".format(epoch + 1, args.epochs, vloss, top1, top5))
            snapshot(epoch + 1, "alternative_training", is_best, "_main")
            model.decay_temperature()
        stage, args.start_epoch = "finetune", 0
        opt, p_opt, sched, p_sched = make_optimizers()
    if stage == "finetune" and args.finetune_epochs > 0:                    # train_adamml.py:521-620
        if rank == 0:
            log("Stage [Post finetuning]: Finetune the main network {} epochs".format(args.finetune_epochs))
        best = os.path.join(log_folder, "model_best.pth.tar")
        if dist.is_initialized():
            dist.barrier()                                                  # rank 0's last snapshot is on disk
        # rank 0 decides whether a best model exists and reads it; every rank then takes part in the SAME collectives (a per-rank
        # os.path.exists over a non-shared or stale file system would let some ranks skip the broadcast and hang the others)
        flag = torch.zeros(2, dtype=torch.float64, device=device)
        if rank == 0 and args.start_epoch == 0 and os.path.exists(best):
            load_reference_checkpoint(model, best)
            flag[0], flag[1] = 1.0, float(model.policy_net.temperature)
        if dist.is_initialized():
            dist.broadcast(flag, 0)
        if flag[0].item() > 0:
            model.policy_net.set_temperature(float(flag[1].item()))
            ddp.broadcast_parameters()
        model.freeze_policy_net()
        model.unfreeze_main_net()
        for epoch in range(args.start_epoch, args.finetune_epochs):
            m, _ = train_epoch(train_loader, ddp, opt, p_opt, epoch + 1, args, zero_cost, rank, log)
            history.append(("finetune", epoch + 1, m["loss"].avg))
            top1, top5, vloss, _ = validate(val_loader, ddp, args, args.val_num_clips)
            sched.step(epoch + 1, vloss)
            p_sched.step(epoch + 1, vloss)
            is_best = top1 > best_top1
            best_top1 = max(top1, best_top1)
            if rank == 0:
                log("Val: [{:03d}/{:03d}]\tLoss {:.4f}\tTop@1 {:.4f}\tTop@5 {:.4f}".format(epoch + 1, args.finetune_epochs, vloss, top1, top5))
            snapshot(epoch + 1, "finetune", is_best, "_finetune")
    return {"history": history, "best_top1": best_top1, "log_folder": log_folder, "temperature": model.policy_net.temperature}


if __name__ == "__main__":
    main()
