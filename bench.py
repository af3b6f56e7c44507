#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: clips/sec, train fwd+bwd, RGB+Audio AdaMML (ResNet-50 + Sound-MobileNetV2 +
MobileNetV2/LSTM policy), 224^2, 8 frames/segment, 5 segments, bf16.

One "step" = one main-net-stage training iteration (policy frozen, train_adamml.py:344-345) over one batch of
synthetic videos already resident in HBM: forward of policy + main nets, CE loss, backward of the main nets,
gradient all-reduce (N > 1), fused SGD step.  N = 1 runs BASELINE.json configs[1] (B = 72 videos = 360 clips per step);
N > 1 runs configs[2]: SyncBN + RCCL gradient all-reduce, B = 72 per GPU (`--scaling weak`, default) or the reference's
own semantics, a GLOBAL batch of 72 split over the ranks (`--scaling strong`, train_adamml.py:122).

    python bench.py --gpus N --steps K --warmup W

Launched under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every process is one rank.
Launched plainly with --gpus N > 1 it spawns the N ranks itself, one process per GPU, as the reference's launcher does
(train_adamml.py:52-63 mp.spawn).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak, same table
CLIP_GFLOP_MAIN_STAGE = 86.7   # algorithmic GFLOP per clip, main-net stage (BASELINE.md section 3 / SURVEY.md section 8d)
CLIP_GB_MAIN_STAGE = 1.12      # algorithmic GB per clip, fwd + bwd, bf16, BN/ReLU/residual fused (same sections)
CHANNELS = {"rgb": 3, "sound": 1, "flow": 10, "rgbdiff": 15}      # per frame (train_adamml.py:86-95)
# newest round's PMC summary (tools/gpu_pmc.sh); accepted only when its kernel-source stamp equals the running build
PMC_TRAFFIC_FILE = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_hbm_traffic.json"))) or
                    [os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")])[-1]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("ADAMML_BENCH_BATCH", 72)),
                    help="videos per GPU (weak scaling) / global batch split over the GPUs (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--segments", type=int, default=5)
    ap.add_argument("--stage", default="main", choices=["main", "policy", "infer"],
                    help="main (headline metric) | policy: the other training stage | infer: eval-mode forward with the main nets "
                         "run only on the clips the policy selected (non-headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sync-bn", action="store_true")
    ap.add_argument("--modalities", nargs="+", default=["rgb", "sound"], choices=["rgb", "sound", "flow", "rgbdiff"],
                    help="non-default: BASELINE.json configs[3] (rgb flow rgbdiff) / configs[4] (rgb sound flow rgbdiff)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: a ONE-rank RCCL communicator with SyncBN and the bucketed gradient all-reduce forced on (every "
                         "collective is an identity): the host / stream choreography of configs[2] measured on one GPU (non-headline)")
    ap.add_argument("--launch-plan", action="store_true",
                    help="replay the static launch sequence of every backbone call from C (adamml_amd/plan.py): for small per-GPU batches, "
                         "where the Python issue of ~1300 launches per step bounds the step (non-headline)")
    ap.add_argument("--u8-input", action="store_true",
                    help="non-headline: the visual modalities arrive as decoded uint8 frames [B, H, W, S*F*C] (what utils/video_transforms.py "
                         "Stack produces) and are normalised inside the input kernel: 1 byte per value through HBM instead of 4")
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: no side streams, so per-kernel durations "
                    "in a rocprofv3 trace are not inflated by concurrently running kernels")
    return ap.parse_args()


def source_stamp():
    """sha256 over the kernel sources + the C-ABI header: identifies the BUILD a PMC traffic file was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "adamml_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(args, device):
    from adamml_amd import adamml, synth
    mod = list(args.modalities)
    model = adamml(groups=8, modality=mod, input_channels=[CHANNELS[m] for m in mod], num_segments=args.segments, rng_policy=False,
                   rng_threshold=0.5, causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False,
                   dropout=0.5, pooling_method="max", fusion_point="logits", unimodality_pretrained=[],
                   learnable_lf_weights=True)
    sd = synth.synth_state_dict(model.state_dict(), seed=1234)       # random-init weights of the named architecture
    model.load_state_dict(sd)
    model.to(device)
    return model


def synth_batch(args, b, device, rank):
    g = torch.Generator(device="cpu").manual_seed(42 + rank)
    s = args.segments
    # generated on the device in chunks (a [72,120,224,224] fp32 clip tensor is 1.7 GB)
    torch.manual_seed(42 + rank)
    xs = []
    for m in args.modalities:
        if m == "sound":
            xs.append(torch.randn(b, s, 256, 256, device=device) * 3.0 - 5.0)
        elif args.u8_input:
            xs.append(torch.randint(0, 256, (b, 224, 224, s * 8 * CHANNELS[m]), dtype=torch.uint8, device=device))
        else:
            xs.append(torch.randn(b, s * 8 * CHANNELS[m], 224, 224, device=device))
    tgt = torch.randint(0, 31, (b,), generator=g).to(device)
    return xs, tgt


def _timed(fn, warm=1, iters=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline(args):
    """ORACLE (CPU fp32 restatement, pinned to the reference goldens) timed on this host on bounded samples of the two
    CPU-runnable workloads of SURVEY.md section 8(d), 1 warm-up + 3 timed iterations each, median:
      C2 (the metric's workload): RGB+Audio AdaMML main-net-stage fwd+bwd+SGD at B = 2 videos x 5 segments = 10 clips;
      C1 (BASELINE.json configs[0]): unimodal RGB ResNet-50, 8 frames, 224^2, b = 4 clips, fwd+bwd+SGD."""
    from adamml_amd import adamml, synth
    from adamml_amd.resnet import resnet
    from oracle import adamml_oracle as O
    ncpu = os.cpu_count() or 1
    # oneDNN convolutions of a 2-video batch stop scaling far below the 128+ hardware threads of the GPU box and then degrade
    # (measured there, C2 sample: 8 threads 2.35 clips/s, 16: 2.63, 64: 1.10, all 128+: 0.30): the baseline uses the best setting
    cores = int(os.environ.get("ADAMML_CPU_THREADS", min(ncpu, 16)))
    torch.set_num_threads(cores)
    mod = ["rgb", "sound"]
    b, s = 2, args.segments
    shapes = adamml(groups=8, modality=mod, input_channels=[3, 1], num_segments=s, rng_policy=False, rng_threshold=0.5,
                    causality_modeling="lstm", num_classes=31, depth=50, without_t_stride=False, dropout=0.5,
                    pooling_method="max", fusion_point="logits", unimodality_pretrained=[], learnable_lf_weights=True).state_dict()
    sd = O.make_leaf_state(synth.synth_state_dict(shapes, seed=1234), ("main_net.",))
    xs = synth.synth_inputs(mod, b, s, 8, 224, 256, seed=42)
    tgt = synth.synth_labels(b, 31, seed=42)
    expo = synth.synth_gumbel_exponential(s, 2, b, seed=7)

    def sgd(state):
        with torch.no_grad():
            for v in state.values():
                if v.grad is not None:
                    v -= 0.01 * v.grad
                    v.grad = None

    def c2_step():
        logits, sel, _ = O.adamml_forward(sd, xs, mod, s, 8, 50, 5.0, expo, "lstm", "max", False, 0.5, True)
        F.cross_entropy(logits, tgt).backward()
        sgd(sd)
    t2, all2 = _timed(c2_step)
    r_shapes = resnet(depth=50, num_classes=31, without_t_stride=False, groups=8, dropout=0.5, pooling_method="max", input_channels=3,
                      imagenet_pretrained=False).state_dict()
    rsd = O.make_leaf_state(synth.synth_state_dict(r_shapes, seed=1234), ("",))
    x1 = synth.synth_inputs(["rgb"], 4, 1, 8, 224, seed=42)[0]
    t1lab = synth.synth_labels(4, 31, seed=42)

    def c1_step():
        F.cross_entropy(O.resnet_forward(rsd, "", x1, 8, 50, "max", False, 0.5, True), t1lab).backward()
        sgd(rsd)
    t1, all1 = _timed(c1_step)
    return {"value": round(b * s / t2, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "oracle (CPU fp32 restatement of the reference, torch %s, %d of %d host threads) main-net-stage train step, "
                      "B=2 videos x %d segments (10 clips) per iteration, 1 warm-up + 3 timed iterations, median"
                      % (torch.__version__, cores, ncpu, s),
            "iterations_s": [round(t, 2) for t in all2],
            "c1_unimodal_resnet50_b4": {"value": round(4 / t1, 3), "unit": "clips/s", "iterations_s": [round(t, 2) for t in all1],
                                        "sample": "BASELINE.json configs[0]: RGB ResNet-50, 8 frames, 224^2, b=4, fwd+bwd+SGD, "
                                                  "1 warm-up + 3 timed iterations, median"}}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: one process per GPU (train_adamml.py:52-63)."""
    import torch.multiprocessing as mp
    backend = os.environ.get("ADAMML_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one device per rank)" % (args.gpus, ndev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    mp.spawn(_spawned_rank, args=(args.gpus, port), nprocs=args.gpus, join=True)


def _spawned_rank(rank, world, port):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    run_rank(parse())


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
        spawn_ranks(args)
        return
    run_rank(args)


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); running %d" % (args.gpus, world, world), file=sys.stderr)
    # (ADAMML_DIST_BACKEND=gloo lets the N>1 code path be exercised with several ranks on ONE GPU; default is RCCL)
    backend = os.environ.get("ADAMML_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)     # "nccl" == RCCL on ROCm
    elif args.force_collectives:
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port_ = s_.getsockname()[1]
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port_, rank=0, world_size=1)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB):
        ge.build()
    comm_ranks = 1
    if world > 1:
        # first collective: proves the communicator spans `world` ranks (every rank contributes 1)
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        comm_ranks = int(one.item())
    from adamml_amd import hip
    from adamml_amd.distributed import HipDDP
    from adamml_amd.optim import FlatSGD, FlatAdam

    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong: the global batch %d does not split over %d ranks" % (args.batch, world))
        per_gpu = args.batch // world                     # train_adamml.py:122
    else:
        per_gpu = args.batch
    model = build(args, device)
    if args.launch_plan:
        from adamml_amd import plan as _plan
        _plan.ENABLED = True
    forced = args.force_collectives and world == 1
    ddp = HipDDP(model, sync_bn=((world > 1 or forced) and not args.no_sync_bn), force_collectives=forced)
    if args.stage == "main":
        model.freeze_policy_net()
    else:
        model.freeze_main_net()
    model.train()
    if args.stage == "infer":
        model.freeze_policy_net()
        model.eval()
    if args.single_stream:
        model.use_side_stream = False
    images, target = synth_batch(args, per_gpu, device, rank)
    opt = p_opt = None

    def step():
        nonlocal opt, p_opt
        if args.stage == "infer":
            with torch.no_grad():
                out, sel = ddp(images)
            return out.sum()
        out, sel = ddp(images)
        loss = F.cross_entropy(out, target)
        if model.update_policy_net:
            usage = sel.mean(dim=1) ** 2
            loss = loss + usage.mean()
        loss.backward()
        ddp.reduce_gradients()
        if model.update_main_net:
            if opt is None:
                opt = FlatSGD(model._flat_main, lr=0.001, momentum=0.9, weight_decay=5e-4)
            opt.step()
            opt.zero_grad()
        if model.update_policy_net:
            if p_opt is None:
                p_opt = FlatAdam(model._flat_policy, lr=1e-4, weight_decay=5e-4)
            p_opt.step()
            p_opt.zero_grad()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # wall clock over exactly K steps (the contract), plus one HIP event per step boundary on the caller's stream -- every
    # side stream is joined into it before a step's optimizer runs -- for the per-step median
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    issue_ms = []
    t0 = time.time()
    marks[0].record()
    for i in range(args.steps):
        ti = time.perf_counter()
        loss = step()
        issue_ms.append((time.perf_counter() - ti) * 1e3)     # host time to ISSUE the step (nothing in step() synchronises)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med_ms = statistics.median(step_ms)
    if world > 1:
        tt = torch.tensor([dt, med_ms], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med_ms = float(tt[0].item()), float(tt[1].item())
    ms_per_step = dt / args.steps * 1e3
    clips = world * per_gpu * args.segments
    value = clips / (ms_per_step / 1e3)
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    roof = None
    breakdown = None
    if not args.no_roofline and args.stage != "infer" and rank != 0:
        # N > 1: the roofline leg below runs two more steps on rank 0; under SyncBN / gradient all-reduce every rank has to
        # issue the same collectives, so the other ranks run the same two (unprofiled) steps
        model.use_side_stream = False
        step()
        step()
        model.use_side_stream = not args.single_stream
    if rank == 0 and not args.no_roofline and args.stage != "infer":
        # per-launch HIP-event timing of one more step, single stream (concurrent streams would inflate each launch)
        model.use_side_stream = False
        step()
        hip.profiler = hip.LaunchProfiler()
        step()
        agg = hip.profiler.summary()
        role_agg = hip.profiler.summary(by_role=True)
        hip.profiler = None
        model.use_side_stream = not args.single_stream
        # launches are grouped by DEVICE kernel where the runtime names it: the forward conv and the data gradient of the
        # implicit-GEMM layers are one kernel (conv_gemm_kernel); the stem and the 3x3/64 layers have their own kernels
        tot_ms = sum(a["ms"] for a in agg.values())
        breakdown = {k: {"launches": a["launches"], "ms": round(a["ms"], 3), "pct": round(100 * a["ms"] / tot_ms, 1),
                         "tflops": round(a["flops"] / (a["ms"] * 1e9), 1) if a["ms"] > 0 else 0,
                         "gbs": round(a["bytes"] / (a["ms"] * 1e6), 1) if a["ms"] > 0 else 0}
                     for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
        per_launch_ms = a["ms"] / a["launches"]
        tfl = a["flops"] / (a["ms"] * 1e9)
        gbs = a["bytes"] / (a["ms"] * 1e6)
        f_mfma, f_hbm = tfl / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
        if f_hbm >= f_mfma:
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}
        else:
            roof = {"bound": "mfma", "achieved": round(tfl, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(f_mfma, 4)}
        # HBM traffic of that kernel: PMC counters cannot be collected from inside the process, so it comes from the committed
        # two-pass rocprofv3 --pmc summary (tools/gpu_pmc.sh) -- accepted only if it was measured on THIS build of the kernels
        traffic, tsrc = None, "no PMC summary for this build (run tools/gpu_pmc.sh)"
        if os.path.exists(PMC_TRAFFIC_FILE):
            tj = json.load(open(PMC_TRAFFIC_FILE))
            t = tj.get(name)
            if tj.get("_source_stamp") != source_stamp():
                tsrc = "%s was measured on another build (stamp %s, this build %s): not reported" % (
                    os.path.relpath(PMC_TRAFFIC_FILE, ROOT), tj.get("_source_stamp"), source_stamp())
            elif t:
                traffic = round(t["per_launch_bytes"] / 1e9, 4)                     # GB per launch (average), as `achieved` is
                tsrc = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 gfx950 correction) of this build "
                        "(source stamp %s), %s" % (tj["_source_stamp"], os.path.relpath(PMC_TRAFFIC_FILE, ROOT)))
        # the "dominant kernel" is a template family that spans regimes (HBM-bound 1x1 layers, MFMA-bound 3x3 layers): the same
        # measurement split by ROLE, each against the roof that bounds it (PMC traffic per role from the same summary file)
        pmc_roles = tj.get("_roles", {}) if (os.path.exists(PMC_TRAFFIC_FILE) and traffic is not None) else {}
        per_role = {}
        for rname, ra in sorted(role_agg.items(), key=lambda kv: -kv[1]["ms"]):
            r_tfl, r_gbs = ra["flops"] / (ra["ms"] * 1e9), ra["bytes"] / (ra["ms"] * 1e6)
            rm, rh = r_tfl / PEAK_BF16_TFLOPS, r_gbs / PEAK_HBM_GBS
            pr = pmc_roles.get(rname)
            per_role[rname] = {"bound": "hbm" if rh >= rm else "mfma", "frac": round(max(rh, rm), 4),
                               "achieved": round(r_gbs if rh >= rm else r_tfl, 1), "peak": PEAK_HBM_GBS if rh >= rm else PEAK_BF16_TFLOPS,
                               "unit": "GB/s" if rh >= rm else "TFLOP/s", "hbm_frac": round(rh, 4), "mfma_frac": round(rm, 4),
                               "c_abi_launches_per_step": ra["launches"], "ms_per_step": round(ra["ms"], 3),
                               "algorithmic_gb_per_step": round(ra["bytes"] / 1e9, 2), "algorithmic_tflop_per_step": round(ra["flops"] / 1e12, 3),
                               "traffic_gb_per_step": round(pr["hbm_bytes_per_step"] / 1e9, 2) if pr else None,
                               "traffic_over_algorithmic": round(pr["hbm_bytes_per_step"] / ra["bytes"], 3) if pr and ra["bytes"] else None}
        roof.update({"per_role": per_role,
                     "launch_count_note": "launches_per_step counts C-ABI calls; a stride-2 data gradient is ONE call that launches its four "
                                          "parity classes (rocprofv3 / PMC count those: +9 device launches per step for ResNet-50)",
                     "traffic": traffic, "traffic_unit": "GB of HBM traffic per launch (PMC, average over this kernel's launches)",
                     "traffic_source": tsrc, "algorithmic_gb_per_launch": round(a["bytes"] / a["launches"] / 1e9, 4),
                     "traffic_gb_per_step": round(traffic * a["launches"], 2) if traffic is not None else None,
                     "algorithmic_gb_per_step": round(a["bytes"] / 1e9, 2), "kernel": name, "launches_per_step": a["launches"],
                     "avg_launch_us": round(per_launch_ms * 1e3, 2), "mfma_frac": round(f_mfma, 4), "hbm_frac": round(f_hbm, 4),
                     "share_of_device_time": round(a["ms"] / tot_ms, 3)})
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.modalities == ["rgb", "sound"]:
        cpu = cpu_baseline(args)

    if rank == 0:
        headline = args.stage == "main" and args.modalities == ["rgb", "sound"] and not args.u8_input
        res = {
            "metric": "clips/sec (train fwd+bwd) RGB+Audio AdaMML @224^2, 5 seg" if args.stage != "infer" else
                      "clips/sec (inference fwd, policy-gated) AdaMML @224^2, 5 seg", "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "ms_per_step_median_hipevent": round(med_ms, 2),
            # host side: median wall time step() takes to return (Python + ctypes issue of ~1300 launches, no device sync inside);
            # when it approaches ms_per_step the step is host-bound
            "host_issue_ms": round(statistics.median(issue_ms), 2), "deterministic": bool(hip.deterministic()),
            "launch_plan": bool(args.launch_plan),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" if not args.u8_input else "synthetic (non-headline: uint8 decoded frames in, normalised by the input kernel)",
            "config": {"workload": ("AdaMML %s (non-headline), eval-mode forward with decision-driven skipping of the main nets, "
                                    "%d segments x 8 frames" % ("+".join(args.modalities), args.segments)) if args.stage == "infer" else
                       ("AdaMML RGB+Audio (ResNet-50 + Sound-MobileNetV2 + MobileNetV2/LSTM policy), %s-net "
                        "stage train step, %d segments x 8 frames, 224^2 / 256^2 spectrogram" % (args.stage, args.segments))
                       if args.modalities == ["rgb", "sound"] else
                       ("AdaMML %s (non-headline config), %s-net stage train step, %d segments x 8 frames" % ("+".join(args.modalities), args.stage, args.segments)),
                       "videos_per_gpu": per_gpu, "global_batch_videos": per_gpu * world, "clips_per_step": clips, "segments": args.segments,
                       "parallelism": "dp%d%s%s" % (world, "+syncbn" if ((world > 1 or forced) and not args.no_sync_bn) else "",
                                                    " (one-rank RCCL communicator, collectives forced on)" if forced else ""),
                       "dist_backend": (backend if world > 1 else None), "communicator_ranks": comm_ranks,
                       "optimizer": "fused flat SGD(momentum 0.9, wd 5e-4)" if args.stage != "infer" else None,
                       "executed_clips_per_modality": getattr(model, "last_skip_stats", None)},
            "videos_per_s": round(value / args.segments, 2),
            # step-level roofline from the SURVEY.md section 8(d) per-clip figures (whole job, all GPUs)
            "step_roofline": {"algorithmic_gflop_per_clip": CLIP_GFLOP_MAIN_STAGE, "algorithmic_gb_per_clip": CLIP_GB_MAIN_STAGE,
                              "mfma_frac": round(value * CLIP_GFLOP_MAIN_STAGE / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                              "hbm_frac": round(value * CLIP_GB_MAIN_STAGE / (PEAK_HBM_GBS * world), 4)} if headline else None,
            "model_tflops": round(value * CLIP_GFLOP_MAIN_STAGE / 1e3, 1) if headline else None,
            "peak_mem_gib": round(peak_mem, 1), "loss": round(float(loss.item()), 4),
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
        }
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
