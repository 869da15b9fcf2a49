#!/usr/bin/env python
"""bench.py — train-step throughput of the acoustic-image generation hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "train-step images/sec"): the `TrainerMask` step of the
reference — modified ResNet-50 image encoder (224x298x3, BN in batch-statistics mode) + UNetAcRes
generator (1 skip) -> 36x48x12, MSE + Huber + 1e-6*KL + slim L2, backward through the generator and
conv_map, TF-1 Adam — per-GPU batch 32, synthetic seeded inputs ALREADY RESIDENT in HBM, random-init
weights.  A "step" is one call of `Trainer.train_step_pipelined`: in steady state it runs ONE batch's worth of every
part of the step — trunk units 1-8 of batch n, trunk units 9-16 of batch n - 1, conv_map + generator + losses +
backward + (exchange) + Adam of batch n - 2 — on three HIP streams: K timed calls = K complete steps,
nothing skipped, every batch's arithmetic bit-identical to the one-stream step (`--no-pipeline`).  N > 1: one process
per GPU, weak scaling by default (32 images per GPU: the series ends on configs[3]'s 8 x 32 = 256), one RCCL all-reduce
of the 43 MB gradient per step, issued from the trained part's stream (a whole pipeline tick of slack behind the trunk
stages); `--no-pipeline`: bucketed, overlapped with backward.  `--scaling strong
--global-batch 256`: the SAME 256 images per step at every N, as shards of 32 (the batch-norm group) run one after
the other on each rank with accumulated gradients — the arithmetic of a step does not depend on N.

The two documented 8-GPU commands (the driver runs the weak one; the >= 6x target of BASELINE.json is the strong one):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W                                      # weak: 8 x 32 = configs[3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W --scaling strong --global-batch 256  # strong: the same 256 images at every N
At N > 1 (or `--dp-force` under torch.distributed.run at N = 1) the line verifies itself: `config.rccl_ranks`,
`config.dp` = lanes obtained beside RCCL (min over ranks), the measured exchange time per step (events around the
collectives, max over ranks), the same loop with the collectives muted, `exchange_hidden`, and the fallback taken if
fewer than three streams run side by side (one-stream step, bucketed exchange overlapped with backward).

One JSON line on rank 0.  `roofline` is for the dominant kernel (the 128x128-tile implicit-GEMM
forward conv that runs the ResNet trunk (LDS-DMA staged, XCD-aware tile order): split-fp16 "f16x3" MFMA by default, exact-f32 MFMA with
--precision f32): ALGORITHMIC FLOPs (2*M*N*K per conv) of its launches divided by their
HIP-event-measured duration, against the dense MFMA peak of the dtype the matrix cores run in.  One stream: the events
of the timed region.  Pipelined: an event pair on a lane also brackets the wait behind the other lanes' kernels, so the
kernel is timed in three one-stream steps right after the timed region (rocprofv3 --kernel-trace serialises kernels
across streams, so its per-kernel average is the kernel alone on the chip too, and agrees); the timed region's event
figure is reported beside it (`timed_region_event_*`).  The f16x3 kernel issues 3 MFMA FLOPs per algorithmic FLOP (`hw_flop_factor`).  `cpu_baseline` times the CPU oracle
(PyTorch restatement of the reference's TF-1 graph; TF-1 itself is unavailable offline) on a bounded
sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "acoustic-image-generation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0   # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]/[3]: 32)")
    ap.add_argument("--num-skip", type=int, default=1)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32", "f16"],
                    help="trunk conv arithmetic: split-fp16 MFMA (fp32-class results), exact-f32 MFMA, or fp16 operand "
                         "storage with fp32 accumulation (BASELINE configs[4]; --workload classifier only: a reduced-"
                         "precision number is never the headline metric)")
    ap.add_argument("--workload", default="trainer_mask", choices=["trainer_mask", "unet_rgb", "unet_sound", "classifier"],
                    help="trainer_mask (default: the north-star path, BASELINE configs[2]/[3]); unet_rgb / unet_sound: the "
                         "single-modality U-Net VAEs of configs[1] / [0]; classifier: DualCamNet on generated images, "
                         "configs[4]'s head")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU (configs[3] = 8 x 32); strong: a fixed --global-batch cut into "
                         "shards of --batch images (the batch-norm group), each rank runs its share one after the other "
                         "with accumulated gradients, one exchange + one Adam per step")
    ap.add_argument("--global-batch", type=int, default=256, help="strong scaling: images per step over all GPUs")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one stream, one batch at a time (Trainer.train_step) instead of the two-lane pipeline "
                         "(Trainer.train_step_pipelined: the frozen trunk of batch t + 1 beside the trained part of batch t)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "bucketed", "whole"],
                    help="data-parallel gradient exchange: bucketed = five buckets fired from hooks inside the backward "
                         "plan on an exchange stream (all-reduce overlapped with backward); whole = one all-reduce of the "
                         "flat gradient behind the backward pass; auto = bucketed on the one-stream step, whole on the "
                         "pipelined step (its trained part runs a pipeline tick behind the trunk stages)")
    ap.add_argument("--dp-force", action="store_true",
                    help="with RANK / WORLD_SIZE = 1 in the environment: initialise RCCL and run the exchange at world "
                         "size 1 (rehearsal of the data-parallel path on a one-GPU box)")
    ap.add_argument("--trunk-stages", type=int, default=0, choices=[0, 1, 2],
                    help="pipeline stages of the frozen trunk (0 = the model's default: 2 for the split-MFMA trunk)")
    ap.add_argument("--stage-cut", type=int, default=8, help="bottleneck units in trunk stage 1 (of 16)")
    ap.add_argument("--two-pass-cin", type=int, default=256,
                    help="conv3 of the stride-1 units in two passes up to this many input channels (0 = never; 256 = shipped)")
    ap.add_argument("--unet-precision", default="split", choices=["split", "bf16", "f32"],
                    help="--workload unet_rgb / unet_sound: arithmetic of the >= 32-channel layers (bf16 = BASELINE "
                         "configs[1] as stated; split = fp32-class)")
    ap.add_argument("--no-gram", action="store_true",
                    help="statistics of the fused-tail conv3 layers from a second K loop (round 3) instead of the Gram matrix "
                         "of the conv's input (round 4): A/B switch")
    ap.add_argument("--lane-priority", type=int, default=0,
                    help="HIP priority of the pipeline's extra streams (0 normal, -1 high): experiment")
    ap.add_argument("--no-side-lane", action="store_true",
                    help="record the plans without the second HIP stream for weight gradients / projection shortcuts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[1] (UNet RGB VAE) and configs[2] (2-skip, batch 64) side measurements")
    ap.add_argument("--cpu-batch", type=int, default=32, help="BASELINE.md §3: the CPU leg runs the GPU leg's batch")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU steps (BASELINE.md §3: >= 10) after --cpu-warmup")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="BASELINE.md §3: >= 3")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every host core this process may run on")
    ap.add_argument("--cpu-budget", type=float, default=150.0, help="seconds of timed CPU work after which no further "
                                                                     "timed step starts")
    return ap.parse_args()


def host_cores():
    """host cores this process can really use: the CPU affinity mask, capped by the cgroup CPU quota (a GPU box gives a
    one-GPU job a share of the host, e.g. 16 of 128 cores; threads beyond the quota only get throttled)"""
    n = len(os.sched_getaffinity(0))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(args):
    """CPU oracle train step: the SAME step as the GPU leg — same generator variant (--num-skip), same
    batch (32) — on a BOUNDED sample: --cpu-warmup (3) warm-up steps, then up to --cpu-steps (10) timed steps (about
    6 s each on the GPU box's 16-core share: ~75 s), stopping early once --cpu-budget seconds (default 150) of timed
    work are spent (at least one timed step), so the default bench.py run stays within a few minutes.  `cores` = the
    thread count actually set with torch.set_num_threads (affinity mask capped by the cgroup CPU quota).  Progress
    goes to stderr."""
    from oracle import trainer as otr

    n = args.cpu_threads if getattr(args, "cpu_threads", 0) else host_cores()
    torch.set_num_threads(n)
    n = torch.get_num_threads()
    budget = float(getattr(args, "cpu_budget", 150.0))
    orc = otr.Oracle(num_skip=args.num_skip, learning_rate=1e-4)
    ac, mf, vid, eps = otr.synthetic_batch(args.cpu_batch, seed=1234)
    warm = int(getattr(args, "cpu_warmup", 1))
    t0 = time.perf_counter()
    for _ in range(warm):
        orc.train_step(ac, mf, vid, eps)
    print("[cpu_baseline] %d warm-up step(s) (batch %d, %d threads): %.1f s" % (warm, args.cpu_batch, n,
                                                                               time.perf_counter() - t0),
          file=sys.stderr, flush=True)
    done = 0
    t0 = time.perf_counter()
    while done < args.cpu_steps and (done == 0 or time.perf_counter() - t0 < budget):
        orc.train_step(ac, mf, vid, eps)
        done += 1
        print("[cpu_baseline] timed step %d: %.1f s so far" % (done, time.perf_counter() - t0), file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    rate = args.cpu_batch * done / dt
    return {"value": rate, "unit": "images/s", "cores": n, "kind": "port",
            "gflops": rate * 41.7,
            "sample": "CPU oracle (PyTorch fp32 restatement of the TF-1 graph; TF-1 unavailable offline): the same "
                      "TrainerMask train step, %d-skip generator, batch %d, %d timed steps (%.0f s) after %d warm-up, "
                      "%d threads" % (args.num_skip, args.cpu_batch, done, dt, warm, n)}


TRAFFIC_TAGS = ("unet_rgb", "b64")    # hbm_traffic_<tag>_*.txt: PMC summaries of the other bench workloads


def load_traffic_profile(kernel_name, tag=None):
    """HBM bytes from the newest PMC summary committed under profiles/ (tools/pmc_traffic.sh + tools/pmc_summary.py:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc passes of THIS bench at batch 32
    f16x3; tag "unet_rgb": of `--workload unet_rgb --unet-precision bf16`, "b64": of `--num-skip 2 --batch 64`).
    Returns {file, commit, step_bytes, kernel_bytes_per_launch} or None: nothing is hard-coded here."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic_*.txt"))
             if (("hbm_traffic_%s_" % tag) in f if tag else not any(("hbm_traffic_%s_" % t) in f for t in TRAFFIC_TAGS))]
    # newest = highest round, then the name (files are named in commit order: r04j < r04z); never the mtime - a fresh
    # clone gives every file the same one
    files = sorted(files, key=lambda f: (int(re.search(r"profiles/r(\d+)/", f).group(1)), os.path.basename(f)))
    if not files:
        return None
    f = files[-1]
    out = {"file": os.path.relpath(f, ROOT), "commit": None, "step_bytes": None, "steady_bytes": None,
           "kernel_bytes_per_launch": None}
    want = kernel_name.replace(" ", "")
    for ln in open(f):
        m = re.match(r"commit (\S+)", ln)
        if m:
            out["commit"] = m.group(1)
        m = re.match(r"total ([0-9.]+) GB/step", ln)
        if m:
            out["step_bytes"] = float(m.group(1)) * 1e9
        m = re.search(r"steady-state step ([0-9.]+) GB", ln)
        if m:
            out["steady_bytes"] = float(m.group(1)) * 1e9
        m = re.match(r"(.+?)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", ln)
        if m and m.group(1).replace(" ", "") == want:
            out["kernel_bytes_per_launch"] = (float(m.group(3)) + float(m.group(4))) * 1e6
    return out


# activation elements per image of the single-modality U-Net VAEs; ~8 accesses of 4 bytes each per
# train step (written once and read once forward, saved activations re-read in backward, gradients written and read)
VAE_ACT_ELEMS = {"unet_rgb": 6.95e6, "unet_sound": 1.78e6}


def vae_roofline(workload, B, ms_per_step, tag=None):
    """the HBM roofline object of a U-Net VAE step: these nets are 8 - 32 channels at up to 224 x 298, so the step is bound
    by bytes, not by a dominant kernel - algorithmic bytes of the whole step against the 8 TB/s peak, beside the counter
    traffic of the committed PMC pass (profiles/r*/hbm_traffic_<tag>_*.txt) when one exists"""
    alg = VAE_ACT_ELEMS[workload] * 8 * 4 * B
    ach = alg / (ms_per_step * 1e-3) / 1e9
    prof = load_traffic_profile("", tag) if tag else None
    sb = prof["steady_bytes"] or prof["step_bytes"] if prof else None
    return {"bound": "hbm", "kernel": "whole step (no dominant kernel: few-channel layers)", "achieved": ach, "peak": 8000.0,
            "unit": "GB/s", "frac": ach / 8000.0, "alg_bytes": alg, "traffic": sb,
            "traffic_ratio": (sb / alg) if sb else None, "traffic_GBps": (sb / (ms_per_step * 1e-3) / 1e9) if sb else None,
            "traffic_source": prof["file"] if prof else None, "traffic_commit": prof["commit"] if prof else None,
            "measured": "algorithmic bytes of one step (SURVEY 8(d): activation elements x 8 accesses x 4 B x batch) over the "
                        "measured step time; traffic: PMC counter bytes of the committed pass, not of this run"}


def secondary_unet_rgb(args):
    """BASELINE configs[1]: the RGB U-Net VAE at batch 32 — as stated ("bf16": operands of the MFMA convs rounded to
    bf16, fp32 accumulation / statistics / loss / Adam) and, beside it, the fp32-class split-MFMA arithmetic that
    holds the 1e-3 parity bar"""
    from acimg.session import Session
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_vae import UNet
    dev = torch.device("cuda", torch.cuda.current_device())
    res = {}
    for precision in ("bf16", "split"):
        tr = TrainerVAE(UNet(precision=precision), learning_rate=1e-4, session=Session(dev))
        g = tr._build_functions(batch_size=32)
        tr.model.initialize(seed=1240)
        g.images.copy_(torch.rand(*g.images.shape, generator=torch.Generator().manual_seed(1234)))
        for _ in range(3):
            tr.train_step(sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            last = tr.train_step(sync=False)
        torch.cuda.synchronize()
        res[precision] = (time.perf_counter() - t0) / 10
        del tr, g
        torch.cuda.empty_cache()
    dt = res["bf16"]
    return {"workload": "BASELINE configs[1]: UNet RGB VAE train step (models/unet_architecture.py + trainer/trainer.py), "
                        "224x298x3, batch 32, bf16", "value": 32 / dt, "unit": "images/s", "ms_per_step": dt * 1e3,
            "dtype": "bf16 operands of the MFMA convs, f32 accumulate / statistics / loss / Adam",
            "roofline": vae_roofline("unet_rgb", 32, dt * 1e3, "unet_rgb"),
            "f32_class": {"value": 32 / res["split"], "unit": "images/s", "ms_per_step": res["split"] * 1e3,
                          "dtype": "f32 (f16x3 / bf16x3 split MFMA)"}}


TILE_THREADS = {(128, 128): "2,4,512", (64, 128): "1,4,256", (128, 64): "2,2,256"}


def dominant_trunk_kernel(g, f16):
    """the trunk forward-conv kernel instance (tile shape / kernel form) that carries the most FLOPs of a recorded step:
    (indices of its launches in g.plan_train, {index: algorithmic FLOP}, {index: algorithmic bytes}, kernel name)"""
    from acimg import ops
    cand = {}
    for i, (name, fn, a) in enumerate(g.plan_train.calls):
        if name in ("conv2d_fwd_split3p", "conv2d_fwd_split1p") and f16:      # the trunk's pre-split LDS-DMA kernels
            d = a[0]._obj
            key = ops.conv2d_fwd_split3_tiling(d)
        elif name == "conv2d_fwd" and not f16:
            d = a[0]._obj
            key = ops.conv2d_fwd_tiling(d)
            if key[2] != 1:
                continue
        else:
            continue
        cand.setdefault(key, []).append((i, 2.0 * d.N * d.OH * d.OW * d.K * d.R * d.S * (3 if d.R == 7 else d.C)))
    tile = max(cand, key=lambda k: sum(f for _, f in cand[k]))
    probe_idx = set(i for i, _ in cand[tile])
    flops = dict(cand[tile])
    alg_bytes = {}
    for i in probe_idx:
        d = g.plan_train.calls[i][2][0]._obj
        alg_bytes[i] = 4.0 * (d.N * d.H * d.W * d.C + d.K * d.R * d.S * d.C + d.N * d.OH * d.OW * d.K)
    if not f16:
        kernel_name = "igemm_f32_kernel<%d,%d,...,false,true>" % (tile[0], tile[1])
    elif len(tile) > 2 and tile[2] == 2:
        kernel_name = "igemm_split3r_kernel<%d>" % (tile[0] // 64)    # the ring kernel (TM = rows / 64)
    elif len(tile) > 2 and tile[2]:
        kernel_name = "igemm_split3dp_kernel<32, 0, 3, 0>"    # the persistent form of the 128x128 trunk kernel
    else:
        kernel_name = "igemm_split3d_kernel<%d,%d,%s,2,2,3>" % (tile[0], tile[1], TILE_THREADS[tile[:2]])
    return probe_idx, flops, alg_bytes, kernel_name


def secondary_configs2(args):
    """BASELINE configs[2]: the TrainerMask step with the 2-skip generator at batch 64 on one GPU"""
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer import Trainer
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model
    dev = torch.device("cuda", torch.cuda.current_device())
    FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, 2
    B = 64
    tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=2),
                 ResNet50Model(input_shape=[224, 298, 3], num_classes=None, precision=args.precision),
                 learning_rate=1e-4, session=Session(dev))
    g = tr._build_functions(batch_size=B)
    tr.modelimages.initialize(seed=1238)
    tr.modelac.initialize(seed=1239)
    fill_inputs(g, B, 4321)
    pipelined = args.precision != "f32" and not args.no_pipeline
    step = tr.train_step_pipelined if pipelined else (lambda: tr.train_step(sync=False))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    tr.flush_pipeline()
    torch.cuda.synchronize()
    # roofline of this configuration's dominant kernel: HIP events around its launches in 3 one-stream steps (the kernel
    # with the chip to itself, as for the headline's `roofline`)
    f16 = args.precision == "f16x3"
    probe_idx, flops, alg_bytes, kernel_name = dominant_trunk_kernel(g, f16)
    ev = []
    for _ in range(3):
        tr.train_step(sync=False, probe=(probe_idx, ev))
    torch.cuda.synchronize()
    ms = sum(e0.elapsed_time(e1) for _, e0, e1 in ev)
    fl = sum(flops[i] for i, _, _ in ev)
    peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
    ach = fl / (ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": kernel_name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "hw_flop_factor": 3 if f16 else 1, "hw_frac": ach * (3 if f16 else 1) / peak,
            "alg_bytes": sum(alg_bytes.values()) / len(alg_bytes), "traffic": None,
            "launches_per_step": len(probe_idx), "avg_launch_ms": ms / len(ev),
            "measured": "HIP events around every launch of the kernel in 3 one-stream steps after the timed region"}
    prof = load_traffic_profile(kernel_name, "b64") if f16 else None
    if prof and prof["kernel_bytes_per_launch"]:
        roof.update(traffic=prof["kernel_bytes_per_launch"], traffic_ratio=prof["kernel_bytes_per_launch"] / roof["alg_bytes"],
                    traffic_source=prof["file"], traffic_commit=prof["commit"],
                    step_hbm_bytes=prof["steady_bytes"] or prof["step_bytes"])
    return {"workload": "BASELINE configs[2]: TrainerMask train step, ResNet-50-mod + UNetAcRes 2-skip "
                        "(models/unet_acresnet2skip.py), batch 64", "value": B / dt, "unit": "images/s",
            "ms_per_step": dt * 1e3, "dtype": "f32", "final_loss": tr._scalars(g)["loss"], "roofline": roof}


def synthetic_inputs(B, seed):
    gen = torch.Generator().manual_seed(seed)
    vid = torch.rand(B, 224, 298, 3, generator=gen)
    mf = torch.rand(B, 12, generator=gen)
    mf = (mf - mf.amin(1, keepdim=True))
    mf = mf / mf.amax(1, keepdim=True)
    ac = torch.rand(B, 36, 48, 12, generator=gen)
    ac = ac - ac.amin((1, 2, 3), keepdim=True)
    ac = ac / ac.amax((1, 2, 3), keepdim=True)
    return ac, mf, vid


def fill_inputs(g, B, seed):
    ac, mf, vid = synthetic_inputs(B, seed)
    g.video.copy_(vid)
    g.mfcc.copy_(mf)
    g.acoustic.copy_(ac)


def other_workload(args):
    """single-GPU timing of the other train steps (same JSON contract; roofline: the U-Net VAEs against the HBM roof -
    no dominant kernel, HBM-bound few-channel layers, DESIGN.md §8 - the classifier step of configs[4] through its
    dominant kernel, the trunk's)"""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from acimg.session import Session
    B = args.batch
    sess = Session(dev)
    gen = torch.Generator().manual_seed(1234)
    if args.workload in ("unet_rgb", "unet_sound"):
        from acimg.trainer_vae import TrainerVAE
        from acimg.unet_vae import UNet, UNetSound
        cls = UNet if args.workload == "unet_rgb" else UNetSound
        tr = TrainerVAE(cls(precision=args.unet_precision), learning_rate=1e-4, session=sess)
        g = tr._build_functions(batch_size=B)
        tr.model.initialize(seed=1240)
        g.images.copy_(torch.rand(*g.images.shape, generator=gen))
        step = lambda: tr.train_step(sync=False)  # noqa: E731
        name = ("%s VAE train step (%s, %dx%dx%d, conv-BN-ReLU U-Net, MSE+Huber+KL+L2, backward, TF-1 Adam)" %
                (cls.__name__, "models/unet_architecture.py" if cls is UNet else "models/unet_sound.py",
                 tr.model.height, tr.model.width, tr.model.channels))
        last = lambda: dict(zip(("mse", "huber", "latent", "reg", "loss"), g.losses[:5].tolist()))  # noqa: E731
        launches = len(g.plan_train) + 1
    else:
        from acimg.dualcamnet import DualCamHybridModel
        from acimg.flags import FLAGS
        from acimg.trainer_class import TrainerClass
        from acimg.unet_acresnet import UNetAc
        from acimg.vision import ResNet50Model
        FLAGS.model = "DualCamNet"
        B = (B // 12) * 12 or 12
        tr = TrainerClass(DualCamHybridModel(input_shape=[36, 48, 12], num_classes=14),
                          ResNet50Model(input_shape=[224, 298, 3], num_classes=None, precision=args.precision),
                          UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=args.num_skip), session=sess)
        g = tr._build_functions(batch_size=B)
        tr.model_encoder_images.initialize(seed=1238)
        tr.model_encoder_acoustic.initialize(seed=1239)
        tr.model.initialize(seed=1241)
        g.video.copy_(torch.rand(B, 224, 298, 3, generator=gen))
        g.mfcc.copy_(torch.rand(B, 12, generator=gen))
        step = lambda: tr.train_step()  # noqa: E731
        name = ("DualCamNet classifier on generated images (trainer_reconstructed_class.py): ResNet-50-mod + UNetAcRes "
                "forward (inference mode) + DualCamNet forward/backward + Adam, %d clips of 12 frames" % (B // 12))
        last = lambda: dict(zip(("loss", "correct"), g.out[:2].tolist()))  # noqa: E731
        launches = len(g.plan_train) + 1
        cls_probe = dominant_trunk_kernel(g, args.precision in ("f16x3", "f16"))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = None
    if args.workload in VAE_ACT_ELEMS:
        roof = vae_roofline(args.workload, B, dt / args.steps * 1e3,
                            "unet_rgb" if (args.workload == "unet_rgb" and args.unet_precision == "bf16" and B == 32) else None)
    else:
        # configs[4]: the dominant kernel is the trunk's (inference-mode forward of the frozen encoder): HIP events around
        # its launches in 3 more steps (one stream: the plan has no second lane here)
        probe_idx, flops, alg_bytes, kernel_name = cls_probe
        ev = []
        for _ in range(3):
            g.plan_train.run_probed(probe_idx, ev)
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in ev)
        fl = sum(flops[i] for i, _, _ in ev)
        terms = 1 if args.precision == "f16" else 3
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS
        ach = fl / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": kernel_name if terms == 3 else kernel_name.replace("0, 3, 0>", "0, 1, 0>"),
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "hw_flop_factor": 1 if args.precision == "f32" else terms, "hw_frac": ach * (1 if args.precision == "f32" else terms) / peak,
                "alg_bytes": sum(alg_bytes.values()) / len(alg_bytes), "traffic": None,
                "launches_per_step": len(probe_idx), "avg_launch_ms": ms / len(ev),
                "measured": "HIP events around every launch of the kernel in 3 steps after the timed region (one stream)"}
    print(json.dumps({
        "metric": "train-step images/sec", "value": B * args.steps / dt, "unit": "images/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": ("f16 trunk operands, f32 accumulate / statistics / loss" if args.precision == "f16" else
                  "bf16 operands of the MFMA convs, f32 accumulate / statistics / loss / Adam"
                  if (args.workload in VAE_ACT_ELEMS and args.unet_precision == "bf16") else "f32"),
        "data": "synthetic",
        "config": {"workload": name, "per_gpu_batch": B, "global_batch": B, "parallelism": "dp1",
                   "launches_per_step": launches},
        "final": last(), "roofline": roof}))


class _QuietStdout(object):
    """The driver reads ONE JSON line from stdout.  Libraries write there too (RCCL prints a version banner when its
    communicator comes up): everything written to file descriptor 1 before `release()` goes to stderr instead."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def release(self):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    args = parse()
    if args.precision == "f16" and args.workload != "classifier":
        raise SystemExit("--precision f16 (fp16 operand storage) is accepted for --workload classifier only")
    quiet = _QuietStdout()
    if args.workload != "trainer_mask":
        try:
            import contextlib
            import io
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                other_workload(args)
        finally:
            quiet.release()
        sys.stdout.write(buf.getvalue())
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = args.dp_force and "RANK" in os.environ   # one-rank RCCL rehearsal
    side_lane = not args.no_side_lane
    if (world > 1 or force_dp) and not args.no_pipeline:
        # streams of a data-parallel pipelined step: trunk stage 1, trunk stage 2, trained part, RCCL's = 4 = the runtime's
        # hardware queues; the trained part's side lane (neutral inside the pipeline: 7.28 ms with or without it) would
        # be a fifth one sharing a queue with another
        side_lane = False
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from acimg import ops
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer import Trainer
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model

    FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, args.num_skip
    B = args.batch
    sess = Session(dev)
    tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=args.num_skip, side_lane=side_lane),
                 ResNet50Model(input_shape=[224, 298, 3], num_classes=None, precision=args.precision,
                               stages=args.trunk_stages or None, stage_cut=args.stage_cut, side_lane=side_lane,
                               two_pass=args.two_pass_cin > 0, two_pass_max_cin=args.two_pass_cin, gram=not args.no_gram),
                 learning_rate=1e-4, session=sess)
    tr.lane_priority = args.lane_priority
    g = tr._build_functions(batch_size=B)
    tr.modelimages.initialize(seed=1238)
    tr.modelac.initialize(seed=1239)
    if world > 1 or force_dp:
        tr.enable_data_parallel(exchange=args.exchange, force=force_dp)
    # synthetic inputs, resident in HBM before the timed region (seed differs per rank)
    fill_inputs(g, B, 1234 + rank)
    strong = args.scaling == "strong"
    shards = None
    if strong:
        from acimg import dp
        nsh = dp.shards_per_rank(args.global_batch, B, world)
        # this rank's shards, resident in HBM: (acoustic, mfcc, video) device tensors, fed by device-to-device copies
        shards = [tuple(t.to(dev) for t in synthetic_inputs(B, 1234 + 1000 * rank + i)) for i in range(nsh)]
    images_per_step = args.global_batch if strong else world * B

    # the dominant kernel = the trunk forward-conv kernel instance (tile shape) that carries the most FLOPs
    f16 = args.precision == "f16x3"
    probe_idx, flops, alg_bytes, kernel_name = dominant_trunk_kernel(g, f16)
    shared = g.plan_train.shared_calls()
    pipelined = not args.no_pipeline and f16
    dp_on = world > 1 or force_dp
    lanes_min, dp_fallback = None, None
    if dp_on and pipelined:
        # SELF-VERIFICATION of the multi-GPU schedule: the pipelined step needs three streams that
        # really run side by side WITH the RCCL communicator up (the runtime exposes ~4 hardware queues and RCCL takes
        # some).  The lanes are measured now, on every rank; if any rank got fewer than three, every rank falls back to
        # the one-stream step with the bucketed exchange overlapped with backward, and the line says so.
        lanes_here = tr._pipeline(g)["lanes"]
        t = torch.tensor([lanes_here], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        lanes_min = int(t.item())
        if lanes_min < 3:
            dp_fallback = ("only %d concurrent HIP streams beside RCCL on at least one rank: one-stream step with the "
                           "bucketed exchange instead of the three-lane pipeline" % lanes_min)
            pipelined = False
            tr._pipe = None
            tr.exchange = "bucketed"

    def one_step(probe=None):
        if strong:
            # the shards of a step go through the same lanes (Trainer.train_step_sharded(pipelined=True)): K calls = K x
            # shards trunks + K x shards trained parts + K exchanges + K Adam updates
            tr.train_step_sharded(shards, probe=probe, pipelined=pipelined)
        elif pipelined:
            # steady state: this call runs the frozen trunk of one batch (lane A) and conv_map + generator + backward +
            # Adam of the previous one (lane B): K calls = K trunks + K optimisation steps, nothing skipped
            tr.train_step_pipelined(probe=probe)
        else:
            tr.train_step(sync=False, probe=probe)

    for _ in range(args.warmup):
        one_step()
    events = []

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    if dp_on:
        tr.comm.start_timing()    # event pairs around every collective of the timed region, on the stream it is issued from
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(probe=(probe_idx, events))
    barrier()
    dt = time.perf_counter() - t0
    n_coll, ex_ms = 0, 0.0
    if dp_on:
        # the collectives of the K timed calls (the pipeline's flush below issues those of the batches still in flight)
        n_coll = len(tr.comm.timing)
        ex_ms = tr.comm.exchange_ms() / max(args.steps, 1)
        tr.comm.timing = None
    if pipelined:
        tr.flush_pipeline()       # the batch whose trunk ran in the last timed call (the first timed call finished one
        torch.cuda.synchronize()  # whose trunk ran during warm-up): outside the timed region on both ends
    last = tr._scalars(g)
    dp_report = None
    if dp_on:
        # what the exchange costs and whether the schedule hides it: (a) device time between the events that bracket the
        # collectives of the timed region, per step, max over ranks; (b) the SAME timed loop once more with the collectives
        # muted (weights diverge between ranks from here on: nothing after this point is a result) - `exchange_hidden` =
        # the step with the exchange took at most 2 % longer than the step without it
        tr.comm.muted = True
        for _ in range(min(args.warmup, 3)):
            one_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        dt_mute = time.perf_counter() - t1
        if pipelined:
            tr.flush_pipeline()
            torch.cuda.synchronize()
        t = torch.tensor([ex_ms, dt_mute], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ex_ms, dt_mute = float(t[0]), float(t[1])
        dp_report = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                     "lanes_min_over_ranks": lanes_min, "fallback": dp_fallback,
                     "collectives_per_step": n_coll / max(args.steps, 1),
                     "exchange_ms_per_step": ex_ms, "no_exchange_ms_per_step": dt_mute / args.steps * 1e3}
    seq_events = []
    if pipelined and events:
        # kernel quality without a second lane on the chip: a few one-stream steps after the timed region
        for _ in range(3):
            tr.train_step(sync=False, probe=(probe_idx, seq_events))
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if dp_report is not None:
        dp_report["exchange_hidden"] = bool(dt / args.steps * 1e3 <= 1.02 * dp_report["no_exchange_ms_per_step"])

    roof = None
    prof = load_traffic_profile(kernel_name) if f16 else None
    prof_ok = bool(prof and prof["kernel_bytes_per_launch"] and B == 32 and not strong)
    if events:
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in events)
        fl = sum(flops[i] for i, _, _ in events)
        achieved = fl / (ms * 1e-3) / 1e12
        peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma",
                "kernel": kernel_name,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "mfma_dtype": "f16 (3-term hi/lo split of fp32 operands, fp32 accumulate)" if f16 else "f32",
                "hw_flop_factor": 3 if f16 else 1,
                # matrix-core utilisation: the 3 MFMAs issued per algorithmic product, against the same peak
                "hw_frac": achieved * (3 if f16 else 1) / peak,
                # algorithmic HBM bytes per launch: the pre-split input tensor (two fp16 planes = 4 B per element) and the
                # split weights read once, the fp32 output written once, averaged over this kernel's launches
                "alg_bytes": sum(alg_bytes[i] for i in probe_idx) / len(probe_idx),
                # HBM bytes per launch from the PMC summary committed under profiles/ (FETCH_SIZE x2 per the gfx950
                # correction + WRITE_SIZE, batch 32, f16x3; collected by tools/pmc_traffic.sh, not by this run)
                "traffic": prof["kernel_bytes_per_launch"] if prof_ok else None,
                "traffic_source": prof["file"] if prof_ok else None,
                "traffic_commit": prof["commit"] if prof_ok else None,
                "launches_per_step": len(probe_idx),
                # launches recorded between a fork and a join of the plan's side lane (the projection shortcuts and the
                # conv1 .. conv3 they run beside): their event durations include the other lane's work
                "launches_sharing_chip": len(probe_idx) if pipelined else len(probe_idx & shared),
                "achieved_exclusive": (sum(flops[i] for i, _, _ in events if i not in shared) /
                                       max(sum(e0.elapsed_time(e1) for i, e0, e1 in events if i not in shared), 1e-9) / 1e9),
                "measured": "HIP events around every launch of the kernel over the timed region (one stream)",
                "avg_launch_ms": ms / len(events), "avg_launch_gflop": fl / len(events) / 1e9,
                "share_of_step_time": (ms / args.steps) / (dt / args.steps * 1e3)}

    if roof is not None and seq_events:
        # Pipelined: an event pair on a lane also brackets the time the launch waits behind the other lanes' kernels, so
        # it over-states the kernel's own duration.  The kernel is therefore timed with the chip to itself, in one-stream
        # steps right after the timed region — which is also what rocprofv3 --kernel-trace of this command reports (it
        # serialises kernels across streams: profiles/r02/bench_b32_kernel_stats_r02p.csv); the event figure of the timed
        # region is kept beside it.
        ms1 = sum(e0.elapsed_time(e1) for _, e0, e1 in seq_events)
        fl1 = sum(flops[i] for i, _, _ in seq_events)
        roof["timed_region_event_ms"] = roof["avg_launch_ms"]
        roof["timed_region_event_tflops"] = roof["achieved"]
        roof["achieved"] = fl1 / (ms1 * 1e-3) / 1e12
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["hw_frac"] = roof["achieved"] * roof["hw_flop_factor"] / roof["peak"]
        roof["avg_launch_ms"] = ms1 / len(seq_events)
        roof["share_of_step_time"] = roof["avg_launch_ms"] * len(probe_idx) / (dt / args.steps * 1e3)
        ex = [(i, e0, e1) for i, e0, e1 in seq_events if i not in shared]
        roof["achieved_exclusive"] = (sum(flops[i] for i, _, _ in ex) /
                                      (sum(e0.elapsed_time(e1) for _, e0, e1 in ex) * 1e-3) / 1e12)
        roof["measured"] = ("HIP events around every launch of the kernel in 3 one-stream steps right after the timed "
                            "region (in the two-lane timed region an event pair also brackets the wait behind the other "
                            "lane: timed_region_event_*); achieved_exclusive: launches outside the side lane's fork/join "
                            "windows only")
    if rank == 0:
        out = {
            "metric": "train-step images/sec", "value": images_per_step * args.steps / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 tensors everywhere; trunk conv products on fp16 matrix cores as a 3-term hi/lo split "
                           "(22 mantissa bits per operand, fp32 accumulate), parity 1e-3 vs the fp32 oracle"
                           if f16 else "fp32 tensors, exact-f32 MFMA"),
            "config": {"workload": "TrainerMask train step: ResNet-50-mod 224x298x3 (BN batch stats) + UNetAcRes "
                                   "%d-skip -> 36x48x12, MSE+Huber+KL+L2, backward, TF-1 Adam" % args.num_skip,
                       "per_gpu_batch": images_per_step // world, "global_batch": images_per_step,
                       "bn_group": B, "shards_per_gpu_per_step": len(shards) if strong else 1,
                       "parallelism": "dp%d" % world, "launches_per_step": len(g.plan_train) + 2,
                       "exchange": (None if not dp_on else
                                    "whole flat gradient, one all-reduce behind the backward pass"
                                    if (strong or ((args.exchange == "whole" or (args.exchange == "auto" and pipelined))
                                                   and dp_fallback is None)) else
                                    "5 buckets fired from hooks in the backward plan on an exchange stream (overlapped)"),
                       # multi-GPU self-verification (None on a plain one-GPU run): ranks RCCL really has, lanes obtained
                       # (min over ranks), measured exchange time and whether the schedule hid it
                       "rccl_ranks": dp_report["rccl_ranks"] if dp_report else None,
                       "dp": dp_report,
                       "lanes": ("%d HIP streams measured to run side by side: trunk units 1-8 of batch n | trunk units 9-16 "
                                 "of batch n-1 | conv_map + generator + backward + exchange + Adam of batch n-2; every batch's "
                                 "arithmetic is the one-stream step's" % tr._pipe["lanes"]) if pipelined else "1"},
            "final_loss": last["loss"], "final_mse": last["mse"],
            "roofline": roof,
        }
        # the whole step against both chip roofs (41.7 GFLOP algorithmic per image; HBM bytes per step
        # from the PMC pass under profiles/, batch 32 f16x3 only)
        per_gpu_rate = images_per_step / world * args.steps / dt
        sb = prof["step_bytes"] if (prof_ok and prof["step_bytes"]) else None
        out["step"] = {"alg_tflops": 41.7e9 * per_gpu_rate / 1e12,
                       # 0.45 GB algorithmic per image with fp32 trunk activations
                       "alg_bytes": 0.45e9 * B,
                       "hbm_bytes": sb, "traffic_ratio": (sb / (0.45e9 * B)) if sb else None,
                       # without the profiled process's one-time dispatches (torch fills of the arenas, trunk weight images)
                       "hbm_bytes_steady": prof["steady_bytes"] if sb else None,
                       "hbm_GBps": (sb / (dt / args.steps) / 1e9) if sb else None,
                       "hbm_frac": (sb / (dt / args.steps) / 8.0e12) if sb else None,
                       "hbm_source": prof["file"] if sb else None, "hbm_commit": prof["commit"] if sb else None,
                       "hbm_note": "counter bytes (incl. Infinity-Cache hits) from the committed PMC pass divided by "
                                   "THIS run's step time: derived, not re-measured"}
        if roof is not None and roof.get("traffic"):
            roof["traffic_ratio"] = roof["traffic"] / roof["alg_bytes"]
            roof["hbm_GBps"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
            roof["hbm_frac"] = roof["hbm_GBps"] * 1e9 / 8.0e12
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        if world == 1 and not args.no_secondary:
            # BASELINE configs[1] (the RGB U-Net VAE at batch 32) timed next to the north-star path, for reference
            for key, fn in (("secondary", secondary_unet_rgb), ("secondary2", secondary_configs2)):
                try:
                    out[key] = fn(args)
                except Exception as e:   # never lose the primary line
                    out[key] = {"error": repr(e)}
    quiet.release()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dp_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
