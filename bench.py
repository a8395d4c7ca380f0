#!/usr/bin/env python3
"""Headline benchmark: AV clips/s, forward + backward (+ gradient all-reduce, clip and AdamW) of the
Whisper-small audio-visual network on synthetic 30 s / 75-frame clips, one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is the whole-job rate with inputs resident in HBM.
`roofline` prices the dominant kernel (the bf16 MFMA GEMM / implicit-GEMM conv kernel) from HIP
events around each of its launches in one extra, untimed step of the same workload; `cpu_baseline`
times the CPU oracle (the fp32 PyTorch restatement of the reference path) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# The step keeps several streams busy at once (the trainable tail, the two frozen encoders replayed from their graph,
# the collective library's stream).  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order, and two
# streams that land on one queue run strictly in issue order: the tail behind the next batch's encoders costs ~2 ms a
# step.  Eight queues keep them apart whatever else the process creates first; it has to be set before HIP starts.
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    pass

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "AV clips/sec (30 s audio+75 video frames) fwd+bwd, Whisper-small"
MODELARGS = (512, 8, 6, 3000, 2048, 0.1)      # reference config.py:12-17
VOCAB = 51865
PEAK_BF16_TFLOPS = 2500.0                      # dense MFMA peak, MI355X
TRAFFIC_FILE = "r03_pmc_traffic.json"          # HBM counters of the dominant kernel (scripts/pmc_traffic.py)


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 points at stderr inside the block (native libraries included): stdout carries the one
    JSON line and nothing else."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def synthetic_batch(B, T, H, W, seed, dev, L=32, mel_frames=3000):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(T // 2 + 1, T + 1, (B,), generator=g)
    lens[0] = T
    video = torch.randn(B, T, 3, H, W, generator=g)
    vmask = torch.arange(T).expand(B, T) < lens.unsqueeze(1)
    video = video * vmask[:, :, None, None, None]
    batch = {
        "audio": torch.randn(B, mel_frames, 80, generator=g),
        "audio_mask": torch.ones(B, mel_frames, dtype=torch.bool),
        "video": video, "video_mask": vmask, "video_lengths": lens,
        "target_ids": torch.randint(1, VOCAB, (B, L), generator=g),
        "target_lengths": torch.full((B, 1), L, dtype=torch.int64),
        "audio_lengths": torch.full((B,), T, dtype=torch.int64),
    }
    return {k: v.to(dev) for k, v in batch.items()}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Threads for the CPU leg: this process's CPU share.  The GPU boxes export OMP_NUM_THREADS = the share of one
    GPU (16) while os.cpu_count() shows the whole machine; without that variable every visible core is used."""
    try:
        return max(1, int(os.environ["OMP_NUM_THREADS"]))
    except (KeyError, ValueError):
        return os.cpu_count() or 1


def cpu_baseline(whisper, threads, mode="train", budget_s=20.0):
    """The CPU oracle's training step (forward + backward of the same network, fp32) on ONE clip, in the same
    BatchNorm mode as the GPU number, plus BASELINE configuration 1 (Whisper-tiny encoder only, 4 x 10 s clips,
    fp32) on the same cores."""
    import oracle
    from mocov2_whisper_flamingo_amd import AVNet, WhisperConfig, WhisperModel
    torch.set_num_threads(threads)
    m = AVNet("audiovisual", None, 96, MODELARGS, VOCAB, whisper_config=whisper, dtype=torch.float32)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    heads = m.whisper_model.config.encoder_attention_heads
    batch = synthetic_batch(1, 75, 96, 96, 4321, torch.device("cpu"))
    times = []
    t_begin = time.perf_counter()
    for i in range(4):
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        t0 = time.perf_counter()
        oracle.training_step(sd, b, heads, MODELARGS[1], training=mode == "train", masked_row="zero")
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
        if time.perf_counter() - t_begin > budget_s and times:
            break
    times.sort()
    med = times[len(times) // 2]
    # configuration 1: Whisper-tiny encoder, 4 clips of 10 s (mel zero-padded to 3000 frames), fp32
    tiny = WhisperModel(WhisperConfig.from_name("tiny", with_decoder=False))
    tsd = {k: v.detach() for k, v in tiny.state_dict().items()}
    mel = torch.randn(4, 80, 3000, generator=torch.Generator().manual_seed(7))
    mel[:, :, 1000:] = 0.0
    t1 = []
    for i in range(4):
        t0 = time.perf_counter()
        oracle.whisper_encoder(tsd, "encoder.", mel, 6)
        if i > 0:
            t1.append(time.perf_counter() - t0)
    t1.sort()
    return {"value": round(1.0 / med, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "host_cpus_visible": os.cpu_count(),
            "sample": f"{len(times)} timed fwd+bwd steps of 1 clip (30 s audio + 75x96x96 video), fp32, {mode}-mode BN "
                      f"(no dropout draws), after 1 warm-up; median {med:.2f} s/step; threads = this process's CPU share",
            "config1": {"value": round(4.0 / t1[len(t1) // 2], 3), "unit": "clips/s",
                        "sample": f"Whisper-tiny encoder forward, 4 x 10 s mel clips, fp32, {len(t1)} timed passes after 1 "
                                  f"warm-up; median {t1[len(t1) // 2] * 1e3:.0f} ms/pass"}}


def _timed(fn, n=10, warm=3):
    """Average seconds per call of ``fn`` from HIP events on the current stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def roofline_extra(dev):
    """Rooflines of the kernels beside the dominant GEMM, each timed alone at the shapes it has in the step:
    the encoder's attention forward (MFMA), the HBM-bound kernels (LayerNorm, batch-norm apply, the 3D stem, the
    head losses) as algorithmic bytes / time against 8 TB/s, and the gated cross-attention block's two entry
    points at the benchmark shape and at the reference's max_frames (B = 8, T = 400: 3200 rows)."""
    from mocov2_whisper_flamingo_amd import GatedCrossAttentionBlock, ops
    from mocov2_whisper_flamingo_amd import functional as Fn
    bf = torch.bfloat16
    out = {}
    # encoder self-attention, one Whisper-small layer of the batch
    B, H, T = 8, 12, 1500
    qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(bf)
    t = _timed(lambda: ops.attn_fwd(qkv[:, :, :H * 64], qkv[:, :, H * 64:2 * H * 64], qkv[:, :, 2 * H * 64:], H,
                                    scale=0.125, use_mfma=True))
    fl = 4.0 * B * H * T * T * 64
    out["attn_fwd_mfma_kernel"] = {"bound": "mfma", "shape": "B=8 H=12 T=1500 hd=64", "us": round(t * 1e6, 1),
                                   "achieved": round(fl / t / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4)}

    def hbm(name, shape, nbytes, fn):
        t = _timed(fn)
        out[name] = {"bound": "hbm", "shape": shape, "us": round(t * 1e6, 1), "algorithmic_mb": round(nbytes / 1e6, 1),
                     "achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(nbytes / t / 8e12, 4)}

    x = torch.randn(12000, 768, device=dev)
    g, b_ = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    hbm("layernorm_fwd_kernel", "12000 x 768 fp32 -> bf16", 12000 * 768 * 6, lambda: ops.layernorm_fwd(x, g, b_, out_dtype=bf))
    raw = torch.randn(375000, 256, device=dev).to(bf)
    sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    hbm("bn_apply_kernel", "375000 x 256 bf16 in place (layer1 bn3 without residual)", 375000 * 256 * 4,
        lambda: ops.bn_apply(raw, sc, sh, relu=True, out=raw))
    video = torch.randn(8, 75, 3, 96, 96, device=dev)
    w = ops.stem_pack_weight(torch.randn(64, 3, 5, 3, 3, device=dev) * 0.1)
    fr = ops.stem_pack_frags(w)
    st = ops.stem_stats_slabs(video)
    hbm("stem3d_mfma_kernel", "8 x 75 x 3 x 96 x 96 fp32 -> 600 x 50 x 50 x 64 bf16 (+ BN partial sums)",
        video.numel() * 4 + 600 * 50 * 50 * 64 * 2, lambda: ops.stem3d_fwd(video, w, bf, col_stats=(st[0], st[1]), frags=fr))
    ld = 51872
    buf = torch.randn(600, ld, device=dev)
    logits = buf.view(8, 75, ld)[:, :, :VOCAB]
    tg = torch.randint(1, VOCAB, (8, 32), device=dev)
    il, tl = torch.full((8,), 75, device=dev), torch.full((8,), 32, device=dev)
    _, ws = ops.head_loss_fwd(logits, tg, il, tl)
    hbm("head_loss_fwd (row_lse + ce + ctc)", "600 x 51865 fp32 logits", 600 * VOCAB * 4, lambda: ops.head_loss_fwd(logits, tg, il, tl))
    hbm("head_loss_bwd (dlogits in bf16)", "600 x 51865 fp32 logits -> bf16 gradient", 600 * VOCAB * 6,
        lambda: ops.head_loss_bwd(logits, tg, il, tl, ws, bf, ld_out=ld))
    # gated cross-attention block through its two entry points (d 512, 8 heads, dropout 0.1, ragged key mask)
    d, Hh = 512, 8
    blk = GatedCrossAttentionBlock(d, Hh, dropout=0.1).to(dev).train()
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(0.5)
    blk._cache = Fn.WeightCache(bf)
    blk._seed_fn = lambda tag: 99
    for p_ in blk.parameters():
        p_.grad = torch.zeros_like(p_)
        p_._avsr_direct_grad = True
    xb = {}
    for Bx, Tx in ((8, 75), (8, 400)):
        xq = torch.randn(Bx, Tx, d, device=dev).to(bf).requires_grad_(True)
        xa = torch.randn(Bx, Tx, d, device=dev).to(bf).requires_grad_(True)
        mask = torch.zeros(Bx, Tx, dtype=torch.bool, device=dev)
        mask[:, Tx - Tx // 8:] = True
        dout = torch.randn(Bx, Tx, d, device=dev).to(bf)
        rows = Bx * Tx
        gemm = rows * (2 * d * d + 2 * d * d + 16 * d * d) + rows * 4 * d * d
        attn = 4 * Bx * Hh * Tx * Tx * 64

        def fwd():
            with torch.no_grad():
                blk(xq, xa, mask)

        def fwdbwd():
            blk(xq, xa, mask).backward(dout)
            xq.grad = xa.grad = None

        tf = _timed(fwd, n=20)
        tb = _timed(fwdbwd, n=20) - tf
        ff, fb = gemm + attn, 2 * gemm + 2.5 * attn
        xb[f"B={Bx} T={Tx} ({rows} rows)"] = {
            "fwd_us": round(tf * 1e6, 1), "fwd_tflops": round(ff / tf / 1e12, 1), "fwd_frac_of_mfma_peak": round(ff / tf / 2.5e15, 4),
            "bwd_us": round(tb * 1e6, 1), "bwd_tflops": round(fb / tb / 1e12, 1), "bwd_frac_of_mfma_peak": round(fb / tb / 2.5e15, 4)}
    out["avsr_gated_xattn_fwd / _bwd (one block)"] = {
        "bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "shapes": xb,
        "note": "wall time per call incl. host enqueue (the small shape is launch bound); counter-based MFMA-busy per "
                "kernel: profiles/r02_xattn_pmc_*.json"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the first batch's encoders and the last batch's tail run un-overlapped inside the timed region (≈11 ms per
    # run): 50 steps (0.4 s) keep that below 3 % of the figure
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches taken in turn")
    ap.add_argument("--whisper", default="small")
    ap.add_argument("--mode", default="train", choices=["train", "eval"],
                    help="train: batch-statistics BN + dropout (what the reference's training_step runs); "
                         "eval: running-statistics BN, no dropout")
    ap.add_argument("--attention", default="bf16", choices=["bf16", "fp8"], help="Whisper self-attention products")
    ap.add_argument("--seconds", type=int, default=30, choices=[30, 60],
                    help="clip length: 60 = long-form clips (two encoder windows, 150 video frames)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="issue each batch's frozen encoders inside its own step instead of one step ahead")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch the frozen encoders kernel by kernel instead of replaying their HIP graph")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="launch each bucket's all-reduce from gradient hooks on a side stream during backward")
    ap.add_argument("--dp-mode", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="gradient exchange: bucketed all-reduce, or reduce-scatter + AdamW on the owned shard + "
                         "all-gather of the parameters")
    ap.add_argument("--rehearse-backend", default=None, help="e.g. gloo: rehearse N ranks on fewer GPUs (not a measurement)")
    ap.add_argument("--grad-stream", default="auto", choices=["auto", "on", "off"],
                    help="parameter-gradient kernels on a stream of their own beside the dX chain; auto = only without "
                         "the pipeline (with it, the step time then hinges on which hardware queue that stream lands on)")
    ap.add_argument("--no-tail-graph", dest="tail_graph", action="store_false",
                    help="enqueue the trainable tail kernel by kernel instead of replaying its two HIP graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-roofline-extra", action="store_true", help="skip the per-kernel rooflines beside the GEMM's")
    args = ap.parse_args()

    if os.environ.get("AVSR_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["AVSR_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the measured side")
    # --rehearse-backend gloo: several ranks share the visible GPU(s) and exchange gradients over gloo - a dry run
    # of the multi-rank control path on a one-GPU box (its number is not a scaling result)
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 or args.rehearse_backend:
        if args.rehearse_backend:
            # also valid with one rank (`--rehearse-backend nccl` without a launcher): the collectives are then
            # issued over a one-rank communicator, which exercises the RCCL call path beside the graph replays
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            with _stdout_to_stderr():
                if args.rehearse_backend == "nccl":
                    dist.init_process_group("nccl", device_id=dev)
                else:
                    dist.init_process_group(args.rehearse_backend)
                dist.barrier()
        else:
            with _stdout_to_stderr():   # the collective library prints a version banner on stdout when a communicator is made
                dist.init_process_group("nccl", device_id=dev)
                dist.barrier()

    from mocov2_whisper_flamingo_amd import AVNet, ops
    from mocov2_whisper_flamingo_amd.dp import FlatTrainer

    torch.manual_seed(1234)        # identical initial weights on every rank
    windows = args.seconds // 30
    model = AVNet("audiovisual", None, 96, MODELARGS, VOCAB, whisper_config=args.whisper, dtype=torch.bfloat16,
                  seed=1234 + rank, attention=args.attention, long_form=windows > 1)
    with torch.no_grad():          # non-zero gates so the attention / FFN branches carry gradient
        for blk in model.fusion_module.layers:
            blk.attn_gate.fill_(0.5)
            blk.ff_gate.fill_(0.5)
    model.to(dev)
    model.train(args.mode == "train")
    # With the next batch's encoders running beside this batch's tail, the gradient exchange is hidden along with
    # the rest of the tail: it is issued in stream order after backward (two 64 MB-class buckets), not from hooks.
    trainer = FlatTrainer(model, lr=1e-3, bucket_mb=64, overlap=args.overlap_allreduce,
                          force_collective=bool(args.rehearse_backend), fused_step=True, dp_mode=args.dp_mode, grad_stream=args.grad_stream == "on" or (args.grad_stream == "auto" and not args.pipeline))
    B, T = args.batch, 75 * windows
    # several distinct synthetic batches, taken in turn: one batch alone (66 MB of video + 7.7 MB of mel) would sit
    # in the 256 MB Infinity Cache from step to step
    batches = [synthetic_batch(B, T, 96, 96, 1234 + rank + 1000 * j, dev, mel_frames=3000 * windows)
               for j in range(max(1, args.batches))]
    batch = batches[0]
    taken = [0]

    def fresh():
        src = batches[taken[0] % len(batches)]
        taken[0] += 1
        b = dict(src)
        b["video_lengths"] = src["video_lengths"].clone()        # forward clamps it in place
        return b

    from mocov2_whisper_flamingo_amd.dp import GraphedTail
    use_tail_graph = (args.tail_graph and args.pipeline and args.dp_mode == "allreduce" and not args.overlap_allreduce
                      and trainer.grad_stream is None)
    tail_graph = GraphedTail(trainer, warmup=2) if use_tail_graph else None

    def step(b=None, encoded=None):
        if tail_graph is not None and encoded is not None:
            return tail_graph.step(b, encoded)[0]
        trainer.zero_grad()
        loss, ctc, ce, _ = model.training_step(b or fresh(), encoded=encoded)
        loss.backward()
        trainer.reduce_gradients()
        trainer.step()
        return loss

    def run(n, marks=None):
        """n complete training steps.  The frozen encoders of batch i+1 are issued (on their own streams) before
        the trainable tail of batch i, so they run beside its small kernels; every batch's encoders, tail,
        backward, all-reduce and optimizer step are inside the call, the first batch's encoders un-overlapped.
        ``marks``: a list that receives one event per step, recorded behind the step's optimizer kernels."""
        if not args.pipeline:
            for _ in range(n):
                loss = step()
                if marks is not None:
                    marks.append(torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)))
            return loss
        cur = fresh()
        enc = model.encode(model.batch_inputs(cur), ahead=True, graph=args.graph)
        for i in range(n):
            nxt = nenc = None
            if i + 1 < n:
                nxt = fresh()
                nenc = model.encode(model.batch_inputs(nxt), ahead=True, graph=args.graph)
            loss = step(cur, enc)
            if marks is not None:
                marks.append(torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)))
            cur, enc = nxt, nenc
        return loss

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # the tail gets a stream of its own (the encoders issued ahead must not queue behind it); no stream priorities:
    # on a GPU shared by two processes a high-priority tail waiting on normal-priority encoder queues stalled for good
    tail = torch.cuda.Stream(device=dev) if args.pipeline else torch.cuda.current_stream()
    with torch.cuda.stream(tail):
        if args.warmup:
            run(args.warmup)
        barrier()
        marks = []
        t0 = time.perf_counter()
        loss = run(args.steps, marks)
        t_issue = time.perf_counter() - t0          # host time to enqueue every step (nothing in run() synchronises)
        barrier()
        dt = time.perf_counter() - t0
    # steady state: from the end of step 0 to the end of step K-2, every step in between has the next batch's
    # encoders beside its tail; the rest of `dt` is the pipeline's fill (the first batch's encoders with nothing
    # beside them) and drain (the last batch's tail alone)
    steady_ms = fill_drain_ms = None
    if len(marks) >= 4:
        steady_ms = marks[0].elapsed_time(marks[-2]) / (len(marks) - 2)
        fill_drain_ms = dt * 1e3 - steady_ms * args.steps
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())

    roofline = None
    if not args.no_roofline:
        # every rank runs the extra step (it contains the gradient all-reduce); rank 0 reports
        ops.GEMM_PROFILE = []
        # per-launch durations of the kernel alone: nothing on a second stream sharing the CUs
        from mocov2_whisper_flamingo_amd import functional as Fn
        model.overlap_encoders, saved = False, Fn.GRAD_STREAM
        Fn.GRAD_STREAM = None
        step()
        torch.cuda.synchronize()
        model.overlap_encoders, Fn.GRAD_STREAM = True, saved
        recs = [(a.elapsed_time(b) * 1e-3, fl, shape[5]) for a, b, fl, tag, shape, *_rest in ops.GEMM_PROFILE if tag == "bf16"]
        ops.GEMM_PROFILE = None
        tsum, fsum = sum(r[0] for r in recs), sum(r[1] for r in recs)
        achieved = fsum / tsum / 1e12
        # HBM bytes per launch of the same kernel come from separate rocprofv3 --pmc passes over this command
        # (a profiler has to wrap the process); the committed summary is read back here
        # (the counters see every launch of the kernel, including the 25 small products the fused cross-attention
        # blocks launch from C, which the event timing above cannot bracket one by one: traffic_launches says so)
        traffic, traffic_launches = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", TRAFFIC_FILE)) as fh:
                rec = json.load(fh)
            traffic, traffic_launches = round(float(rec["hbm_bytes_per_launch"])), int(rec["launches"])
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "traffic_launches": traffic_launches,
                    "algorithmic_bytes_per_launch": round(sum(r[2] for r in recs) / len(recs)),
                    # the two per-launch figures average over different launch sets (the counters also see the small
                    # products launched from C inside the fused blocks): compare the per-step totals
                    "traffic_bytes_per_step": None if traffic is None else traffic * traffic_launches,
                    "algorithmic_bytes_per_step": round(sum(r[2] for r in recs)),
                    "traffic_source": None if traffic is None else
                    f"profiles/{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, committed "
                    "with the code and read back here (not measured by this run)",
                    "kernel": "gemm_bf16_kernel (dense + implicit-GEMM conv instantiations)",
                    "launches_per_step": len(recs), "avg_launch_us": round(tsum / len(recs) * 1e6, 2),
                    "algorithmic_gflop_per_step": round(fsum / 1e9, 1),
                    "kernel_time_share_of_step": round(tsum / (dt / args.steps), 3)}

    extra = None
    if not args.no_roofline and rank == 0 and not args.no_roofline_extra:
        extra = roofline_extra(dev)

    # the data-parallel term: the gradient all-reduce alone (the flat fp32 gradient buffer, bucket by bucket as the
    # step issues it), against the 7 x 153 GB/s of xGMI links a GPU has
    dp_term = None
    if dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl":
        nbytes = trainer.flat_grad.numel() * 4

        def allreduce():
            for bk in trainer.buckets:
                lo, hi = bk["range"]
                dist.all_reduce(trainer.flat_grad[lo:hi])

        barrier()
        t_ar = _timed(allreduce, n=10, warm=3)
        tt = torch.tensor([t_ar], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_ar = tt.item()
        trainer.flat_grad.zero_()
        dp_term = {"rccl_ranks": world, "dp_mode": args.dp_mode, "gradient_mb": round(nbytes / 1e6, 1), "buckets": len(trainer.buckets),
                   "allreduce_ms": round(t_ar * 1e3, 3),
                   "bus_gbps": round(2 * (world - 1) / world * nbytes / t_ar / 1e9, 1), "xgmi_peak_gbps": 7 * 153,
                   "share_of_step": round(t_ar / (dt / args.steps), 3),
                   "note": "all-reduce timed alone after the run; in the step it is issued in stream order after "
                           "backward and runs beside the next batch's frozen encoders"}

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(args.whisper, threads=host_threads(), mode=args.mode)

    if rank == 0:
        clips = B * world * args.steps
        out = {
            "metric": METRIC if (args.whisper, args.seconds) == ("small", 30) else
            f"AV clips/sec ({args.seconds} s audio+{T} video frames) fwd+bwd, Whisper-{args.whisper}", "value": round(clips / dt, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_steady": None if steady_ms is None else round(steady_ms, 3),
            "fill_drain_ms_per_run": None if fill_drain_ms is None else round(fill_drain_ms, 3),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Whisper-{args.whisper} + MoCo ResNet-50 frontend + gated xattn, batch {B} x {args.seconds} s "
                                   f"clips ({T} frames 96x96) per GPU, {args.mode}-mode BN/dropout, "
                                   + ("fp8 encoder attention, " if args.attention == "fp8" else "")
                                   + "fwd+bwd+all-reduce+clip+AdamW",
                       "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": round(final_loss, 4),
                       "distinct_batches": len(batches),
                       "tail": "two HIP graphs (forward+backward, clip+AdamW+weight refresh)"
                       if (tail_graph is not None and tail_graph.rec is not None) else "eager launches",
                       "host_enqueue_ms_per_step": round(t_issue / args.steps * 1e3, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "roofline_extra": extra, "dp": dp_term,
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
