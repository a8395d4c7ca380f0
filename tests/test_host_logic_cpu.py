"""CPU tests of the host-side logic: checkpoint key compatibility, the trainable-parameter set,
the learning-rate schedule, and the data-parallel gradient reducer over gloo (world size 2)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from mocov2_whisper_flamingo_amd import AVNet, WhisperConfig
from mocov2_whisper_flamingo_amd.dp import FlatTrainer, one_cycle_lr, trainable_parameters
from tests.helpers import seed_module, synthetic_batch

MICRO = WhisperConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=1,
                      decoder_attention_heads=2)


def micro_model(seed=21):
    m = AVNet("audiovisual", None, 96, (128, 2, 4, 3000, 512, 0.1), 503, whisper_config=MICRO, dtype=torch.float32)
    return m, seed_module(m, seed)


def test_state_dict_keys_match_reference_layout():
    m, sd = micro_model()
    keys = set(sd)
    # models/av_net.py:22-63, gate_cross_attention.py:16-27,56-59, moco_visual_frontend.py:14-26, utils.py:18,34
    for k in ["whisper_model.encoder.conv1.weight", "whisper_model.encoder.embed_positions.weight",
              "whisper_model.encoder.layers.0.self_attn.k_proj.weight", "whisper_model.encoder.layers.1.fc2.bias",
              "whisper_model.encoder.layer_norm.weight", "whisper_model.decoder.embed_tokens.weight",
              "whisper_model.decoder.layers.0.encoder_attn.q_proj.weight", "audio_proj.weight", "audio_ln.bias",
              "visual_model.frontend3D.0.weight", "visual_model.frontend3D.1.running_var",
              "visual_model.frontend3D.1.num_batches_tracked", "visual_model.MoCoModel.layer1.0.downsample.0.weight",
              "visual_model.MoCoModel.layer1.0.downsample.1.running_mean", "visual_model.MoCoModel.layer3.5.bn3.weight",
              "visual_model.MoCoModel.layer4.2.conv3.weight", "video_proj.bias", "video_ln.weight",
              "fusion_module.audio_proj.weight", "fusion_module.video_proj.bias", "fusion_module.ln_post.weight",
              "fusion_module.layers.0.attn.in_proj_weight", "fusion_module.layers.0.attn.in_proj_bias",
              "fusion_module.layers.1.attn.out_proj.weight", "fusion_module.layers.0.attn_ln.weight",
              "fusion_module.layers.0.ff_ln.bias", "fusion_module.layers.0.attn_gate", "fusion_module.layers.1.ff_gate",
              "fusion_module.layers.0.ff.0.weight", "fusion_module.layers.1.ff.2.bias", "fusion_scalar",
              "pos_enc_audio.pe", "pos_enc_video.pe", "decoder.linear.weight", "decoder.linear.bias"]:
        assert k in keys, k
    assert "whisper_model.encoder.layers.0.self_attn.k_proj.bias" not in keys      # k_proj has no bias
    assert sd["pos_enc_audio.pe"].shape == (1, 5000, 128)
    assert sd["visual_model.frontend3D.0.weight"].shape == (64, 3, 5, 3, 3)
    assert len(m.fusion_module.layers) == 2                                          # n_layers // 2
    # resnet50 trunk parameter count (no stem conv/bn, no fc): 23,508,032 - 9,408 - 128 (+ nothing for fc=Identity)
    n_trunk = sum(p.numel() for p in m.visual_model.MoCoModel.parameters())
    assert n_trunk == 23508032 - 9408 - 128
    # a Lightning checkpoint prefixes keys with "model."; stripping it loads cleanly
    m2, _ = micro_model(seed=5)
    m2.load_state_dict({k: v for k, v in sd.items()}, strict=True)


def test_frozen_and_trainable_sets():
    m, _ = micro_model()
    names = [n for n, _ in trainable_parameters(m)]
    assert all(not n.startswith(("whisper_model.", "visual_model.")) for n in names)
    assert "fusion_scalar" not in names and "decoder.linear.weight" in names
    assert all(not p.requires_grad for p in m.whisper_model.encoder.parameters())
    assert all(not p.requires_grad for p in m.visual_model.parameters())
    assert set(names) == {k for k in m.state_dict() if k.startswith(oracle.ref_avnet.TRAINABLE_PREFIXES)
                          and not k.endswith(".pe")}


def test_full_size_parameter_counts():
    # the figures the gradient exchange is sized from: fusion 9,983,494; head 26,606,745; projections
    from mocov2_whisper_flamingo_amd import Decoder, GatedCrossModalFusion
    assert sum(p.numel() for p in GatedCrossModalFusion(512, 8, 3).parameters()) == 9983494
    assert sum(p.numel() for p in Decoder(512, 51865).parameters()) == 26606745


def test_one_cycle_matches_torch():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=50, pct_start=0.1, div_factor=25.0,
                                              final_div_factor=1e4, anneal_strategy="linear")
    for i in range(50):
        assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(i, 50)) < 1e-12
        opt.step()
        if i < 49:
            sch.step()


def test_ops_refuse_cpu_tensors():
    from mocov2_whisper_flamingo_amd import _lib
    m, _ = micro_model()
    b = synthetic_batch(B=1, Tv=2, H=32, W=32)
    with pytest.raises(_lib.AvsrHipError):
        m((b["audio"], b["audio_mask"], b["video"], b["video_mask"], b["video_lengths"]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q, fused=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m, sd = micro_model()
    tr = FlatTrainer(m, bucket_mb=0.25, overlap=False, fused_step=fused)
    full = synthetic_batch(B=2 * world, Tv=6, H=32, W=32, L=4, vocab=503, seed=3)
    mine = {k: v[rank * 2:(rank + 1) * 2] for k, v in full.items()}
    _, _, _, _, grads, _ = oracle.training_step(sd, mine, 2, 2, masked_row="zero")
    tr.zero_grad()
    params = dict(m.named_parameters())
    with torch.no_grad():
        for k, g in grads.items():
            params[k].grad.copy_(g)          # .grad is a view into the flat bucket buffer
    tr.reduce_gradients()
    if rank == 0:
        _, _, _, _, ref, _ = oracle.training_step(sd, full, 2, 2, masked_row="zero")
        # fused_step leaves the SUM over ranks in the buffer and hands the 1 / world to the optimizer kernels
        mean = tr._pre_scale if fused else 1.0
        assert mean == (1.0 / world if fused else 1.0)
        worst = max(((params[k].grad * mean - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-6)).item() for k in ref)
        q.put((worst, len(tr.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
def test_dp_gradient_mean_over_gloo_equals_full_batch(fused):
    """N-rank averaged gradients == single-rank gradients on the concatenated batch (eval-mode BN)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    worst, nb = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert nb > 1 and worst < 1e-4


def _accum_worker(rank, world, port, q, overlap):
    """Gradient accumulation over k = 3 micro-batches on 2 ranks: a loss that is linear in every trainable
    parameter gives known gradients (its coefficients), through autograd's own accumulation and hooks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        m, _ = micro_model()
        k = 3
        tr = FlatTrainer(m, bucket_mb=0.25, overlap=overlap, accumulate=k)
        named = trainable_parameters(m)

        def coeffs(r, micro):
            g = torch.Generator().manual_seed(1000 * r + micro)
            return [torch.randn(p.shape, generator=g) for _, p in named]

        launched_before_last = None
        for cycle in range(2):                                   # two optimizer steps: the pass counter re-arms
            tr.zero_grad()
            for micro in range(k):
                c = coeffs(rank, 10 * cycle + micro)
                loss = sum((p * ci).sum() for (_, p), ci in zip(named, c))
                last = tr.backward(loss)
                assert last == (micro == k - 1)
                if micro == k - 2:
                    launched_before_last = any(b["launched"] for b in tr.buckets)
            tr.reduce_gradients()
            want = [sum(coeffs(r, 10 * cycle + micro)[i] for r in range(world) for micro in range(k)) / (world * k)
                    for i in range(len(named))]
            worst = max(((p.grad - w).abs().max() / w.abs().max()).item() for (_, p), w in zip(named, want))
            assert worst < 1e-5, worst
            assert not launched_before_last                      # nothing was exchanged before the k-th pass
        err = None
        if overlap:          # one pass too many after the buckets went out: refused, not silently wrong
            try:
                tr.backward(sum((p * 1.0).sum() for _, p in named))
            except RuntimeError as e:
                err = str(e)
            assert err is not None and "accumulate" in err
        if rank == 0:
            q.put(("ok", worst, len(tr.buckets)))
        dist.barrier()
    except Exception as e:
        if rank == 0:
            q.put(("error", repr(e), 0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_dp_gradient_accumulation_over_gloo(overlap):
    """accumulate_grad_batches (config.py:42, train.py:322): k backward passes per optimizer step, the loss of
    each divided by k, ONE gradient exchange - launched from the hooks of the k-th pass (overlap) or after it."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    status, worst, nb = q.get(timeout=240)
    for p in procs:
        p.join(60)
    assert status == "ok", worst
    assert all(p.exitcode == 0 for p in procs)
    assert nb > 1


def _cpu_optimizer_ops(monkeypatch_target):
    """Plain-torch stand-ins for the three optimizer kernels (same arithmetic as csrc/elementwise.hip), so the
    host logic of FlatTrainer.step() - sharding, collectives, buffer bookkeeping - can run on CPU tensors."""
    import math

    def sumsq_accumulate(x, out):
        out += x.double().pow(2).sum().float()

    def clip_scale(sumsq, max_norm, scale, norm_out=None, pre_scale=1.0):
        nrm = pre_scale * math.sqrt(float(sumsq[0]))
        if norm_out is not None:
            norm_out[0] = nrm
        scale[0] = pre_scale * min(1.0, max_norm / (nrm + 1e-6))

    def adamw_step(param, grad, exp_avg, exp_avg_sq, step, *, lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01,
                   grad_scale=None, bf16_shadow=None, clear_grad=False):
        g = grad * (grad_scale[0] if grad_scale is not None else 1.0)
        if clear_grad:
            grad.zero_()
        bc1, bc2 = 1.0 - betas[0] ** step, 1.0 - betas[1] ** step
        param.mul_(1.0 - lr * weight_decay)
        exp_avg.mul_(betas[0]).add_(g, alpha=1.0 - betas[0])
        exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1.0 - betas[1])
        param.addcdiv_(exp_avg, exp_avg_sq.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)

    monkeypatch_target.sumsq_accumulate, monkeypatch_target.clip_scale = sumsq_accumulate, clip_scale
    monkeypatch_target.adamw_step = adamw_step


def _rs_ag_worker(rank, world, port, q, fused):
    """Three optimizer steps with known per-rank gradients, once with the bucketed all-reduce and once with
    reduce-scatter + sharded AdamW + all-gather: the same parameters after every step, on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        from mocov2_whisper_flamingo_amd import ops
        _cpu_optimizer_ops(ops)
        ma, _ = micro_model()
        mb, _ = micro_model()
        ta = FlatTrainer(ma, bucket_mb=0.25, overlap=False, fused_step=fused, lr=1e-2, max_grad_norm=0.5)
        tb = FlatTrainer(mb, overlap=False, fused_step=fused, lr=1e-2, max_grad_norm=0.5, dp_mode="rs_ag")
        assert tb.flat_param.numel() % (4 * world) == 0 and tb.exp_avg.numel() * world == tb.flat_param.numel()
        assert tb.exp_avg.numel() < ta.exp_avg.numel()                   # optimizer state for the owned shard only
        worst = 0.0
        for step in range(3):
            for tr in (ta, tb):
                tr.zero_grad()
                g = torch.Generator().manual_seed(100 * step + rank)
                with torch.no_grad():
                    for _, p in tr.named:
                        p.grad.add_(torch.randn(p.shape, generator=g) * (0.1 + step))    # some steps clip, some do not
                tr.reduce_gradients()
                tr.step()
            for (na, pa), (nb, pb) in zip(ta.named, tb.named):
                assert na == nb
                worst = max(worst, ((pa - pb).abs().max() / pa.abs().max().clamp_min(1e-6)).item())
            assert abs(ta.grad_norm().item() - tb.grad_norm().item()) <= 1e-5 * ta.grad_norm().item()
            # every rank holds the same parameters after the all-gather
            mine = tb.flat_param.clone()
            dist.broadcast(mine, src=0)
            assert torch.equal(mine, tb.flat_param)
        if rank == 0:
            q.put(("ok", worst))
        dist.barrier()
    except Exception as e:
        if rank == 0:
            q.put(("error", repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fused", [(2, False), (2, True), (4, True)])
def test_dp_reduce_scatter_all_gather_mode_follows_the_all_reduce_trajectory(world, fused):
    """FlatTrainer(dp_mode="rs_ag") (train.py:312-326 as reduce-scatter + AdamW on the owned shard + all-gather)
    against the all-reduce mode over three steps, world size 2 and 4 on gloo."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rs_ag_worker, args=(r, world, port, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    status, worst = q.get(timeout=300)
    for p in procs:
        p.join(60)
    assert status == "ok", worst
    assert all(p.exitcode == 0 for p in procs)
    assert worst < 1e-5


def test_adaptive_time_mask_draws_like_the_reference():
    """Host-side draw of AdaptiveTimeMask (datamodule/transforms.py:27-45): one candidate per started 25 frames,
    mask lengths below the window, nothing outside the clip; a zero first draw skips the mask after its start
    has been drawn."""
    import random
    from mocov2_whisper_flamingo_amd.transforms import adaptive_time_mask

    class Fixed:
        def __init__(self, vals):
            self.vals = list(vals)

        def randrange(self, a, b):
            v = self.vals.pop(0)
            assert a <= v < b
            return v

    # length 75 -> int((75 + 25 - 0.1) // 25) = 3 candidates; pairs (t, len): (4, 6), (0, 9), (9, 3).  The pair with
    # t = 0 still draws its start (33, transforms.py:40) before it is skipped (:41-42)
    z = adaptive_time_mask(75, 10, 25, Fixed([4, 6, 0, 9, 9, 3, 70, 33, 10]))
    assert [i for i, v in enumerate(z) if v] == [10, 11, 12, 70, 71, 72, 73, 74]
    for seed in range(20):
        z = adaptive_time_mask(60, 10, 25, random.Random(seed))
        assert len(z) == 60 and sum(z) <= 3 * 9


def test_spec_augment_draws_follow_the_reference_order_and_ranges():
    """LogMelFrontend.draw without a GPU: per clip two frame-band starts in [0, frames - 48) are drawn first, then
    two mel-bin band starts in [0, 70), then the noise offset and the SNR level (datamodule/transforms.py:157-177
    on the [80, frames] tensor of :127, then :55-57)."""
    from mocov2_whisper_flamingo_amd.logmel import LogMelFrontend

    class Recorder:
        def __init__(self):
            self.calls = []

        def randrange(self, a, b):
            self.calls.append(("randrange", a, b))
            return b - 1

        def randint(self, a, b):
            self.calls.append(("randint", a, b))
            return b

        def choice(self, seq):
            self.calls.append(("choice", len(seq)))
            return seq[-1]

    fe = LogMelFrontend.__new__(LogMelFrontend)          # draw() needs no device state
    rec = Recorder()
    fs, ts, off, snr = fe.draw(2, 480000, 20000, rec)
    per_clip = [("randrange", 0, 3001 - 48)] * 2 + [("randrange", 0, 70)] * 2 + [("randint", 0, 17000), ("choice", 7)]
    assert rec.calls == per_clip * 2
    assert ts == [[2952, 2952]] * 2 and fs == [[69, 69]] * 2 and off == [17000] * 2 and snr == [999999.0] * 2
    fs, ts, _, _ = fe.draw(1, 160 * 40, 20000, Recorder())        # 41 frames: no room for a 48-frame band
    assert ts == [[-1, -1]] and fs == [[69, 69]]


def test_dropout_generator_statistics():
    """The integer-hash dropout generator (host restatement of csrc/common.h avsr_rand4): keep rate, and no visible
    correlation between neighbouring elements, neighbouring rows, the four draws of a counter, or the masks of two
    consecutive training steps (seeds one SEED_STEP_MUL apart)."""
    import numpy as np
    from tests.philox_ref import attn_keep_mask, keep_mask, rand4
    seed = 0x1234567812345678
    p = 0.1
    m = keep_mask(seed, (1 << 20,), p).astype(np.float64)
    assert abs(m.mean() - 0.9) < 1.5e-3
    x = m - m.mean()
    for lag in (1, 2, 3, 4, 8, 512, 513):
        assert abs((x[:-lag] * x[lag:]).mean() / x.var()) < 5e-3, lag
    step = AVNet.SEED_STEP_MUL
    m2 = keep_mask((seed + step) & 0xFFFFFFFFFFFFFFFF, (1 << 20,), p).astype(np.float64)
    assert abs(((m2 - m2.mean()) * x).mean() / x.var()) < 5e-3
    a = attn_keep_mask(seed, (2, 8, 75, 75), p).astype(np.float64)
    assert abs(a.mean() - 0.9) < 3e-3
    xa = a - a.mean()
    assert abs((xa[:, :, :-1] * xa[:, :, 1:]).mean() / xa.var()) < 1e-2           # neighbouring query rows
    assert abs((xa[..., :-1] * xa[..., 1:]).mean() / xa.var()) < 1e-2             # neighbouring keys
    # the 16-bit draws themselves are uniform: mean and variance of 2^20 draws per output
    for o in rand4(seed, np.arange(1 << 20, dtype=np.uint64)):
        u = o.astype(np.float64) / 65536.0
        assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1.0 / 12.0) < 2e-3


def test_layernorm_fold_algebra():
    """ops.ln_fold_weights: LayerNorm(x) @ W^T + b equals rstd * (x @ W_g^T) - rstd * mean * colsum + bias_b, and a value
    kept as two bf16 planes hi = bf16(x), lo = bf16(x - hi) is x to 2^-16 relative (what the folded encoder path relies
    on between an out-projection / fc2 and the QKV / fc1 product after it; HF modeling_whisper.py
    WhisperEncoderLayer.forward: self_attn_layer_norm -> q/k/v_proj, final_layer_norm -> fc1)."""
    from mocov2_whisper_flamingo_amd import ops
    g = torch.Generator().manual_seed(11)
    M, D, N = 50, 96, 40
    x = torch.randn(M, D, generator=g) * 3 + torch.randn(M, 1, generator=g)
    w, b = torch.randn(N, D, generator=g) / D ** 0.5, torch.randn(N, generator=g)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.3
    want = torch.nn.functional.layer_norm(x, (D,), gamma, beta) @ w.t() + b
    w_g, colsum, bias_b = ops.ln_fold_weights(w, b, gamma, beta, dtype=torch.float32)
    mean = x.mean(1, keepdim=True)
    rstd = torch.rsqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    got = rstd * (x @ w_g.t()) - rstd * mean * colsum[None, :] + bias_b[None, :]
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
    # with bf16 weights the column sums are those of the ROUNDED weights: the mean's term cancels what the product holds
    w_b, colsum_b, _ = ops.ln_fold_weights(w, b, gamma, beta)
    assert w_b.dtype == torch.bfloat16 and torch.equal(colsum_b, w_b.float().sum(1))
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert ((hi.float() + lo.float()) - x).abs().max() <= 2.0 ** -16 * x.abs().max()
