"""Parity at the sizes and in the modes the benchmark runs (BASELINE.json configurations 2, 4 and 5), against the
CPU oracle on identical seeded weights and inputs:

* the Whisper-small encoder at full size on one 30 s clip, and two-layer stacks at Whisper-medium (D = 1024,
  16 heads) and Whisper-large-v2 (D = 1280, 20 heads) widths - fp32 within 1e-3, bf16 within 2e-2;
* the whole visual frontend on one clip of 75 frames 96 x 96 (running statistics, and batch statistics);
* AVNet.training_step in the benchmark's mode - bf16 compute, batch-statistics BatchNorm, both dropouts with the
  device generator replayed on the host - end to end: loss, logits and every gradient.

A randomly initialised ResNet-50 with batch-statistics BN amplifies any perturbation about 1.3x per block, which
is a property of random weights, not of the kernels.  These tests seed the trunk the way trained residual
networks look (the last BN gain of every block small: 0.15 x), so end-to-end tolerances are meaningful."""
import pytest
import torch

import oracle
from mocov2_whisper_flamingo_amd import AVNet, MoCoVisualFrontend, WhisperConfig, WhisperModel
from tests.helpers import assert_close, closeness, max_err, record, seed_module, synthetic_batch, to_device

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}
MICRO = WhisperConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=1,
                      decoder_attention_heads=2)


def tame_residual_branches(module, sd, prefix=""):
    """Scale the last BatchNorm gain of every Bottleneck (bn3.weight) by 0.15 in the module and in its fp32 state
    dict: each block then adds a small correction to its input, as in a trained network."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("bn3.weight"):
                p.mul_(0.15)
                sd[prefix + name] = p.detach().cpu().clone().float()
    return sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name,layers", [("small", 12), ("medium", 2), ("large-v2", 2)])
def test_whisper_encoder_at_benchmark_widths(dev, dtype, name, layers):
    torch.set_num_threads(16)
    full = WhisperConfig.from_name(name)
    cfg = WhisperConfig(d_model=full.d_model, encoder_layers=layers, encoder_attention_heads=full.encoder_attention_heads,
                        with_decoder=False)
    m = WhisperModel(cfg)
    sd = seed_module(m, 7)
    m.to(dev)
    mel = torch.randn(1, 80, 3000, generator=torch.Generator().manual_seed(3))
    ref = oracle.whisper_encoder(sd, "encoder.", mel, cfg.encoder_attention_heads)
    got = m.encoder(mel.to(dev), dtype=dtype)[0]
    assert got.shape == ref.shape == (1, 1500, full.d_model)
    assert_close(got, ref, TOL[dtype], f"whisper-{name} encoder ({layers} layers) {dtype}")


def test_whisper_encoder_batch_8_with_split_attention_rounds(dev):
    """Whisper-small at the benchmark batch (8 clips: 1152 attention items on 768 resident workgroups).  With
    ``attn_split`` the items of the half-empty last round are computed by two workgroups each and merged; the encoder
    output must agree with the unsplit run to bf16 rounding, call after call on the same workspace."""
    full = WhisperConfig.from_name("small")
    cfg = WhisperConfig(d_model=full.d_model, encoder_layers=2, encoder_attention_heads=full.encoder_attention_heads,
                        with_decoder=False)
    m = WhisperModel(cfg)
    seed_module(m, 9)
    m.to(dev)
    mel = torch.randn(8, 80, 3000, generator=torch.Generator().manual_seed(4)).to(dev)
    base = m.encoder(mel, dtype=torch.bfloat16)[0]
    m.encoder.attn_split = True
    for _ in range(2):
        got = m.encoder(mel, dtype=torch.bfloat16)[0]
        assert m.encoder._attn_ws._buf is not None, "this shape is split"
        assert max_err(got, base) < 1e-2
    assert bool((m.encoder._attn_ws._buf[:4 * 384] == 0).all()), "the 384 counters come back to zero"
    # the pruned last layer (75 queries per clip over 1500 keys: 96 items, every one cut four ways)
    m.encoder.attn_split = False
    base = m.encoder(mel, dtype=torch.bfloat16, rows=75)[0]
    m.encoder.attn_split = True
    got = m.encoder(mel, dtype=torch.bfloat16, rows=75)[0]
    assert got.shape == base.shape == (8, 75, full.d_model)
    assert max_err(got, base) < 1e-2


@pytest.mark.parametrize("dtype,training", [(torch.float32, False), (torch.bfloat16, False), (torch.float32, True),
                                            (torch.bfloat16, True)])
def test_visual_frontend_full_clip(dev, dtype, training):
    """One clip of 75 frames 96 x 96 through the 3D stem and all 16 Bottlenecks: 25 x 25 -> 13 x 13 -> 7 x 7 ->
    4 x 4 feature maps, ragged length (frames past it come out zero)."""
    torch.set_num_threads(16)
    m = MoCoVisualFrontend()
    sd = tame_residual_branches(m, seed_module(m, 11))
    m.compute_dtype = dtype
    m.to(dev).train(training)
    video = torch.randn(1, 75, 3, 96, 96, generator=torch.Generator().manual_seed(5))
    lens = torch.tensor([61])
    video[0, 61:] = 0
    new_stats = {}
    ref = oracle.visual_frontend(sd, "", video, lens, training, new_stats)
    got = m(video.to(dev), lens.to(dev))
    assert got.shape == ref.shape == (1, 75, 2048)
    assert (got[0, 61:] == 0).all()
    assert_close(got, ref, TOL[dtype] * (2 if training else 1), f"visual frontend full clip {dtype} training={training}")
    if training:
        now = m.state_dict()
        for k in ("frontend3D.1.running_var", "MoCoModel.layer3.5.bn2.running_mean", "MoCoModel.layer4.2.bn3.running_var"):
            assert max_err(now[k], new_stats[k]) < (5e-3 if dtype == torch.float32 else 2e-2), k


def test_training_step_in_the_benchmark_mode(dev):
    """bf16 compute + batch-statistics BN + attention / feed-forward dropout, end to end against the oracle with
    the keep masks replayed from the device generator: loss, logits, every gradient of the trainable tail."""
    from tests.philox_ref import attn_keep_mask, keep_mask
    dtype, p = torch.bfloat16, 0.1
    m = AVNet("audiovisual", None, 96, (128, 2, 4, 3000, 512, p), 503, whisper_config=MICRO, dtype=dtype)
    sd = seed_module(m, 21)
    sd = tame_residual_branches(m.visual_model, sd, "visual_model.")
    m.to(dev).train()
    batch = synthetic_batch(B=2, Tv=6, H=32, W=32, L=4, vocab=503, seed=17)
    loss, ctc, ce, logits = m.training_step(to_device(batch, dev))
    B, T, H, d = 2, 6, 2, 128
    drop = {"attn": [], "ff": []}
    for i in range(2):
        drop["attn"].append(torch.from_numpy(attn_keep_mask(m._dropout_seed(f"fusion.{i}.attn"), (B, H, T, T), p)))
        drop["ff"].append(torch.from_numpy(keep_mask(m._dropout_seed(f"fusion.{i}.ff"), (B * T, d), p)).view(B, T, d))
    loss_r, ctc_r, ce_r, logits_r, grads_r, _ = oracle.training_step(sd, batch, 2, 2, training=True, drop=drop, p_drop=p,
                                                                     masked_row="zero")
    tol = TOL[dtype]
    assert max_err(logits, logits_r) < tol
    assert abs(loss.item() - loss_r.item()) < tol * max(1.0, abs(loss_r.item()))
    assert abs(ctc.item() - ctc_r.item()) < tol * max(1.0, abs(ctc_r.item()))
    loss.backward()
    params = dict(m.named_parameters())
    for k, gr in grads_r.items():
        scale = max(1.0, gr.abs().max().item())
        assert (params[k].grad.float().cpu() - gr).abs().max().item() < 5 * tol * scale, k


def _replayed_dropout(m, n_blocks, B, H, T, d, p):
    """The keep masks the device generator draws for the fusion blocks of the model's NEXT training forward."""
    from tests.philox_ref import attn_keep_mask, keep_mask
    drop = {"attn": [], "ff": []}
    for i in range(n_blocks):
        drop["attn"].append(torch.from_numpy(attn_keep_mask(m._dropout_seed(f"fusion.{i}.attn"), (B, H, T, T), p)))
        drop["ff"].append(torch.from_numpy(keep_mask(m._dropout_seed(f"fusion.{i}.ff"), (B * T, d), p)).view(B, T, d))
    return drop


def _argmax_agreement(what, m, logits, logits_r, tol):
    """Greedy ids: asserted equal wherever the reference's top-2 margin exceeds 2 tol max|logit|; the agreement
    rate over ALL positions is recorded (bf16 ids are not bit-exact near ties)."""
    ids, ref_ids = m.decode_ids(logits).cpu(), logits_r.argmax(-1)
    top2 = logits_r.topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 2 * tol * logits_r.abs().max()
    assert sure.any() and torch.equal(ids[sure], ref_ids[sure])
    record(what + ": bf16 arg-max agreement over all positions", agree=float((ids == ref_ids).float().mean()),
           positions=ids.numel(), sure_positions=int(sure.sum()))


@pytest.mark.parametrize("training,B", [(False, 8), (True, 2), (True, 8)])
def test_config2_model_end_to_end(dev, training, B):
    """BASELINE.json configuration 2 at its real sizes, batch 8 included (the benchmark's tile choices): Whisper-small
    (D 768, 12 layers, 12 heads), 75 frames 96 x 96, fusion width 512 x 8 heads x 3 blocks, vocabulary 51865,
    bf16 compute; eval mode, and the benchmark's train mode (batch-statistics BN + dropout, masks replayed).
    Loss, logits, greedy ids where the reference's top-2 margin exceeds the tolerance, and the gradients of the
    head and of the first fusion block against the oracle."""
    from tests.helpers import MODELARGS
    torch.set_num_threads(16)
    dtype, p, V = torch.bfloat16, 0.1, 51865
    cfg = WhisperConfig.from_name("small", with_decoder=False)
    m = AVNet("audiovisual", None, 96, MODELARGS, V, whisper_config=cfg, dtype=dtype)
    sd = seed_module(m, 33)
    sd = tame_residual_branches(m.visual_model, sd, "visual_model.")
    m.to(dev).train(training)
    batch = synthetic_batch(B=B, Tv=75, H=96, W=96, L=32, vocab=V, seed=19)
    loss, ctc, ce, logits = m.training_step(to_device(batch, dev))
    T, H, d = 75, 8, 512
    drop = _replayed_dropout(m, 3, B, H, T, d, p) if training else None
    loss_r, ctc_r, ce_r, logits_r, grads_r, _ = oracle.training_step(sd, batch, 12, 8, training=training, drop=drop,
                                                                     p_drop=p if training else 0.0, masked_row="zero")
    tol = TOL[dtype]
    tag = f"config 2 batch {B} {'train' if training else 'eval'}"
    assert logits.shape == logits_r.shape == (B, 75, V)
    assert_close(logits, logits_r, tol, tag + ": logits")
    assert abs(loss.item() - loss_r.item()) < tol * max(1.0, abs(loss_r.item()))
    _argmax_agreement(tag, m, logits, logits_r, tol)
    loss.backward()
    params = dict(m.named_parameters())
    for k in ("decoder.linear.weight", "decoder.linear.bias", "fusion_module.layers.0.attn.in_proj_weight",
              "fusion_module.layers.0.ff.0.weight", "fusion_module.layers.0.attn_ln.weight", "audio_proj.weight",
              "video_proj.weight"):
        gr = grads_r[k]
        assert (params[k].grad.float().cpu() - gr).abs().max().item() < 5 * tol * max(1.0, gr.abs().max().item()), k
        if gr.numel() >= 4096:
            c = closeness(params[k].grad, gr, 5 * tol)
            record(tag + f": gradient {k}", **c)


def test_config2_untamed_random_trunk_is_reported(dev):
    """The same end-to-end comparison on the seeded RANDOM trunk (no scaling of the last BN gain): bf16 rounding is
    amplified block after block by batch-statistics BN on random weights, so no tolerance is asserted here - the
    error is measured and reported, eval and train mode."""
    from tests.helpers import MODELARGS
    torch.set_num_threads(16)
    dtype, V, B = torch.bfloat16, 51865, 2
    cfg = WhisperConfig.from_name("small", with_decoder=False)
    for training in (False, True):
        m = AVNet("audiovisual", None, 96, (MODELARGS[0], MODELARGS[1], MODELARGS[2], MODELARGS[3], MODELARGS[4], 0.0), V,
                  whisper_config=cfg, dtype=dtype)
        sd = seed_module(m, 33)
        m.to(dev).train(training)
        batch = synthetic_batch(B=B, Tv=75, H=96, W=96, L=32, vocab=V, seed=19)
        inp = (batch["audio"], batch["audio_mask"], batch["video"], batch["video_mask"], batch["video_lengths"].clone())
        ref, parts = oracle.avnet_forward(sd, inp, 12, 8, training=training, new_stats={} if training else None,
                                          masked_row="zero", return_parts=True)
        bd = to_device(batch, dev)
        got = m((bd["audio"], bd["audio_mask"], bd["video"], bd["video_mask"], bd["video_lengths"].clone()))
        assert torch.isfinite(got).all()
        vis = m.visual_model(bd["video"], bd["video_lengths"]) if not training else None
        record(f"config 2 UNTAMED random trunk, {'train' if training else 'eval'} mode (reported, not asserted): logits",
               **closeness(got, ref, TOL[dtype]))
        if vis is not None:
            record("config 2 UNTAMED random trunk, eval mode: visual features", **closeness(vis, parts["visual"], TOL[dtype]))


@pytest.mark.parametrize("training", [False, True])
def test_config4_medium_with_noise_augmented_mel(dev, training):
    """BASELINE.json configuration 4 on one clip: the training-time mel pipeline (SpecAugment bands + babble-style
    noise at a drawn SNR + clip layer norm; the noise recording is synthetic, the reference's file is not in the
    checkout) feeding the Whisper-medium AV-net (D 1024, 24 layers, 16 heads) in bf16, against the oracle run on
    the oracle's own mel of the same waveform and draws; eval mode, and train mode (batch-statistics BN over the one
    clip's frames, both dropouts replayed).  PARITY UNPINNED for the mel part (torchaudio absent)."""
    from mocov2_whisper_flamingo_amd import LogMelFrontend
    from tests.helpers import MODELARGS
    torch.set_num_threads(16)
    V, p = 51865, 0.1
    g = torch.Generator().manual_seed(41)
    wave = 0.1 * torch.randn(1, 480000, generator=g)
    noise_rec = 0.3 * torch.randn(20000, generator=g)
    fs, ts, off, snr = [[5, 40]], [[100, 2000]], [123], [5.0]
    mel_ref = oracle.torchaudio_melspec_train(wave, fs, ts, [noise_rec[off[0]:off[0] + 3000]], snr)      # [1,3000,80]
    mel = LogMelFrontend("melspec_norm", dev).train(wave.to(dev), noise_rec.to(dev), draws=(fs, ts, off, snr))
    assert max_err(mel, mel_ref) < 1e-3 * max(1.0, float(mel_ref.abs().max()))
    cfg = WhisperConfig.from_name("medium", with_decoder=False)
    m = AVNet("audiovisual", None, 96, MODELARGS, V, whisper_config=cfg, dtype=torch.bfloat16)
    sd = tame_residual_branches(m.visual_model, seed_module(m, 35), "visual_model.")
    m.to(dev).train(training)
    batch = synthetic_batch(B=1, Tv=75, H=96, W=96, L=32, vocab=V, seed=23)
    inp = (mel_ref, batch["audio_mask"], batch["video"], batch["video_mask"], batch["video_lengths"].clone())
    got = m((mel, batch["audio_mask"].to(dev), batch["video"].to(dev), batch["video_mask"].to(dev),
             batch["video_lengths"].clone().to(dev)))
    drop = _replayed_dropout(m, 3, 1, 8, 75, 512, p) if training else None
    ref = oracle.avnet_forward(sd, inp, 16, 8, training=training, new_stats={} if training else None, drop=drop,
                               p_drop=p if training else 0.0, masked_row="zero")
    assert got.shape == ref.shape == (1, 75, V)
    assert_close(got, ref, TOL[torch.bfloat16], f"config 4 one clip {'train' if training else 'eval'}: logits")


@pytest.mark.parametrize("attention,tol,training", [("bf16", 2e-2, False), ("fp8", 5e-2, False), ("bf16", 2e-2, True)])
def test_config5_large_v2_long_form(dev, attention, tol, training):
    """BASELINE.json configuration 5 on one clip: Whisper-large-v2 (D 1280, 32 layers, 20 heads), 60 s of audio
    encoded as two 30 s windows, 150 video frames, bf16 with the encoder's attention products in bf16 or in fp8
    (e4m3 operands: three mantissa bits, tolerance 5e-2 - proposed here, the task states none for fp8); eval mode,
    and (bf16) train mode with batch-statistics BN and replayed dropout."""
    from tests.helpers import MODELARGS
    torch.set_num_threads(16)
    V, p = 51865, 0.1
    cfg = WhisperConfig.from_name("large-v2", with_decoder=False)
    m = AVNet("audiovisual", None, 96, MODELARGS, V, whisper_config=cfg, dtype=torch.bfloat16, attention=attention,
              long_form=True)
    sd = tame_residual_branches(m.visual_model, seed_module(m, 37), "visual_model.")
    m.to(dev).train(training)
    batch = synthetic_batch(B=1, Tv=150, H=96, W=96, L=32, vocab=V, seed=29)
    g = torch.Generator().manual_seed(30)
    audio = torch.randn(1, 6000, 80, generator=g)
    mask = torch.ones(1, 6000, dtype=torch.bool)
    inp = (audio, mask, batch["video"], batch["video_mask"], batch["video_lengths"].clone())
    got = m((audio.to(dev), mask.to(dev), batch["video"].to(dev), batch["video_mask"].to(dev),
             batch["video_lengths"].clone().to(dev)))
    drop = _replayed_dropout(m, 3, 1, 8, 150, 512, p) if training else None
    ref = oracle.avnet_forward(sd, inp, 20, 8, training=training, new_stats={} if training else None, drop=drop,
                               p_drop=p if training else 0.0, masked_row="zero", long_form=True)
    assert got.shape == ref.shape == (1, 150, V)
    assert_close(got, ref, tol, f"config 5 one clip {attention} attention {'train' if training else 'eval'}: logits")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unread_frames_pruned_in_the_last_encoder_layer(dev, dtype):
    """encoder(..., rows=n) returns bit for bit the first n frames of the full output (only the last layer's
    row-wise work on the frames nobody reads is skipped), and AVNet's logits do not change when it uses it."""
    cfg = WhisperConfig.from_name("tiny", with_decoder=False)
    m = WhisperModel(cfg)
    seed_module(m, 7)
    m.to(dev)
    mel = torch.randn(2, 3000, 80, generator=torch.Generator().manual_seed(3)).to(dev)
    full = m.encoder(mel, dtype=dtype, channels_last=True)[0]
    for n in (1, 75, 400, 1499):
        part = m.encoder(mel, dtype=dtype, channels_last=True, rows=n)[0]
        assert part.shape == (2, n, cfg.d_model)
        assert torch.equal(part, full[:, :n]), n
    assert torch.equal(m.encoder(mel, dtype=dtype, channels_last=True, rows=5000)[0], full)
    net = AVNet("audiovisual", None, 96, (128, 2, 4, 3000, 512, 0.1), 503, whisper_config=MICRO, dtype=dtype)
    seed_module(net, 21)
    net.to(dev).eval()
    bd = to_device(synthetic_batch(B=2, Tv=6, H=32, W=32, L=4, vocab=503, seed=5), dev)
    outs = []
    for prune in (True, False):
        net.prune_unread_frames = prune
        outs.append(net((bd["audio"], bd["audio_mask"], bd["video"], bd["video_mask"], bd["video_lengths"].clone())))
    assert torch.equal(outs[0], outs[1])


def test_two_pass_batch_norm_for_expanding_convolutions(dev):
    """Train-mode BN of the expanding 1x1 convolutions as statistics-only pass + product with the normalisation in
    its epilogue (an option, off by default; forced here for every eligible convolution): the first block's batch
    statistics - hence its running statistics - are bit for bit those of the store-raw-then-apply path, and the
    features meet the bf16 tolerance against the oracle."""
    import copy
    torch.set_num_threads(16)
    m = MoCoVisualFrontend()
    sd = tame_residual_branches(m, seed_module(m, 11))
    m.compute_dtype = torch.bfloat16
    m.to(dev).train()
    m2 = copy.deepcopy(m)
    m2.two_pass_min_bytes = 0
    m.two_pass_min_bytes = 1 << 60
    video = torch.randn(2, 20, 3, 96, 96, generator=torch.Generator().manual_seed(5))
    lens = torch.tensor([20, 13])
    ref = oracle.visual_frontend(sd, "", video, lens, True, {})
    a, b = m(video.to(dev), lens.to(dev)), m2(video.to(dev), lens.to(dev))
    assert max_err(a, ref) < 2 * TOL[torch.bfloat16] and max_err(b, ref) < 2 * TOL[torch.bfloat16]
    sa, sb = m.state_dict(), m2.state_dict()
    for k in ("MoCoModel.layer1.0.bn3.running_mean", "MoCoModel.layer1.0.downsample.1.running_var"):
        assert torch.equal(sa[k], sb[k]), k                     # same input, same tiles, same fp32 accumulators
    for k in ("MoCoModel.layer1.2.bn3.running_var", "MoCoModel.layer2.1.bn3.running_mean"):
        assert max_err(sa[k], sb[k]) < 2e-3, k                   # later blocks see inputs with one rounding less
