"""MoCo v2 visual frontend (3D stem + ResNet-50 trunk) on the HIP kernels, with
the reference's parameter names (``frontend3D.{0,1}.*``, ``MoCoModel.layer{1-4}.
{j}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}.*``;
models/moco_visual_frontend.py:14-26, torchvision ``resnet50`` with
conv1/bn1/relu/maxpool/fc = Identity).

Frozen in the reference (models/av_net.py:45-46): forward only.  Activations are
NHWC; every convolution of the trunk is an implicit GEMM.  BatchNorm follows
``self.training`` like the reference does: eval = running statistics folded
into the GEMM epilogue; train = batch statistics accumulated in the epilogue,
one finalize + one in-place apply kernel per BN, running statistics updated in
place (the statistics come from the fp32 accumulators, before rounding).
"""
import math
import os

import torch
import torch.nn as nn

from . import ops

BLOCKS = (3, 4, 6, 3)
WIDTHS = (64, 128, 256, 512)


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, width, stride):
        super().__init__()
        out = width * 4
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        self.stride = stride
        if stride != 1 or inplanes != out:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(out))
        else:
            self.downsample = None


class _ResNet50Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        inplanes = 64
        for li, (n, w) in enumerate(zip(BLOCKS, WIDTHS)):
            blocks = []
            for j in range(n):
                blocks.append(_Bottleneck(inplanes, w, 2 if (li > 0 and j == 0) else 1))
                inplanes = w * 4
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        # torchvision's resnet50 initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def blocks(self):
        for li in range(4):
            yield from getattr(self, f"layer{li + 1}")


class MoCoVisualFrontend(nn.Module):
    def __init__(self, d_model=512, num_classes=500, frame_length=96, vid_feature_dim=512, enable_logging=False):
        super().__init__()
        self.enable_logging = enable_logging
        self.frontend3D = nn.Sequential(
            nn.Conv3d(3, 64, kernel_size=(5, 3, 3), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64),
            nn.ReLU(True),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)),
        )
        self.MoCoModel = _ResNet50Trunk()
        self.compute_dtype = torch.bfloat16
        self.two_pass_min_bytes = int(os.environ.get("AVSR_TWO_PASS_MIN_BYTES", 1 << 60))   # see _conv_bn
        # batch statistics of bn3 from the Gram matrix of conv3's input (see run_block); AVSR_GRAM_BN3=0 turns it off
        self.gram_bn3 = os.environ.get("AVSR_GRAM_BN3", "1") != "0"
        self.gram_widths = (64,) if os.environ.get("AVSR_GRAM_BN3") == "64" else (64, 128)
        self.gram_downsample = os.environ.get("AVSR_GRAM_DS", "1") != "0"      # layer1.0's shortcut BN the same way
        self.fuse_finalize = os.environ.get("AVSR_BN_FUSE_FINALIZE", "1") != "0"
        self.gram_applies_bn2 = os.environ.get("AVSR_GRAM_APPLIES_BN2", "1") != "0"
        self._packed = None
        self._stats_epoch = 0                   # bumped by every batch-statistics forward (the kernels write the buffers raw)
        self._tracked_pending = 0               # training-mode forwards not yet added to num_batches_tracked
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_counters())
        self._blocks = list(self.MoCoModel.blocks())

    def flush_counters(self):
        """nn.BatchNorm's ``num_batches_tracked`` buffers count training-mode forwards and feed nothing here (momentum is
        a number, not None): the count is kept on the host and added to the 54 buffers when someone looks - every
        ``state_dict()`` does - instead of by a launch at the end of every forward."""
        if self._tracked_pending:
            with torch.no_grad():
                torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                                     if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d))], self._tracked_pending)
            self._tracked_pending = 0

    def stats_version(self):
        """Changes whenever the BN running statistics may have: a training-mode forward (here or replayed from a
        recording) or an in-place write by the caller (load_state_dict, copy_)."""
        return (self._stats_epoch, sum(b._version for b in self.buffers()))

    # ---- weight packing -------------------------------------------------
    def _pack(self, dtype):
        key = (dtype, self.training, sum(p._version for p in self.parameters()),
               None if self.training else self.stats_version(), next(self.parameters()).data_ptr())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]

        def cw(conv):   # [Co, Ci, kh, kw] -> [Co, kh*kw*Ci] in the compute dtype
            w = conv.weight.detach()
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous()

        pk = {"stem_w": ops.stem_pack_weight(self.frontend3D[0].weight), "blocks": [], "fold": {}}
        # bf16: the stem runs as an implicit GEMM on MFMA and needs the weight in fragment order
        pk["stem_f"] = ops.stem_pack_frags(pk["stem_w"]) if dtype == torch.bfloat16 else None
        if not self.training:      # eval: fold every BN's running statistics once per weight version
            for m in self.modules():
                if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                    pk["fold"][id(m)] = ops.bn_fold(m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var, m.eps)
        for blk in self.MoCoModel.blocks():
            e = {"w1": cw(blk.conv1), "w2": cw(blk.conv2), "w3": cw(blk.conv3), "stride": blk.stride}
            if blk.downsample is not None:
                e["wd"] = cw(blk.downsample[0])
            pk["blocks"].append(e)
        self._packed = (key, pk)
        return pk

    def _fold(self, bn):
        return self._packed[1]["fold"][id(bn)]

    # ---- one conv + BN (+ReLU / + residual) ------------------------------------
    def _conv_bn(self, x, w, k, stride, pad, bn, relu, residual=None, residual_affine=None, defer=False):
        """``defer`` (train mode): return (raw conv output, (scale, shift)) without the apply pass, for a consumer that
        applies the affine itself; ``residual_affine``: such a deferred pair for the residual operand."""
        if not self.training:
            sc, sh = self._fold(bn)
            if residual is not None:
                act = ops.ACT_RELU_AFTER_RESIDUAL
                res2d = residual.view(-1, residual.shape[-1])
            else:
                act, res2d = (ops.ACT_RELU if relu else ops.ACT_NONE), None
            return ops.conv_nhwc(x, w, k, k, stride, pad, scale=sc, bias=sh, act=act, residual=res2d)
        C = w.shape[0]
        rows = ops.conv_out_rows(x, k, k, stride, pad)
        stats = ops.gemm_stats_slabs(rows, C, x.dtype, x.device)
        aff = torch.empty((2, C), dtype=torch.float32, device=x.device)
        # Option (off unless two_pass_min_bytes is lowered): expanding 1x1 convolutions with a short reduction
        # (layer1's conv3 / downsample: 64 -> 256 channels over 375 000 pixels) are bound by the 192 MB output
        # they store.  Instead of storing the raw output and re-reading it for the normalisation, run the product
        # twice: a first pass that only accumulates the column sums (no store), then the product again with scale /
        # shift, ReLU and the residual in its epilogue.  Same sums (same tiles, same fp32 accumulators), one
        # rounding less.  Measured in the whole step it does not pay (7.87 ms against 7.81 ms without): the
        # normalisation pass is pure HBM traffic that overlaps with the audio encoder's MFMA-bound kernels, the
        # second product is not.
        if (k == 1 and stride == 1 and pad == 0 and x.dtype == torch.bfloat16 and residual_affine is None
                and C >= 4 * w.shape[1] and rows * C * 2 >= self.two_pass_min_bytes):
            ops.conv_nhwc(x, w, 1, 1, 1, 0, col_stats=(stats[0], stats[1]), stats_only=True)
            ops.bn_finalize(stats[0], stats[1], rows, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                            bn.running_var, aff[0], aff[1], bn.eps, bn.momentum)
            if residual is not None:
                act, res2d = ops.ACT_RELU_AFTER_RESIDUAL, residual.view(-1, residual.shape[-1])
            else:
                act, res2d = (ops.ACT_RELU if relu else ops.ACT_NONE), None
            out = ops.conv_nhwc(x, w, 1, 1, 1, 0, scale=aff[0], bias=aff[1], act=act, residual=res2d)
            return (out, None) if defer else out
        raw = ops.conv_nhwc(x, w, k, k, stride, pad, col_stats=(stats[0], stats[1]))
        if (self.fuse_finalize and not defer and residual_affine is None and raw.dtype == torch.bfloat16 and C % 64 == 0
                and stats.shape[1] <= ops.BN_FUSED_APPLY_MAX_PARTS):
            # short slabs (the deep, small feature maps): the apply pass folds them itself, one launch instead of two
            return ops.bn_finalize_apply(raw, stats[0], stats[1], rows, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                         bn.running_var, residual=residual, relu=relu, out=raw, eps=bn.eps,
                                         momentum=bn.momentum)
        ops.bn_finalize(stats[0], stats[1], rows, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                        bn.running_var, aff[0], aff[1], bn.eps, bn.momentum)
        if defer:
            return raw, (aff[0], aff[1])
        return ops.bn_apply(raw, aff[0], aff[1], residual=residual, residual_affine=residual_affine, relu=relu, out=raw)

    @torch.no_grad()
    def run_stem(self, video, pk=None, video_slot=None):
        """frontend3D on fp32 video [B,T,3,H,W] -> NHWC [B*T, H/4, W/4, 64] in the compute dtype.
        ``video_slot``: see ops.stem3d_fwd (the input's address is read from that device word when the kernel runs)."""
        dtype = self.compute_dtype
        pk = pk or self._pack(dtype)
        bn0 = self.frontend3D[1]
        if not self.training:
            sc, sh = self._fold(bn0)
            return ops.maxpool3x3s2(ops.stem3d_fwd(video, pk["stem_w"], dtype, scale=sc, shift=sh, frags=pk["stem_f"],
                                                   video_slot=video_slot))
        stats = ops.stem_stats_slabs(video)
        raw = ops.stem3d_fwd(video, pk["stem_w"], dtype, col_stats=(stats[0], stats[1]), frags=pk["stem_f"],
                             video_slot=video_slot)
        aff = torch.empty((2, 64), dtype=torch.float32, device=video.device)
        ops.bn_finalize(stats[0], stats[1], raw.numel() // 64, bn0.weight.detach(), bn0.bias.detach(),
                        bn0.running_mean, bn0.running_var, aff[0], aff[1], bn0.eps, bn0.momentum)
        return ops.maxpool3x3s2(raw, scale=aff[0], shift=aff[1])

    @torch.no_grad()
    def run_block(self, index, y, pk=None, pool=None):
        """One Bottleneck (index 0..15 over layer1..layer4) on an NHWC tensor.  ``pool`` = (x_len, B, T), last block
        only: return the pooled, masked [B, T, C] features where the block can fold the pool into its last pass
        (the caller checks the rank of what comes back)."""
        pk = pk or self._pack(self.compute_dtype)
        e = pk["blocks"][index]
        blk = self._blocks[index]
        idt, idt_aff = y, None
        gram_ds = False
        if ("wd" in e and self.training and self.gram_bn3 and self.gram_downsample and e["stride"] == 1 and y.dtype == torch.bfloat16
                and y.shape[-1] in self.gram_widths and e["w3"].shape[1] in self.gram_widths):
            # layer1's first block: its 1 x 1 shortcut convolution reads the 64-channel block input at stride 1, so
            # the shortcut BN's batch statistics follow from the Gram matrix of that input as bn3's do from conv3's -
            # the shortcut is stored normalised, conv3 adds it in its epilogue, and neither raw 256-channel tensor
            # nor the three-stream apply pass over them exists
            gram_ds = True
            bnd = blk.downsample[1]
            Cin, C = y.shape[-1], e["wd"].shape[0]
            rows = y.numel() // Cin
            gp, cp = ops.gram_stats(y.view(rows, Cin))
            affd = torch.empty((2, C), dtype=torch.float32, device=y.device)
            red = ops.bn_gram_finalize(gp, cp, rows, e["wd"], bnd.weight.detach(), bnd.bias.detach(), bnd.running_mean,
                                       bnd.running_var, affd[0], affd[1], bnd.eps, bnd.momentum)
            idt = ops.conv_nhwc(y, e["wd"], 1, 1, 1, 0, scale=affd[0], bias=affd[1])
            # conv1 reads the same input: bn1's statistics come from the same Gram matrix, bn1 + ReLU ride in conv1's epilogue
            bn1 = blk.bn1
            aff1 = torch.empty((2, e["w1"].shape[0]), dtype=torch.float32, device=y.device)
            ops.bn_gram_finalize(gp, cp, rows, e["w1"], bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean,
                                 bn1.running_var, aff1[0], aff1[1], bn1.eps, bn1.momentum, reduced=red)
            h1 = ops.conv_nhwc(y, e["w1"], 1, 1, 1, 0, scale=aff1[0], bias=aff1[1], act=ops.ACT_RELU)
        elif "wd" in e:
            # batch statistics: the downsample BN has no activation, so its apply pass is folded into bn3's
            if self.training:
                idt, idt_aff = self._conv_bn(y, e["wd"], 1, e["stride"], 0, blk.downsample[1], relu=False, defer=True)
            else:
                idt = self._conv_bn(y, e["wd"], 1, e["stride"], 0, blk.downsample[1], relu=False)
        h = h1 if gram_ds else self._conv_bn(y, e["w1"], 1, 1, 0, blk.bn1, relu=True)
        width = e["w3"].shape[1]
        gram3 = (self.training and self.gram_bn3 and ("wd" not in e or gram_ds) and h.dtype == torch.bfloat16
                 and width in self.gram_widths)
        aff2 = None
        if gram3 and self.gram_applies_bn2:
            # bn2's apply pass rides in the Gram statistics pass below, which reads conv2's output anyway
            h, aff2 = self._conv_bn(h, e["w2"], 3, e["stride"], 1, blk.bn2, relu=True, defer=True)
        else:
            h = self._conv_bn(h, e["w2"], 3, e["stride"], 1, blk.bn2, relu=True)
        if gram3:
            # Identity-shortcut blocks of layer1 / layer2 (64 / 128 channels into conv3, whose 256- / 512-channel output
            # is what the block's HBM traffic consists of): the batch statistics of conv3's output follow from the
            # 64 x 64 / 128 x 128 Gram matrix of its INPUT (one pass over a tensor a quarter of the output's size), so
            # conv3 applies bn3, the shortcut and the ReLU in its epilogue and the raw product is never stored or
            # re-read (per block: one write and one read of the widest tensor less, and no apply pass).
            bn3 = blk.bn3
            C = e["w3"].shape[0]
            rows = h.numel() // width
            gp, cp = ops.gram_stats(h.view(rows, width), affine=aff2)
            aff = torch.empty((2, C), dtype=torch.float32, device=h.device)
            ops.bn_gram_finalize(gp, cp, rows, e["w3"], bn3.weight.detach(), bn3.bias.detach(), bn3.running_mean,
                                 bn3.running_var, aff[0], aff[1], bn3.eps, bn3.momentum)
            return ops.conv_nhwc(h, e["w3"], 1, 1, 1, 0, scale=aff[0], bias=aff[1], act=ops.ACT_RELU_AFTER_RESIDUAL,
                                 residual=idt.view(-1, C))
        if pool is not None and self.training and idt_aff is None and h.dtype == torch.bfloat16:
            # the trunk's last block: nothing but the average pool reads its output, so bn3's apply pass, the
            # shortcut and the ReLU ride in the pool and the normalised 2048-channel tensor is never written
            raw, aff3 = self._conv_bn(h, e["w3"], 1, 1, 0, blk.bn3, relu=False, defer=True)
            x_len, B, T = pool
            if aff3 is not None:
                return ops.bn_apply_avgpool_mask(raw, idt.view(raw.shape), aff3[0], aff3[1], x_len, B, T, self.compute_dtype)
            return ops.bn_apply(raw, torch.ones_like(blk.bn3.weight), torch.zeros_like(blk.bn3.weight), residual=idt,
                                relu=True, out=raw)                      # (two-pass option: already normalised)
        return self._conv_bn(h, e["w3"], 1, 1, 0, blk.bn3, relu=True, residual=idt, residual_affine=idt_aff)

    @torch.no_grad()
    def forward(self, x, x_len, video_slot=None):
        """x: [B,T,3,H,W] (or [B,1,T,3,H,W]) fp32, x_len: int64 [B] -> [B,T,2048]
        in the compute dtype, rows t >= x_len[b] zero.
        ``video_slot`` (one-element int64 GPU tensor): the stem reads the video's address from it when it runs and
        ``x`` only gives the shape - for a caller that records this forward once and replays it on other batches."""
        if x.dim() == 6 and x.shape[1] == 1:
            x = x.squeeze(1)
        B, T = x.shape[0], x.shape[1]
        video = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        if video.data_ptr() % 16 and video_slot is None:
            video = video.clone()           # the MFMA stem reads 16-byte chunks: an odd view gets an aligned copy, not another kernel
        if video_slot is not None and video is not x:
            raise ValueError("video_slot needs contiguous fp32 video (a converted copy would be read at a fixed address)")
        pk = self._pack(self.compute_dtype)        # one weight-version check per forward
        y = self.run_stem(video, pk, video_slot)
        last = len(self._blocks) - 1
        for i in range(last + 1):
            y = self.run_block(i, y, pk, pool=(x_len, B, T) if i == last else None)
        if self.training:
            self._stats_epoch += 1
            self._tracked_pending += 1      # the 54 num_batches_tracked counters: see flush_counters
        return y if y.dim() == 3 else ops.avgpool_mask(y, x_len, B, T, self.compute_dtype)
