"""Device-side counterpart of the reference's ``VideoTransform``
(datamodule/transforms.py:61-96) for decoded frames that are already on the GPU.

``VideoTransform("val" | "test")``: x / 255 then ImageNet normalisation (:94-95).
``VideoTransform("train")``: bilinear resize to 64 x 64, / 255, random horizontal
flip (p = 0.5), ColorJitter(0.4, 0.4, 0.4, 0.1), RandomGrayscale(0.2),
``AdaptiveTimeMask(10, 25)`` and the normalisation (:65-75), the random choices
drawn on the host.  The colour steps follow torchvision 0.19.1's tensor
implementation, restated without torchvision at hand (parity unpinned).
"""
import random

import torch

from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def adaptive_time_mask(length, window=10, stride=25, rng=random):
    """Frames ``AdaptiveTimeMask(window, stride)`` zeroes (datamodule/transforms.py:27-45), as a list of
    booleans.  One candidate mask per started ``stride`` frames.  All (t, t_end) pairs are drawn first, below
    ``window`` (:36); then, in order, each candidate with ``length - t > 0`` draws a start below ``length - t``
    (:40) and only after that draw is skipped when ``t`` is 0 (:41-42); otherwise ``t_end`` frames from the
    start are zeroed (:43-44: the second draw is the mask length)."""
    zero = [False] * length
    n_mask = int((length + stride - 0.1) // stride)
    pairs = [(rng.randrange(0, window), rng.randrange(0, window)) for _ in range(n_mask)]
    for t, t_len in pairs:
        if length - t <= 0:
            continue
        t_start = rng.randrange(0, length - t)
        if t == 0:
            continue
        for i in range(t_start, min(t_start + t_len, length)):
            zero[i] = True
    return zero


def color_draws(rng=random):
    """The draws of torchvision's ColorJitter(0.4, 0.4, 0.4, 0.1).get_params followed by RandomGrayscale(0.2), in
    their order (datamodule/transforms.py:71-72): a permutation of the four adjustments, brightness / contrast /
    saturation factors uniform in [0.6, 1.4], a hue shift uniform in [-0.1, 0.1], then the grayscale coin.
    Returns (order [4], factors [4], grey)."""
    order = list(range(4))
    rng.shuffle(order)
    factors = [rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(-0.1, 0.1)]
    return order, factors, rng.random() < 0.2


class VideoTransform:
    def __init__(self, subset, train_size=(64, 64)):
        if subset not in ("train", "val", "test"):
            raise ValueError("subset must be train, val or test")
        self.subset = subset
        self.train_size = train_size

    def __call__(self, video, *, rng=random, draws=None):
        """video: [T,3,H,W] or [B,T,3,H,W], uint8 or fp32 in 0..255, on the GPU.  ``draws`` overrides the random
        choices of the train subset: (flip [B], frame_zero [B][T]) - no colour steps - or (flip, frame_zero,
        order [B][4], factors [B][4], grey [B]) as ``color_draws`` returns them per clip."""
        if video.dim() not in (4, 5):
            raise ValueError(f"Expected 4D tensor (T,C,H,W), got shape {tuple(video.shape)}")
        squeeze = video.dim() == 4
        if squeeze:
            video = video.unsqueeze(0)
        if video.size(2) != 3:
            raise ValueError(f"Expected 3 channels, got {video.size(2)} channels")
        video = video.contiguous()
        if self.subset != "train":
            out = ops.video_preprocess(video, mean=IMAGENET_MEAN, std=IMAGENET_STD)
        else:
            B, T = video.shape[:2]
            if draws is None:                 # per clip, in the pipeline's order: flip, jitter, grayscale, time mask
                flips, zeros, orders, facs, greys = [], [], [], [], []
                for _ in range(B):
                    flips.append(rng.random() < 0.5)
                    o, f, g = color_draws(rng)
                    orders.append(o); facs.append(f); greys.append(g)
                    zeros.append(adaptive_time_mask(T, 10, 25, rng))
                draws = (flips, zeros, orders, facs, greys)
            dev = video.device
            flip = torch.tensor(draws[0], dtype=torch.uint8, device=dev)
            zero = torch.tensor(draws[1], dtype=torch.uint8, device=dev)
            if len(draws) == 2:
                out = ops.video_preprocess(video, size=self.train_size, flip=flip, frame_zero=zero, mean=IMAGENET_MEAN,
                                           std=IMAGENET_STD)
            else:
                # resize, /255 and flip first (values in [0, 1]); then the colour steps, the zeroed frames and the
                # normalisation in one pass over each frame
                out = ops.video_preprocess(video, size=self.train_size, flip=flip, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
                ops.video_color(out, order=torch.tensor(draws[2], dtype=torch.int32, device=dev),
                                factors=torch.tensor(draws[3], dtype=torch.float32, device=dev),
                                grey=torch.tensor(draws[4], dtype=torch.uint8, device=dev), frame_zero=zero,
                                mean=IMAGENET_MEAN, std=IMAGENET_STD)
        return out[0] if squeeze else out
