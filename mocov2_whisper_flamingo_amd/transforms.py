"""Device-side counterpart of the reference's ``VideoTransform``
(datamodule/transforms.py:61-96) for decoded frames that are already on the GPU.

``VideoTransform("val" | "test")``: x / 255 then ImageNet normalisation (:94-95).
``VideoTransform("train")``: bilinear resize to 64 x 64, / 255, random horizontal
flip (p = 0.5), ``AdaptiveTimeMask(10, 25)`` and the normalisation (:65-75), the
random choices drawn on the host exactly as the reference draws them.
ColorJitter and RandomGrayscale (:71-72; torchvision, not installed here) are not
reproduced: the train pipeline here is the reference's minus those two steps.
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


class VideoTransform:
    def __init__(self, subset, train_size=(64, 64)):
        if subset not in ("train", "val", "test"):
            raise ValueError("subset must be train, val or test")
        self.subset = subset
        self.train_size = train_size

    def __call__(self, video, *, rng=random, draws=None):
        """video: [T,3,H,W] or [B,T,3,H,W], uint8 or fp32 in 0..255, on the GPU.  ``draws`` = (flip list [B],
        frame_zero lists [B][T]) overrides the random choices of the train subset."""
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
            if draws is None:
                draws = ([rng.random() < 0.5 for _ in range(B)], [adaptive_time_mask(T, 10, 25, rng) for _ in range(B)])
            flip = torch.tensor(draws[0], dtype=torch.uint8, device=video.device)
            zero = torch.tensor(draws[1], dtype=torch.uint8, device=video.device)
            out = ops.video_preprocess(video, size=self.train_size, flip=flip, frame_zero=zero, mean=IMAGENET_MEAN,
                                       std=IMAGENET_STD)
        return out[0] if squeeze else out
