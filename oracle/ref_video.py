"""Oracle (test infrastructure): the training-time VideoTransform of the reference
(datamodule/transforms.py:61-96) on one clip [T,3,H,W], with every random draw passed in.

The reference builds the pipeline from torchvision 0.19.1 transforms (RandomHorizontalFlip, ColorJitter(0.4, 0.4,
0.4, 0.1), RandomGrayscale(0.2), Normalize).  torchvision is not installed here and the reference holds no fixture
for this path: the colour operations below restate torchvision/transforms/_functional_tensor.py (adjust_brightness /
adjust_contrast / adjust_saturation / adjust_hue, rgb_to_grayscale, _rgb2hsv, _hsv2rgb, _blend) for float images with
bound 1.0 - PARITY UNPINNED.  Only core torch is used."""
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def _blend(img1, img2, ratio):
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0)


def adjust_brightness(img, factor):
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)      # one grey level per image
    return _blend(img, mean, factor)


def adjust_saturation(img, factor):
    return _blend(img, rgb_to_grayscale(img), factor)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - (s * f)), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, shift):
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + shift) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def color_jitter(clip, order, factors):
    """torchvision ColorJitter.forward on a clip [T,3,H,W] in [0,1]: the four adjustments in the drawn ``order``
    (a permutation of 0 brightness, 1 contrast, 2 saturation, 3 hue; negative = skipped) with ``factors`` =
    (brightness, contrast, saturation, hue shift).  The contrast step uses one mean grey level PER FRAME."""
    fns = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)
    for op in order:
        if 0 <= op < 4:
            clip = fns[op](clip, factors[op])
    return clip


def video_transform_train(clip, flip, order, factors, grey, frame_zero, size=(64, 64)):
    """transforms.py:65-75 on a clip [T,3,H,W] with values 0..255: bilinear resize, /255, horizontal flip,
    ColorJitter, RandomGrayscale, AdaptiveTimeMask (``frame_zero``: booleans per frame), ImageNet normalisation."""
    x = F.interpolate(clip.float(), size=size, mode="bilinear", align_corners=False) / 255.0
    if flip:
        x = x.flip(-1)
    x = color_jitter(x, order, factors)
    if grey:
        x = rgb_to_grayscale(x).expand(-1, 3, -1, -1).clone()
    x = x.clone()
    x[torch.tensor(frame_zero, dtype=torch.bool)] = 0
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (x - mean) / std
