"""Oracle (test infrastructure): MoCo v2 visual frontend - 3D stem + ResNet-50
trunk.  fp32 CPU restatement of ``models/moco_visual_frontend.py:14-52`` and of
the torchvision 0.19.1 ``resnet50`` architecture it instantiates (Bottleneck,
expansion 4, blocks [3,4,6,3], stride on the 3x3 conv, eps 1e-5, momentum 0.1).

PARITY UNPINNED for the trunk: torchvision is not installed and the reference
holds no fixture for it; the restatement is built from core ``torch.nn.functional``
convolution / batch-norm / pooling calls.
"""
import torch
import torch.nn.functional as F

RESNET50_BLOCKS = (3, 4, 6, 3)
RESNET50_WIDTHS = (64, 128, 256, 512)


def _bn(sd, pfx, x, training, new_stats, momentum=0.1, eps=1e-5):
    """BatchNorm{2,3}d.  In training mode batch statistics normalise and the
    running statistics are updated (unbiased variance) into ``new_stats``."""
    w, b = sd[pfx + "weight"], sd[pfx + "bias"]
    rm, rv = sd[pfx + "running_mean"], sd[pfx + "running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)
    rm2, rv2 = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm2, rv2, w, b, True, momentum, eps)
    if new_stats is not None:
        new_stats[pfx + "running_mean"] = rm2
        new_stats[pfx + "running_var"] = rv2
        nbt = sd.get(pfx + "num_batches_tracked")
        if nbt is not None:
            new_stats[pfx + "num_batches_tracked"] = nbt + 1
    return y


def bottleneck(sd, pfx, x, stride, training, new_stats):
    """torchvision ``Bottleneck.forward`` (v1.5: the stride sits on conv2)."""
    idt = x
    y = F.conv2d(x, sd[pfx + "conv1.weight"])
    y = F.relu(_bn(sd, pfx + "bn1.", y, training, new_stats))
    y = F.conv2d(y, sd[pfx + "conv2.weight"], stride=stride, padding=1)
    y = F.relu(_bn(sd, pfx + "bn2.", y, training, new_stats))
    y = F.conv2d(y, sd[pfx + "conv3.weight"])
    y = _bn(sd, pfx + "bn3.", y, training, new_stats)
    if (pfx + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[pfx + "downsample.0.weight"], stride=stride)
        idt = _bn(sd, pfx + "downsample.1.", idt, training, new_stats)
    return F.relu(y + idt)


def stem3d(sd, pfx, x, training, new_stats):
    """``frontend3D`` (moco_visual_frontend.py:14-19) on [B,3,T,H,W]."""
    y = F.conv3d(x, sd[pfx + "0.weight"], stride=(1, 2, 2), padding=(2, 3, 3))
    y = F.relu(_bn(sd, pfx + "1.", y, training, new_stats))
    return F.max_pool3d(y, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))


def resnet_trunk(sd, pfx, x, training, new_stats):
    """layers1-4 + global average pool of ``resnet50`` whose conv1/bn1/relu/
    maxpool/fc are Identity (moco_visual_frontend.py:20-26)."""
    for li, nblk in enumerate(RESNET50_BLOCKS):
        for j in range(nblk):
            stride = 2 if (li > 0 and j == 0) else 1
            x = bottleneck(sd, pfx + f"layer{li + 1}.{j}.", x, stride, training, new_stats)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def visual_frontend(sd, pfx, video, x_len, training=False, new_stats=None):
    """``MoCoVisualFrontend.forward`` (moco_visual_frontend.py:29-52):
    [B,T,3,H,W] (or [B,1,T,3,H,W]) -> [B,T,2048] with rows t >= x_len zeroed."""
    x = video
    if x.dim() == 6 and x.shape[1] == 1:
        x = x.squeeze(1)
    x = x.permute(0, 2, 1, 3, 4)
    x = stem3d(sd, pfx + "frontend3D.", x, training, new_stats)
    B, C, T, H, W = x.shape
    x = x.transpose(1, 2).contiguous().view(B * T, C, H, W)
    mask = torch.arange(T).expand(B, T) >= x_len.unsqueeze(1)
    x = resnet_trunk(sd, pfx + "MoCoModel.", x, training, new_stats)
    x = x.view(B, T, -1)
    return x.masked_fill(mask.unsqueeze(-1), 0.0)
