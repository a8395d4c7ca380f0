"""Host restatement (numpy) of the counter-based generator the HIP kernels use
for dropout (csrc/common.h: avsr_rand4 / avsr_keep / avsr_attn_keep), so tests can replay the
exact keep masks.  (The module keeps its first name: an earlier generator was Philox-4x32-10.)"""
import numpy as np

_U32 = np.uint32


def _mix32(h):
    with np.errstate(over="ignore"):
        h = h ^ (h >> _U32(16))
        h = h * _U32(0x7FEB352D)
        h = h ^ (h >> _U32(15))
        h = h * _U32(0x846CA68B)
        h = h ^ (h >> _U32(16))
    return h


def rand4(seed, idx):
    """Four 16-bit draws per counter: arrays shaped like ``idx``."""
    idx = np.asarray(idx, dtype=np.uint64)
    c0 = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (idx >> np.uint64(32)).astype(np.uint32)
    s0, s1 = _U32(seed & 0xFFFFFFFF), _U32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        base = s0 + c1 * _U32(0x85EBCA77)
        h0 = _mix32((c0 * _U32(0x9E3779B1) + base) ^ s1)
        h1 = _mix32((c0 * _U32(0xC2B2AE3D) + base) ^ (~s1))
    return h0 & _U32(0xFFFF), h0 >> _U32(16), h1 & _U32(0xFFFF), h1 >> _U32(16)


def _keep(draws, p):
    return draws >= _U32(int(np.float32(p) * np.float32(65536.0)))


def attn_keep_mask(seed, shape, p):
    """Keep mask of attention-probability dropout for a [B,H,Tq,Tk] score tensor (common.h: avsr_attn_keep):
    element (row, key) uses output key&3 of counter row*ceil(Tk/4) + key//4."""
    B, H, Tq, Tk = shape
    tk4 = (Tk + 3) // 4
    rows = np.arange(B * H * Tq, dtype=np.uint64)[:, None]
    ctr = rows * np.uint64(tk4) + np.arange(tk4, dtype=np.uint64)[None, :]
    outs = rand4(seed, ctr.reshape(-1))
    bits = np.stack([o.reshape(B * H * Tq, tk4) for o in outs], axis=-1).reshape(B * H * Tq, tk4 * 4)[:, :Tk]
    return _keep(bits, p).reshape(shape)


def keep_mask(seed, shape, p):
    """Keep mask of a flat tensor (common.h: avsr_keep): element i uses output i&3 of counter i//4."""
    n = int(np.prod(shape))
    outs = rand4(seed, np.arange((n + 3) // 4, dtype=np.uint64))
    draws = np.stack(outs, axis=-1).reshape(-1)[:n]
    return _keep(draws, p).reshape(shape)
