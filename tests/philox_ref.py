"""Host restatement (numpy) of the counter-based generator the HIP kernels use
for dropout (csrc/common.h: avsr_philox / avsr_keep), so tests can replay the
exact keep masks."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4(seed, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    c0 = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (idx >> np.uint64(32)).astype(np.uint32)
    c2 = np.full_like(c0, 0x243F6A88)
    c3 = np.full_like(c0, 0x85A308D3)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox(seed, idx):
    return philox4(seed, idx)[0]


def _keep(bits, p):
    return (bits >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) >= np.float32(p)


def attn_keep_mask(seed, shape, p):
    """Keep mask of attention-probability dropout for a [B,H,Tq,Tk] score tensor (common.h: avsr_attn_keep):
    element (row, key) uses output key&3 of counter row*ceil(Tk/4) + key//4."""
    B, H, Tq, Tk = shape
    tk4 = (Tk + 3) // 4
    rows = np.arange(B * H * Tq, dtype=np.uint64)[:, None]
    ctr = rows * np.uint64(tk4) + np.arange(tk4, dtype=np.uint64)[None, :]
    outs = philox4(seed, ctr.reshape(-1))
    bits = np.stack([o.reshape(B * H * Tq, tk4) for o in outs], axis=-1).reshape(B * H * Tq, tk4 * 4)[:, :Tk]
    return _keep(bits, p).reshape(shape)


def keep_mask(seed, shape, p):
    n = int(np.prod(shape))
    u = (philox(seed, np.arange(n, dtype=np.uint64)) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(shape)
