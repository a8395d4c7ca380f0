"""Per-kernel parity tests: every C-ABI entry point against a plain PyTorch fp32
CPU computation of the same op.  Tolerances: fp32 kernels 1e-4 relative to the
output scale, bf16 kernels 2e-2 (bf16 inputs are rounded BEFORE the reference
is computed, so only accumulation order and output rounding differ)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
from mocov2_whisper_flamingo_amd import ops  # noqa: E402
from tests.philox_ref import attn_keep_mask, keep_mask  # noqa: E402
from tests.helpers import assert_close  # noqa: E402


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def tol(dtype):
    return 2e-2 if dtype == torch.bfloat16 else 2e-4


def rnd(gen, *shape, dtype=torch.float32, scale=1.0):
    x = torch.randn(*shape, generator=gen) * scale
    return x.to(dtype).float() if dtype == torch.bfloat16 else x


@pytest.mark.parametrize("M,N,K,odt", [(12000, 768, 768, torch.float32), (11990, 700, 264, torch.bfloat16),
                                       (12000, 768, 3072, torch.float32)])
def test_gemm_row_tile_96_160(dev, M, N, K, odt):
    """Shapes for which the host leaves the 128-row tile for one that fills the last round of workgroups better
    (564 tiles of 128 rows -> 450 of 160; 96-row tiles from 1000 tiles on): bias + GELU
    (bf16 out) or bias + in-place fp32 residual, ragged M / N / K included."""
    g = torch.Generator().manual_seed(M + N + K)
    a, b = rnd(g, M, K, dtype=torch.bfloat16), rnd(g, N, K, dtype=torch.bfloat16, scale=0.05)
    bias = torch.randn(N, generator=g)
    ref = a @ b.T + bias
    ld = (N + 7) // 8 * 8
    if odt == torch.float32:
        res = torch.randn(M, N, generator=g)
        ref = res + ref
        buf = torch.zeros(M, ld, device=dev)
        buf[:, :N] = res.to(dev)
        out = ops.gemm_nt(a.to(dev, torch.bfloat16), b.to(dev, torch.bfloat16), out=buf[:, :N], bias=bias.to(dev),
                          residual=buf[:, :N])
    else:
        ref = F.gelu(ref)
        out = ops.gemm_nt(a.to(dev, torch.bfloat16), b.to(dev, torch.bfloat16), bias=bias.to(dev), act=ops.ACT_GELU)
    assert rel_err(out, ref) < 1e-2


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_gemm_row_tile_256(dev, odt):
    """A product large enough for the 256-row tile (>= 1000 such tiles, K >= 2048), ragged in M and N, with bias +
    residual; checked on a sample of rows (the full fp32 product on the CPU would take minutes)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 8192 - 37, 4096 - 24, 2048
    a, b = rnd(g, M, K, dtype=torch.bfloat16), rnd(g, N, K, dtype=torch.bfloat16, scale=0.05)
    bias, res = torch.randn(N, generator=g), rnd(g, M, N, dtype=odt)
    out = ops.gemm_nt(a.to(dev, torch.bfloat16), b.to(dev, torch.bfloat16), bias=bias.to(dev), residual=res.to(dev, odt),
                      out_dtype=odt)
    rows = torch.cat([torch.arange(0, 300), torch.randint(300, M - 300, (500,), generator=g), torch.arange(M - 300, M)])
    ref = a[rows] @ b.T + bias + res[rows]
    assert rel_err(out[rows.to(dev)], ref) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(300, 200, 64), (128, 128, 128), (257, 1001, 512), (75, 2048, 520), (1, 64, 8)])
def test_gemm_plain(dev, dtype, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a, b = rnd(g, M, K, dtype=dtype), rnd(g, N, K, dtype=dtype)
    ref = a @ b.T
    out = ops.gemm_nt(a.to(dev, dtype), b.to(dev, dtype), out_dtype=torch.float32)
    assert rel_err(out, ref) < tol(dtype) * 0.5
    out2 = ops.gemm_nt(a.to(dev, dtype), b.to(dev, dtype))
    assert out2.dtype == dtype and rel_err(out2, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_asymmetric_identity(dev, dtype):
    # A = I with an asymmetric B catches a transposed accumulator write
    n = 160
    a = torch.eye(n)
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 100.0
    out = ops.gemm_nt(a.to(dev, dtype), b.to(dev, dtype), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), b.T.contiguous())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_epilogue(dev, dtype):
    g = torch.Generator().manual_seed(3)
    M, N, K = 333, 264, 192
    a, b = rnd(g, M, K, dtype=dtype), rnd(g, N, K, dtype=dtype, scale=0.1)
    bias, scale = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    res = torch.randn(M, N, generator=g)
    gate = torch.tensor([0.7])
    pre = (a @ b.T) * 0.5 * scale + bias
    ref = res + math.tanh(0.7) * F.gelu(pre)
    out = ops.gemm_nt(a.to(dev, dtype), b.to(dev, dtype), out_dtype=torch.float32, bias=bias.to(dev), scale=scale.to(dev),
                      act=ops.ACT_GELU, residual=res.to(dev), gate=gate.to(dev), alpha=0.5)
    assert rel_err(out, ref) < tol(dtype)
    # periodic residual (positional table) + relu + column statistics
    table = torch.randn(37, N, generator=g)
    cs = ops.gemm_stats_slabs(M, N, dtype, dev)
    out = ops.gemm_nt(a.to(dev, dtype), b.to(dev, dtype), out_dtype=torch.float32, bias=bias.to(dev), act=ops.ACT_RELU,
                      residual=table.to(dev), res_period=37, col_stats=(cs[0], cs[1]))
    cs = cs.sum(1)
    pre = a @ b.T + bias
    ref = table[torch.arange(M) % 37] + F.relu(pre)
    assert rel_err(out, ref) < tol(dtype)
    assert rel_err(cs[0], pre.sum(0)) < 1e-3
    assert rel_err(cs[1], (pre * pre).sum(0)) < 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_batched_strided(dev, dtype):
    g = torch.Generator().manual_seed(4)
    Bt, T, D, N = 3, 50, 64, 96
    big = rnd(g, Bt, 200, D, dtype=dtype)         # use the first T rows of each batch item
    w = rnd(g, N, D, dtype=dtype)
    ref = big[:, :T] @ w.T
    a = big.to(dev, dtype)[:, :T]
    out = ops.gemm_nt(a, w.to(dev, dtype), out_dtype=torch.float32)
    assert out.shape == (Bt, T, N) and rel_err(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cfg", [
    # n, H, W, C, Cout, KH, KW, stride, pad
    (3, 25, 25, 64, 64, 3, 3, 1, 1),
    (2, 25, 25, 128, 128, 3, 3, 2, 1),
    (2, 13, 13, 256, 512, 1, 1, 2, 0),
    (5, 7, 7, 64, 256, 1, 1, 1, 0),
    (2, 1, 300, 80, 96, 1, 3, 1, 1),      # Whisper conv1 as a 1-row image
    (2, 1, 300, 96, 96, 1, 3, 2, 1),      # Whisper conv2 (stride 2)
])
def test_conv_implicit_gemm(dev, dtype, cfg):
    n, H, W, C, Co, KH, KW, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = rnd(g, n, C, H, W, dtype=dtype)
    w = rnd(g, Co, C, KH, KW, dtype=dtype, scale=1.0 / math.sqrt(C * KH * KW))
    bias = torch.randn(Co, generator=g)
    pad = (0, p) if H == 1 else p
    ref = F.relu(F.conv2d(x, w, bias, stride=s, padding=pad)).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    wd = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(dev, dtype)
    out = ops.conv_nhwc(xd, wd, KH, KW, s, pad, out_dtype=torch.float32, bias=bias.to(dev), act=ops.ACT_RELU)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < tol(dtype)


@pytest.mark.parametrize("cfg", [
    # n, H, W, C, Cout, KH, KW, stride, pad: 9600 output rows x 512 channels = 300 tiles of 128 rows, 240 of 160
    (600, 4, 4, 64, 512, 3, 3, 1, 1),
    (600, 7, 7, 64, 512, 3, 3, 2, 1),
    (600, 4, 4, 512, 512, 1, 1, 1, 0),
    (601, 4, 4, 64, 512, 3, 3, 1, 1),      # 9616 rows: the last 160-row tile is ragged
    # 101400 rows x 128 channels = 793 tiles of 128 rows, 634 of 160: one round at three workgroups per CU
    (600, 13, 13, 64, 128, 3, 3, 1, 1),
    (600, 13, 13, 128, 128, 1, 1, 1, 0),
    (599, 13, 13, 64, 128, 3, 3, 1, 1),    # 101231 rows: ragged last tile
])
def test_conv_one_round_of_tall_tiles_with_statistics(dev, cfg):
    """Grids of slightly more than one 128-row tile per CU (or between two and four of them) run 160-row tiles; the
    statistics slabs keep their 128-row-tile height, the rows no tile owns must come back as zeros (poisoned here)."""
    n, H, W, C, Co, KH, KW, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = rnd(g, n, C, H, W, dtype=torch.bfloat16)
    w = rnd(g, Co, C, KH, KW, dtype=torch.bfloat16, scale=1.0 / math.sqrt(C * KH * KW))
    scale = torch.rand(Co, generator=g) + 0.5
    raw = F.conv2d(x, w, None, stride=s, padding=p).permute(0, 2, 3, 1)
    M = raw.numel() // Co
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    wd = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(dev, torch.bfloat16)
    cs = ops.gemm_stats_slabs(M, Co, torch.bfloat16, dev)
    cs.fill_(float("nan"))
    out = ops.conv_nhwc(xd, wd, KH, KW, s, p, out_dtype=torch.bfloat16, col_stats=(cs[0], cs[1]))
    assert rel_err(out, raw) < tol(torch.bfloat16)
    tot = cs.sum(1)
    assert torch.isfinite(tot).all()
    flat = raw.reshape(M, Co)
    assert rel_err(tot[0], flat.sum(0)) < 1e-3
    assert rel_err(tot[1], (flat * flat).sum(0)) < 1e-3
    # with a residual and the ReLU behind it (the identity blocks' last convolution)
    res = rnd(g, M, Co, dtype=torch.bfloat16)
    out = ops.conv_nhwc(xd, wd, KH, KW, s, p, out_dtype=torch.bfloat16, scale=scale.to(dev),
                        act=ops.ACT_RELU_AFTER_RESIDUAL, residual=res.to(dev, torch.bfloat16).view(out.shape))
    ref = F.relu(flat * scale + res).view(raw.shape)
    assert rel_err(out, ref) < tol(torch.bfloat16)


@pytest.mark.parametrize("idt,odt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("D", [384, 512, 768, 1024, 1280])
def test_layernorm_fwd_bwd(dev, idt, odt, D):
    g = torch.Generator().manual_seed(D)
    rows = 77
    x = rnd(g, rows, D, dtype=idt) * 2 + 0.3
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    pe = torch.randn(10, D, generator=g)
    ref = F.layer_norm(x, (D,), gamma, beta) + pe[torch.arange(rows) % 10]
    y, mean, rstd = ops.layernorm_fwd(x.to(dev, idt), gamma.to(dev), beta.to(dev), out_dtype=odt, add=pe.to(dev),
                                      save_stats=True)
    assert rel_err(y, ref) < tol(odt)
    if idt == odt:
        xr = x.clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        dy = rnd(g, rows, D, dtype=odt)
        F.layer_norm(xr, (D,), gr, br).backward(dy)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dx = ops.layernorm_bwd(dy.to(dev, odt), x.to(dev, idt), gamma.to(dev), mean, rstd, dg, db)
        assert rel_err(dx, xr.grad) < tol(odt)
        gt = 1e-3 if odt == torch.float32 else 1e-2
        assert rel_err(dg, gr.grad) < gt and rel_err(db, br.grad) < gt


def _attn_ref(q, k, v, H, mask, scale, keep=None, p=0.0):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qh = q.view(B, Tq, H, 64).transpose(1, 2)
    kh = k.view(B, Tk, H, 64).transpose(1, 2)
    vh = v.view(B, Tk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :].bool(), float("-inf"))
    dead = torch.isinf(s).all(-1, keepdim=True)
    pr = torch.softmax(s.masked_fill(dead, 0.0), -1).masked_fill(dead, 0.0)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ vh).transpose(1, 2).reshape(B, Tq, H * 64)


@pytest.mark.parametrize("Tq,Tk", [(1500, 1500), (75, 75), (130, 257), (20, 20), (64, 64), (33, 100), (200, 129)])
def test_attn_mfma_fwd(dev, Tq, Tk):
    g = torch.Generator().manual_seed(Tq)
    B, H = 2, 3
    qkv = rnd(g, B, max(Tq, Tk), 3 * H * 64, dtype=torch.bfloat16)
    mask = None
    if Tq == 75:
        mask = torch.zeros(B, Tk, dtype=torch.uint8)
        mask[0, 40:] = 1
        mask[1, :] = 1      # every key masked: rows must come out as zeros
    q, k, v = qkv[:, :Tq, :H * 64], qkv[:, :Tk, H * 64:2 * H * 64], qkv[:, :Tk, 2 * H * 64:]
    ref = _attn_ref(q.contiguous(), k.contiguous(), v.contiguous(), H, mask, 0.125)
    d = qkv.to(dev, torch.bfloat16)
    dq = d[:, :Tq].contiguous() if Tq != Tk else d      # column slices of a fused QKV buffer stay views
    dkv = d[:, :Tk].contiguous() if Tq != Tk else d
    o, lse = ops.attn_fwd(dq[:, :, :H * 64], dkv[:, :, H * 64:2 * H * 64], dkv[:, :, 2 * H * 64:], H,
                          key_mask=None if mask is None else mask.to(dev), scale=0.125, use_mfma=True, save_lse=True)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < 2e-2
    if mask is None:
        s = (q.view(B, Tq, H, 64).transpose(1, 2) @ k.reshape(B, Tk, H, 64).transpose(1, 2).transpose(-1, -2)) * 0.125
        assert rel_err(lse, torch.logsumexp(s, -1)) < 1e-2


@pytest.mark.parametrize("M,N,K", [(600, 512, 512), (600, 2048, 512), (600, 51865, 512), (75, 100, 72), (1, 64, 64),
                                   (1000, 520, 2048), (3200, 512, 512), (5000, 200, 136), (12000, 768, 768)])
def test_gemm_tn_weight_gradient(dev, M, N, K):
    """dW = dY^T X and db = column sums of dY from the row-major operands (no transposed copies), written or
    accumulated; ragged M / N / K and a padded row pitch (N = 51865 in rows of 51872)."""
    g = torch.Generator().manual_seed(M + N)
    ldn, ldk = (N + 7) // 8 * 8, (K + 7) // 8 * 8
    dy = torch.full((M, ldn), 7.0)                 # padding columns hold junk: they must not reach the results
    x = torch.full((M, ldk), -3.0)
    dy[:, :N] = rnd(g, M, N, dtype=torch.bfloat16)
    x[:, :K] = rnd(g, M, K, dtype=torch.bfloat16)
    dyd, xd = dy.to(dev, torch.bfloat16)[:, :N], x.to(dev, torch.bfloat16)[:, :K]
    assert ops.gemm_tn_ok(dyd, xd)
    ref_w = dy[:, :N].double().t() @ x[:, :K].double()
    ref_b = dy[:, :N].double().sum(0)
    cs = torch.empty(N, device=dev)
    w = ops.gemm_tn(dyd, xd, colsum=cs)
    assert rel_err(w, ref_w) < 1e-5 and rel_err(cs, ref_b) < 1e-5
    w0 = rnd(g, N, K).to(dev)
    b0 = rnd(g, N).to(dev)
    w1, b1 = w0.clone(), b0.clone()
    ops.gemm_tn(dyd, xd, out=w1, accumulate=True, colsum=b1, accumulate_colsum=True)
    assert rel_err(w1, w0.cpu().double() + ref_w) < 1e-5 and rel_err(b1, b0.cpu().double() + ref_b) < 1e-5
    ld = K + 3                                      # a row slice of a wider fp32 buffer (the flat gradient buffer)
    wide = torch.zeros(N, ld, device=dev)
    ops.gemm_tn(dyd, xd, out=wide[:, :K])
    assert rel_err(wide[:, :K], ref_w) < 1e-5 and wide[:, K:].abs().max() == 0


@pytest.mark.parametrize("B,H,Tq,Tk,masked,p", [
    (8, 12, 1500, 1500, False, 0.0),     # Whisper-small at batch 8: 144 items per XCD = one full round + half a round
    (5, 8, 2560, 1100, True, 0.0),       # 100 items per XCD: the last 4 are cut four ways; masked keys in one share
    (4, 6, 500, 500, False, 0.0),        # fewer items than resident workgroups: every item cut in two
    (4, 6, 500, 520, True, 0.25),        # the dropout form; the mask is a function of the absolute key index
])
def test_attn_fwd_split_last_round(dev, B, H, Tq, Tk, masked, p):
    """With a workspace the forward hands the items of a half-empty last round of workgroups to several workgroups
    each (shares of the keys, summed by the last to finish).  Same numbers as the one-workgroup form up to the fp32
    summation order, twice in a row on one workspace (the counters must come back to zero)."""
    lib = ops._lib.load()
    assert lib.avsr_attn_fwd_workspace_bytes(B, H, Tq, Tk, p, 0) > 0, "shape chosen to be split"
    assert lib.avsr_attn_fwd_workspace_bytes(B, H, Tq, Tk, p, 1) == 0, "never with a causal mask"
    g = torch.Generator().manual_seed(B * Tq)
    q = rnd(g, B, Tq, H * 64).to(dev, torch.bfloat16)
    kv = rnd(g, B, Tk, 2 * H * 64).to(dev, torch.bfloat16)
    mask = None
    if masked:
        mask = torch.zeros(B, Tk, dtype=torch.uint8)
        mask[0, Tk // 3:] = 1                   # a whole share of the keys is masked for clip 0
        mask[1, :] = 1                          # and clip 1 has no key at all: zeros, lse = +inf
        mask = mask.to(dev)
    kw = dict(key_mask=mask, scale=0.125, use_mfma=True, save_lse=True, p_drop=p, seed=77)
    o0, l0 = ops.attn_fwd(q, kv[:, :, :H * 64], kv[:, :, H * 64:], H, **kw)
    ws = ops.AttnWorkspace()
    for _ in range(2):
        o1, l1 = ops.attn_fwd(q, kv[:, :, :H * 64], kv[:, :, H * 64:], H, workspace=ws, **kw)
        assert ws._buf is not None
        assert torch.isfinite(o1.float()).all()
        assert (o1.float() - o0.float()).abs().max() <= 2 ** -7 * o0.float().abs().max()      # a bf16 ulp or two
        assert rel_err(o1, o0) < 1e-2      # the shares round P to bf16 against their own reference maxima
        fin = torch.isfinite(l0)
        assert torch.equal(fin, torch.isfinite(l1))
        assert (l1[fin] - l0[fin]).abs().max() < 1e-4
    if masked:
        assert o1[1].abs().max() == 0


def test_attn_fwd_split_not_taken_on_a_cu_masked_stream(dev):
    """The key split relies on workgroups with equal (id & 7) sharing an XCD; a stream restricted to part of the CUs
    (hipExtStreamCreateWithCUMask) gives no such placement, so the forward must run unsplit there: bit-identical
    to the call without a workspace, and the workspace's tickets stay untouched."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)(*([0xffffffff] * 4 + [0] * 4))          # the lower half of the CUs
    sp = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words) == 0
    stream = torch.cuda.ExternalStream(sp.value)
    B, H, Tq, Tk = 4, 6, 500, 500
    g = torch.Generator().manual_seed(5)
    q = rnd(g, B, Tq, H * 64).to(dev, torch.bfloat16)
    kv = rnd(g, B, Tk, 2 * H * 64).to(dev, torch.bfloat16)
    kw = dict(scale=0.125, use_mfma=True, save_lse=True)
    o0, l0 = ops.attn_fwd(q, kv[:, :, :H * 64], kv[:, :, H * 64:], H, **kw)
    ws = ops.AttnWorkspace()
    ops.attn_fwd(q, kv[:, :, :H * 64], kv[:, :, H * 64:], H, workspace=ws, **kw)      # allocates and zeroes it
    torch.cuda.synchronize()
    ws._buf.fill_(0x5A)                            # a split launch would spin on (or trip over) these tickets
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        o1, l1 = ops.attn_fwd(q, kv[:, :, :H * 64], kv[:, :, H * 64:], H, workspace=ws, **kw)
    stream.synchronize()
    assert torch.equal(o1, o0) and torch.equal(l1, l0)
    assert int((ws._buf != 0x5A).sum()) == 0
    ws._buf.zero_()
    assert hip.hipStreamDestroy(sp) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_attn_valu_fwd_bwd(dev, dtype, p):
    g = torch.Generator().manual_seed(11)
    B, H, Tq, Tk = 2, 2, 75, 70
    q, k, v = (rnd(g, B, t, H * 64, dtype=dtype) for t in (Tq, Tk, Tk))
    mask = torch.zeros(B, Tk, dtype=torch.uint8)
    mask[0, 50:] = 1
    seed = 0x1234ABCD5678
    keep = torch.from_numpy(attn_keep_mask(seed, (B, H, Tq, Tk), p)).float() if p > 0 else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, H, mask, 0.125, keep, p)
    do = rnd(g, B, Tq, H * 64, dtype=dtype)
    ref.backward(do)
    qd, kd, vd = (t.to(dev, dtype) for t in (q, k, v))
    o, lse = ops.attn_fwd(qd, kd, vd, H, key_mask=mask.to(dev), scale=0.125, save_lse=True, p_drop=p, seed=seed)
    assert rel_err(o, ref) < tol(dtype)
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, o, do.to(dev, dtype), lse, H, key_mask=mask.to(dev), scale=0.125, p_drop=p,
                              seed=seed)
    assert rel_err(dq, qr.grad) < tol(dtype) * 2
    assert rel_err(dk, kr.grad) < tol(dtype) * 2
    assert rel_err(dv, vr.grad) < tol(dtype) * 2


@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(2, 2, 75, 70, True), (1, 3, 200, 333, True), (2, 8, 75, 75, False),
                                              (1, 1, 129, 64, True)])
def test_attn_mfma_fwd_bwd(dev, p, B, H, Tq, Tk, masked):
    """bf16 MFMA forward + backward with key mask and replayed dropout against the fp32 computation on the
    bf16-rounded operands; one batch row has every key masked (output and gradients must be zero there)."""
    g = torch.Generator().manual_seed(17 + Tq)
    dtype = torch.bfloat16
    q, k, v = (rnd(g, B, t, H * 64, dtype=dtype) for t in (Tq, Tk, Tk))
    mask = torch.zeros(B, Tk, dtype=torch.uint8)
    if masked:
        mask[0, Tk // 2:] = 1
        mask[0, 3] = 1
        if B > 1:
            mask[1, :] = 1
    seed = 0x9E3779B97F4A7C15
    keep = torch.from_numpy(attn_keep_mask(seed, (B, H, Tq, Tk), p)).float() if p > 0 else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, H, mask, 0.125, keep, p)
    do = rnd(g, B, Tq, H * 64, dtype=dtype)
    ref.backward(do)
    qd, kd, vd = (t.to(dev, dtype) for t in (q, k, v))
    md = mask.to(dev) if masked else None
    o, lse = ops.attn_fwd(qd, kd, vd, H, key_mask=md, scale=0.125, use_mfma=True, save_lse=True, p_drop=p, seed=seed)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < 2e-2
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, o, do.to(dev, dtype), lse, H, key_mask=md, scale=0.125, use_mfma=True,
                              p_drop=p, seed=seed)
    for got, want in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        assert torch.isfinite(got.float()).all()
        assert rel_err(got, want) < 3e-2
    if masked and B > 1:
        assert float(o[1].float().abs().max()) == 0.0 and float(dq[1].float().abs().max()) == 0.0
        assert float(dk[1].float().abs().max()) == 0.0 and float(dv[1].float().abs().max()) == 0.0
    # the MFMA and the fp32-math kernels draw the same dropout mask
    o2, lse2 = ops.attn_fwd(qd, kd, vd, H, key_mask=md, scale=0.125, use_mfma=False, save_lse=True, p_drop=p, seed=seed)
    assert rel_err(o, o2) < 2e-2
    fin = torch.isfinite(lse2)
    assert torch.equal(fin, torch.isfinite(lse)) and (lse[fin] - lse2[fin]).abs().max() < 2e-2


@pytest.mark.parametrize("Tq,Tk,masked", [(12, 12, False), (200, 200, True), (448, 448, False), (70, 300, True), (1, 129, False)])
@pytest.mark.parametrize("dtype,use_mfma", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_attn_causal_fwd(dev, Tq, Tk, masked, dtype, use_mfma):
    """Causal forward (queries = the last Tq positions of the Tk keys), alone and combined with a key mask,
    against the masked softmax in fp32 on the same (rounded) operands; the log-sum-exp is checked too."""
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    B, H = 2, 3
    q, k, v = (rnd(g, B, t, H * 64, dtype=dtype) for t in (Tq, Tk, Tk))
    mask = torch.zeros(B, Tk, dtype=torch.uint8)
    if masked:
        mask[1, Tk - Tq + Tq // 2:] = 1                      # padding tokens at the end of one sequence
        mask[0, 0] = 1
    qh, kh, vh = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * 0.125
    future = torch.arange(Tk).view(1, Tk) > torch.arange(Tq).view(Tq, 1) + (Tk - Tq)
    s = s.masked_fill(future, float("-inf")).masked_fill(mask[:, None, None, :].bool(), float("-inf"))
    dead = torch.isinf(s).all(-1, keepdim=True)
    ref = (torch.softmax(s.masked_fill(dead, 0.0), -1).masked_fill(dead, 0.0) @ vh).transpose(1, 2).reshape(B, Tq, H * 64)
    qd, kd, vd = (t.to(dev, dtype) for t in (q, k, v))
    o, lse = ops.attn_fwd(qd, kd, vd, H, key_mask=mask.to(dev) if masked else None, scale=0.125, use_mfma=use_mfma,
                          save_lse=True, causal=True)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < tol(dtype)
    want = torch.logsumexp(s, -1)
    fin = torch.isfinite(want)
    assert (lse.cpu()[fin] - want[fin]).abs().max() < (1e-3 if dtype == torch.float32 else 2e-2)
    if Tk > Tq:                                              # more queries than keys has no causal meaning
        with pytest.raises(RuntimeError):
            ops.attn_fwd(kd, qd, qd, H, causal=True)


@pytest.mark.parametrize("Tq,Tk,masked,causal", [(1500, 1500, False, False), (75, 75, True, False), (130, 257, True, False),
                                                  (448, 448, False, True)])
def test_attn_fp8_fwd(dev, Tq, Tk, masked, causal):
    """fp8 (e4m3) MFMA forward.  Two checks: tight against an emulation that rounds Q, K, V and the scaled
    probabilities to e4m3 exactly where the kernel does (so the MFMA data path, masks and layout are right),
    and loose against exact attention on the bf16 operands - the accuracy a caller trades for the format (three
    mantissa bits per operand; on random data the output is an average of up to 1500 values, so it is small
    against the rounding noise: 15 % of the output maximum is the bound here, the network-level effect is
    checked in test_avnet_long_form_chunked_encoder)."""
    g = torch.Generator().manual_seed(Tq + 3 * Tk)
    B, H = 2, 3
    q, k, v = (rnd(g, B, t, H * 64, dtype=torch.bfloat16) for t in (Tq, Tk, Tk))
    q = q * 0.125
    mask = torch.zeros(B, Tk, dtype=torch.uint8)
    if masked:
        mask[1, Tk // 2:] = 1
        mask[0, 5] = 1
    f8 = lambda t: t.to(torch.float8_e4m3fn).float()
    def attn(qx, kx, vx, quant_p):
        qh, kh, vh = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (qx, kx, vx))
        s = qh @ kh.transpose(-1, -2)
        if causal:
            s = s.masked_fill(torch.arange(Tk).view(1, Tk) > torch.arange(Tq).view(Tq, 1) + (Tk - Tq), float("-inf"))
        s = s.masked_fill(mask[:, None, None, :].bool(), float("-inf"))
        p = torch.exp(s - s.amax(-1, keepdim=True))
        l = p.sum(-1, keepdim=True)
        if quant_p:
            p = f8(p * 128.0) / 128.0
        return ((p @ vh) / l).transpose(1, 2).reshape(B, Tq, H * 64), torch.logsumexp(s, -1)
    emu, lse_emu = attn(f8(q), f8(k), f8(v), True)
    exact, _ = attn(q, k, v, False)
    qd, kd, vd = (t.to(dev, torch.bfloat16) for t in (q, k, v))
    o, lse = ops.attn_fwd(qd, kd, vd, H, key_mask=mask.to(dev) if masked else None, scale=1.0, use_mfma="fp8",
                          save_lse=True, causal=causal)
    assert torch.isfinite(o.float()).all()
    print("fp8 attention error vs emulation / vs exact:", rel_err(o, emu), rel_err(o, exact))
    assert rel_err(o, emu) < 4e-2          # differs by where the running maximum stood when P was rounded
    assert rel_err(o, exact) < 0.15
    assert (lse.cpu() - lse_emu).abs().max() < 2e-2
    with pytest.raises(RuntimeError):
        ops.attn_fwd(qd, kd, vd, H, use_mfma="fp8", p_drop=0.1, seed=1)


def test_embed_tokens(dev):
    g = torch.Generator().manual_seed(4)
    V, D, P, B, T = 97, 64, 20, 3, 7
    table, pos = torch.randn(V, D, generator=g), torch.randn(P, D, generator=g)
    ids = torch.randint(0, V, (B, T), generator=g)
    got = ops.embed_tokens(ids.to(dev), table.to(dev), pos.to(dev), past=5)
    assert torch.equal(got.cpu(), (table[ids] + pos[5:5 + T]).reshape(B * T, D))
    got = ops.embed_tokens(ids.to(dev), table.to(dev).bfloat16(), pos.to(dev))
    assert torch.equal(got.cpu(), (table.bfloat16().float()[ids] + pos[:T]).reshape(B * T, D))
    bad = ids.clone()
    bad[1, 2] = V
    with pytest.raises(IndexError):
        ops.embed_tokens(bad.to(dev), table.to(dev), pos.to(dev))
    with pytest.raises(RuntimeError):
        ops.embed_tokens(ids.to(dev), table.to(dev), pos.to(dev), past=P - T + 1)     # runs past the position table


@pytest.mark.parametrize("H,W", [(32, 36), (96, 96), (18, 52), (18, 50)])
@pytest.mark.parametrize("train", [False, True])
def test_stem_mfma(dev, H, W, train):
    """bf16 MFMA stem against conv3d on the bf16-rounded operands (inputs and weights are rounded by the kernel)."""
    g = torch.Generator().manual_seed(H + W)
    B, T = 2, 6
    video = torch.randn(B, T, 3, H, W, generator=g)
    w = torch.randn(64, 3, 5, 3, 3, generator=g) * 0.1
    vr, wr = video.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    conv = F.conv3d(vr.permute(0, 2, 1, 3, 4), wr, stride=(1, 2, 2), padding=(2, 3, 3))      # [B,64,T,OH,OW]
    wp = ops.stem_pack_weight(w.to(dev))
    fr = ops.stem_pack_frags(wp)
    if not train:
        sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
        y = ops.stem3d_fwd(video.to(dev), wp, torch.bfloat16, scale=sc, shift=sh, frags=fr)
        ref = F.relu(conv * sc.cpu().view(1, 64, 1, 1, 1) + sh.cpu().view(1, 64, 1, 1, 1))
    else:
        cs = ops.stem_stats_slabs(video.to(dev))
        cs.fill_(float("nan"))                              # every slab row must be written
        y = ops.stem3d_fwd(video.to(dev), wp, torch.bfloat16, col_stats=(cs[0], cs[1]), frags=fr)
        ref = conv
        tot, tot2 = cs[0].sum(0).cpu(), cs[1].sum(0).cpu()
        stol = 1e-3 if W % 4 == 0 else 1e-2     # W % 4 != 0 falls back to the direct kernel, which does not round the input
        assert rel_err(tot, conv.sum((0, 2, 3, 4))) < stol and rel_err(tot2, (conv * conv).sum((0, 2, 3, 4))) < stol
    ref = ref.permute(0, 2, 3, 4, 1).reshape(B * T, conv.shape[3], conv.shape[4], 64)
    assert y.shape == ref.shape and rel_err(y, ref) < 1e-2
    # and against the direct fp32-input kernel (only the input rounding differs)
    y2 = ops.stem3d_fwd(video.to(dev), wp, torch.bfloat16, **({} if train else dict(scale=sc, shift=sh)),
                        **(dict(col_stats=(ops.stem_stats_slabs(video.to(dev))[0], ops.stem_stats_slabs(video.to(dev))[1])) if train else {}))
    assert rel_err(y, y2) < 2e-2


@pytest.mark.parametrize("M,D,N2", [(300, 384, 1152), (12000, 768, 2304), (1500, 768, 3072), (777, 1280, 640)])
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_layernorm_folded_into_the_products_around_it(dev, M, D, N2, act):
    """x = residual + o @ Wo^T + bo (fp32), y = act(LayerNorm(x) @ W^T + b): the first product leaves a bf16 copy of x and
    per-tile row sums, the second one normalises in its epilogue - against the same chain in fp32 on the CPU, at the
    bf16 tolerance, and against the unfolded device path (LayerNorm kernel between the two products)."""
    g = torch.Generator().manual_seed(M + D)
    o = rnd(g, M, D, dtype=torch.bfloat16)
    wo = (torch.randn(D, D, generator=g) * D ** -0.5).to(torch.bfloat16)
    bo = torch.randn(D, generator=g) * 0.1
    # rows with a mean well away from zero and a few large channels, as a transformer's residual stream has
    res = torch.randn(M, D, generator=g) + torch.randn(M, 1, generator=g) * 0.5
    res[:, 7] += 20.0
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.2
    w = torch.randn(N2, D, generator=g) * D ** -0.5
    b = torch.randn(N2, generator=g) * 0.1
    x_ref = res + o.float() @ wo.float().t() + bo
    y_ref = F.layer_norm(x_ref, (D,), gamma, beta) @ w.t() + b
    if act == "gelu":
        y_ref = F.gelu(y_ref)
    # folded
    xs = res.to(dev).clone()
    parts = torch.full((ops.row_lse_parts(D), M, 2), float("nan"), device=dev)
    xb = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(o.to(dev).to(torch.bfloat16), wo.to(dev), bias=bo.to(dev), residual=xs, out=xs, ln_out=(parts, xb))
    assert rel_err(xs, x_ref) < 1e-2 and torch.equal(xb, xs.to(torch.bfloat16))
    assert torch.allclose(parts[:, :, 0].sum(0).cpu(), x_ref.sum(1), rtol=1e-3, atol=1e-2)
    assert torch.allclose(parts[:, :, 1].sum(0).cpu(), (x_ref * x_ref).sum(1), rtol=1e-3, atol=1e-2)
    w_g, colsum, bias_b = ops.ln_fold_weights(w.to(dev), b.to(dev), gamma.to(dev), beta.to(dev))
    y = ops.gemm_nt(xb, w_g, bias=bias_b, act=ops.ACT_GELU if act == "gelu" else ops.ACT_NONE, ln_in=(parts, colsum, 1e-5))
    assert y.dtype == torch.bfloat16 and y.shape == (M, N2)
    assert_close(y, y_ref, 2e-2, f"LN-folded product {M}x{D}x{N2} {act}")
    # the stored rows as two bf16 planes (hi + lo = the fp32 value to 2^-18), updated in place by a second product
    pl = torch.empty(2, M, D, dtype=torch.bfloat16, device=dev)
    parts2 = torch.full_like(parts, float("nan"))
    ops.gemm_nt(o.to(dev).to(torch.bfloat16), wo.to(dev), bias=bo.to(dev), residual=res.to(dev), no_out=True,
                planes_out=(pl[0], pl[1]), ln_out=(parts2, None))
    assert torch.equal(pl[0], xb) and torch.equal(parts2, parts)
    assert ((pl[0].float() + pl[1].float()) - xs).abs().max() <= 2.0 ** -16 * xs.abs().max()
    o2 = rnd(g, M, D, dtype=torch.bfloat16).to(dev).to(torch.bfloat16)
    x2_ref = x_ref + o2.float().cpu() @ wo.float().t() + bo
    ops.gemm_nt(o2, wo.to(dev), bias=bo.to(dev), no_out=True, planes_res=(pl[0], pl[1]), planes_out=(pl[0], pl[1]),
                ln_out=(parts2, None))
    assert rel_err(pl[0].float() + pl[1].float(), x2_ref) < 1e-2
    assert torch.allclose(parts2[:, :, 0].sum(0).cpu(), x2_ref.sum(1), rtol=1e-3, atol=2e-2)
    back = torch.empty(M, D, device=dev)
    ops.gemm_nt(o2, wo.to(dev), bias=bo.to(dev), planes_res=(pl[0], pl[1]), out=back)     # planes -> fp32, no statistics
    assert rel_err(back, x2_ref + o2.float().cpu() @ wo.float().t() + bo) < 1e-2
    # the unfolded path on the same stored rows
    h = ops.layernorm_fwd(xs, gamma.to(dev), beta.to(dev), out_dtype=torch.bfloat16)
    y2 = ops.gemm_nt(h, w.to(dev).to(torch.bfloat16), bias=b.to(dev), act=ops.ACT_GELU if act == "gelu" else ops.ACT_NONE)
    e_fold = (y.float().cpu() - y_ref).norm() / y_ref.norm()
    e_plain = (y2.float().cpu() - y_ref).norm() / y_ref.norm()
    print(f"LN fold {M}x{D}x{N2} {act}: l2 folded {e_fold:.2e}  unfolded {e_plain:.2e}")
    assert e_fold < 2.0 * e_plain + 1e-3


def test_zero_regions_one_launch(dev):
    """avsr_zero_regions: strided row sets of different dtypes and pitches, nothing else touched."""
    a = torch.ones(3, 7, 10, device=dev, dtype=torch.bfloat16)
    b = torch.ones(5, 4, 6, device=dev)
    ops.zero_regions(a[:, 0], a[:, 6], b[:, 2])
    wa, wb = torch.ones_like(a), torch.ones_like(b)
    wa[:, 0] = 0; wa[:, 6] = 0; wb[:, 2] = 0
    assert torch.equal(a, wa) and torch.equal(b, wb)
    with pytest.raises(ValueError):
        ops.zero_regions(a[:, :, 0])


def test_stem_reads_its_input_through_a_device_word(dev):
    """avsr_stem3d_fwd_indirect: the video's address comes from a device word when the kernel runs - same bits as the
    direct call, whichever batch the word names; the forms that cannot take the indirection say so."""
    g = torch.Generator().manual_seed(77)
    B, T, H, W = 2, 5, 24, 32
    vids = [torch.randn(B, T, 3, H, W, generator=g).to(dev) for _ in range(2)]
    wp = ops.stem_pack_weight((torch.randn(64, 3, 5, 3, 3, generator=g) * 0.1).to(dev))
    fr = ops.stem_pack_frags(wp)
    slot = torch.zeros(1, dtype=torch.int64, device=dev)
    for v in vids:
        cs, cs2 = ops.stem_stats_slabs(v), ops.stem_stats_slabs(v)
        want = ops.stem3d_fwd(v, wp, torch.bfloat16, col_stats=(cs[0], cs[1]), frags=fr)
        ops.store_u64(slot, v.data_ptr())
        got = ops.stem3d_fwd(vids[0], wp, torch.bfloat16, col_stats=(cs2[0], cs2[1]), frags=fr, video_slot=slot)
        assert torch.equal(got, want) and torch.equal(cs, cs2)
    assert slot.item() == vids[1].data_ptr()
    src, dst = torch.arange(8, device=dev) * 3 + 1, torch.zeros(8, dtype=torch.int64, device=dev)
    ops.store_u64(slot, 12345, copy=(dst, src))             # a few more words by the same launch
    assert slot.item() == 12345 and torch.equal(dst, src)
    with pytest.raises(ops._lib.AvsrHipError):                    # the direct fp32 kernel has no indirect form
        ops.stem3d_fwd(vids[0], wp, torch.float32, video_slot=slot)
    with pytest.raises(ValueError):
        ops.stem3d_fwd(vids[0], wp, torch.bfloat16, frags=fr, video_slot=torch.zeros(2, dtype=torch.int64, device=dev))


@pytest.mark.parametrize("nparts,C", [(5, 64), (33, 100), (793, 128), (2930, 64), (2930, 256), (5860, 200), (9000, 256)])
def test_bn_finalize_slabs(dev, nparts, C):
    """Training-mode BN statistics from [parts, C] partial-sum slabs (direct, single-launch and two-stage paths)."""
    g = torch.Generator().manual_seed(nparts)
    rows_per_part = 16
    x = torch.randn(nparts * rows_per_part, C, generator=g) * 2 + 0.5
    xs = x.view(nparts, rows_per_part, C)
    s1, s2 = xs.sum(1), (xs * xs).sum(1)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train()
    ref = bn(x)
    rmd, rvd = rm.to(dev), rv.to(dev)
    sc, sh = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ops.bn_finalize(s1.to(dev), s2.to(dev), x.shape[0], w.to(dev), b.to(dev), rmd, rvd, sc, sh)
    got = x * sc.cpu() + sh.cpu()
    assert rel_err(got, ref) < 2e-4
    assert rel_err(rmd, bn.running_mean) < 1e-4 and rel_err(rvd, bn.running_var) < 1e-4
    # the split reduction combines its slices in a fixed order: repeated launches agree to the bit, and the ticket
    # counters are back at zero for the next one
    sc2, sh2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
    for _ in range(3):
        ops.bn_finalize(s1.to(dev), s2.to(dev), x.shape[0], w.to(dev), b.to(dev), None, None, sc2, sh2)
        assert torch.equal(sc2, sc) and torch.equal(sh2, sh)
    assert all(int(t.abs().sum()) == 0 for t in ops._BN_TICKETS.values())


@pytest.mark.parametrize("rows,C,nparts,res", [(29400, 256, 230, False), (29400, 1024, 230, True), (9600, 512, 75, False),
                                               (9600, 2048, 75, True), (300, 64, 3, True), (1000, 128, 256, False)])
def test_bn_finalize_and_apply_in_one_launch(dev, rows, C, nparts, res):
    """avsr_bn_finalize_apply against the two launches it replaces: same output (bf16 ulp: the slab rows are added in
    another order), same affine and running statistics; against the definition on the CPU as well."""
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).to(torch.bfloat16).to(dev)
    r = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev) if res else None
    xf = x.float().cpu()
    # slabs as a producer would leave them: row p holds the sums of rows p, p + nparts, ...
    per = (rows + nparts - 1) // nparts
    pad = torch.zeros(per * nparts, C)
    pad[:rows] = xf
    cs, cq = pad.view(nparts, per, C).sum(1).to(dev), (pad * pad).view(nparts, per, C).sum(1).to(dev)
    w, b = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    rm, rv = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    rm2, rv2 = rm.clone(), rv.clone()
    sc, sh = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ops.bn_finalize(cs, cq, rows, w, b, rm, rv, sc, sh)
    want = ops.bn_apply(x, sc, sh, residual=r, relu=True)
    sc2, sh2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
    got = ops.bn_finalize_apply(x, cs, cq, rows, w, b, rm2, rv2, residual=r, relu=True, affine_out=(sc2, sh2))
    assert torch.allclose(sc2, sc, rtol=1e-5, atol=1e-6) and torch.allclose(sh2, sh, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rm2, rm, rtol=1e-5, atol=1e-6) and torch.allclose(rv2, rv, rtol=1e-5, atol=1e-6)
    d = (got.float() - want.float()).abs()
    assert (d <= 2.0 ** -7 * want.float().abs() + 1e-6).all() and (d > 0).float().mean() < 1e-2
    ref = F.relu(F.batch_norm(xf.t().unsqueeze(0), None, None, w.cpu(), b.cpu(), True).squeeze(0).t()
                 + (r.float().cpu() if res else 0))
    assert rel_err(got, ref) < 2e-2
    # in place, and without the optional outputs
    x2 = x.clone()
    assert ops.bn_finalize_apply(x2, cs, cq, rows, w, b, None, None, residual=r, relu=True, out=x2) is x2
    assert torch.equal(x2, got)


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("hw,C", [((4, 4), 2048), ((3, 3), 64), ((1, 5), 4096)])
def test_bn_apply_folded_into_the_average_pool(dev, odt, hw, C):
    g = torch.Generator().manual_seed(C)
    B, T = 2, 5
    x = rnd(g, B * T, hw[0], hw[1], C, dtype=torch.bfloat16)
    r = rnd(g, B * T, hw[0], hw[1], C, dtype=torch.bfloat16)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    lens = torch.tensor([5, 3])
    ref = F.relu(x * sc + sh + r).mean((1, 2)).view(B, T, C)
    ref[1, 3:] = 0
    got = ops.bn_apply_avgpool_mask(x.to(dev, torch.bfloat16), r.to(dev, torch.bfloat16), sc.to(dev), sh.to(dev),
                                    lens.to(dev), B, T, odt)
    assert got.dtype == odt and rel_err(got, ref) < tol(odt if odt == torch.bfloat16 else torch.float32) * (1 if odt == torch.bfloat16 else 5)
    assert (got[1, 3:] == 0).all()


@pytest.mark.parametrize("shape", [(3, 50, 50, 64), (2, 13, 17, 128), (1, 7, 9, 256), (2, 5, 70, 64), (2, 24, 24, 8)])
def test_maxpool_bf16_bands(dev, shape):
    """The 16-byte bf16 max-pool (bands of output rows, the shared input row carried in registers): odd sizes,
    more output columns than one pass covers, negative scales, with and without the fused affine + ReLU."""
    n, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = rnd(g, n, H, W, C, dtype=torch.bfloat16)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    xd = x.to(dev, torch.bfloat16)
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    out = ops.maxpool3x3s2(xd)
    assert out.shape == ref.shape and torch.equal(out.float().cpu(), ref)
    ref = F.max_pool2d(F.relu(x * sc + sh).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    out = ops.maxpool3x3s2(xd, scale=sc.to(dev), shift=sh.to(dev))
    assert rel_err(out, ref) < tol(torch.bfloat16)


@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_stem_pool_bn(dev, odt):
    g = torch.Generator().manual_seed(5)
    B, T, H, W = 2, 4, 32, 36
    video = torch.randn(B, T, 3, H, W, generator=g)
    w = torch.randn(64, 3, 5, 3, 3, generator=g) * 0.1
    bw, bb = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    conv = F.conv3d(video.permute(0, 2, 1, 3, 4), w, stride=(1, 2, 2), padding=(2, 3, 3))
    # eval mode
    ref = F.max_pool3d(F.relu(F.batch_norm(conv, rm, rv, bw, bb, False)), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    ref = ref.permute(0, 2, 3, 4, 1).reshape(B * T, ref.shape[3], ref.shape[4], 64)
    sc, sh = ops.bn_fold(bw.to(dev), bb.to(dev), rm.to(dev), rv.to(dev))
    wp = ops.stem_pack_weight(w.to(dev))
    y = ops.stem3d_fwd(video.to(dev), wp, odt, scale=sc, shift=sh)
    out = ops.maxpool3x3s2(y)
    assert out.shape == ref.shape and rel_err(out, ref) < tol(odt)
    # training mode: raw conv + batch statistics, affine+relu inside the pool
    rm2, rv2 = rm.clone(), rv.clone()
    ref_t = F.max_pool3d(F.relu(F.batch_norm(conv, rm2, rv2, bw, bb, True)), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    ref_t = ref_t.permute(0, 2, 3, 4, 1).reshape(B * T, ref.shape[1], ref.shape[2], 64)
    cs = ops.stem_stats_slabs(video.to(dev))
    raw = ops.stem3d_fwd(video.to(dev), wp, odt, col_stats=(cs[0], cs[1]))
    rmd, rvd = rm.to(dev), rv.to(dev)
    sc2, sh2 = torch.empty(64, device=dev), torch.empty(64, device=dev)
    ops.bn_finalize(cs[0], cs[1], raw.numel() // 64, bw.to(dev), bb.to(dev), rmd, rvd, sc2, sh2)
    out_t = ops.maxpool3x3s2(raw, scale=sc2, shift=sh2)
    assert rel_err(out_t, ref_t) < tol(odt) * 2
    assert rel_err(rmd, rm2) < 1e-3 and rel_err(rvd, rv2) < 1e-3
    # bn_apply with residual
    x = rnd(g, 50, 64, dtype=odt)
    r = rnd(g, 50, 64, dtype=odt)
    ya = ops.bn_apply(x.to(dev, odt), sc, sh, residual=r.to(dev, odt), relu=True)
    assert rel_err(ya, F.relu(x * sc.cpu() + sh.cpu() + r)) < tol(odt)
    # ... and a residual that still needs its own affine (the downsample branch's BN, folded into this pass);
    # 4096 rows take the 16-byte bf16 kernel through more than one grid stride
    for rows in (50, 4096):
        x, r = rnd(g, rows, 64, dtype=odt), rnd(g, rows, 64, dtype=odt)
        rs, rb = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
        yb = ops.bn_apply(x.to(dev, odt), sc, sh, residual=r.to(dev, odt), residual_affine=(rs.to(dev), rb.to(dev)), relu=True)
        assert rel_err(yb, F.relu(x * sc.cpu() + sh.cpu() + r * rs + rb)) < tol(odt)
    # avgpool + mask
    feat = rnd(g, B * T, 4, 4, 128, dtype=odt)
    lens = torch.tensor([3, 4])
    pooled = ops.avgpool_mask(feat.to(dev, odt), lens.to(dev), B, T, torch.float32)
    refp = feat.mean((1, 2)).view(B, T, 128)
    refp[0, 3:] = 0
    assert rel_err(pooled, refp) < tol(odt)


@pytest.mark.parametrize("T,V,L,ld", [(20, 501, 6, 504), (300, 203, 40, 203), (75, 1030, 32, 1032)])
def test_head_loss(dev, T, V, L, ld):
    """Log-softmax + label-smoothed CE + CTC and their fused gradient against torch.  (300, 40) is past what the
    CTC kernel keeps in LDS (global-gather path), ld = 203 exercises the unaligned-row (scalar) paths."""
    g = torch.Generator().manual_seed(9 + T)
    B = 3
    logits = torch.randn(B, T, V, generator=g) * 2
    targets = torch.randint(1, V, (B, L), generator=g)
    targets[1, 2] = targets[1, 1]                     # repeated label
    tl = torch.tensor([L, L - 2, L - 1])
    il = torch.tensor([T, T - 5, L - 2])              # last one infeasible (L-1 labels in L-2 frames) -> zero_infinity
    lr = logits.clone().requires_grad_(True)
    lp = F.log_softmax(lr, -1).transpose(0, 1)
    ctc = F.ctc_loss(lp, targets, il, tl, blank=0, reduction="mean", zero_infinity=True)
    ce = F.cross_entropy(lr[:, :L].reshape(-1, V), targets.reshape(-1), ignore_index=-100, label_smoothing=0.1)
    (ctc + ce).backward()
    buf = torch.zeros(B, T, ld, device=dev)
    buf[..., :V] = logits.to(dev)
    lg = buf[..., :V]
    out3, ws = ops.head_loss_fwd(lg, targets.to(dev), il.to(dev), tl.to(dev))
    o = out3.cpu()
    assert abs(o[0] - ctc.item()) < 1e-4 * max(1, abs(ctc.item()))
    assert abs(o[1] - ce.item()) < 1e-4 * max(1, abs(ce.item()))
    d = ops.head_loss_bwd(lg, targets.to(dev), il.to(dev), tl.to(dev), ws, torch.float32, ld_out=ld)
    assert rel_err(d.view(B, T, ld)[..., :V], lr.grad) < 1e-3
    am = ops.argmax_rows(lg)
    assert torch.equal(am.cpu(), logits.argmax(-1))


@pytest.mark.parametrize("M,V,K", [(600, 1000, 128), (77, 5000, 64), (600, 51865, 512), (10, 129, 64)])
def test_head_losses_from_the_products_row_partials(dev, M, V, K):
    """The head product leaves {max, sum exp, sum} per row and 128-column tile (gemm_nt(row_part=...)); merged, they
    are the row log-sum-exp and row sum the losses need, so the forward never streams the logits: same losses, same
    workspace rows as the pass over the logits, including a ragged last tile and grids of very few tiles."""
    g = torch.Generator().manual_seed(M + V)
    B, T = (8, M // 8) if M % 8 == 0 else (1, M)
    x = rnd(g, M, K, dtype=torch.bfloat16).to(dev, torch.bfloat16)
    w = (rnd(g, V, K, dtype=torch.bfloat16) * (4.0 / math.sqrt(K))).to(dev, torch.bfloat16)
    b = torch.randn(V, generator=g).to(dev)
    ld = (V + 31) // 32 * 32
    buf = torch.empty(M, ld, device=dev)
    part = torch.full((ops.row_lse_parts(V), M, 4), float("nan"), device=dev)
    ops.gemm_nt(x, w, bias=b, out=buf[:, :V], row_part=part)
    plain = torch.empty(M, ld, device=dev)
    ops.gemm_nt(x, w, bias=b, out=plain[:, :V])
    assert torch.equal(buf[:, :V], plain[:, :V])                   # the product itself is untouched
    logits = buf[:, :V]
    tiles = list(logits.split(128, dim=1))
    ref_max = torch.stack([t.max(1).values for t in tiles])
    ref_se = torch.stack([(t - t.max(1, keepdim=True).values).exp().sum(1) for t in tiles])
    ref_sx = torch.stack([t.sum(1) for t in tiles])
    assert torch.equal(part[..., 0], ref_max)
    assert rel_err(part[..., 1], ref_se) < 1e-5 and rel_err(part[..., 2], ref_sx) < 1e-5
    L = 5
    targets = torch.randint(1, V, (B, L), generator=g).to(dev)
    il, tl = torch.full((B,), T, device=dev), torch.full((B,), min(L, max(1, T // 3)), device=dev)
    lg = buf.view(B, T, ld)[:, :, :V]
    o0, ws0 = ops.head_loss_fwd(lg, targets, il, tl)
    o1, ws1 = ops.head_loss_fwd(lg, targets, il, tl, row_part=part)
    assert rel_err(o1, o0) < 1e-5
    assert rel_err(ws1[:2 * M], ws0[:2 * M]) < 1e-5                 # lse and row sums
    d0 = ops.head_loss_bwd(lg, targets, il, tl, ws0, torch.float32, ld_out=ld)
    d1 = ops.head_loss_bwd(lg, targets, il, tl, ws1, torch.float32, ld_out=ld)
    assert rel_err(d1, d0) < 5e-4                                  # exp(logit - lse): the merge order shows in the last bits


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_elementwise(dev, dtype):
    g = torch.Generator().manual_seed(1)
    x = rnd(g, 70, 130, dtype=dtype)
    xd = x.to(dev, dtype)
    assert rel_err(ops.transpose2d(xd), x.T) < 1e-6
    assert rel_err(ops.cast(xd, torch.float32, 2.0), x * 2) < 1e-6
    assert rel_err(ops.colsum(xd), x.sum(0)) < 1e-3
    assert rel_err(ops.gelu_fwd(xd), F.gelu(x)) < tol(dtype)
    xr = x.clone().requires_grad_(True)
    dy = rnd(g, 70, 130, dtype=dtype)
    F.gelu(xr).backward(dy)
    assert rel_err(ops.gelu_bwd(xd, dy.to(dev, dtype)), xr.grad) < tol(dtype)
    y = rnd(g, 70, 130, dtype=dtype)
    gate = torch.tensor([0.4])
    seed, p = 77, 0.3
    keep = torch.from_numpy(keep_mask(seed, (70, 130), p)).float()
    ref = x + math.tanh(0.4) * y * keep / (1 - p)
    out = ops.gated_residual_fwd(xd, y.to(dev, dtype), gate.to(dev), p, seed)
    assert rel_err(out, ref) < tol(dtype)
    dg = torch.zeros(1, device=dev)
    dyk = ops.gated_residual_bwd(dy.to(dev, dtype), y.to(dev, dtype), gate.to(dev), dg, p, seed)
    assert rel_err(dyk, math.tanh(0.4) * dy * keep / (1 - p)) < tol(dtype)
    ref_dg = (1 - math.tanh(0.4) ** 2) * (dy * y * keep / (1 - p)).sum()
    assert abs(dg.item() - ref_dg.item()) < 2e-2 * abs(ref_dg.item()) + 1e-2
    assert rel_err(ops.add3(xd, y.to(dev, dtype), dy.to(dev, dtype)), x + y + dy) < tol(dtype)


def test_adamw_and_clip(dev):
    g = torch.Generator().manual_seed(2)
    n = 10007
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01)
    pd, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in (1, 2, 3):
        ref_p.grad = gr.clone() * step
        norm = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        gd = (gr * step).to(dev)
        ss, sc, nrm = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        ops.sumsq_accumulate(gd, ss)
        ops.clip_scale(ss, 1.0, sc, nrm)
        assert abs(nrm.item() - norm.item()) < 1e-3 * norm.item()
        ops.adamw_step(pd, gd, m, v, step, lr=1e-3, grad_scale=sc)
    assert rel_err(pd, ref_p.data) < 1e-5


# ---------------------------------------------------------------- mel front end
def _seeded_wave(n, seed, batch=1):
    g = torch.Generator().manual_seed(seed)
    return 0.1 * torch.randn(batch, n, generator=g)


@pytest.mark.gpu
def test_logmel_whisper_matches_oracle_and_golden(dev):
    """fp32 tolerance 1e-3 on the (log10 + 4) / 4 features; the committed golden is the output of
    transformers' WhisperFeatureExtractor on the same seeded 10 s waveform."""
    from mocov2_whisper_flamingo_amd import LogMelFrontend
    import numpy as np, os
    fe = LogMelFrontend("whisper", dev)
    wave = _seeded_wave(160000, 5)
    got = fe(wave.to(dev))                                              # [1,80,3000], zero-padded to 30 s
    padded = torch.zeros(1, 480000); padded[0, :160000] = wave[0]
    ref = oracle.whisper_logmel(padded)
    assert got.shape == (1, 80, 3000)
    assert (got.cpu() - ref).abs().max() < 1e-3
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logmel.npz"))
    assert np.abs(got[0, :, ::7].cpu().numpy() - g["feats_sub"]).max() < 1e-3
    assert abs(float(got.double().sum()) - float(g["total"])) < 1e-3 * 240000 * 0.05
    # full-length batch, channels-last bf16 (the layout / dtype the encoder consumes)
    wave = _seeded_wave(480000, 6, batch=3)
    wave[1] *= 0.01
    wave[2, 300000:] = 0
    ref = oracle.whisper_logmel(wave)
    got = fe(wave.to(dev))
    assert (got.cpu() - ref).abs().max() < 1e-3
    got_cl = fe(wave.to(dev), dtype=torch.bfloat16, channels_last=True)
    assert got_cl.shape == (3, 3000, 80)
    assert (got_cl.float().cpu() - ref.transpose(1, 2)).abs().max() < 2e-2


@pytest.mark.gpu
def test_melspec_layernorm_matches_oracle(dev):
    """PARITY UNPINNED for this variant (torchaudio is not installed anywhere the goldens could be made):
    checked against the CPU restatement only.  Tolerance 1e-3 relative to the unit-variance output."""
    from mocov2_whisper_flamingo_amd import LogMelFrontend
    fe = LogMelFrontend("melspec_norm", dev)
    for n in (480000, 123457, 16000):
        wave = _seeded_wave(n, 7, batch=2)
        wave[1] *= 3.0
        ref = oracle.torchaudio_melspec_norm(wave)
        got = fe(wave.to(dev))
        assert got.shape == (2, 3000, 80)
        err = (got.cpu() - ref).abs().max()
        assert err < 1e-3 * max(1.0, float(ref.abs().max())), (n, float(err))


@pytest.mark.gpu
def test_melspec_training_pipeline_matches_oracle(dev):
    """Training-time pipeline (SpecAugment bands + noise at a target SNR per mel row + clip layer norm) with fixed
    draws against the CPU restatement.  PARITY UNPINNED like the evaluation variant (torchaudio absent): the
    add_noise arithmetic follows torchaudio's documented formula.  Bands as the reference applies them to its
    [80, frames] tensor (datamodule/transforms.py:127,152-177): 48 frames and 10 mel bins.  Covers overlapping /
    skipped bands, the 'clean' SNR level 999999, a 10 s clip and a band reaching the trimmed last frame."""
    import random
    from mocov2_whisper_flamingo_amd import LogMelFrontend
    fe = LogMelFrontend("melspec_norm", dev)
    g = torch.Generator().manual_seed(12)
    noise_rec = 0.3 * torch.randn(20000, generator=g)
    # fs = starts of the 10-bin mel bands in [0, 70), ts = starts of the 48-frame bands in [0, frames - 48)
    for n, fs, ts, off, snr in (
            (480000, [[0, 69], [10, 15], [-1, 5]], [[0, 2952], [100, 130], [-1, -1]], [0, 17000, 5], [-5.0, 999999.0, 10.0]),
            (160000, [[3, 3], [69, 0], [7, -1]], [[0, 952], [400, 500], [10, -1]], [100, 200, 300], [0.0, 20.0, 15.0])):
        wave = _seeded_wave(n, 8, batch=3)
        wave[2] *= 0.05
        segs = [noise_rec[o:o + 3000] for o in off]
        ref = oracle.torchaudio_melspec_train(wave, fs, ts, segs, snr)
        got = fe.train(wave.to(dev), noise_rec.to(dev), draws=(fs, ts, off, snr))
        assert got.shape == (3, 3000, 80)
        err = (got.cpu() - ref).abs().max()
        assert err < 1e-3 * max(1.0, float(ref.abs().max())), (n, float(err))
        got16 = fe.train(wave.to(dev), noise_rec.to(dev), draws=(fs, ts, off, snr), dtype=torch.bfloat16, channels_last=False)
        assert got16.shape == (3, 80, 3000)
        assert (got16.float().cpu() - ref.transpose(1, 2)).abs().max() < 2e-2 * max(1.0, float(ref.abs().max()))
    # the host-side draws stay in the reference's ranges
    fs, ts, off, snr = fe.draw(64, 480000, 20000, random.Random(1))
    assert all(0 <= f < 70 for r in fs for f in r) and all(0 <= t < 3001 - 48 for r in ts for t in r)
    assert all(0 <= o <= 17000 for o in off) and set(snr) <= set(map(float, fe.SNR_LEVELS))
    out = fe.train(_seeded_wave(480000, 9, batch=2).to(dev), noise_rec.to(dev), rng=random.Random(2))
    assert torch.isfinite(out).all() and abs(float(out[0].mean())) < 1e-3
    with pytest.raises(ValueError):
        LogMelFrontend("whisper", dev).train(wave.to(dev), noise_rec.to(dev))


@pytest.mark.gpu
def test_video_transform(dev):
    """Device VideoTransform against the same steps in core torch (interpolate / flip / normalise): the val path
    on uint8 and fp32 frames, and the train path (resize 64x64, flip, zeroed frames) with fixed draws."""
    import random
    from mocov2_whisper_flamingo_amd import VideoTransform
    from mocov2_whisper_flamingo_amd.transforms import adaptive_time_mask
    g = torch.Generator().manual_seed(14)
    B, T, H, W = 2, 9, 96, 88
    u8 = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 1, 3, 1, 1)
    ref = (u8.float() / 255.0 - mean) / std
    for x in (u8, u8.float()):
        got = VideoTransform("val")(x.to(dev))
        assert (got.cpu() - ref).abs().max() < 1e-5
    assert (VideoTransform("test")(u8[0].to(dev)).cpu() - ref[0]).abs().max() < 1e-5
    flip = [True, False]
    zero = [[t in (2, 3, 7) for t in range(T)], [False] * T]
    x = F.interpolate(u8.float().view(B * T, 3, H, W), size=(64, 64), mode="bilinear", align_corners=False)
    x = (x / 255.0).view(B, T, 3, 64, 64)
    x[0] = x[0].flip(-1)
    x[0, [2, 3, 7]] = 0
    ref_t = (x - mean) / std
    got = VideoTransform("train")(u8.float().to(dev), draws=(flip, zero))
    assert got.shape == (B, T, 3, 64, 64) and (got.cpu() - ref_t).abs().max() < 1e-4
    z = adaptive_time_mask(75, 10, 25, random.Random(3))
    assert len(z) == 75 and sum(z) <= 27
    out = VideoTransform("train")(u8.to(dev), rng=random.Random(4))
    assert out.shape == (B, T, 3, 64, 64) and torch.isfinite(out).all()
    with pytest.raises(ValueError):
        VideoTransform("val")(torch.zeros(2, 4, 8, 8, device=dev))


@pytest.mark.gpu
def test_video_transform_colour_steps(dev):
    """The train pipeline with ColorJitter and RandomGrayscale (datamodule/transforms.py:65-75) against the CPU
    restatement with fixed draws: every permutation position, skipped steps, a grey clip, zeroed frames, flips.
    PARITY UNPINNED (torchvision absent): the restatement follows torchvision 0.19.1's tensor ops."""
    import random
    from mocov2_whisper_flamingo_amd import VideoTransform
    g = torch.Generator().manual_seed(15)
    B, T, H, W = 4, 6, 80, 72
    u8 = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8)
    u8[3, :, :, :10] = 77                                        # grey pixels (max == min) exercise the hue edge case
    flips = [True, False, True, False]
    zeros = [[t in (1, 4) for t in range(T)], [False] * T, [t == 0 for t in range(T)], [False] * T]
    orders = [[3, 0, 2, 1], [1, 2, 3, 0], [0, -1, 3, -1], [2, 1, 0, 3]]
    facs = [[1.4, 0.6, 1.3, 0.1], [0.6, 1.4, 0.7, -0.1], [1.1, 1.0, 1.0, 0.05], [0.9, 1.2, 0.0, -0.03]]
    greys = [False, True, False, False]
    got = VideoTransform("train")(u8.to(dev), draws=(flips, zeros, orders, facs, greys))
    assert got.shape == (B, T, 3, 64, 64)
    for b in range(B):
        ref = oracle.video_transform_train(u8[b].float(), flips[b], orders[b], facs[b], greys[b], zeros[b])
        # hue: a sector boundary (floor(6h)) can flip on a rounding difference; allow a handful of such pixels
        diff = (got[b].cpu() - ref).abs()
        assert diff.max() < 5e-2 and (diff > 1e-3).float().mean() < 1e-3, (b, float(diff.max()))
    out = VideoTransform("train")(u8.to(dev), rng=random.Random(4))
    assert out.shape == (B, T, 3, 64, 64) and torch.isfinite(out).all()
    from mocov2_whisper_flamingo_amd._lib import AvsrHipError
    with pytest.raises(AvsrHipError):                            # a frame's planes must fit in LDS
        ops.video_color(torch.zeros(1, 1, 3, 128, 128, device=dev))


@pytest.mark.gpu
def test_logmel_rejects_bad_arguments(dev):
    from mocov2_whisper_flamingo_amd._lib import AvsrHipError
    tables = ops.logmel_tables(ops.MEL_WHISPER, dev)
    with pytest.raises(AvsrHipError):
        ops.logmel_fwd(torch.zeros(1, 100, device=dev), tables, ops.MEL_WHISPER, n_samples=100, n_frames=1)
    with pytest.raises(AvsrHipError):
        ops.logmel_fwd(torch.zeros(1, 16000, device=dev), tables, ops.MEL_WHISPER, n_frames=101)
    with pytest.raises(ValueError):
        ops.logmel_fwd(torch.zeros(16000, device=dev), tables, ops.MEL_WHISPER)


@pytest.mark.parametrize("M,K,N", [(5000, 64, 256), (375000, 64, 256), (101400, 128, 512), (77, 128, 512)])
def test_bn_statistics_from_the_input_gram_matrix(dev, M, K, N):
    """avsr_gram_stats + avsr_bn_gram_finalize: training-mode BN scale / shift and running statistics of a 1x1
    convolution's output, computed from the K x K Gram matrix and the column sums of its INPUT, against the
    statistics of the product itself (fp64 on the same bf16 operands)."""
    g = torch.Generator().manual_seed(M + K)
    h = torch.relu(torch.randn(M, K, generator=g) * 0.7 + 0.2).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    rm, rv = 0.1 * torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    y = h.double() @ w.double().t()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    want_scale = gamma.double() / torch.sqrt(var + 1e-5)
    want_shift = beta.double() - mean * want_scale
    want_rm = 0.9 * rm.double() + 0.1 * mean
    want_rv = 0.9 * rv.double() + 0.1 * var * (M / (M - 1))
    hd, wd = h.to(dev), w.to(dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    aff = torch.empty(2, N, device=dev)
    gp, cp = ops.gram_stats(hd)
    assert gp.shape[1:] == (K, K) and cp.shape == (gp.shape[0], K)
    ops.bn_gram_finalize(gp, cp, M, wd, gamma.to(dev), beta.to(dev), rmd, rvd, aff[0], aff[1])
    for got, want, tol in ((aff[0], want_scale, 2e-4), (aff[1], want_shift, 2e-4), (rmd, want_rm, 1e-5), (rvd, want_rv, 1e-4)):
        err = ((got.double().cpu() - want).abs().max() / want.abs().max().clamp_min(1.0)).item()
        assert err < tol, err
    # the shares are added in a fixed order: a second run gives the same bits
    aff2 = torch.empty(2, N, device=dev)
    gp2, cp2 = ops.gram_stats(hd)
    ops.bn_gram_finalize(gp2, cp2, M, wd, gamma.to(dev), beta.to(dev), None, None, aff2[0], aff2[1])
    assert torch.equal(aff, aff2)
    # the same statistics from the RAW tensor one BatchNorm earlier: the pass normalises it in flight (and in place)
    raw = (torch.randn(M, K, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    sc[3] = -sc[3]                                          # a negative gain is legal
    want_h = ops.bn_apply(raw, sc, sh, relu=True)
    gp_w, cp_w = ops.gram_stats(want_h)
    x = raw.clone()
    gp_g, cp_g = ops.gram_stats(x, affine=(sc, sh))
    assert torch.equal(x, want_h)
    assert torch.equal(gp_g, gp_w) and torch.equal(cp_g, cp_w)
