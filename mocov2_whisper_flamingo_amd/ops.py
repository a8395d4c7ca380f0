"""Tensor-level wrappers over the C ABI (no autograd here): each function takes
torch tensors that already live on the GPU, checks what the kernels assume,
and launches on torch's current stream.  torch only provides device memory and
the stream.
"""
import ctypes
import os
import torch

from . import _lib
from ._lib import F32, BF16, ACT_NONE, ACT_GELU, ACT_RELU, ACT_RELU_AFTER_RESIDUAL, GemmDesc, check

_DT = {torch.float32: F32, torch.bfloat16: BF16}

# When set to a list, every gemm_nt launch appends (start_event, end_event, flops, tag): HIP events
# recorded on the launch stream, used by bench.py to time the dominant kernel per launch.
GEMM_PROFILE = None
GEMM_PROFILE_REPEAT = False


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    # raw hipStream_t of torch's current stream (the fast path: no Stream object is built)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AvsrHipError("tensor is not on the GPU - there is no CPU path in this package")


def gemm_nt(a, b, *, out=None, out_dtype=None, bias=None, scale=None, act=ACT_NONE, residual=None,
            res_period=0, gate=None, alpha=1.0, col_stats=None, conv=None, M=None, stats_only=False, row_part=None,
            ln_out=None, ln_in=None, planes_out=None, planes_res=None, no_out=False):
    """out[m,n] = epi(sum_k a[m,k] b[n,k]).  a: [M,K] or [batch,M,K]; b: [N,K] or [batch,N,K].
    conv = (H, W, C, KH, KW, stride, pad_h, pad_w, OH, OW): a is an NHWC image batch, M = n_img*OH*OW.
    ``stats_only``: fill ``col_stats`` and store no product (returns None).
    A LayerNorm between two products, folded into them (see ``ln_fold_weights``):
    ``ln_out`` = (parts [ceil(N/128), M, 2] fp32, copy [M, N] bf16) on the product that writes the LayerNorm's input
    (fp32 ``out`` with ``residual``): also stores the rows' per-tile {sum, sum of squares} and their bf16 copy;
    ``ln_in`` = (parts, colsum [N] fp32, eps) on the product that reads the normalised rows: ``a`` is that bf16 copy,
    ``b`` / ``bias`` come from ``ln_fold_weights``.
    ``planes_out`` = (hi, lo) bf16 [M, N]: the fp32 result as hi = bf16(v), lo = bf16(v - hi) (with ``no_out`` instead of
    ``out``; ``ln_out`` = (parts, None) then adds the statistics); ``planes_res`` = (hi, lo): the residual in that form
    (may be the same tensors as ``planes_out``)."""
    lib = _lib.load()
    _need_cuda(a, b, out, bias, scale, residual, gate)
    d = GemmDesc()
    batched = b.dim() == 3
    N, K = b.shape[-2], b.shape[-1]
    if conv is not None:
        H, W, C, KH, KW, stride, pad_h, pad_w, OH, OW = conv
        if not a.is_contiguous():
            raise ValueError("conv input must be contiguous NHWC")
        d.conv.enabled, d.conv.H, d.conv.W, d.conv.C = 1, H, W, C
        d.conv.KH, d.conv.KW, d.conv.stride, d.conv.pad_h, d.conv.pad_w = KH, KW, stride, pad_h, pad_w
        d.conv.OH, d.conv.OW = OH, OW
        Mv = a.numel() // (H * W * C) * OH * OW
        lda, sA, batch = 0, 0, 1
    else:
        if a.stride(-1) != 1 or b.stride(-1) != 1:
            raise ValueError("gemm_nt operands must be K-contiguous")
        Mv = a.shape[-2] if M is None else M
        lda = a.stride(-2)
        batch = a.shape[0] if a.dim() == 3 else 1
        sA = a.stride(0) if a.dim() == 3 else 0
        if a.shape[-1] != K:
            raise ValueError(f"K mismatch {a.shape} vs {b.shape}")
    if batched and b.shape[0] != batch:
        raise ValueError("batch mismatch")
    odt = out_dtype if out_dtype is not None else (out.dtype if out is not None else a.dtype)
    if stats_only:
        if col_stats is None or out is not None or residual is not None:
            raise ValueError("a statistics-only product takes col_stats and neither out nor residual")
    elif out is None and not no_out:
        shape = (batch, Mv, N) if (a.dim() == 3 and conv is None) else (Mv, N)
        out = torch.empty(shape, dtype=odt, device=a.device)
    if out is not None and out.stride(-1) != 1:
        raise ValueError("out must be row-major")
    d.in_dtype, d.out_dtype = _dt(a), _DT[(torch.float32 if no_out else odt) if out is None else out.dtype]
    if b.dtype != a.dtype:
        raise TypeError("a and b dtypes differ")
    d.M, d.N, d.K = Mv, N, K
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), lda, b.data_ptr(), b.stride(-2)
    d.C, d.ldc = (None, N) if out is None else (out.data_ptr(), out.stride(-2))
    d.batch, d.strideA = batch, sA
    d.strideB = b.stride(0) if batched else 0
    d.strideC = out.stride(0) if (out is not None and out.dim() == 3) else 0
    d.alpha = alpha
    d.scale, d.bias, d.act = _p(scale), _p(bias), act
    if residual is not None:
        if residual.dtype != (torch.float32 if out is None else out.dtype) or residual.stride(-1) != 1:
            raise ValueError("residual must match the output dtype and be row-major")
        d.residual, d.ldr, d.res_period = residual.data_ptr(), residual.stride(-2), res_period
    d.gate = _p(gate)
    if col_stats is not None:
        d.col_sum, d.col_sumsq = col_stats[0].data_ptr(), col_stats[1].data_ptr()
    if row_part is not None:        # [ceil(N/128), M, 4] fp32: per row and 128-column tile {max, sum exp, sum}
        if row_part.dtype != torch.float32 or not row_part.is_contiguous() or row_part.numel() < row_lse_parts(N) * Mv * 4:
            raise ValueError("row_part must be a contiguous fp32 buffer of row_lse_parts(N) * M * 4 floats")
        d.row_part = row_part.data_ptr()
    def _plane(t):
        if t.dtype != torch.bfloat16 or t.stride(-1) != 1 or t.dim() != 2 or t.shape[0] < Mv or t.shape[1] < N:
            raise ValueError("a bf16 plane is a row-major [M, N] bf16 tensor")
        _need_cuda(t)
        return t
    if planes_out is not None:
        hi, lo = _plane(planes_out[0]), _plane(planes_out[1])
        if hi.stride(0) != lo.stride(0):
            raise ValueError("the two planes share one row stride")
        d.copy_bf16, d.copy_lo, d.ld_copy = hi.data_ptr(), lo.data_ptr(), hi.stride(0)
    if planes_res is not None:
        hi, lo = _plane(planes_res[0]), _plane(planes_res[1])
        if hi.stride(0) != lo.stride(0) or residual is not None:
            raise ValueError("the two planes share one row stride, and replace `residual`")
        d.residual_hi, d.residual_lo, d.ld_res_planes = hi.data_ptr(), lo.data_ptr(), hi.stride(0)
    if ln_out is not None:
        parts, copy = ln_out
        if parts.dtype != torch.float32 or not parts.is_contiguous() or parts.numel() < row_lse_parts(N) * Mv * 2:
            raise ValueError("ln_out = (fp32 [ceil(N/128), M, 2] contiguous, bf16 [M, N] row-major or None with planes_out)")
        _need_cuda(parts)
        d.ln_out = parts.data_ptr()
        if copy is not None:
            d.copy_bf16, d.ld_copy = _plane(copy).data_ptr(), copy.stride(-2)
        elif planes_out is None:
            raise ValueError("ln_out needs the bf16 copy of the rows (its second element, or planes_out)")
    if ln_in is not None:
        parts, colsum, eps = ln_in
        n_parts = row_lse_parts(K)
        if (parts.dtype != torch.float32 or not parts.is_contiguous() or parts.numel() < n_parts * Mv * 2
                or colsum.dtype != torch.float32 or colsum.numel() != N or not colsum.is_contiguous()):
            raise ValueError("ln_in = (fp32 [ceil(K/128), M, 2] contiguous, fp32 [N], eps)")
        _need_cuda(parts, colsum)
        d.ln_in, d.ln_parts, d.ln_eps, d.ln_colsum = parts.data_ptr(), n_parts, eps, colsum.data_ptr()
    if GEMM_PROFILE is None:
        check(lib.avsr_gemm_nt(_stream(), ctypes.byref(d)), "avsr_gemm_nt")
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.avsr_gemm_nt(_stream(), ctypes.byref(d)), "avsr_gemm_nt")
    e1.record()
    # algorithmic bytes: each operand read once (a convolution's input image once, not once per tap), the output
    # written once, the residual read once where the period does not make it a broadcast
    a_elems = a.numel() if conv is not None else Mv * K * batch
    nbytes = a.element_size() * (a_elems + N * K * (batch if batched else 1))
    if out is not None:
        nbytes += out.element_size() * Mv * N * batch
        if residual is not None:
            nbytes += out.element_size() * (res_period if res_period else Mv) * N * batch
    elif residual is not None:
        nbytes += residual.element_size() * Mv * N
    if planes_out is not None:          # two bf16 planes: the bytes of an fp32 result
        nbytes += 4 * Mv * N
    elif ln_out is not None:            # the bf16 copy beside the fp32 result
        nbytes += 2 * Mv * N
    if planes_res is not None:
        nbytes += 4 * Mv * N
    rec = (e0, e1, 2.0 * Mv * N * K * batch, "bf16" if a.dtype == torch.bfloat16 else "f32",
           (Mv, N, K, batch, None if conv is None else tuple(conv), nbytes))
    if GEMM_PROFILE_REPEAT:        # scripts only: time an immediate second launch of the same product (cache-warm)
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2.record()
        check(lib.avsr_gemm_nt(_stream(), ctypes.byref(d)), "avsr_gemm_nt")
        e3.record()
        rec = rec + (e2, e3)
    GEMM_PROFILE.append(rec)
    return out


def ln_fold_weights(weight, bias, gamma, beta, dtype=torch.bfloat16):
    """LayerNorm(x; gamma, beta) @ weight^T + bias as a product on the UN-normalised x:
        rstd * (x @ w_g^T) - rstd * mean * colsum + bias_b,
    w_g = weight * gamma (rounded to ``dtype``: what the MFMA multiplies), colsum[n] = sum_k w_g[n, k] (of the rounded
    values, so the mean's term cancels what the product accumulated), bias_b = bias + weight @ beta.
    Returns (w_g, colsum, bias_b) for ``gemm_nt(..., ln_in=)``."""
    w = weight.detach().float()
    w_g = (w * gamma.detach().float()[None, :]).to(dtype).contiguous()
    colsum = w_g.float().sum(1).contiguous()
    bias_b = (w @ beta.detach().float())
    if bias is not None:
        bias_b = bias_b + bias.detach().float()
    return w_g, colsum, bias_b.contiguous()


def gemm_nt_splitk(a, b, S, out_dtype=None):
    """a [M,K] @ b [N,K]^T with K cut into S slices computed as S batched GEMMs (fp32 partials) and
    summed: for products with few output tiles and a very long K.  K/S must be a multiple of 8."""
    lib = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    Kc = K // S
    if Kc * S != K or Kc % 8 or a.stride(1) != 1 or b.stride(1) != 1 or (M * N) % 4:
        raise ValueError("unsupported split")
    a3 = torch.as_strided(a, (S, M, Kc), (Kc, a.stride(0), 1))
    b3 = torch.as_strided(b, (S, N, Kc), (Kc, b.stride(0), 1))
    parts = gemm_nt(a3, b3, out_dtype=torch.float32)                  # [S, M, N]
    out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
    check(lib.avsr_sum_parts(_stream(), _DT[out.dtype], parts.data_ptr(), M * N, S, None, 0, out.data_ptr(), M * N),
          "avsr_sum_parts")
    return out


def gemm_stats_slabs(M, N, dtype, device):
    """Zero-free allocation of the [2, parts, N] partial-sum slabs a gemm_nt / conv_nhwc launch with
    M output rows fills completely (pass ``col_stats=(s[0], s[1])``)."""
    parts = _lib.load().avsr_gemm_stats_parts(_DT[dtype], M)
    return torch.empty((2, parts, N), dtype=torch.float32, device=device)


def conv_out_rows(x, KH, KW, stride, pad):
    n, H, W, _ = x.shape
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    return n * ((H + 2 * ph - KH) // stride + 1) * ((W + 2 * pw - KW) // stride + 1)


def conv_nhwc(x, w, KH, KW, stride, pad, **kw):
    """x: [N,H,W,C] contiguous; w: [Cout, KH*KW*C] -> [N,OH,OW,Cout].  ``pad`` is an int or (pad_h, pad_w)."""
    n, H, W, C = x.shape
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    OH = (H + 2 * ph - KH) // stride + 1
    OW = (W + 2 * pw - KW) // stride + 1
    y = gemm_nt(x, w, conv=(H, W, C, KH, KW, stride, ph, pw, OH, OW), **kw)
    return None if y is None else y.view(n, OH, OW, w.shape[0])


def layernorm_fwd(x, gamma, beta, *, out_dtype=None, add=None, save_stats=False, eps=1e-5):
    lib = _lib.load()
    _need_cuda(x, gamma, beta, add)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    rows = x2.shape[0]
    y = torch.empty((rows, D), dtype=out_dtype or x.dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    period = add.shape[0] if add is not None else 0
    check(lib.avsr_layernorm_fwd(_stream(), _dt(x2), _DT[y.dtype], x2.data_ptr(), x2.stride(0), y.data_ptr(), D,
                                 gamma.data_ptr(), beta.data_ptr(), _p(add), period, _p(mean), _p(rstd), rows, D, eps),
          "avsr_layernorm_fwd")
    y = y.view(*x.shape[:-1], D)
    return (y, mean, rstd) if save_stats else y


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta):
    lib = _lib.load()
    D = x.shape[-1]
    x2, dy2 = x.reshape(-1, D), dy.reshape(-1, D)
    dx = torch.empty_like(dy2)
    check(lib.avsr_layernorm_bwd(_stream(), _dt(x2), _dt(dy2), dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0),
                                 gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), D, dgamma.data_ptr(),
                                 dbeta.data_ptr(), x2.shape[0], D), "avsr_layernorm_bwd")
    return dx.view(dy.shape)


def _bthd(t, H):
    """[B,T,*] view whose last dim holds H heads of 64 starting at the view's offset."""
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("attention operands must be [B,T,H*64] views with unit inner stride")
    if t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError("attention operands must have batch stride = T * row stride")
    return t.stride(1)


# AVSR_ATTN_SPLIT=0 keeps attn_fwd on the one-workgroup-per-item grid even when a workspace is given (A/B switch).
ATTN_SPLIT = os.environ.get("AVSR_ATTN_SPLIT", "1") != "0"


class AttnWorkspace:
    """Scratch for attn_fwd's split last round (avsr_attn_fwd_ws): a byte buffer zeroed when it is allocated - the
    kernel leaves its counters zero - and reused by every call of one stream of calls (e.g. the layers of an encoder).
    Calls that share one must not overlap on the device."""

    def __init__(self):
        self._buf = None

    def get(self, nbytes, device):
        if self._buf is None or self._buf.numel() < nbytes or self._buf.device != device:
            self._buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        return self._buf


def attn_fwd(q, k, v, H, *, key_mask=None, scale=1.0, use_mfma=False, save_lse=False, p_drop=0.0, seed=0, causal=False,
             workspace=None):
    """q: [B,Tq,H*64] view, k/v: [B,Tk,H*64] views (may be column slices of a fused QKV buffer).
    ``causal``: query i sees keys <= i + (Tk - Tq); forward only (attn_bwd has no causal form).
    ``use_mfma``: False = fp32-math kernel, True = bf16 MFMA, "fp8" = e4m3 MFMA products (forward only).
    ``workspace``: an AttnWorkspace lets the bf16 MFMA kernel split the items of a half-empty last round of
    workgroups over the keys (same result up to fp32 summation order)."""
    use_mfma = 2 if use_mfma == "fp8" else int(bool(use_mfma))
    lib = _lib.load()
    _need_cuda(q, k, v, key_mask)
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    o = torch.empty((B, Tq, H * 64), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if save_lse else None
    if key_mask is not None and (key_mask.dtype != torch.uint8 or not key_mask.is_contiguous()):
        raise ValueError("key_mask must be contiguous uint8 [B,Tk]")
    ws, ws_bytes = None, 0
    if workspace is not None and use_mfma == 1 and ATTN_SPLIT:
        ws_bytes = int(lib.avsr_attn_fwd_workspace_bytes(B, H, Tq, Tk, p_drop, int(causal)))
        if ws_bytes:
            ws = workspace.get(ws_bytes, q.device)
    check(lib.avsr_attn_fwd_ws(_stream(), _dt(q), int(use_mfma), q.data_ptr(), _bthd(q, H), k.data_ptr(), _bthd(k, H),
                               v.data_ptr(), _bthd(v, H), o.data_ptr(), H * 64, _p(key_mask), _p(lse), B, H, Tq, Tk, 64,
                               scale, p_drop, seed, int(causal), _p(ws), ws.numel() if ws is not None else 0), "avsr_attn_fwd_ws")
    return (o, lse) if save_lse else o


def attn_bwd(q, k, v, o, do, lse, H, *, key_mask=None, scale=1.0, use_mfma=False, p_drop=0.0, seed=0):
    lib = _lib.load()
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    dq = torch.empty((B, Tq, H * 64), dtype=q.dtype, device=q.device)
    dk = torch.empty((B, Tk, H * 64), dtype=q.dtype, device=q.device)
    dv = torch.empty((B, Tk, H * 64), dtype=q.dtype, device=q.device)
    ws = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    do = do.contiguous()
    check(lib.avsr_attn_bwd(_stream(), _dt(q), int(use_mfma), q.data_ptr(), _bthd(q, H), k.data_ptr(), _bthd(k, H), v.data_ptr(),
                            _bthd(v, H), o.data_ptr(), _bthd(o, H), do.data_ptr(), _bthd(do, H), dq.data_ptr(), H * 64,
                            dk.data_ptr(), H * 64, dv.data_ptr(), H * 64, _p(key_mask), lse.data_ptr(), ws.data_ptr(),
                            B, H, Tq, Tk, 64, scale, p_drop, seed), "avsr_attn_bwd")
    return dq, dk, dv


def stem_pack_weight(weight):
    """[64,3,5,3,3] conv weight -> [135 taps, 64] fp32, the layout the stem kernel stages."""
    if tuple(weight.shape) != (64, 3, 5, 3, 3):
        raise ValueError("stem weight must be [64,3,5,3,3]")
    return weight.detach().float().reshape(64, 135).t().contiguous()


def stem_pack_frags(weight_packed):
    """bf16 MFMA-fragment copy of a stem_pack_weight() result (the operand of the MFMA stem kernel)."""
    lib = _lib.load()
    _need_cuda(weight_packed)
    frags = torch.empty(int(lib.avsr_stem3d_frag_elems()), dtype=torch.bfloat16, device=weight_packed.device)
    check(lib.avsr_stem3d_pack_frags(_stream(), weight_packed.data_ptr(), frags.data_ptr()), "avsr_stem3d_pack_frags")
    return frags


def stem_stats_slabs(video):
    B, T, _, H, W = video.shape
    parts = _lib.load().avsr_stem3d_stats_parts(B, T, H, W)
    return torch.empty((2, parts, 64), dtype=torch.float32, device=video.device)


def store_u64(slot, value, copy=None):
    """slot: int64 tensor of one element on the GPU; ``value`` is written to it in stream order.
    ``copy`` = (dst, src): contiguous int64 GPU tensors of one size, copied by the same launch."""
    _need_cuda(slot)
    if slot.dtype != torch.int64 or slot.numel() != 1:
        raise ValueError("store_u64 expects a one-element int64 tensor")
    cd = cs = None
    n = 0
    if copy is not None:
        cd, cs = copy
        _need_cuda(cd, cs)
        if (cd.dtype != torch.int64 or cs.dtype != torch.int64 or cd.numel() != cs.numel() or not cd.is_contiguous()
                or not cs.is_contiguous()):
            raise ValueError("store_u64 copies between contiguous int64 tensors of one size")
        n = cd.numel()
    check(_lib.load().avsr_store_u64(_stream(), slot.data_ptr(), int(value), _p(cd), _p(cs), n), "avsr_store_u64")


def stem3d_fwd(video, weight, out_dtype, *, scale=None, shift=None, col_stats=None, frags=None, video_slot=None):
    """video [B,T,3,H,W] fp32 contiguous -> NHWC [B*T,OH,OW,64]; weight = stem_pack_weight(w);
    frags = stem_pack_frags(weight) selects the MFMA kernel for bf16 outputs.
    video_slot: one-element int64 GPU tensor; the kernel reads the input's address from it when it runs (store_u64)
    and ``video`` only gives the shape - the launch can be recorded once and replayed on other batches in place."""
    lib = _lib.load()
    _need_cuda(video, weight)
    if video.dtype != torch.float32 or not video.is_contiguous() or weight.dtype != torch.float32:
        raise ValueError("stem expects contiguous fp32 video and fp32 weight")
    B, T, C, H, W = video.shape
    if C != 3 or tuple(weight.shape) != (135, 64) or not weight.is_contiguous():
        raise ValueError("stem shape mismatch (weight must be packed with stem_pack_weight)")
    OH, OW = (H + 6 - 3) // 2 + 1, (W + 6 - 3) // 2 + 1
    out = torch.empty((B * T, OH, OW, 64), dtype=out_dtype, device=video.device)
    cs = col_stats or (None, None)
    if video_slot is not None:
        _need_cuda(video_slot)
        if video_slot.dtype != torch.int64 or video_slot.numel() != 1 or video.data_ptr() % 16:
            raise ValueError("video_slot must be a one-element int64 tensor and the video 16-byte aligned")
        check(lib.avsr_stem3d_fwd_indirect(_stream(), _DT[out_dtype], video_slot.data_ptr(), weight.data_ptr(), _p(frags),
                                           out.data_ptr(), _p(scale), _p(shift), _p(cs[0]), _p(cs[1]), B, T, H, W),
              "avsr_stem3d_fwd_indirect")
        return out
    check(lib.avsr_stem3d_fwd(_stream(), _DT[out_dtype], video.data_ptr(), weight.contiguous().data_ptr(), _p(frags), out.data_ptr(),
                              _p(scale), _p(shift), _p(cs[0]), _p(cs[1]), B, T, H, W), "avsr_stem3d_fwd")
    return out


MEL_WHISPER, MEL_SPEC_LN = 0, 1


def logmel_tables(variant, device):
    """Constant tables of avsr_logmel_fwd for one variant: (DFT matrix with the hann window folded in, mel bank)."""
    lib = _lib.load()
    dft = torch.empty(int(lib.avsr_logmel_dft_table_floats()), dtype=torch.float32, device=device)
    mel = torch.empty(int(lib.avsr_logmel_mel_table_floats()), dtype=torch.float32, device=device)
    _need_cuda(dft, mel)
    check(lib.avsr_logmel_tables(_stream(), variant, dft.data_ptr(), mel.data_ptr()), "avsr_logmel_tables")
    return dft, mel


def logmel_fwd(wave, tables, variant, *, n_samples=None, n_frames=None, out_dtype=torch.float32, channels_last=False,
               out=None, static_workspace=False):
    """wave [B, n_valid] fp32 (read as zero-padded to ``n_samples``) -> mel features [B,80,n_frames], or
    [B,n_frames,80] with ``channels_last``.  ``static_workspace``: use the library's fixed per-GPU scratch array
    instead of allocating a workspace (with a caller-provided ``out`` the call allocates nothing)."""
    lib = _lib.load()
    _need_cuda(wave, *tables)
    if wave.dim() != 2 or wave.dtype != torch.float32 or wave.stride(1) != 1:
        raise ValueError("logmel expects a [B, n] fp32 waveform with unit sample stride")
    B, n_valid = wave.shape
    n_samples = int(n_samples or n_valid)
    if n_frames is None:
        n_frames = n_samples // 160 if variant == MEL_WHISPER else 3000
    shape = (B, n_frames, 80) if channels_last else (B, 80, n_frames)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=wave.device)
    elif tuple(out.shape) != shape or out.dtype != out_dtype or not out.is_contiguous() or not out.is_cuda:
        raise ValueError("out must be a contiguous GPU tensor of the result's shape and dtype")
    st, sm = (80, 1) if channels_last else (1, n_frames)
    if static_workspace:
        check(lib.avsr_logmel_fwd_static(_stream(), variant, _DT[out_dtype], wave.data_ptr(), wave.stride(0), n_samples,
                                         n_valid, B, tables[0].data_ptr(), tables[1].data_ptr(), out.data_ptr(),
                                         80 * n_frames, st, sm, n_frames), "avsr_logmel_fwd_static")
        return out
    ws = torch.empty(int(lib.avsr_logmel_workspace_floats(B, n_frames)), dtype=torch.float32, device=wave.device)
    check(lib.avsr_logmel_fwd(_stream(), variant, _DT[out_dtype], wave.data_ptr(), wave.stride(0), n_samples, n_valid, B,
                              tables[0].data_ptr(), tables[1].data_ptr(), out.data_ptr(), 80 * n_frames, st, sm,
                              n_frames, ws.data_ptr()), "avsr_logmel_fwd")
    return out


def logmel_augment_fwd(wave, tables, *, noise, snr_db, freq_start=None, time_start=None, freq_width=10, time_width=48,
                       n_frames=3000, out_dtype=torch.float32, channels_last=True):
    """Training-time mel-spectrogram pipeline: mel power -> SpecAugment bands -> noise at a target SNR per mel
    row -> clip layer norm.  wave [B,n] fp32; noise [B,n_frames] or [n_frames] fp32; snr_db [B] fp32;
    freq_start / time_start int32 [B,2] (negative = no band): starts of the bands of ``freq_width`` mel bins and
    of ``time_width`` frames.  The defaults are what the reference's SpecAugment does to its [80, frames] tensor
    (datamodule/transforms.py:127,152-177): 80 // 8 = 10 mel bins and 48 frames."""
    lib = _lib.load()
    _need_cuda(wave, noise, snr_db, freq_start, time_start, *tables)
    if wave.dim() != 2 or wave.dtype != torch.float32 or wave.stride(1) != 1:
        raise ValueError("logmel expects a [B, n] fp32 waveform with unit sample stride")
    B, n = wave.shape
    if noise.dtype != torch.float32 or noise.shape[-1] != n_frames or noise.stride(-1) != 1:
        raise ValueError("noise must be fp32 with n_frames values per clip")
    if noise.dim() == 2 and noise.shape[0] != B:
        raise ValueError("noise must be [n_frames] or [B, n_frames]")
    if snr_db.dtype != torch.float32 or snr_db.numel() != B:
        raise ValueError("snr_db must be fp32 [B]")
    for st_ in (freq_start, time_start):
        if st_ is not None and (st_.dtype != torch.int32 or tuple(st_.shape) != (B, 2) or not st_.is_contiguous()):
            raise ValueError("band starts must be contiguous int32 [B,2]")
    shape = (B, n_frames, 80) if channels_last else (B, 80, n_frames)
    out = torch.empty(shape, dtype=out_dtype, device=wave.device)
    st, sm = (80, 1) if channels_last else (1, n_frames)
    ws = torch.empty(int(lib.avsr_logmel_augment_workspace_floats(B, n_frames)), dtype=torch.float32, device=wave.device)
    check(lib.avsr_logmel_augment_fwd(_stream(), _DT[out_dtype], wave.data_ptr(), wave.stride(0), n, n, B,
                                      tables[0].data_ptr(), tables[1].data_ptr(), out.data_ptr(), 80 * n_frames, st, sm,
                                      n_frames, ws.data_ptr(), _p(freq_start), freq_width, _p(time_start), time_width,
                                      noise.data_ptr(), noise.stride(0) if noise.dim() == 2 else 0,
                                      snr_db.contiguous().data_ptr()), "avsr_logmel_augment_fwd")
    return out


def video_preprocess(video, *, size=None, flip=None, frame_zero=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """video [B,T,3,H,W] uint8 or fp32 (0..255) -> fp32 [B,T,3,OH,OW]: resize (``size`` = (OH, OW) or None),
    /255, per-clip horizontal flip (uint8 [B]), zeroed frames (uint8 [B,T]), ImageNet normalisation."""
    import ctypes
    lib = _lib.load()
    _need_cuda(video, flip, frame_zero)
    if video.dim() != 5 or video.shape[2] != 3 or not video.is_contiguous() or video.dtype not in (torch.uint8, torch.float32):
        raise ValueError("video must be a contiguous [B,T,3,H,W] uint8 or fp32 tensor")
    B, T, _, H, W = video.shape
    OH, OW = size if size is not None else (H, W)
    for m_, shp in ((flip, (B,)), (frame_zero, (B, T))):
        if m_ is not None and (m_.dtype != torch.uint8 or tuple(m_.shape) != shp or not m_.is_contiguous()):
            raise ValueError("flip must be uint8 [B] and frame_zero uint8 [B,T], contiguous")
    out = torch.empty((B, T, 3, OH, OW), dtype=torch.float32, device=video.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib.avsr_video_preprocess(_stream(), int(video.dtype == torch.uint8), video.data_ptr(), out.data_ptr(), B, T, H, W,
                                    OH, OW, _p(flip), _p(frame_zero), ctypes.cast(m3, ctypes.c_void_p),
                                    ctypes.cast(s3, ctypes.c_void_p)), "avsr_video_preprocess")
    return out


def video_color(video, *, order=None, factors=None, grey=None, frame_zero=None, mean=(0.485, 0.456, 0.406),
                std=(0.229, 0.224, 0.225)):
    """In place on video fp32 [B,T,3,H,W] with values in [0, 1]: ColorJitter steps in the drawn order (int32 [B,4],
    factors fp32 [B,4]), RandomGrayscale (uint8 [B]), zeroed frames (uint8 [B,T]), normalisation."""
    lib = _lib.load()
    _need_cuda(video, order, factors, grey, frame_zero)
    if video.dim() != 5 or video.shape[2] != 3 or not video.is_contiguous() or video.dtype != torch.float32:
        raise ValueError("video must be a contiguous fp32 [B,T,3,H,W] tensor")
    B, T, _, H, W = video.shape
    for t_, dt_, shp in ((order, torch.int32, (B, 4)), (factors, torch.float32, (B, 4)), (grey, torch.uint8, (B,)),
                         (frame_zero, torch.uint8, (B, T))):
        if t_ is not None and (t_.dtype != dt_ or tuple(t_.shape) != shp or not t_.is_contiguous()):
            raise ValueError("order int32 [B,4], factors fp32 [B,4], grey uint8 [B], frame_zero uint8 [B,T], contiguous")
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib.avsr_video_color(_stream(), video.data_ptr(), B, T, H, W, _p(order), _p(factors), _p(grey), _p(frame_zero),
                               ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p)), "avsr_video_color")
    return video


def maxpool3x3s2(x, *, scale=None, shift=None, out_dtype=None):
    lib = _lib.load()
    N, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, OH, OW, C), dtype=out_dtype or x.dtype, device=x.device)
    check(lib.avsr_maxpool3x3s2(_stream(), _dt(x), _DT[y.dtype], x.data_ptr(), y.data_ptr(), _p(scale), _p(shift), N, H, W, C),
          "avsr_maxpool3x3s2")
    return y


def avgpool_mask(x, lens, B, T, out_dtype):
    """x NHWC [B*T,h,w,C] -> [B,T,C] mean over pixels, frames t>=lens[b] zeroed."""
    lib = _lib.load()
    n, h, w, C = x.shape
    y = torch.empty((B, T, C), dtype=out_dtype, device=x.device)
    check(lib.avsr_avgpool_mask_fwd(_stream(), _dt(x), _DT[out_dtype], x.data_ptr(), y.data_ptr(), _p(lens), B, T, h * w, C),
          "avsr_avgpool_mask_fwd")
    return y


def bn_apply_avgpool_mask(x, residual, scale, shift, lens, B, T, out_dtype):
    """mean over pixels of relu(x * scale + shift + residual) for NHWC bf16 x / residual [B*T,h,w,C] -> [B,T,C], frames
    t >= lens[b] zeroed: the trunk's last BN apply and the average pool in one pass."""
    lib = _lib.load()
    _need_cuda(x, residual, scale, shift)
    n, h, w, C = x.shape
    if x.dtype != torch.bfloat16 or residual.dtype != torch.bfloat16 or residual.shape != x.shape:
        raise ValueError("bn_apply_avgpool_mask takes bf16 NHWC tensors of one shape")
    if not (x.is_contiguous() and residual.is_contiguous()):
        raise ValueError("bn_apply_avgpool_mask takes contiguous tensors")
    y = torch.empty((B, T, C), dtype=out_dtype, device=x.device)
    check(lib.avsr_bn_apply_avgpool_mask(_stream(), _DT[out_dtype], x.data_ptr(), residual.data_ptr(), scale.data_ptr(),
                                         shift.data_ptr(), y.data_ptr(), _p(lens), B, T, h * w, C),
          "avsr_bn_apply_avgpool_mask")
    return y


def bn_finalize(col_sum, col_sumsq, count, weight, bias, running_mean, running_var, scale, shift, eps=1e-5, momentum=0.1):
    """col_sum / col_sumsq: [parts, C] partial-sum slabs (or [C])."""
    lib = _lib.load()
    C = weight.numel()
    nparts = col_sum.numel() // C
    ws = tickets = None
    stream = _stream()
    if nparts >= 512:          # many slab rows: a workspace lets the reduction spread over several workgroups
        ws = torch.empty(lib.avsr_bn_finalize_workspace_floats(C), dtype=torch.float32, device=weight.device)
        key = (weight.device, stream)
        tickets = _BN_TICKETS.get(key)
        if tickets is None or tickets.numel() * 64 < C:      # zero once; every launch leaves them zero
            tickets = _BN_TICKETS[key] = torch.zeros(max(64, (C + 63) // 64), dtype=torch.int32, device=weight.device)
    check(lib.avsr_bn_finalize(stream, col_sum.data_ptr(), col_sumsq.data_ptr(), nparts, float(count), weight.data_ptr(),
                               bias.data_ptr(), _p(running_mean), _p(running_var), scale.data_ptr(), shift.data_ptr(),
                               C, eps, momentum, _p(ws), _p(tickets)), "avsr_bn_finalize")


BN_FUSED_APPLY_MAX_PARTS = 256


def bn_finalize_apply(x, col_sum, col_sumsq, count, weight, bias, running_mean, running_var, *, residual=None, relu=True,
                      out=None, affine_out=None, eps=1e-5, momentum=0.1):
    """bn_finalize + bn_apply in one launch: bf16 x [..., C] (C a multiple of 64), slabs of at most
    BN_FUSED_APPLY_MAX_PARTS rows.  ``affine_out`` = (scale, shift) also receives the affine."""
    lib = _lib.load()
    C = weight.numel()
    nparts = col_sum.numel() // C
    y = out if out is not None else torch.empty_like(x)
    _need_cuda(x, col_sum, col_sumsq, weight, bias, residual, y)
    if x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16 or not x.is_contiguous() or not y.is_contiguous() or (
            residual is not None and (residual.dtype != torch.bfloat16 or not residual.is_contiguous())):
        raise ValueError("bn_finalize_apply takes contiguous bf16 tensors")
    sc, sh = affine_out if affine_out is not None else (None, None)
    check(lib.avsr_bn_finalize_apply(_stream(), x.data_ptr(), _p(residual), y.data_ptr(), col_sum.data_ptr(),
                                     col_sumsq.data_ptr(), nparts, float(count), weight.data_ptr(), bias.data_ptr(),
                                     _p(running_mean), _p(running_var), _p(sc), _p(sh), x.numel() // C, C, eps, momentum,
                                     int(relu)), "avsr_bn_finalize_apply")
    return y


def gram_stats(x, affine=None):
    """Per-share partial Gram matrices and column sums of x [M, K] bf16 (K = 64 or 128, unit inner stride):
    (gram_parts fp32 [parts, K, K], colsum_parts fp32 [parts, K]) for bn_gram_finalize.
    ``affine`` = (scale, shift): x is a raw convolution output; the statistics are those of relu(x * scale + shift),
    which replaces x in place."""
    lib = _lib.load()
    _need_cuda(x)
    M, K = x.shape
    if x.dtype != torch.bfloat16 or x.stride(1) != 1:
        raise ValueError("gram_stats takes a bf16 [M, K] matrix with unit inner stride")
    parts = int(lib.avsr_gram_parts(M, K))
    if parts <= 0:
        raise _lib.AvsrHipError("gram_stats: K must be 64 or 128")
    gp = torch.empty((parts, K, K), dtype=torch.float32, device=x.device)
    cp = torch.empty((parts, K), dtype=torch.float32, device=x.device)
    if affine is not None:
        _need_cuda(*affine)
        check(lib.avsr_gram_stats_bn(_stream(), x.data_ptr(), x.stride(0), M, K, affine[0].data_ptr(), affine[1].data_ptr(),
                                     x.data_ptr(), x.stride(0), gp.data_ptr(), cp.data_ptr()), "avsr_gram_stats_bn")
        return gp, cp
    check(lib.avsr_gram_stats(_stream(), x.data_ptr(), x.stride(0), M, K, gp.data_ptr(), cp.data_ptr()), "avsr_gram_stats")
    return gp, cp


def bn_gram_finalize(gram_parts, colsum_parts, count, w, weight, bias, running_mean, running_var, scale, shift,
                     eps=1e-5, momentum=0.1, reduced=None):
    """Training-mode BN affine (and running-statistics update) of the 1x1 convolution with weight ``w`` (bf16 [N, K])
    from the input's Gram statistics (gram_stats).  Returns the workspace holding the reduced statistics; pass it back
    as ``reduced`` for another convolution over the SAME input (the partials are then not reduced again)."""
    lib = _lib.load()
    parts, K = colsum_parts.shape
    N = w.shape[0]
    if w.dtype != torch.bfloat16 or w.stride(1) != 1 or w.shape[1] != K:
        raise ValueError("bn_gram_finalize: w must be bf16 [N, K]")
    ws = reduced if reduced is not None else torch.empty(int(lib.avsr_bn_gram_workspace_doubles(K)), dtype=torch.float64,
                                                          device=w.device)
    check(lib.avsr_bn_gram_finalize(_stream(), None if reduced is not None else gram_parts.data_ptr(),
                                    None if reduced is not None else colsum_parts.data_ptr(), parts, K, float(count),
                                    w.data_ptr(), w.stride(0), N, weight.data_ptr(), bias.data_ptr(), _p(running_mean),
                                    _p(running_var), scale.data_ptr(), shift.data_ptr(), eps, momentum, ws.data_ptr()),
          "avsr_bn_gram_finalize")
    return ws


_BN_TICKETS = {}     # (device, stream) -> int32 ticket counters of the split batch-norm reduction


def bn_fold(weight, bias, running_mean, running_var, eps=1e-5):
    lib = _lib.load()
    scale, shift = torch.empty_like(weight), torch.empty_like(weight)
    check(lib.avsr_bn_fold(_stream(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                           scale.data_ptr(), shift.data_ptr(), weight.numel(), eps), "avsr_bn_fold")
    return scale, shift


def bn_apply(x, scale, shift, *, residual=None, residual_affine=None, relu=True, out=None, out_dtype=None):
    """y = act(x*scale + shift (+ residual)).  ``residual_affine=(scale, shift)``: the residual is a raw
    convolution output whose own (activation-free) BN is applied on the fly."""
    lib = _lib.load()
    C = x.shape[-1]
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    if residual is not None and residual.dtype != y.dtype:
        raise TypeError("residual must have the output dtype")
    rs, rb = residual_affine if residual_affine is not None else (None, None)
    check(lib.avsr_bn_apply(_stream(), _dt(x), _DT[y.dtype], x.data_ptr(), _p(residual), y.data_ptr(), scale.data_ptr(),
                            shift.data_ptr(), _p(rs), _p(rb), x.numel() // C, C, int(relu)), "avsr_bn_apply")
    return y


def row_lse_parts(N):
    """Column tiles of the per-row log-sum-exp partials a product with N columns writes (gemm_nt(row_part=...))."""
    return (N + 127) // 128


def head_loss_fwd(logits, targets, input_lengths, target_lengths, label_smoothing=0.1, ignore_index=-100, blank=0,
                  row_part=None):
    """logits fp32 [B,T,V] (row stride may exceed V).  Returns (out4, ws): out4 = [ctc, ce, n_ce, ctc + ce].
    ``row_part``: the log-sum-exp partials the product that made the logits wrote (gemm_nt(row_part=...)); the
    forward then does not read the logits row by row."""
    lib = _lib.load()
    B, T, V = logits.shape
    if logits.dtype != torch.float32 or logits.stride(2) != 1 or logits.stride(0) != T * logits.stride(1):
        raise ValueError("logits must be fp32 [B,T,V] with uniform row stride")
    Lw = targets.shape[1]
    nws = lib.avsr_head_loss_workspace_floats(B, T, Lw)
    ws = torch.empty(nws, dtype=torch.float32, device=logits.device)
    out3 = torch.empty(4, dtype=torch.float32, device=logits.device)
    check(lib.avsr_head_loss_fwd(_stream(), logits.data_ptr(), logits.stride(1), B, T, V, targets.data_ptr(),
                                 targets.stride(0), Lw, input_lengths.data_ptr(), target_lengths.data_ptr(),
                                 label_smoothing, ignore_index, blank, ws.data_ptr(), out3.data_ptr(), _p(row_part),
                                 0 if row_part is None else row_lse_parts(V)), "avsr_head_loss_fwd")
    return out3, ws


def head_loss_bwd(logits, targets, input_lengths, target_lengths, ws, grad_dtype, *, label_smoothing=0.1,
                  ignore_index=-100, blank=0, w_ctc=1.0, w_ce=1.0, up_ctc=None, up_ce=None, ld_out=None):
    lib = _lib.load()
    B, T, V = logits.shape
    ld = ld_out or V
    buf = torch.empty((B * T, ld), dtype=grad_dtype, device=logits.device)      # the kernel zeroes the columns past V
    check(lib.avsr_head_loss_bwd(_stream(), _DT[grad_dtype], logits.data_ptr(), logits.stride(1), B, T, V,
                                 targets.data_ptr(), targets.stride(0), targets.shape[1], input_lengths.data_ptr(),
                                 target_lengths.data_ptr(), label_smoothing, ignore_index, blank, ws.data_ptr(), w_ctc,
                                 w_ce, _p(up_ctc), _p(up_ce), buf.data_ptr(), ld), "avsr_head_loss_bwd")
    return buf


def argmax_rows(logits):
    lib = _lib.load()
    V = logits.shape[-1]
    rows = logits.numel() // V if logits.is_contiguous() else logits.shape[0] * logits.shape[1]
    ld = logits.stride(-2)
    out = torch.empty(logits.shape[:-1], dtype=torch.int64, device=logits.device)
    check(lib.avsr_argmax_rows(_stream(), logits.data_ptr(), ld, rows, V, out.data_ptr()), "avsr_argmax_rows")
    return out


def cast(x, dtype, scale=1.0):
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    if x.numel():
        check(lib.avsr_cast(_stream(), _dt(x), _DT[dtype], x.data_ptr(), y.data_ptr(), x.numel(), scale), "avsr_cast")
    return y


def zero_regions(*views):
    """Sets up to four 2-D views (unit inner stride, any row stride) to zero with one launch."""
    lib = _lib.load()
    _need_cuda(*views)
    n = len(views)
    if not 1 <= n <= 4 or any(v.dim() != 2 or v.stride(1) != 1 for v in views):
        raise ValueError("zero_regions takes one to four [rows, cols] views with unit inner stride")
    ptr = (ctypes.c_void_p * n)(*[v.data_ptr() for v in views])
    ld = (ctypes.c_int64 * n)(*[v.stride(0) * v.element_size() for v in views])
    rows = (ctypes.c_int64 * n)(*[v.shape[0] for v in views])
    rb = (ctypes.c_int64 * n)(*[v.shape[1] * v.element_size() for v in views])
    check(lib.avsr_zero_regions(_stream(), n, ptr, ld, rows, rb), "avsr_zero_regions")


def cast_rows(x, out, scale=1.0):
    """out[r, :] = x[r, :] (dtype conversion) for 2-D views with unit inner stride and any row stride."""
    lib = _lib.load()
    _need_cuda(x, out)
    if x.dim() != 2 or out.shape != x.shape or x.stride(1) != 1 or out.stride(1) != 1:
        raise ValueError("cast_rows takes two [rows, cols] views with unit inner stride")
    check(lib.avsr_cast_rows(_stream(), _dt(x), _DT[out.dtype], x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
                             x.shape[0], x.shape[1], scale), "avsr_cast_rows")
    return out


def transpose2d(x, dtype=None, scale=1.0, pad_to=1, with_copy=False, out=None, out_copy=None):
    """[..., R, C] (row-major, row stride may exceed C) -> [..., C, Rp] contiguous, Rp = R rounded up
    to a multiple of ``pad_to`` with the padding columns zero (optionally converting dtype).
    ``with_copy`` (2-D only) also returns the untransposed tensor in the output dtype: (y^T, y).
    ``out`` / ``out_copy``: buffers of an earlier call with the same arguments, rewritten in place (their padding
    columns are already zero and are not touched)."""
    lib = _lib.load()
    if x.stride(-1) != 1 or (x.dim() > 2 and not x.is_contiguous()):
        x = x.contiguous()
    R, C = x.shape[-2], x.shape[-1]
    ldx = x.stride(-2)
    Rp = (R + pad_to - 1) // pad_to * pad_to
    batch = x.numel() // (R * C)
    if out is not None:
        if tuple(out.shape) != (*x.shape[:-2], C, Rp) or not out.is_contiguous() or (dtype and out.dtype != dtype):
            raise ValueError("transpose2d: `out` does not match")
        y = out
    else:
        alloc = torch.zeros if Rp != R else torch.empty
        y = alloc((*x.shape[:-2], C, Rp), dtype=dtype or x.dtype, device=x.device)
    if with_copy and out_copy is not None:
        if tuple(out_copy.shape) != (R, C) or not out_copy.is_contiguous() or out_copy.dtype != y.dtype:
            raise ValueError("transpose2d: `out_copy` does not match")
        cp = out_copy
    else:
        cp = torch.empty((R, C), dtype=y.dtype, device=x.device) if with_copy else None
    check(lib.avsr_transpose(_stream(), _dt(x), _DT[y.dtype], x.data_ptr(), ldx, R * ldx, y.data_ptr(), Rp, C * Rp, batch,
                             R, C, scale, _p(cp), C), "avsr_transpose")
    return (y, cp) if with_copy else y


def gemm_tn_ok(a, b):
    """True when gemm_tn takes these operands as they are (bf16, unit inner stride, 16-byte rows)."""
    return (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
            and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def gemm_tn(a, b, out=None, accumulate=False, colsum=None, accumulate_colsum=False):
    """out[N,K] (+)= a^T b for row-major bf16 a [M,N], b [M,K] (fp32 result), and colsum[n] (+)= sum_m a[m][n]:
    the weight and bias gradients of a linear layer from dY and X as they lie in memory."""
    lib = _lib.load()
    _need_cuda(a, b, out, colsum)
    (M, N), K = a.shape, b.shape[1]
    if b.shape[0] != M:
        raise ValueError("gemm_tn: operands must have the same number of rows")
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=a.device)
        accumulate = False
    if out.dtype != torch.float32 or out.stride(1) != 1 or tuple(out.shape) != (N, K):
        raise ValueError("gemm_tn: out must be fp32 [N,K] with unit inner stride")
    if colsum is not None and (colsum.dtype != torch.float32 or not colsum.is_contiguous() or colsum.numel() != N):
        raise ValueError("gemm_tn: colsum must be contiguous fp32 [N]")
    check(lib.avsr_gemm_tn(_stream(), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                           M, N, K, int(accumulate), _p(colsum), int(accumulate_colsum)), "avsr_gemm_tn")
    return out


def colsum(x, out=None, accumulate=False):
    lib = _lib.load()
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    check(lib.avsr_colsum(_stream(), _dt(x), x.data_ptr(), x.stride(0), M, N, out.data_ptr(), int(accumulate)), "avsr_colsum")
    return out


def gelu_fwd(x):
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.avsr_gelu_fwd(_stream(), _dt(x), x.data_ptr(), y.data_ptr(), x.numel()), "avsr_gelu_fwd")
    return y


def gelu_bwd(x, dy):
    lib = _lib.load()
    dx = torch.empty_like(x)
    check(lib.avsr_gelu_bwd(_stream(), _dt(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel()), "avsr_gelu_bwd")
    return dx


def gated_residual_fwd(x, y, gate, p_drop=0.0, seed=0):
    lib = _lib.load()
    out = torch.empty_like(x)
    check(lib.avsr_gated_residual_fwd(_stream(), _dt(x), x.data_ptr(), y.data_ptr(), gate.data_ptr(), out.data_ptr(),
                                      x.numel(), p_drop, seed), "avsr_gated_residual_fwd")
    return out


def gated_residual_bwd(dout, y, gate, dgate, p_drop=0.0, seed=0):
    lib = _lib.load()
    dy = torch.empty_like(dout)
    check(lib.avsr_gated_residual_bwd(_stream(), _dt(dout), dout.data_ptr(), y.data_ptr(), gate.data_ptr(), dy.data_ptr(),
                                      dgate.data_ptr(), dout.numel(), p_drop, seed), "avsr_gated_residual_bwd")
    return dy


def length_mask(lengths, T):
    """lengths int64 [B] is clamped to T IN PLACE (models/av_net.py:96); returns the bool [B, T] padding mask
    (True = padded, models/av_net.py:111-114)."""
    lib = _lib.load()
    _need_cuda(lengths)
    if lengths.dtype != torch.int64 or not lengths.is_contiguous():
        raise ValueError("lengths must be a contiguous int64 tensor")
    B = lengths.numel()
    mask = torch.empty((B, T), dtype=torch.uint8, device=lengths.device)
    check(lib.avsr_length_mask(_stream(), lengths.data_ptr(), B, T, mask.data_ptr()), "avsr_length_mask")
    return mask.view(torch.bool)


def add3(a, b, c=None):
    lib = _lib.load()
    out = torch.empty_like(a)
    check(lib.avsr_add3(_stream(), _dt(a), a.data_ptr(), b.data_ptr(), _p(c), out.data_ptr(), a.numel()), "avsr_add3")
    return out


def xattn_block_desc(dtype, B, Tq, Tk, d, H, d_ff, p_drop, seed_attn, seed_ff, weights, vectors, seed_delta=None):
    """Descriptor of one gated cross-attention block for gated_xattn_fwd / _bwd.  ``weights`` = four (W, W^T)
    pairs in ``dtype`` for in_proj, out_proj, ff.0, ff.2 (W as nn.Linear stores it, W^T [in, ld >= out]);
    ``vectors`` = fp32 tensors (b_in, b_out, b_fc1, b_fc2, ln1_g, ln1_b, ln2_g, ln2_b, gate_attn, gate_ff).
    The descriptor keeps the tensors alive."""
    blk = _lib.XattnBlock()
    blk.dtype, blk.B, blk.Tq, blk.Tk, blk.d, blk.H, blk.d_ff = _DT[dtype], B, Tq, Tk, d, H, d_ff
    blk.p_drop, blk.seed_attn, blk.seed_ff = float(p_drop), int(seed_attn), int(seed_ff)
    # seed_delta: a device tensor whose first 8 bytes are added to both seeds when the kernels run (graph replays)
    blk.seed_delta = _p(seed_delta)
    _need_cuda(*vectors)
    for name, (w, wt) in zip(("in", "out", "fc1", "fc2"), weights):
        _need_cuda(w, wt)
        if w.dtype != dtype or wt.dtype != dtype or not w.is_contiguous() or wt.stride(-1) != 1:
            raise ValueError(f"weight copies of {name} must be contiguous tensors of the compute dtype")
        setattr(blk, "w_" + name, w.data_ptr())
        setattr(blk, f"w_{name}T", wt.data_ptr())
        setattr(blk, f"ld_{name}T", wt.stride(0))
    for name, v in zip(("b_in", "b_out", "b_fc1", "b_fc2", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "gate_attn", "gate_ff"),
                       vectors):
        if v.dtype != torch.float32 or not v.is_contiguous():
            raise ValueError(f"{name} must be a contiguous fp32 tensor")
        setattr(blk, name, v.data_ptr())
    blk._keep = (weights, vectors, seed_delta)
    return blk


def gated_xattn_fwd(blk, x, xa, key_mask):
    """One gated cross-attention block forward.  x [B*Tq, d], xa [B*Tk, d] or None, key_mask uint8 [B, Tk] or
    None -> (out [B*Tq, d], saved activations for gated_xattn_bwd)."""
    lib = _lib.load()
    _need_cuda(x, xa, key_mask)
    if not x.is_contiguous() or (xa is not None and not xa.is_contiguous()):
        raise ValueError("gated_xattn operands must be dense [rows, d] tensors")
    if _dt(x) != blk.dtype or tuple(x.shape) != (blk.B * blk.Tq, blk.d):
        raise ValueError("x does not match the block descriptor")
    if xa is not None and (_dt(xa) != blk.dtype or tuple(xa.shape) != (blk.B * blk.Tk, blk.d)):
        raise ValueError("xa does not match the block descriptor")
    if key_mask is not None and (key_mask.dtype != torch.uint8 or tuple(key_mask.shape) != (blk.B, blk.Tk)
                                 or not key_mask.is_contiguous()):
        raise ValueError("key_mask must be a contiguous uint8 [B, Tk] tensor")
    out = torch.empty_like(x)
    saved = torch.empty(int(lib.avsr_gated_xattn_fwd_workspace_bytes(ctypes.byref(blk))), dtype=torch.uint8, device=x.device)
    check(lib.avsr_gated_xattn_fwd(_stream(), ctypes.byref(blk), x.data_ptr(), _p(xa), _p(key_mask), out.data_ptr(),
                                   saved.data_ptr()), "avsr_gated_xattn_fwd")
    return out, saved


XATTN_GRAD_NAMES = ("dw_in", "db_in", "dw_out", "db_out", "dw_fc1", "db_fc1", "dw_fc2", "db_fc2",
                    "dln1_g", "dln1_b", "dln2_g", "dln2_b", "dgate_attn", "dgate_ff")


def gated_xattn_bwd(blk, x, xa, key_mask, saved, dout, grads, dxa=None):
    """Backward of gated_xattn_fwd.  ``grads``: 14 fp32 tensors (order XATTN_GRAD_NAMES) the parameter gradients
    are accumulated into.  ``dxa``: buffer to ADD d(loss)/d(xa) to, or None for a fresh tensor.
    Returns (dx, dxa)."""
    lib = _lib.load()
    _need_cuda(dout, dxa, *grads)
    if not dout.is_contiguous() or _dt(dout) != blk.dtype or dout.shape != x.shape:
        raise ValueError("dout must be a dense tensor shaped like x, in the compute dtype")
    g = _lib.XattnGrads()
    for name, t in zip(XATTN_GRAD_NAMES, grads):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous fp32 tensor")
        setattr(g, name, t.data_ptr())
    dx = torch.empty_like(x)
    accumulate = dxa is not None
    if xa is not None and dxa is None:
        dxa = torch.empty_like(xa)
    ws = torch.empty(int(lib.avsr_gated_xattn_bwd_workspace_bytes(ctypes.byref(blk))), dtype=torch.uint8, device=x.device)
    check(lib.avsr_gated_xattn_bwd(_stream(), ctypes.byref(blk), x.data_ptr(), _p(xa), _p(key_mask), saved.data_ptr(),
                                   dout.data_ptr(), dx.data_ptr(), _p(dxa) if xa is not None else None, int(accumulate),
                                   ctypes.byref(g), ws.data_ptr()), "avsr_gated_xattn_bwd")
    return dx, (dxa if xa is not None else None)


def add_table(x, table):
    """x [..., T, d] + table [T, d] (fp32, contiguous) broadcast over the leading dimensions."""
    lib = _lib.load()
    _need_cuda(x, table)
    if table.dtype != torch.float32 or not table.is_contiguous() or tuple(x.shape[-table.dim():]) != tuple(table.shape):
        raise ValueError("table must be a contiguous fp32 tensor matching x's trailing dimensions")
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib.avsr_add_table(_stream(), _dt(x), x.data_ptr(), table.data_ptr(), out.data_ptr(), x.numel(), table.numel()),
          "avsr_add_table")
    return out


def embed_tokens(ids, table, pos, past=0, scale=1.0):
    """ids [B,T] int64 -> fp32 [B*T, D]: table[ids] * scale + pos[past:past+T] (Whisper decoder input).
    Raises IndexError on an id outside the table, as nn.Embedding does."""
    lib = _lib.load()
    _need_cuda(ids, table, pos)
    if ids.dtype != torch.int64 or ids.dim() != 2:
        raise ValueError("ids must be int64 [B,T]")
    B, T = ids.shape
    V, D = table.shape
    ids = ids.contiguous()
    out = torch.empty((B * T, D), dtype=torch.float32, device=ids.device)
    bad = torch.zeros(1, dtype=torch.int32, device=ids.device)
    check(lib.avsr_embed_tokens(_stream(), _dt(table), ids.data_ptr(), table.data_ptr(), pos.data_ptr(), out.data_ptr(),
                                B * T, T, past, D, V, pos.shape[0], scale, bad.data_ptr()), "avsr_embed_tokens")
    if int(bad.item()):
        raise IndexError("token id out of range for the embedding table")
    return out


def sumsq_accumulate(x, out):
    lib = _lib.load()
    check(lib.avsr_sumsq_accumulate(_stream(), x.data_ptr(), x.numel(), out.data_ptr()), "avsr_sumsq_accumulate")


def clip_scale(sumsq, max_norm, scale, norm_out=None, pre_scale=1.0):
    """scale[0] = pre_scale * min(1, max_norm / (pre_scale * sqrt(sumsq[0]) + 1e-6)); norm_out[0] = the norm."""
    lib = _lib.load()
    check(lib.avsr_clip_scale(_stream(), sumsq.data_ptr(), max_norm, pre_scale, scale.data_ptr(), _p(norm_out)),
          "avsr_clip_scale")


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, *, lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01,
               grad_scale=None, bf16_shadow=None, clear_grad=False):
    lib = _lib.load()
    check(lib.avsr_adamw_step(_stream(), param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                              _p(bf16_shadow), param.numel(), lr, betas[0], betas[1], eps, weight_decay, step,
                              _p(grad_scale), int(bool(clear_grad))), "avsr_adamw_step")


def step_params(record, seed_delta, lr, betas, step):
    """Write one training step's scalars (dropout seed offset, learning rate, AdamW bias corrections for ``step``)
    into the 32-byte device ``record`` that graph-recorded kernels read; the values travel as kernel arguments."""
    lib = _lib.load()
    check(lib.avsr_step_params(_stream(), record.data_ptr(), int(seed_delta) & 0xFFFFFFFFFFFFFFFF, float(lr), betas[0],
                               betas[1], int(step)), "avsr_step_params")


def adamw_step_dev(param, grad, exp_avg, exp_avg_sq, record, *, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01,
                   grad_scale=None, clear_grad=False, reset=None):
    """adamw_step with lr and the bias corrections read from ``record`` (see step_params); ``reset``: a device float
    set to zero by the launch (the sum of squares of the step's clip factor)."""
    lib = _lib.load()
    check(lib.avsr_adamw_step_dev(_stream(), param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                  None, param.numel(), record.data_ptr() + 8, betas[0], betas[1], eps, weight_decay,
                                  _p(grad_scale), int(bool(clear_grad)), _p(reset)), "avsr_adamw_step_dev")
