// NT GEMM with fused epilogue and an implicit-GEMM (NHWC conv) A-operand
// loader.  Two kernels:
//   * bf16 inputs: 128x128x64 tile, 4 waves (2x2), 64x64 per wave as 4x4
//     v_mfma_f32_16x16x32_bf16, register-staged global->LDS double buffering,
//     XOR-swizzled 128-byte LDS rows read with ds_read_b128.
//   * f32 inputs: 64x64x16 tile on v_mfma_f32_16x16x4_f32 (exact fp32 fma
//     chain) - the reference-precision path used for fp32 parity.
// The weight operand B feeds the MFMA "A" slot, so an accumulator register
// quad holds 4 consecutive output columns n of one row m: stores, bias and
// residual reads are 8/16-byte vectors per lane.
#include "common.h"

struct GemmP {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc, sA, sB, sC;
    float alpha; const float* scale; const float* bias; int act;
    const void* res; long ldr; int res_period; const float* gate; float* csum; float* csq;
    int H, W, Cc, KH, KW, stride, pad_h, pad_w, OH, OW;
    int tiles_m, tiles_n, vec_ok;
};

// Block id -> (tile_m, tile_n).  Blocks are dealt round-robin over the 8 XCDs;
// first undo that so each XCD owns a contiguous run of ids, then walk the tile
// grid in groups of 8 row-tiles so the ~64 blocks resident on one XCD share
// 8 A panels and 8 B panels through that XCD's L2.
__device__ __forceinline__ void tile_coords(const GemmP& p, int& tm, int& tn) {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int g = wg / per_group;
    const int first_m = g * GM;
    const int gm = min(GM, p.tiles_m - first_m);
    const int in_g = wg - g * per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

// Implicit-GEMM addressing.  A row of the implicit matrix is an output pixel;
// its image base and top-left input coordinate are fixed for the whole K loop.
struct ConvRow { int base, ih0, iw0; bool ok; };
__device__ __forceinline__ ConvRow conv_row(const GemmP& p, int m) {
    ConvRow r;
    r.ok = m < p.M;
    const int mm = r.ok ? m : 0;
    const int ohw = p.OH * p.OW;
    const int img = mm / ohw, rem = mm - img * ohw;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    r.base = img * p.H * p.W;
    r.ih0 = oh * p.stride - p.pad_h;
    r.iw0 = ow * p.stride - p.pad_w;
    return r;
}
// Column kk of the implicit matrix = filter tap (kh,kw) and channel cc.
__device__ __forceinline__ void conv_col(const GemmP& p, int kk, int& kh, int& kw, int& cc) {
    const int tap = kk / p.Cc;
    cc = kk - tap * p.Cc;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
}
template <typename T>
__device__ __forceinline__ const T* conv_src(const GemmP& p, const T* base, const ConvRow& r, int kh, int kw,
                                             int cc, bool& ok) {
    const int ih = r.ih0 + kh, iw = r.iw0 + kw;
    ok = ok && r.ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
    return base + ((long)(r.base + (ok ? ih * p.W + iw : 0))) * p.Cc + cc;
}

// Epilogue for one 16x16 accumulator tile: this lane owns row m, columns
// nb..nb+3.  Returns the pre-activation values in `pre` for the statistics.
template <typename OutT>
__device__ __forceinline__ void epilogue_quad(const GemmP& p, f32x4 acc, int m, int nb, float g,
                                              long cbase, f32x4& pre) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        float x = acc[r] * p.alpha;
        if (n < p.N) {
            if (p.scale) x *= p.scale[n];
            if (p.bias) x += p.bias[n];
        }
        pre[r] = x;
        if (p.act == AVSR_ACT_GELU) x = gelu_erf(x);
        else if (p.act == AVSR_ACT_RELU) x = fmaxf(x, 0.f);
        v[r] = x;
    }
    const bool post_relu = p.act == AVSR_ACT_RELU_AFTER_RESIDUAL;
    if (m >= p.M) return;
    OutT* cp = (OutT*)p.C + cbase + (long)m * p.ldc + nb;
    const OutT* rp = p.res ? (const OutT*)p.res + (long)(p.res_period > 0 ? m % p.res_period : m) * p.ldr + nb : nullptr;
    if (p.vec_ok && nb + 3 < p.N) {
        if constexpr (sizeof(OutT) == 4) {
            if (rp) { const f32x4 rv = *(const f32x4*)rp;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rv[r] + g * v[r]; }
            if (post_relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)cp = o;
        } else {
            if (rp) { const bf16x4 rv = *(const bf16x4*)rp;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (float)rv[r] + g * v[r]; }
            if (post_relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *(bf16x4*)cp = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (nb + r < p.N) {
                float x = v[r];
                if (rp) x = to_f32<OutT>(rp[r]) + g * x;
                if (post_relu) x = fmaxf(x, 0.f);
                cp[r] = from_f32<OutT>(x);
            }
        }
    }
}

// Column sums of the pre-activation values over the valid rows of this wave's
// sub-tile; one atomic per column per wave.
__device__ __forceinline__ void stats_quad(const GemmP& p, f32x4 s, f32x4 s2, int nb, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = s[r], b = s2[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if ((lane & 15) == 0 && nb + r < p.N) {
            atomicAdd(p.csum + nb + r, a);
            atomicAdd(p.csq + nb + r, b);
        }
    }
}

// ---------------------------------------------------------------- bf16 ----
#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int lds_off(int row, int chunk) {   // in bf16 elements
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

template <bool CONV, typename OutT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][BM * BK];
    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;
    const bf16_t* A = (const bf16_t*)p.A + (long)bz * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long)bz * p.sB;
    const long cbase = (long)bz * p.sC;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lc = t & 7, lr = t >> 3;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[4], rb[4];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (p.K + BK - 1) / BK;

    ConvRow crow[4];
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) crow[i] = conv_row(p, m0 + lr + 32 * i);
    }
    auto load_tile = [&](int kt) {
        const int kk = kt * BK + lc * 8;
        const bool kin = kk < p.K;
        int kh = 0, kw = 0, cc = 0;
        if constexpr (CONV) conv_col(p, kin ? kk : 0, kh, kw, cc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lr + 32 * i;
            bool ok = kin && m < p.M;
            const bf16_t* src;
            if constexpr (CONV) src = conv_src<bf16_t>(p, A, crow[i], kh, kw, cc, ok);
            else src = A + (long)m * p.lda + kk;
            ra[i] = ok ? *(const bf16x8*)src : zero8;
            const int n = n0 + lr + 32 * i;
            const bool okb = kin && n < p.N;
            rb[i] = okb ? *(const bf16x8*)(B + (long)n * p.ldb + kk) : zero8;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 32 * i;
            *(bf16x8*)&lds[buf][0][lds_off(r, lc)] = ra[i];
            *(bf16x8*)&lds[buf][1][lds_off(r, lc)] = rb[i];
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const bf16_t* As = lds[buf][0];
        const bf16_t* Bs = lds[buf][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4], wf[4];
            const int ch = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + (lane & 15);
                xf[i] = *(const bf16x8*)&As[lds_off(row, ch)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + (lane & 15);
                wf[j] = *(const bf16x8*)&Bs[lds_off(row, ch)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float g = p.gate ? tanhf(*p.gate) : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            f32x4 pre;
            epilogue_quad<OutT>(p, acc[i][j], m, nb, g, cbase, pre);
            if (m < p.M) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[r] += pre[r]; s2[r] += pre[r] * pre[r]; }
            }
        }
        if (p.csum) stats_quad(p, s, s2, nb, lane);
    }
}

// ----------------------------------------------------------------- f32 ----
#define FM 64
#define FN 64
#define FK 16
#define FLD 17

template <bool CONV, typename OutT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    __shared__ float As[FM * FLD];
    __shared__ float Bs[FN * FLD];
    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * FM, n0 = tn * FN;
    const int bz = blockIdx.z;
    const float* A = (const float*)p.A + (long)bz * p.sA;
    const float* B = (const float*)p.B + (long)bz * p.sB;
    const long cbase = (long)bz * p.sC;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = t >> 2, lc = t & 3;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + FK - 1) / FK;
    ConvRow crow;
    if constexpr (CONV) crow = conv_row(p, m0 + lr);
    for (int kt = 0; kt < nk; ++kt) {
        const int kk = kt * FK + lc * 4;
        const bool kin = kk < p.K;
        f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        {
            const int m = m0 + lr;
            bool ok = kin && m < p.M;
            const float* src;
            if constexpr (CONV) {
                int kh, kw, cc;
                conv_col(p, kin ? kk : 0, kh, kw, cc);
                src = conv_src<float>(p, A, crow, kh, kw, cc, ok);
            } else src = A + (long)m * p.lda + kk;
            if (ok) va = *(const f32x4*)src;
            const int n = n0 + lr;
            if (kin && n < p.N) vb = *(const f32x4*)(B + (long)n * p.ldb + kk);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[lr * FLD + lc * 4 + e] = va[e];
            Bs[lr * FLD + lc * 4 + e] = vb[e];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            const int k = ks * 4 + (lane >> 4);
            float xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = As[(wm * 32 + i * 16 + (lane & 15)) * FLD + k];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = Bs[(wn * 32 + j * 16 + (lane & 15)) * FLD + k];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    const float g = p.gate ? tanhf(*p.gate) : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nb = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 32 + i * 16 + (lane & 15);
            f32x4 pre;
            epilogue_quad<OutT>(p, acc[i][j], m, nb, g, cbase, pre);
            if (m < p.M) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[r] += pre[r]; s2[r] += pre[r] * pre[r]; }
            }
        }
        if (p.csum) stats_quad(p, s, s2, nb, lane);
    }
}

// ---------------------------------------------------------------- host ----
static bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

extern "C" int avsr_gemm_nt(avsr_stream_t stream, const avsr_gemm_t* g) {
    if (!g || !g->A || !g->B || !g->C) return AVSR_EINVAL;
    if (g->M <= 0 || g->N <= 0 || g->K <= 0 || g->batch <= 0) return AVSR_EINVAL;
    if (g->M > 0x7fffffff || g->N > 0x7fffffff || g->K > 0x7fffffff) return AVSR_EUNSUPPORTED;
    if (g->in_dtype != AVSR_F32 && g->in_dtype != AVSR_BF16) return AVSR_EINVAL;
    if (g->out_dtype != AVSR_F32 && g->out_dtype != AVSR_BF16) return AVSR_EINVAL;
    if ((g->col_sum == nullptr) != (g->col_sumsq == nullptr)) return AVSR_EINVAL;
    const bool bf = g->in_dtype == AVSR_BF16;
    const int chunk = bf ? 8 : 4;              // elements per 16-byte load
    const size_t esz = bf ? 2 : 4;
    if (g->K % chunk || g->ldb % chunk) return AVSR_EUNSUPPORTED;
    if (!aligned_to(g->A, 16) || !aligned_to(g->B, 16)) return AVSR_EUNSUPPORTED;
    if ((g->strideA * esz) % 16 || (g->strideB * esz) % 16) return AVSR_EUNSUPPORTED;
    const bool conv = g->conv.enabled != 0;
    if (conv) {
        const auto& c = g->conv;
        if (c.C % chunk || c.H <= 0 || c.W <= 0 || c.KH <= 0 || c.KW <= 0 || c.stride <= 0) return AVSR_EINVAL;
        if ((int64_t)c.KH * c.KW * c.C != g->K) return AVSR_EINVAL;
        if (c.OH != (c.H + 2 * c.pad_h - c.KH) / c.stride + 1) return AVSR_EINVAL;
        if (c.OW != (c.W + 2 * c.pad_w - c.KW) / c.stride + 1) return AVSR_EINVAL;
        if (g->M % ((int64_t)c.OH * c.OW)) return AVSR_EINVAL;
    } else if (g->lda % chunk) return AVSR_EUNSUPPORTED;

    GemmP p;
    p.A = g->A; p.B = g->B; p.C = g->C;
    p.M = (int)g->M; p.N = (int)g->N; p.K = (int)g->K;
    p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc;
    p.sA = g->strideA; p.sB = g->strideB; p.sC = g->strideC;
    p.alpha = g->alpha; p.scale = g->scale; p.bias = g->bias; p.act = g->act;
    p.res = g->residual; p.ldr = g->ldr; p.res_period = (int)g->res_period; p.gate = g->gate; p.csum = g->col_sum; p.csq = g->col_sumsq;
    p.H = g->conv.H; p.W = g->conv.W; p.Cc = g->conv.C; p.KH = g->conv.KH; p.KW = g->conv.KW;
    p.stride = g->conv.stride; p.pad_h = g->conv.pad_h; p.pad_w = g->conv.pad_w; p.OH = g->conv.OH; p.OW = g->conv.OW;
    const size_t osz = g->out_dtype == AVSR_BF16 ? 2 : 4;
    const size_t vbytes = 4 * osz;
    p.vec_ok = (g->ldc % 4 == 0) && aligned_to(g->C, vbytes) && ((g->strideC * osz) % vbytes == 0) &&
               (!g->residual || ((g->ldr % 4 == 0) && aligned_to(g->residual, vbytes)));
    if (g->residual && g->batch > 1) return AVSR_EUNSUPPORTED;
    const int tmS = bf ? BM : FM, tnS = bf ? BN : FN;
    p.tiles_m = (p.M + tmS - 1) / tmS;
    p.tiles_n = (p.N + tnS - 1) / tnS;
    const long ntiles = (long)p.tiles_m * p.tiles_n;
    if (ntiles > 0x7fffffff || g->batch > 65535) return AVSR_EUNSUPPORTED;
    dim3 grid((unsigned)ntiles, 1, (unsigned)g->batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool of = g->out_dtype == AVSR_F32;
#define LAUNCH(K, CV, OT) hipLaunchKernelGGL((K<CV, OT>), grid, block, 0, s, p)
    if (bf) {
        if (conv) { if (of) LAUNCH(gemm_bf16_kernel, true, float); else LAUNCH(gemm_bf16_kernel, true, bf16_t); }
        else      { if (of) LAUNCH(gemm_bf16_kernel, false, float); else LAUNCH(gemm_bf16_kernel, false, bf16_t); }
    } else {
        if (conv) { if (of) LAUNCH(gemm_f32_kernel, true, float); else LAUNCH(gemm_f32_kernel, true, bf16_t); }
        else      { if (of) LAUNCH(gemm_f32_kernel, false, float); else LAUNCH(gemm_f32_kernel, false, bf16_t); }
    }
#undef LAUNCH
    return avsr_launch_status();
}
