// Visual-frontend kernels outside the implicit-GEMM trunk: the 3D stem
// (direct convolution, HBM/VALU bound: 135 taps x 64 output channels per
// pixel), the 3x3/2 max-pool with the batch-norm affine + ReLU applied on the
// fly, and the global average pool with the padded-frame mask.
#include "common.h"

#define ST_CO 64
#define ST_KT 5
#define ST_KH 3
#define ST_KW 3
#define ST_TAPS (3 * ST_KT * ST_KH * ST_KW)   // 135

// video: [B, T, 3, H, W] fp32 (the layout the reference's dataloader emits,
// datamodule/data_module.py:142-152).  Output NHWC [B*T, OH, OW, 64].
// Block = 4 waves, 256 output pixels of one frame in 4 rounds of 64; wave w
// computes channels 16w..16w+15 of each round's pixels, so a lane keeps 16
// accumulators and every weight read is a wave-uniform LDS broadcast; the four
// waves re-read the same input taps through L1.  `wpacked` is the weight
// transposed to [tap][cout] (tap = ((ci*5+kt)*3+kh)*3+kw) so staging is a copy.
// scale/shift != NULL: y = relu(conv*scale + shift) (eval-mode BN folded);
// else the raw convolution is stored and per-channel sum / sum-of-squares of this
// block go to row (frame*gridDim.x + blockIdx.x) of the [parts, 64] slabs.
#define ST_CPW 16   // channels per wave
#define ST_ROUNDS 4
template <typename TO>
__global__ __launch_bounds__(256) void stem3d_kernel(const float* __restrict__ video, const float* __restrict__ wpacked,
                                                     TO* __restrict__ out, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ csum,
                                                     float* __restrict__ csq, int T, int H, int W, int OH, int OW) {
    __shared__ __attribute__((aligned(16))) float ws[ST_TAPS * ST_CO];
    for (int i = threadIdx.x; i < ST_TAPS * ST_CO / 4; i += 256) ((f32x4*)ws)[i] = ((const f32x4*)wpacked)[i];
    __syncthreads();
    const int frame = blockIdx.y;                  // b*T + t
    const int b = frame / T, t = frame - b * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = wave * ST_CPW;
    const long fsz = (long)3 * H * W;
    float st1[ST_CPW], st2[ST_CPW];
#pragma unroll
    for (int c = 0; c < ST_CPW; ++c) { st1[c] = 0.f; st2[c] = 0.f; }
    for (int rd = 0; rd < ST_ROUNDS; ++rd) {
        const int pix = (blockIdx.x * ST_ROUNDS + rd) * 64 + lane;
        const bool live = pix < OH * OW;
        const int oh = live ? pix / OW : 0, ow = live ? pix - oh * OW : 0;
        float acc[ST_CPW];
#pragma unroll
        for (int c = 0; c < ST_CPW; ++c) acc[c] = 0.f;
        for (int kt = 0; kt < ST_KT; ++kt) {
            const int tt = t + kt - 2;
            if (tt < 0 || tt >= T) continue;           // uniform over the block
            const float* fp = video + ((long)b * T + tt) * fsz;
#pragma unroll 1
            for (int ci = 0; ci < 3; ++ci) {
#pragma unroll 1
                for (int kh = 0; kh < ST_KH; ++kh) {
                    const int ih = oh * 2 - 3 + kh;
                    const bool rok = live && ih >= 0 && ih < H;
                    const float* rp = fp + ((long)ci * H + (rok ? ih : 0)) * W;
                    const int tap0 = ((ci * ST_KT + kt) * ST_KH + kh) * ST_KW;
#pragma unroll
                    for (int kw = 0; kw < ST_KW; ++kw) {
                        const int iw = ow * 2 - 3 + kw;
                        const float x = (rok && iw >= 0 && iw < W) ? rp[iw] : 0.f;
                        const f32x4* wp = (const f32x4*)&ws[(tap0 + kw) * ST_CO + c0];
#pragma unroll
                        for (int c4 = 0; c4 < ST_CPW / 4; ++c4) {
                            const f32x4 w = wp[c4];
                            acc[c4 * 4 + 0] = fmaf(x, w[0], acc[c4 * 4 + 0]);
                            acc[c4 * 4 + 1] = fmaf(x, w[1], acc[c4 * 4 + 1]);
                            acc[c4 * 4 + 2] = fmaf(x, w[2], acc[c4 * 4 + 2]);
                            acc[c4 * 4 + 3] = fmaf(x, w[3], acc[c4 * 4 + 3]);
                        }
                    }
                }
            }
        }
        if (!live) continue;
#pragma unroll
        for (int c = 0; c < ST_CPW; ++c) { st1[c] += acc[c]; st2[c] += acc[c] * acc[c]; }
        TO* op = out + ((long)frame * OH * OW + pix) * ST_CO + c0;
#pragma unroll
        for (int c4 = 0; c4 < ST_CPW / 4; ++c4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[c4 * 4 + e];
                if (scale) x = fmaxf(x * scale[c0 + c4 * 4 + e] + shift[c0 + c4 * 4 + e], 0.f);
                v[e] = x;
            }
            if constexpr (sizeof(TO) == 4) *(f32x4*)(op + c4 * 4) = (f32x4){v[0], v[1], v[2], v[3]};
            else *(bf16x4*)(op + c4 * 4) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    }
    if (csum) {
        const long part = (long)frame * gridDim.x + blockIdx.x;
#pragma unroll
        for (int c = 0; c < ST_CPW; ++c) {
            const float s = wave_sum(st1[c]), s2 = wave_sum(st2[c]);
            if (lane == 0) { csum[part * ST_CO + c0 + c] = s; csq[part * ST_CO + c0 + c] = s2; }
        }
    }
}

// MaxPool (1,3,3)/(1,2,2)/(0,1,1) over NHWC; optional affine + ReLU applied to
// every element before the max (training-mode BN, whose scale may be negative).
template <typename T, typename TO>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, TO* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int N, int H, int W, int C, int OH, int OW) {
    const int c4n = C >> 2;
    const long total = (long)N * OH * OW * c4n;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % c4n);
        long pp = i / c4n;
        const int ow = (int)(pp % OW); pp /= OW;
        const int oh = (int)(pp % OH);
        const int n = (int)(pp / OH);
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (scale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = scale[c4 * 4 + e]; sh[e] = shift[c4 * 4 + e]; }
        }
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                const T* p = x + (((long)n * H + ih) * W + iw) * C + c4 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = to_f32<T>(p[e]);
                    if (scale) v = fmaxf(v * sc[e] + sh[e], 0.f);
                    m[e] = fmaxf(m[e], v);
                }
            }
        TO* q = y + (((long)n * OH + oh) * OW + ow) * C + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = from_f32<TO>(m[e]);
    }
}

// Global average pool over HW pixels of an NHWC tensor [B*T, HW, C] -> [B*T, C];
// frames t >= len[b] are written as zeros (moco_visual_frontend.py:42,49).
template <typename T, typename TO>
__global__ __launch_bounds__(256) void avgpool_mask_kernel(const T* __restrict__ x, TO* __restrict__ y,
                                                           const int64_t* __restrict__ lens, int Tn, int HW, int C) {
    const int frame = blockIdx.x;
    const int b = frame / Tn, t = frame - b * Tn;
    const bool dead = lens && t >= lens[b];
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        if (!dead)
            for (int p = 0; p < HW; ++p) s += to_f32<T>(x[((long)frame * HW + p) * C + c]);
        y[(long)frame * C + c] = from_f32<TO>(dead ? 0.f : s * inv);
    }
}

// number of rows of the [parts, 64] statistics slabs avsr_stem3d_fwd fills
extern "C" int64_t avsr_stem3d_stats_parts(int64_t B, int64_t T, int64_t H, int64_t W) {
    const int64_t OH = (H + 6 - 3) / 2 + 1, OW = (W + 6 - 3) / 2 + 1;
    return B * T * ((OH * OW + 64 * ST_ROUNDS - 1) / (64 * ST_ROUNDS));
}

extern "C" int avsr_stem3d_fwd(avsr_stream_t stream, int out_dtype, const float* video, const float* weight_packed,
                               void* out, const float* scale, const float* shift, float* col_sum, float* col_sumsq,
                               int64_t B, int64_t T, int64_t H, int64_t W) {
    if (!video || !weight_packed || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0) return AVSR_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return AVSR_EINVAL;
    if ((col_sum == nullptr) != (col_sumsq == nullptr)) return AVSR_EINVAL;
    if ((uintptr_t)weight_packed % 16) return AVSR_EUNSUPPORTED;
    const int OH = (int)((H + 6 - 3) / 2 + 1), OW = (int)((W + 6 - 3) / 2 + 1);
    if (B * T > 65535) return AVSR_EUNSUPPORTED;
    dim3 grid((unsigned)((OH * OW + 64 * ST_ROUNDS - 1) / (64 * ST_ROUNDS)), (unsigned)(B * T));
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == AVSR_F32)
        hipLaunchKernelGGL(stem3d_kernel<float>, grid, dim3(256), 0, s, video, weight_packed, (float*)out, scale, shift,
                           col_sum, col_sumsq, (int)T, (int)H, (int)W, OH, OW);
    else if (out_dtype == AVSR_BF16)
        hipLaunchKernelGGL(stem3d_kernel<bf16_t>, grid, dim3(256), 0, s, video, weight_packed, (bf16_t*)out, scale, shift,
                           col_sum, col_sumsq, (int)T, (int)H, (int)W, OH, OW);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_maxpool3x3s2(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y,
                                 const float* scale, const float* shift, int64_t N, int64_t H, int64_t W, int64_t C) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return AVSR_EINVAL;
    if (C % 4) return AVSR_EUNSUPPORTED;
    if ((scale == nullptr) != (shift == nullptr)) return AVSR_EINVAL;
    const int OH = (int)((H + 2 - 3) / 2 + 1), OW = (int)((W + 2 - 3) / 2 + 1);
    const long total = N * OH * OW * (C / 4);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipStream_t s = (hipStream_t)stream;
#define L(TI, TO) hipLaunchKernelGGL((maxpool3x3s2_kernel<TI, TO>), dim3((unsigned)nb), dim3(256), 0, s, (const TI*)x, (TO*)y, \
                                     scale, shift, (int)N, (int)H, (int)W, (int)C, OH, OW)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

extern "C" int avsr_avgpool_mask_fwd(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y,
                                     const int64_t* lens, int64_t B, int64_t T, int64_t HW, int64_t C) {
    if (!x || !y || B <= 0 || T <= 0 || HW <= 0 || C <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * T)), block(256);
#define L(TI, TO) hipLaunchKernelGGL((avgpool_mask_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, (TO*)y, lens, (int)T, \
                                     (int)HW, (int)C)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_F32) L(bf16_t, float);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}
