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
// One thread = one output pixel, all 64 channels; weights sit in LDS as
// w[tap][cout] so every lane reads the same address (broadcast).
// scale/shift != NULL: y = relu(conv*scale + shift) (eval-mode BN folded);
// else the raw convolution is stored and per-channel sum / sum-of-squares are
// accumulated for the batch statistics.
template <typename TO>
__global__ __launch_bounds__(256) void stem3d_kernel(const float* __restrict__ video, const float* __restrict__ weight,
                                                     TO* __restrict__ out, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ csum,
                                                     float* __restrict__ csq, int T, int H, int W, int OH, int OW) {
    __shared__ __attribute__((aligned(16))) float ws[ST_TAPS * ST_CO];
    for (int i = threadIdx.x; i < ST_TAPS * ST_CO; i += 256) {
        const int tap = i / ST_CO, co = i - tap * ST_CO;
        ws[i] = weight[co * ST_TAPS + tap];       // torch layout [co][ci][kt][kh][kw] = [co][tap]
    }
    __syncthreads();
    const int frame = blockIdx.y;                  // b*T + t
    const int b = frame / T, t = frame - b * T;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const bool live = pix < OH * OW;
    const int oh = live ? pix / OW : 0, ow = live ? pix - oh * OW : 0;
    float acc[ST_CO];
#pragma unroll
    for (int c = 0; c < ST_CO; ++c) acc[c] = 0.f;
    const long fsz = (long)3 * H * W;
    for (int kt = 0; kt < ST_KT; ++kt) {
        const int tt = t + kt - 2;
        if (tt < 0 || tt >= T) continue;           // uniform over the block
        const float* fp = video + ((long)b * T + tt) * fsz;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < ST_KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < ST_KW; ++kw) {
                    const int ih = oh * 2 - 3 + kh, iw = ow * 2 - 3 + kw;
                    float x = 0.f;
                    if (live && ih >= 0 && ih < H && iw >= 0 && iw < W) x = fp[((long)ci * H + ih) * W + iw];
                    const int tap = ((ci * ST_KT + kt) * ST_KH + kh) * ST_KW + kw;
                    const f32x4* wp = (const f32x4*)&ws[tap * ST_CO];
#pragma unroll
                    for (int c4 = 0; c4 < ST_CO / 4; ++c4) {
                        const f32x4 w = wp[c4];
                        acc[c4 * 4 + 0] = fmaf(x, w[0], acc[c4 * 4 + 0]);
                        acc[c4 * 4 + 1] = fmaf(x, w[1], acc[c4 * 4 + 1]);
                        acc[c4 * 4 + 2] = fmaf(x, w[2], acc[c4 * 4 + 2]);
                        acc[c4 * 4 + 3] = fmaf(x, w[3], acc[c4 * 4 + 3]);
                    }
                }
    }
    if (csum) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int c = 0; c < ST_CO; ++c) {
            const float v = live ? acc[c] : 0.f;
            const float s = wave_sum(v), s2 = wave_sum(v * v);
            if (lane == 0) { atomicAdd(csum + c, s); atomicAdd(csq + c, s2); }
        }
    }
    if (!live) return;
    TO* op = out + ((long)frame * OH * OW + pix) * ST_CO;
#pragma unroll
    for (int c4 = 0; c4 < ST_CO / 4; ++c4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[c4 * 4 + e];
            if (scale) x = fmaxf(x * scale[c4 * 4 + e] + shift[c4 * 4 + e], 0.f);
            v[e] = x;
        }
        if constexpr (sizeof(TO) == 4) *(f32x4*)(op + c4 * 4) = (f32x4){v[0], v[1], v[2], v[3]};
        else *(bf16x4*)(op + c4 * 4) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
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

extern "C" int avsr_stem3d_fwd(avsr_stream_t stream, int out_dtype, const float* video, const float* weight, void* out,
                               const float* scale, const float* shift, float* col_sum, float* col_sumsq, int64_t B,
                               int64_t T, int64_t H, int64_t W) {
    if (!video || !weight || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0) return AVSR_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return AVSR_EINVAL;
    if ((col_sum == nullptr) != (col_sumsq == nullptr)) return AVSR_EINVAL;
    const int OH = (int)((H + 6 - 3) / 2 + 1), OW = (int)((W + 6 - 3) / 2 + 1);
    if (B * T > 65535) return AVSR_EUNSUPPORTED;
    dim3 grid((unsigned)((OH * OW + 255) / 256), (unsigned)(B * T));
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == AVSR_F32)
        hipLaunchKernelGGL(stem3d_kernel<float>, grid, dim3(256), 0, s, video, weight, (float*)out, scale, shift, col_sum,
                           col_sumsq, (int)T, (int)H, (int)W, OH, OW);
    else if (out_dtype == AVSR_BF16)
        hipLaunchKernelGGL(stem3d_kernel<bf16_t>, grid, dim3(256), 0, s, video, weight, (bf16_t*)out, scale, shift, col_sum,
                           col_sumsq, (int)T, (int)H, (int)W, OH, OW);
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
