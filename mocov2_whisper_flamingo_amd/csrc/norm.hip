// Row-wise LayerNorm (forward / backward) with wave64 shuffle reductions, and
// the batch-norm statistics / apply kernels of the visual trunk.
#include "common.h"

// One wave per row.  Each lane owns columns lane*4 + 256*i (float4 / bf16x4
// chunks), so a wave reads a row as contiguous 1 KiB (fp32) segments and the
// same lane sees the same columns in every row (used by the backward for the
// dgamma / dbeta partial sums).
#define LN_MAXV 8   // up to 8 chunks of 4 per lane -> D <= 2048

template <typename T>
__device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    const f32x4 x = *(const f32x4*)p; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
    const bf16x4 x = *(const bf16x4*)p; v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
    *(bf16x4*)p = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

// One wave per PAIR of rows: both rows' loads are issued before the first reduction, so a wave has twice the bytes
// in flight while it waits (the kernel is a stream of 3 KB rows: bandwidth comes from loads in flight).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ add, int add_period,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const int nr = min(2, rows - row0);
    float v[2][LN_MAXV][4];
    const int nch = D >> 2;   // chunks of 4
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const TI* xr = x + (long)(row0 + (r < nr ? r : 0)) * ldx;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) { load4<TI>(xr + c * 4, v[r][i]); s[r] += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r >= nr) break;
        const int row = row0 + r;
        const float mean = wave_sum(s[r]) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
        TO* yr = y + (long)row * ldy;
        const float* ar = add ? add + (long)(row % add_period) * D : nullptr;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float g[4], b[4], o[4];
                load4<float>(gamma + c * 4, g);
                load4<float>(beta + c * 4, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mean) * rstd * g[e] + b[e];
                if (ar) { float a[4]; load4<float>(ar + c * 4, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += a[e]; }
                store4<TO>(yr + c * 4, o);
            }
        }
    }
}

// Tall operands (the encoder's residual stream: thousands of 3-KB rows).  NV = chunks of 4 columns per lane, exact
// for the width (D <= 256*NV) instead of the general kernel's eight guarded ones, gamma and beta loaded once into
// registers: 14.0 -> 10.0 us on [12000, 768] fp32 -> bf16 (5.5 TB/s).  A wave walks row pairs with the stride of
// the grid and issues the loads of its next pair before it reduces and writes the current one; measured, the grid
// that gives every wave ONE pair (2048 workgroups at 12000 rows) is as fast as any streaming grid (512: 11.7 us,
// 1024: 10.3 us), so the loop only matters above 16384 rows.
#ifndef LN_STREAM
#define LN_STREAM 1
#endif
#ifndef LN_STREAM_BLOCKS
#define LN_STREAM_BLOCKS 2048
#endif
template <typename TI, typename TO, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_stream_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                   int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int npairs = (rows + 1) >> 1, nch = D >> 2;
    float g[NV][4], b[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) { load4<float>(gamma + c * 4, g[i]); load4<float>(beta + c * 4, b[i]); }
    }
    float cur[2][NV][4], nxt[2][NV][4];
    auto load_pair = [&](int p, float (&v)[2][NV][4]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = min(2 * p + r, rows - 1);
            const TI* xr = x + (long)row * ldx;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) load4<TI>(xr + c * 4, v[r][i]);
            }
        }
    };
    if (wid < npairs) load_pair(wid, cur);
    for (int p = wid; p < npairs; p += nw) {
        if (p + nw < npairs) load_pair(p + nw, nxt);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 2 * p + r;
            if (row >= rows) break;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < nch) s += cur[r][i][0] + cur[r][i][1] + cur[r][i][2] + cur[r][i][3];
            const float mean = wave_sum(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < nch) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = cur[r][i][e] - mean; q += d * d; }
                }
            const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
            if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
            TO* yr = y + (long)row * ldy;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (cur[r][i][e] - mean) * rstd * g[i][e] + b[i][e];
                    store4<TO>(yr + c * 4, o);
                }
            }
        }
        if (p + nw < npairs) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cur[r][i][e] = nxt[r][i][e];
        }
    }
}

// Backward: dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); dgamma +=
// dy*xhat, dbeta += dy.  Each wave walks rows row0, row0+stride, ... and keeps its
// dgamma/dbeta partials in registers; one atomic per column per wave at the end.
template <typename TX, typename TG>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TG* __restrict__ dy, long lddy, const TX* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TG* __restrict__ dx, long lddx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int D, const TG* __restrict__ dres, long lddr) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const int nch = D >> 2;
    float dg[LN_MAXV][4], db[LN_MAXV][4], g[LN_MAXV][4];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; g[i][e] = 0.f; }
        if (c < nch) load4<float>(gamma + c * 4, g[i]);
    }
    for (int row = wid; row < rows; row += nw) {
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXV][4], gy[LN_MAXV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float xv[4], dv[4];
                load4<TX>(x + (long)row * ldx + c * 4, xv);
                load4<TG>(dy + (long)row * lddy + c * 4, dv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    gy[i][e] = dv[e] * g[i][e];
                    s1 += gy[i][e];
                    s2 += gy[i][e] * xh[i][e];
                    dg[i][e] += dv[e] * xh[i][e];
                    db[i][e] += dv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2);
                if (dres) {                     // the gradient that reaches x past the norm (a residual connection)
                    float rv[4];
                    load4<TG>(dres + (long)row * lddr + c * 4, rv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += rv[e];
                }
                store4<TG>(dx + (long)row * lddx + c * 4, o);
            }
        }
    }
    // fold the block's 4 waves through LDS (lane l of every wave owns the same columns), then one
    // atomic per column per block
    __shared__ float red[2][3][64 * 4 * LN_MAXV];
    const int w = threadIdx.x >> 6;
    if (w > 0) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[0][w - 1][c * 4 + e] = dg[i][e]; red[1][w - 1][c * 4 + e] = db[i][e]; }
            }
        }
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = c * 4 + e;
                    atomicAdd(dgamma + k, dg[i][e] + red[0][0][k] + red[0][1][k] + red[0][2][k]);
                    atomicAdd(dbeta + k, db[i][e] + red[1][0][k] + red[1][1][k] + red[1][2][k]);
                }
            }
        }
    }
}

extern "C" int avsr_layernorm_fwd(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx,
                                  void* y, int64_t ldy, const float* gamma, const float* beta, const float* add,
                                  int64_t add_period, float* mean, float* rstd, int64_t rows, int64_t D, float eps) {
    if (!x || !y || !gamma || !beta || rows <= 0) return AVSR_EINVAL;
    if (D % 4 || D > 64 * 4 * LN_MAXV || ldx % 4 || ldy % 4) return AVSR_EUNSUPPORTED;
    if (add && add_period <= 0) return AVSR_EINVAL;
    dim3 grid((unsigned)((rows + 7) / 8)), block(256);
    hipStream_t s = (hipStream_t)stream;
#if LN_STREAM
    if (!add && rows >= 8192 && in_dtype == AVSR_F32 && out_dtype == AVSR_BF16 && D <= 1280) {
        const dim3 sg((unsigned)std::min<int64_t>(LN_STREAM_BLOCKS, (rows + 7) / 8));
#define LS(NV) hipLaunchKernelGGL((layernorm_fwd_stream_kernel<float, bf16_t, NV>), sg, block, 0, s, (const float*)x, (long)ldx, \
                                  (bf16_t*)y, (long)ldy, gamma, beta, mean, rstd, (int)rows, (int)D, eps)
        if (D <= 512) LS(2);
        else if (D <= 768) LS(3);
        else if (D <= 1024) LS(4);
        else LS(5);
#undef LS
        return avsr_launch_status();
    }
#endif
#define L(TI, TO) hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, (long)ldx, (TO*)y, \
                                     (long)ldy, gamma, beta, add, (int)add_period, mean, rstd, (int)rows, (int)D, eps)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_F32) L(bf16_t, float);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

#ifndef LN_BWD_CAP_SHORT
#define LN_BWD_CAP_SHORT 64
#endif
#ifndef LN_BWD_CAP_TALL
#define LN_BWD_CAP_TALL 512
#endif
extern "C" int avsr_layernorm_bwd(avsr_stream_t stream, int x_dtype, int g_dtype, const void* dy, int64_t lddy,
                                  const void* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int64_t D) {
    return avsr_layernorm_bwd_res(stream, x_dtype, g_dtype, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dgamma, dbeta,
                                  rows, D, nullptr, 0);
}

extern "C" int avsr_layernorm_bwd_res(avsr_stream_t stream, int x_dtype, int g_dtype, const void* dy, int64_t lddy,
                                      const void* x, int64_t ldx, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, int64_t lddx, float* dgamma, float* dbeta,
                                      int64_t rows, int64_t D, const void* dres, int64_t lddr) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0) return AVSR_EINVAL;
    if (D % 4 || D > 64 * 4 * LN_MAXV || ldx % 4 || lddy % 4 || lddx % 4 || (dres && lddr % 4)) return AVSR_EUNSUPPORTED;
    long nb = (rows + 7) / 8;        // ~2 rows per wave; one atomic per column per block at the end
    // every workgroup ends in one atomic per column on the SAME 2 D words, and those serialise at the memory side
    // (25 600 rows x 512: 124 us with 1024 workgroups; 3200 rows: 41 us with 256, 22 us with 64; 600 rows: 13.5 / 11.8 us)
    const long cap = rows > 8192 ? LN_BWD_CAP_TALL : LN_BWD_CAP_SHORT;
    if (nb > cap) nb = cap;
    dim3 grid((unsigned)nb), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(TX, TG) hipLaunchKernelGGL((layernorm_bwd_kernel<TX, TG>), grid, block, 0, s, (const TG*)dy, (long)lddy, (const TX*)x, \
                                     (long)ldx, gamma, mean, rstd, (TG*)dx, (long)lddx, dgamma, dbeta, (int)rows, (int)D, \
                                     (const TG*)dres, (long)lddr)
    if (x_dtype == AVSR_F32 && g_dtype == AVSR_F32) L(float, float);
    else if (x_dtype == AVSR_F32 && g_dtype == AVSR_BF16) L(float, bf16_t);
    else if (x_dtype == AVSR_BF16 && g_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (x_dtype == AVSR_BF16 && g_dtype == AVSR_F32) L(bf16_t, float);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

// ------------------------------------------------------------ batch norm ---
// Per-channel statistics arrive as [nparts, C] slabs of partial sums (one row per
// producing workgroup: plain stores, fixed summation order, no atomics).
#define BN_RED_PARTS 32

// Fold nparts rows into BN_RED_PARTS rows: block (x = 64-channel group, y = output row s)
// sums rows s, s+32, s+64, ... with its 4 waves and combines them through LDS.
__global__ __launch_bounds__(256) void bn_reduce_parts_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                              int nparts, int C, float* __restrict__ ws) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, s = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int p = s + BN_RED_PARTS * w; p < nparts; p += BN_RED_PARTS * 4) {
            a += sum[(long)p * C + c];
            b += sumsq[(long)p * C + c];
        }
    red[0][w][lane] = a; red[1][w][lane] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        ws[((long)s * 2 + 0) * C + c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        ws[((long)s * 2 + 1) * C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    }
}

// Turn the partial sums into the affine (scale, shift) the next kernel applies, and update
// the running statistics the way nn.BatchNorm does in training mode (momentum update with
// the UNBIASED variance).  sum row r / sumsq row r live at sum + r*row_stride.
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, int nparts,
                                   long row_stride, float count, const float* __restrict__ weight,
                                   const float* __restrict__ bias, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ scale, float* __restrict__ shift,
                                   int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, q = 0.f;
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) { s += sum[(long)p * row_stride + c]; q += sumsq[(long)p * row_stride + c]; }
    const float mean = s / count;
    float var = q / count - mean * mean;
    var = fmaxf(var, 0.f);
    const float sc = weight[c] * rsqrtf(var + eps);
    scale[c] = sc;
    shift[c] = bias[c] - mean * sc;
    if (running_mean) {
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

// Both steps in one launch for up to a few thousand slab rows.  grid = (ceil(C/64), S): workgroup (g, s) sums
// slice s of the slab rows for channels 64g..64g+63 - its NW waves take interleaved rows (16 loads in flight per
// lane) and fold through LDS in a fixed order.  S = 1: it finalises directly.  S > 1 (thousands of rows: one
// workgroup per 64 channels would read megabytes alone): it publishes its partial sums in `ws`, takes a ticket,
// and the workgroup that draws the last ticket adds the S partials in slice order (so the result does not depend
// on which workgroup that is), finalises and puts the ticket counter back to zero.
#define BN_FUSED_MAX_PARTS 8192
#define BN_SPLIT_MAX 16
template <int NW>
__global__ __launch_bounds__(64 * NW) void bn_reduce_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                                  int nparts, float count, const float* __restrict__ weight,
                                                                  const float* __restrict__ bias, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float* __restrict__ scale,
                                                                  float* __restrict__ shift, int C, float eps, float momentum,
                                                                  float* __restrict__ ws, int* __restrict__ tickets) {
    __shared__ float red[2][NW][64];
    __shared__ int last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, C - 1);
    const int S = gridDim.y, slice = blockIdx.y;
    const int per = (nparts + S - 1) / S, p_end = min(nparts, (slice + 1) * per);
    // the channel's parameters do not depend on the sums: request them now, not behind the reduction
    float pw = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    if (w == 0) { pw = weight[cc]; pb = bias[cc]; if (running_mean) { prm = running_mean[cc]; prv = running_var[cc]; } }
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = 0.f; b[u] = 0.f; }
    int p = slice * per + w;
    for (; p + NW * 7 < p_end; p += NW * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] += sum[(long)(p + NW * u) * C + cc];
            b[u] += sumsq[(long)(p + NW * u) * C + cc];
        }
    }
    for (; p < p_end; p += NW) { a[0] += sum[(long)p * C + cc]; b[0] += sumsq[(long)p * C + cc]; }
    red[0][w][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    red[1][w][lane] = ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
    __syncthreads();
    float s = 0.f, q = 0.f;
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { s += red[0][i][lane]; q += red[1][i][lane]; }
    }
    if (S > 1) {
        if (w == 0 && c < C) {
            __hip_atomic_store(&ws[((long)slice * 2 + 0) * C + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws[((long)slice * 2 + 1) * C + c], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // The partial sums are device-scope atomic stores (written through to where every XCD sees them) and are
        // waited for before the ticket is drawn; the last workgroup reads them with device-scope atomic loads.  A
        // release / acquire fence pair would order the same thing, but at device scope it writes back and
        // invalidates this XCD's whole L2 - with the convolution outputs of the layer sitting dirty in it.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            last = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
        __syncthreads();
        if (!last) return;
        if (threadIdx.x == 0) __hip_atomic_store(&tickets[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0 && c < C) {
            s = 0.f; q = 0.f;
            for (int i = 0; i < S; ++i) {
                s += __hip_atomic_load(&ws[((long)i * 2 + 0) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q += __hip_atomic_load(&ws[((long)i * 2 + 1) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (w != 0 || c >= C) return;
    const float mean = s / count;
    const float var = fmaxf(q / count - mean * mean, 0.f);
    const float sc = pw * rsqrtf(var + eps);
    scale[c] = sc;
    shift[c] = pb - mean * sc;
    if (running_mean) {
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * prm + momentum * mean;
        running_var[c] = (1.f - momentum) * prv + momentum * unb;
    }
}

// Eval-mode fold: scale = w / sqrt(running_var + eps), shift = b - running_mean*scale.
__global__ void bn_fold_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                               const float* __restrict__ running_mean, const float* __restrict__ running_var,
                               float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = weight[c] * rsqrtf(running_var[c] + eps);
    scale[c] = sc;
    shift[c] = bias[c] - running_mean[c] * sc;
}

// y = act(x*scale[c] + shift[c] (+ residual)) over an NHWC tensor [rows, C];
// 8 (bf16) or 4 (f32) channels per thread, grid-stride.
template <typename TI, typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TI* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                       long n4, int C, int relu) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % C);
        float v[4], sc[4], sh[4];
        load4<TI>(x + i * 4, v);
        load4<float>(scale + c, sc);
        load4<float>(shift + c, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) { float r[4]; load4<T>(res + i * 4, r);
            if (rscale) { float a[4], b[4]; load4<float>(rscale + c, a); load4<float>(rshift + c, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = r[e] * a[e] + b[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e]; }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f); }
        store4<T>(y + i * 4, v);
    }
}

#ifndef BN_APPLY_BLOCKS
#define BN_APPLY_BLOCKS 4096
#endif
// bf16 -> bf16 form for channel counts that are a multiple of 8 with 16-byte-aligned operands: 8 channels per
// thread as one 16-byte load / store.  A thread's channel group is fixed when the grid stride is a multiple of
// C/8 (the host rounds it), so scale and shift live in registers for the whole kernel and the loop is load,
// 8 fma, (add, max), store: two independent 16-byte loads in flight per thread and iteration.
// RA: the residual is itself a raw convolution output with an affine of its own (the downsample branch's BN,
// which has no activation): y = act(x*scale + shift + res*rscale + rshift).
template <bool RA>
__global__ __launch_bounds__(256) void bn_apply_bf16x8_kernel(const bf16_t* x, const bf16_t* res,   // y may be x
                                                              bf16_t* y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ rscale,
                                                              const float* __restrict__ rshift, long n8, int C8, int relu) {
    const long stride = (long)gridDim.x * blockDim.x;          // multiple of C8
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i % C8) * 8;
    float sc[8], sh[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[c + e]; sh[e] = shift[c + e]; rs[e] = 1.f; }
    if (RA) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { rs[e] = rscale[c + e]; sh[e] += rshift[c + e]; }   // both shifts add up front
    }
    for (; i < n8; i += 2 * stride) {
        const long j = i + stride;
        const bool two = j < n8;
        const bf16x8 a0 = *(const bf16x8*)(x + i * 8);
        const bf16x8 a1 = two ? *(const bf16x8*)(x + j * 8) : a0;
        bf16x8 r0 = a0, r1 = a0;
        if (res) { r0 = *(const bf16x8*)(res + i * 8); if (two) r1 = *(const bf16x8*)(res + j * 8); }
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v0 = fmaf((float)a0[e], sc[e], sh[e]), v1 = fmaf((float)a1[e], sc[e], sh[e]);
            if (RA) { v0 = fmaf((float)r0[e], rs[e], v0); v1 = fmaf((float)r1[e], rs[e], v1); }
            else if (res) { v0 += (float)r0[e]; v1 += (float)r1[e]; }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            o0[e] = (bf16_t)v0; o1[e] = (bf16_t)v1;
        }
        *(bf16x8*)(y + i * 8) = o0;
        if (two) *(bf16x8*)(y + j * 8) = o1;
    }
}

// Training-mode BatchNorm of a convolution output with FEW slab rows (the deep, small feature maps: a few hundred row
// tiles), finalize and apply in one launch: workgroup (g, chunk) folds the slab rows of channels 64g .. 64g + 63 itself
// (its waves take interleaved rows; every workgroup of a channel group adds the same numbers in the same order, so they
// all hold the same affine) and then normalises its chunk of rows of that 64-channel slice - 128 contiguous bytes per
// row.  Chunk 0 of each channel group also updates the running statistics and publishes scale / shift.  A launch
// (and the dependent-launch gap before it) less per BatchNorm; the slabs are read once per workgroup, which is why
// this form is for short slabs only.
#define BFA_MAX_PARTS 256
#ifndef BFA_TARGET_WGS
#define BFA_TARGET_WGS 256
#endif
// Waves per workgroup: 16 is fastest when the visual frontend runs alone (3.63 -> 3.56 ms per pass), but beside the audio
// encoder's GEMMs a 1024-thread workgroup waits for a CU with sixteen free wave slots: in the whole step 16 waves lose
// 0.02 ms against the two separate launches, 8 (or 4) waves gain 0.08 ms.
#define BFA_WAVES 8
template <int NW>
__global__ __launch_bounds__(64 * NW) void bn_finalize_apply_kernel(const bf16_t* x, const bf16_t* res, bf16_t* y,   // y may be x
                                                                 const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                                 int nparts, float count, const float* __restrict__ weight,
                                                                 const float* __restrict__ bias, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, float* __restrict__ scale_out,
                                                                 float* __restrict__ shift_out, long rows, int C, float eps,
                                                                 float momentum, int relu, long rows_per_chunk) {
    __shared__ float red[2][NW][64];
    constexpr int RPI = 8 * NW;                                 // rows per pass of the workgroup (8 threads per row)
    __shared__ float aff[2][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int c = blockIdx.x * 64 + lane;
    float pw = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    const bool first = blockIdx.y == 0;
    if (w == 0) { pw = weight[c]; pb = bias[c]; if (first && running_mean) { prm = running_mean[c]; prv = running_var[c]; } }
    // the first rows of the chunk do not depend on the statistics: request them ahead of the reduction
    const int q8 = t & 7, r = t >> 3;
    const long r_begin = (long)blockIdx.y * rows_per_chunk, r_end = min(rows, r_begin + rows_per_chunk);
    const long col = (long)blockIdx.x * 64 + q8 * 8;
    long row = r_begin + r;
    bf16x8 a0 = {}, a1 = {}, r0 = {}, r1 = {};
    if (row < r_end) { a0 = *(const bf16x8*)(x + row * C + col); if (res) r0 = *(const bf16x8*)(res + row * C + col); }
    if (row + RPI < r_end) { a1 = *(const bf16x8*)(x + (row + RPI) * C + col); if (res) r1 = *(const bf16x8*)(res + (row + RPI) * C + col); }
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    int p = w;
    for (; p + NW * 3 < nparts; p += NW * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { sa[u] += sum[(long)(p + NW * u) * C + c]; sb[u] += sumsq[(long)(p + NW * u) * C + c]; }
    }
    for (; p < nparts; p += NW) { sa[0] += sum[(long)p * C + c]; sb[0] += sumsq[(long)p * C + c]; }
    red[0][w][lane] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    red[1][w][lane] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    __syncthreads();
    if (w == 0) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) { s += red[0][i][lane]; q += red[1][i][lane]; }
        const float mean = s / count;
        const float var = fmaxf(q / count - mean * mean, 0.f);
        const float sc = pw * rsqrtf(var + eps), sh = pb - mean * sc;
        aff[0][lane] = sc; aff[1][lane] = sh;
        if (first) {
            if (scale_out) { scale_out[c] = sc; shift_out[c] = sh; }
            if (running_mean) {
                const float unb = count > 1.f ? var * count / (count - 1.f) : var;
                running_mean[c] = (1.f - momentum) * prm + momentum * mean;
                running_var[c] = (1.f - momentum) * prv + momentum * unb;
            }
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = aff[0][q8 * 8 + e]; sh[e] = aff[1][q8 * 8 + e]; }
    for (; row < r_end; row += 2 * RPI) {
        const bool two = row + RPI < r_end;
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v0 = fmaf((float)a0[e], sc[e], sh[e]), v1 = fmaf((float)a1[e], sc[e], sh[e]);
            if (res) { v0 += (float)r0[e]; v1 += (float)r1[e]; }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            o0[e] = (bf16_t)v0; o1[e] = (bf16_t)v1;
        }
        // the next pair of rows is requested before this pair is stored
        const long nrow = row + 2 * RPI;
        bf16x8 n0 = {}, n1 = {}, m0 = {}, m1 = {};
        if (nrow < r_end) { n0 = *(const bf16x8*)(x + nrow * C + col); if (res) m0 = *(const bf16x8*)(res + nrow * C + col); }
        if (nrow + RPI < r_end) { n1 = *(const bf16x8*)(x + (nrow + RPI) * C + col); if (res) m1 = *(const bf16x8*)(res + (nrow + RPI) * C + col); }
        *(bf16x8*)(y + row * C + col) = o0;
        if (two) *(bf16x8*)(y + (row + RPI) * C + col) = o1;
        a0 = n0; a1 = n1; r0 = m0; r1 = m1;
    }
}

extern "C" int avsr_bn_finalize_apply(avsr_stream_t stream, const void* x, const void* residual, void* y, const float* sum,
                                      const float* sumsq, int64_t nparts, double count, const float* weight,
                                      const float* bias, float* running_mean, float* running_var, float* scale_out,
                                      float* shift_out, int64_t rows, int64_t C, float eps, float momentum, int relu) {
    if (!x || !y || !sum || !sumsq || !weight || !bias || rows <= 0 || C <= 0 || nparts <= 0 || count <= 0) return AVSR_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr) || (scale_out == nullptr) != (shift_out == nullptr)) return AVSR_EINVAL;
    if (C % 64 || nparts > BFA_MAX_PARTS || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) % 16)) return AVSR_EUNSUPPORTED;
    const long groups = C / 64;
    long chunks = std::max<long>(1, BFA_TARGET_WGS / groups);
    chunks = std::min<long>(chunks, (rows + 255) / 256);
    const long rows_per_chunk = (rows + chunks - 1) / chunks;
    chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    if (chunks > 65535) return AVSR_EUNSUPPORTED;
#define L(NWV) hipLaunchKernelGGL(bn_finalize_apply_kernel<NWV>, dim3((unsigned)groups, (unsigned)chunks), dim3(64 * NWV), 0, \
                       (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, sum, sumsq, (int)nparts, \
                       (float)count, weight, bias, running_mean, running_var, scale_out, shift_out, (long)rows, (int)C, eps, \
                       momentum, relu, rows_per_chunk)
    L(BFA_WAVES);
#undef L
    return avsr_launch_status();
}

// ---- batch statistics of a 1x1 convolution's OUTPUT from the second moments of its INPUT -----------------------
// y = W h per pixel, so mean(y_c) = w_c . mean(h) and mean(y_c^2) = w_c^T mean(h h^T) w_c: the K x K Gram matrix of the
// input (K = 64 / 128 channels) gives the statistics of all 4K output channels without the output existing yet, and the
// convolution can then apply the normalisation, the residual and the ReLU in its own epilogue instead of storing the
// raw product for a second pass.  Step 1 (avsr_gram_stats, gemm_tn.hip) leaves per-share partial Gram matrices and
// column sums; gram_reduce adds the shares in share order (16 waves take interleaved shares, folded in wave order:
// deterministic) into doubles scaled by 1 / count; bn_gram_finalize evaluates the two forms per output channel.
__global__ __launch_bounds__(1024) void gram_reduce_kernel(const float* __restrict__ gram_parts, const float* __restrict__ colsum_parts,
                                                           int parts, int KK, int K, double inv_n, float* __restrict__ out) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool live = e < KK + K;
    const float* src = e < KK ? gram_parts + e : colsum_parts + (live ? e - KK : 0);
    const long stride = e < KK ? KK : K;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live) {
        int p = w;
        for (; p + 48 < parts; p += 64) {
            const float v0 = src[(long)p * stride], v1 = src[(long)(p + 16) * stride];
            const float v2 = src[(long)(p + 32) * stride], v3 = src[(long)(p + 48) * stride];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; p < parts; p += 16) a0 += src[(long)p * stride];
    }
    red[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && live) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][lane];
        out[e] = (float)(s * inv_n);
    }
}

// 8 output channels per workgroup (2 per wave, one after the other); the K x K mean-of-products matrix (fp32: the forms
// below lose about 50 x 2^-24 of relative accuracy to cancellation, far inside what batch normalisation notices) is
// staged in LDS with 16-byte loads and read column-wise (it is symmetric), so the lanes of a wave read consecutive words.
#define GRAM_FIN_CH 8
template <int K>
__global__ __launch_bounds__(256) void bn_gram_finalize_kernel(const float* __restrict__ gm, const bf16_t* __restrict__ W, long ldw,
                                                               int N, float count, const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ scale,
                                                               float* __restrict__ shift, float eps, float momentum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* G = (float*)smem;                        // [K][K]
    float* mu = G + K * K;                          // [K]
    float* wv = mu + K;                             // [4 waves][K]
    constexpr int R = K / 64;                       // rows (= columns) of the matrix per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {                                               // mu follows G in both layouts; (K*K + K) / 4 float4, several in flight
        constexpr int NV = (K * K + K) / 4;
        const f32x4* src = (const f32x4*)gm;
        f32x4* dst = (f32x4*)G;
#pragma unroll 8
        for (int i = threadIdx.x; i < NV; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (int cc = 0; cc < GRAM_FIN_CH / 4; ++cc) {
        const int c = blockIdx.x * GRAM_FIN_CH + wave * (GRAM_FIN_CH / 4) + cc;
        if (c >= N) break;                          // wave-uniform
        float wl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { wl[r] = (float)W[(long)c * ldw + lane + 64 * r]; wv[wave * K + lane + 64 * r] = wl[r]; }
        __builtin_amdgcn_wave_barrier();
        double t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = 0.0;
        for (int j = 0; j < K; ++j) {
            const float wj = wv[wave * K + j];
#pragma unroll
            for (int r = 0; r < R; ++r) t[r] += (double)(G[j * K + lane + 64 * r] * wj);
        }
        double q = 0.0, m = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) { q += (double)wl[r] * t[r]; m += (double)wl[r] * (double)mu[lane + 64 * r]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { q += __shfl_xor(q, o, 64); m += __shfl_xor(m, o, 64); }
        __builtin_amdgcn_wave_barrier();            // every lane has read wv before the next channel overwrites it
        if (lane == 0) {
            const float mean = (float)m;
            const float var = fmaxf((float)(q - m * m), 0.f);
            const float sc = weight[c] * rsqrtf(var + eps);
            scale[c] = sc;
            shift[c] = bias[c] - mean * sc;
            if (running_mean) {
                const float unb = count > 1.f ? var * count / (count - 1.f) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
            }
        }
    }
}

extern "C" int64_t avsr_bn_gram_workspace_doubles(int64_t K) { return (K * K + K + 1) / 2; }    // holds K*K + K floats

extern "C" int avsr_bn_gram_finalize(avsr_stream_t stream, const float* gram_parts, const float* colsum_parts, int64_t parts,
                                     int64_t K, double count, const void* W_bf16, int64_t ldw, int64_t N, const float* weight,
                                     const float* bias, float* running_mean, float* running_var, float* scale, float* shift,
                                     float eps, float momentum, double* workspace) {
    // gram_parts == NULL (with colsum_parts == NULL): the workspace already holds the reduced statistics of this input
    // from an earlier call (several convolutions reading one tensor: one reduction serves them all)
    if ((gram_parts == nullptr) != (colsum_parts == nullptr)) return AVSR_EINVAL;
    if (!W_bf16 || !weight || !bias || !scale || !shift || !workspace) return AVSR_EINVAL;
    if ((gram_parts && parts <= 0) || N <= 0 || count <= 0 || ldw < K) return AVSR_EINVAL;
    if (K != 64 && K != 128) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int KK = (int)(K * K);
    if (gram_parts)
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((KK + K + 63) / 64)), dim3(1024), 0, s, gram_parts, colsum_parts,
                           (int)parts, KK, (int)K, 1.0 / count, (float*)workspace);
    const size_t lds = (size_t)(KK + K + 4 * K) * sizeof(float);
    const dim3 grid((unsigned)((N + GRAM_FIN_CH - 1) / GRAM_FIN_CH));
    const float* gm = (const float*)workspace;
    if (K == 64) {
        hipLaunchKernelGGL(bn_gram_finalize_kernel<64>, grid, dim3(256), lds, s, gm, (const bf16_t*)W_bf16, (long)ldw, (int)N,
                           (float)count, weight, bias, running_mean, running_var, scale, shift, eps, momentum);
    } else {
        static avsr_dev_flags attr_set;
        if (!avsr_allow_lds((const void*)bn_gram_finalize_kernel<128>, (int)lds, attr_set)) return AVSR_EHIP;
        hipLaunchKernelGGL(bn_gram_finalize_kernel<128>, grid, dim3(256), lds, s, gm, (const bf16_t*)W_bf16, (long)ldw, (int)N,
                           (float)count, weight, bias, running_mean, running_var, scale, shift, eps, momentum);
    }
    return avsr_launch_status();
}

extern "C" int64_t avsr_bn_finalize_workspace_floats(int64_t C) { return (int64_t)BN_RED_PARTS * 2 * C; }

extern "C" int avsr_bn_finalize(avsr_stream_t stream, const float* sum, const float* sumsq, int64_t nparts, double count,
                                const float* weight, const float* bias, float* running_mean, float* running_var,
                                float* scale, float* shift, int64_t C, float eps, float momentum, float* ws,
                                int32_t* tickets) {
    if (!sum || !sumsq || !weight || !bias || !scale || !shift || C <= 0 || count <= 0 || nparts <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 fgrid((unsigned)((C + 255) / 256)), fblock(256);
    if (nparts > BN_RED_PARTS && nparts <= BN_FUSED_MAX_PARTS) {
        // thousands of rows: split them over up to 16 workgroups per 64 channels when the caller lends the buffers
        int S = 1;
        if (ws && tickets && nparts >= 512) S = (int)std::min<int64_t>(BN_SPLIT_MAX, nparts / 256);
        // 8 waves per workgroup: sixteen fold a slab a little faster alone, but a 1024-thread workgroup waits for a CU
        // with sixteen free wave slots while the other streams' GEMMs hold the chip (0.04 ms of the step)
#define L(NWV) hipLaunchKernelGGL(bn_reduce_finalize_kernel<NWV>, dim3((unsigned)((C + 63) / 64), (unsigned)S), dim3(64 * NWV), 0, s, sum, sumsq, \
                           (int)nparts, (float)count, weight, bias, running_mean, running_var, scale, shift, (int)C, eps, \
                           momentum, ws, tickets)
        L(8);
#undef L
    } else if (nparts > BN_RED_PARTS) {
        if (!ws) return AVSR_EINVAL;
        hipLaunchKernelGGL(bn_reduce_parts_kernel, dim3((unsigned)((C + 63) / 64), BN_RED_PARTS), dim3(256), 0, s, sum, sumsq,
                           (int)nparts, (int)C, ws);
        hipLaunchKernelGGL(bn_finalize_kernel, fgrid, fblock, 0, s, ws, ws + C, BN_RED_PARTS, (long)(2 * C), (float)count,
                           weight, bias, running_mean, running_var, scale, shift, (int)C, eps, momentum);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel, fgrid, fblock, 0, s, sum, sumsq, (int)nparts, (long)C, (float)count, weight,
                           bias, running_mean, running_var, scale, shift, (int)C, eps, momentum);
    }
    return avsr_launch_status();
}

extern "C" int avsr_bn_fold(avsr_stream_t stream, const float* weight, const float* bias, const float* running_mean,
                            const float* running_var, float* scale, float* shift, int64_t C, float eps) {
    if (!weight || !bias || !running_mean || !running_var || !scale || !shift || C <= 0) return AVSR_EINVAL;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, bias,
                       running_mean, running_var, scale, shift, (int)C, eps);
    return avsr_launch_status();
}

extern "C" int avsr_bn_apply(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, const void* residual,
                             void* y, const float* scale, const float* shift, const float* res_scale,
                             const float* res_shift, int64_t rows, int64_t C, int relu) {
    if (!x || !y || !scale || !shift || rows <= 0 || C <= 0) return AVSR_EINVAL;
    if ((res_scale != nullptr) != (res_shift != nullptr) || (res_scale && !residual)) return AVSR_EINVAL;
    if (C % 4) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16 && C % 8 == 0 && (2048 % (C / 8) == 0 || (C / 8) % 256 == 0) &&
        (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) % 16) == 0) {
        const long n8 = rows * C / 8;
        const int C8 = (int)(C / 8);
        // grid stride = 256 * blocks must be a multiple of C8: blocks a multiple of C8 / gcd(C8, 256)
        long unit = 1;
        if (C8 > 256) unit = C8 / 256;                          // C8 is then a multiple of 256
        long blocks = std::min<long>((n8 + 511) / 512, BN_APPLY_BLOCKS);
        blocks = std::max<long>(unit, blocks / unit * unit);
        if (C8 <= 256 && 256 % C8 != 0) blocks = 0;             // not reached: the test above admits powers of two only
        if (blocks > 0) {
            if (res_scale)
                hipLaunchKernelGGL(bn_apply_bf16x8_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x,
                                   (const bf16_t*)residual, (bf16_t*)y, scale, shift, res_scale, res_shift, n8, C8, relu);
            else
                hipLaunchKernelGGL(bn_apply_bf16x8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x,
                                   (const bf16_t*)residual, (bf16_t*)y, scale, shift, res_scale, res_shift, n8, C8, relu);
            return avsr_launch_status();
        }
    }
    const long n4 = rows * C / 4;
    long nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
#define L(TI, TO) hipLaunchKernelGGL((bn_apply_kernel<TI, TO>), dim3((unsigned)nb), dim3(256), 0, s, (const TI*)x, \
                                     (const TO*)residual, (TO*)y, scale, shift, res_scale, res_shift, n4, (int)C, relu)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}
