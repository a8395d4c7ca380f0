// Small HBM-bound helpers around the GEMMs: dtype casts, tiled transposes,
// column sums (bias gradients), GELU forward / backward, the tanh-gated
// residual with dropout (forward / backward), sums, and the fused AdamW step.
#include "common.h"

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n, float scale) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i + e] = from_f32<TO>(to_f32<TI>(x[i + e]) * scale);
        } else {
            for (long j = i; j < n; ++j) y[j] = from_f32<TO>(to_f32<TI>(x[j]) * scale);
        }
    }
}

// y[r][c] = x[r][c] * scale for row-strided operands (cols a multiple of 4): a cast into, or out of, a padded buffer
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_rows_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                        long rows, long cols4, float scale) {
    const long n = rows * cols4, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long r = i / cols4, c = (i - r * cols4) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[r * ldy + c + e] = from_f32<TO>(to_f32<TI>(x[r * ldx + c + e]) * scale);
    }
}

// y[b][c][r] = x[b][r][c] * scale  (32x32 LDS tile, padded against bank conflicts); `copy`, when given,
// also receives the untransposed x * scale in the output type ([R, ldc] rows, batch 1 only): a weight's
// compute-dtype copy and its transpose in one pass.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ x, long ldx, long sx, TO* __restrict__ y, long ldy,
                                                        long sy, int R, int C, float scale, TO* __restrict__ copy, long ldc) {
    __shared__ float tile[32][33];
    const TI* xb = x + (long)blockIdx.z * sx;
    TO* yb = y + (long)blockIdx.z * sy;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const bool in = r < R && c < C;
        const float v = in ? to_f32<TI>(xb[(long)r * ldx + c]) * scale : 0.f;
        tile[i][tx] = v;
        if (copy && in) copy[(long)r * ldc + c] = from_f32<TO>(v);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) yb[(long)c * ldy + r] = from_f32<TO>(tile[tx][i]);
    }
}

// out[n] (+)= sum_m x[m][n]; a 1024-thread block per 64 columns and row slice, 16 waves split the slice's rows.
// One slice (gridDim.y == 1): plain store, fixed summation order.  Several slices (tall operands): each block adds
// its partial with one atomic per column; `out` then has to hold the value to add to (zero for a plain sum).
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* __restrict__ x, long ld, int M, int N, float* __restrict__ out,
                                                      int accumulate, int rows_per_slice) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int m_begin = blockIdx.y * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = m_begin + w;
        for (; m + 48 < m_end; m += 64) {
            s0 += to_f32<T>(x[(long)m * ld + n]);
            s1 += to_f32<T>(x[(long)(m + 16) * ld + n]);
            s2 += to_f32<T>(x[(long)(m + 32) * ld + n]);
            s3 += to_f32<T>(x[(long)(m + 48) * ld + n]);
        }
        for (; m < m_end; m += 16) s0 += to_f32<T>(x[(long)m * ld + n]);
    }
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][lane];
        if (gridDim.y > 1) atomicAdd(out + n, t);
        else out[n] = accumulate ? out[n] + t : t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f32<T>(gelu_erf(to_f32<T>(x[i])));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = from_f32<T>(to_f32<T>(dy[i]) * gelu_erf_grad(to_f32<T>(x[i])));
}

// out = x + tanh(gate) * drop(y)   (models/gate_cross_attention.py:43,47 with the
// nn.Dropout of :26 folded in; keep decision = avsr_keep(seed, element index)).
template <typename T>
__global__ __launch_bounds__(256) void gated_residual_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                 const float* __restrict__ gate, T* __restrict__ out, long n,
                                                                 float pdrop, uint64_t seed) {
    const float g = tanhf(gate[0]);
    const float ks = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = to_f32<T>(y[i]);
        if (pdrop > 0.f) v = avsr_keep(seed, (uint64_t)i, pdrop) ? v * ks : 0.f;
        out[i] = from_f32<T>(to_f32<T>(x[i]) + g * v);
    }
}
// dy = tanh(g) * drop(dout);  dgate += (1 - tanh(g)^2) * sum(dout * drop(y)).  VEC: 8 consecutive elements per thread
// and step (16-byte accesses for bf16); every workgroup ends in ONE atomic on the gate gradient.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void gated_residual_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                                 const float* __restrict__ gate, T* __restrict__ dy,
                                                                 float* __restrict__ dgate, long n, float pdrop, uint64_t seed_in,
                                                                 const uint64_t* __restrict__ seed_delta) {
    __shared__ float red[16];
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);
    const float g = tanhf(gate[0]);
    const float ks = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
    const long stride = (long)gridDim.x * blockDim.x;
    float acc = 0.f;
    if constexpr (VEC) {
        typedef __attribute__((ext_vector_type(8))) T vec8;
        for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride * 8) {
            const vec8 dv = *(const vec8*)(dout + i), yv = *(const vec8*)(y + i);
            vec8 o;
            uint32_t draws[8];                            // i is a multiple of 8: two generator calls cover the eight
            if (pdrop > 0.f) { avsr_rand4(seed, (uint64_t)i >> 2, draws); avsr_rand4(seed, ((uint64_t)i >> 2) + 1, draws + 4); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = to_f32<T>(dv[e]);
                if (pdrop > 0.f) d = avsr_bits_keep(draws[e], pdrop) ? d * ks : 0.f;
                acc += d * to_f32<T>(yv[e]);
                o[e] = from_f32<T>(g * d);
            }
            *(vec8*)(dy + i) = o;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float d = to_f32<T>(dout[i]);
            if (pdrop > 0.f) d = avsr_keep(seed, (uint64_t)i, pdrop) ? d * ks : 0.f;
            acc += d * to_f32<T>(y[i]);
            dy[i] = from_f32<T>(g * d);
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(dgate, (1.f - g * g) * acc);
}

// out = a + b + c  (models/av_net.py:106)
template <typename T>
__global__ __launch_bounds__(256) void add3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                   T* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]) + (c ? to_f32<T>(c[i]) : 0.f));
}

// out[i] = x[i] + table[i % period]: a [T, d] fp32 table added to every batch item (models/utils.py:25)
template <typename T>
__global__ __launch_bounds__(256) void add_table_kernel(const T* __restrict__ x, const float* __restrict__ table,
                                                        T* __restrict__ out, long n, long period) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = from_f32<T>(to_f32<T>(x[i]) + table[i % period]);
}

// out[i] = sum_s parts[s][i] (+ bias[i % ncol]); combines split-K partial products
template <typename TO>
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, long part_stride, int S,
                                                        const float* __restrict__ bias, int ncol, TO* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        f32x4 a = *(const f32x4*)(parts + i);
        for (int s = 1; s < S; ++s) { const f32x4 b = *(const f32x4*)(parts + (long)s * part_stride + i); a += b; }
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += bias[(i + e) % ncol];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) out[i + e] = from_f32<TO>(a[e]);
    }
}

// out[0] += sum(x^2)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[16];
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (((uintptr_t)x & 15) == 0) {                       // 16-byte loads over the aligned body, scalars over the tail
        const long n4 = n >> 2;
        const float4* x4 = (const float4*)x;
        float a0 = 0.f, a1 = 0.f;
        long i = tid;
        for (; i + stride < n4; i += 2 * stride) {
            const float4 u = x4[i], w = x4[i + stride];
            a0 += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            a1 += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
        }
        if (i < n4) { const float4 u = x4[i]; a0 += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w; }
        acc = a0 + a1;
        for (long j = (n4 << 2) + tid; j < n; j += stride) acc += x[j] * x[j];
    } else {
        for (long i = tid; i < n; i += stride) acc += x[i] * x[i];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, acc);
}

// AdamW over a flat fp32 parameter buffer (torch.optim.AdamW semantics:
// decoupled weight decay, bias-corrected moments, eps added to sqrt(v_hat)).
// grad_scale[0] multiplies the gradient first (global-norm clipping factor);
// a bf16 shadow copy of the new weights is written when `shadow` is set; `clear_grad` leaves the gradient
// buffer zeroed for the next accumulation (saves the separate fill pass).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ grad_scale, int clear_grad,
                                                    const float* __restrict__ hyper, float* __restrict__ reset_word) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }     // per-step scalars of a step replayed from a graph
    if (reset_word && blockIdx.x == 0 && threadIdx.x == 0) *reset_word = 0.f;
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] * gs;
        if (clear_grad) g[i] = 0.f;
        float w = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gr;
        const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        w -= (lr / bc1) * mi / denom;
        p[i] = w;
        if (shadow) shadow[i] = (bf16_t)w;
    }
}

// norm = pre_scale * sqrt(sumsq[0]);  scale[0] = pre_scale * min(1, max_norm / (norm + 1e-6))   (torch
// clip_grad_norm_ applied to pre_scale * gradient: pre_scale = 1 / world turns summed gradients into means)
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float max_norm, float pre_scale,
                                  float* __restrict__ scale, float* __restrict__ norm_out) {
    const float nrm = pre_scale * sqrtf(sumsq[0]);
    if (norm_out) norm_out[0] = nrm;
    scale[0] = pre_scale * fminf(1.f, max_norm / (nrm + 1e-6f));
}

// --------------------------------------------------------------- host ------
static inline unsigned grid_for(long n, int per_block, long cap) {
    long nb = (n + per_block - 1) / per_block;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

// out[row, :] = table[ids[row], :] * scale + pos[past + row % T, :]: token + learned position embeddings of
// the Whisper decoder, written to the fp32 residual stream.  An id outside [0, V) writes the position row
// alone and raises the error flag (the host reports it), never reads outside the table.
template <typename TW>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long* __restrict__ ids, const TW* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ out, long rows,
                                                           int T, int past, int D, long V, float scale, int* __restrict__ bad) {
    const int d4 = D >> 2;
    const long n = rows * d4, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long row = i / d4;
        const int c = (int)(i - row * d4) * 4;
        const long id = ids[row];
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (id >= 0 && id < V) {
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = to_f32<TW>(table[id * D + c + k]);
        } else if (c == 0 && bad) atomicOr(bad, 1);
        const f32x4 pp = *(const f32x4*)(pos + (long)(past + (int)(row % T)) * D + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = e[k] * scale + pp[k];
        *(f32x4*)(out + row * D + c) = o;
    }
}

extern "C" int avsr_cast(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y, int64_t n, float scale) {
    if (!x || !y || n <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 1024, 8192)), block(256);
#define L(TI, TO) hipLaunchKernelGGL((cast_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, (TO*)y, (long)n, scale)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_F32) L(bf16_t, float);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

struct ZeroRegions { void* ptr[4]; long ld[4]; long rows[4]; long words[4]; };
__global__ void zero_regions_kernel(ZeroRegions z) {
    const int r = blockIdx.y;
    const long n = z.rows[r] * z.words[r];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long row = i / z.words[r], w = i - row * z.words[r];
        *(unsigned*)((char*)z.ptr[r] + row * z.ld[r] + 4 * w) = 0u;
    }
}

// Up to four strided regions (rows x row_bytes each, row pitch ld_bytes) set to zero by ONE launch: the padding frames
// around the clips of a batch (the zero frame before / after each clip that turns the Whisper stem's padded
// convolutions, HF modeling_whisper.py WhisperEncoder.forward conv1 / conv2, into plain products).
extern "C" int avsr_zero_regions(avsr_stream_t stream, int64_t n, void* const* ptr, const int64_t* ld_bytes, const int64_t* rows,
                                 const int64_t* row_bytes) {
    if (n <= 0 || n > 4 || !ptr || !ld_bytes || !rows || !row_bytes) return AVSR_EINVAL;
    ZeroRegions z;
    long most = 0;
    for (int i = 0; i < 4; ++i) { z.ptr[i] = nullptr; z.ld[i] = 0; z.rows[i] = 0; z.words[i] = 0; }
    for (int i = 0; i < n; ++i) {
        if (!ptr[i] || rows[i] <= 0 || row_bytes[i] <= 0 || ld_bytes[i] < row_bytes[i]) return AVSR_EINVAL;
        if (row_bytes[i] % 4 || ld_bytes[i] % 4 || (uintptr_t)ptr[i] % 4) return AVSR_EUNSUPPORTED;
        z.ptr[i] = ptr[i]; z.ld[i] = ld_bytes[i]; z.rows[i] = rows[i]; z.words[i] = row_bytes[i] / 4;
        most = std::max<long>(most, rows[i] * (row_bytes[i] / 4));
    }
    dim3 grid(grid_for(most, 256, 1024), (unsigned)n);
    hipLaunchKernelGGL(zero_regions_kernel, grid, dim3(256), 0, (hipStream_t)stream, z);
    return avsr_launch_status();
}

extern "C" int avsr_cast_rows(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx, void* y,
                              int64_t ldy, int64_t rows, int64_t cols, float scale) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return AVSR_EINVAL;
    if (cols % 4) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(rows * (cols / 4), 256, 8192)), block(256);
#define L(TI, TO) hipLaunchKernelGGL((cast_rows_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, (long)ldx, (TO*)y, (long)ldy, \
                                     (long)rows, (long)(cols / 4), scale)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_F32) L(bf16_t, float);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

extern "C" int avsr_transpose(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx, int64_t sx,
                              void* y, int64_t ldy, int64_t sy, int64_t batch, int64_t R, int64_t C, float scale,
                              void* copy, int64_t ldc) {
    if (!x || !y || batch <= 0 || R <= 0 || C <= 0 || batch > 65535) return AVSR_EINVAL;
    if (copy && (batch != 1 || ldc < C)) return AVSR_EINVAL;
    if ((R + 31) / 32 > 65535) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)batch), block(256);
#define L(TI, TO) hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, (long)ldx, (long)sx, (TO*)y, \
                                     (long)ldy, (long)sy, (int)R, (int)C, scale, (TO*)copy, (long)ldc)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_F32) L(bf16_t, float);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

extern "C" int avsr_colsum(avsr_stream_t stream, int dtype, const void* x, int64_t ld, int64_t M, int64_t N, float* out,
                           int accumulate) {
    if (!x || !out || M <= 0 || N <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype != AVSR_F32 && dtype != AVSR_BF16) return AVSR_EINVAL;
    // tall operands: slices of 1024 rows, as many as fill the chip about twice with the column groups
    const long col_groups = (N + 63) / 64;
    long slices = 1;
    if (M > 2048) {
        slices = (M + 1023) / 1024;
        const long want = (512 + col_groups - 1) / col_groups;
        if (slices > want) slices = want;
        if (slices < 1) slices = 1;
    }
    const int rows_per_slice = (int)((M + slices - 1) / slices);
    if (slices > 1 && !accumulate && hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s) != hipSuccess) return AVSR_EHIP;
    dim3 grid((unsigned)col_groups, (unsigned)slices), block(1024);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)x, (long)ld, (int)M, (int)N, out, accumulate, rows_per_slice);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (long)ld, (int)M, (int)N, out, accumulate, rows_per_slice);
    return avsr_launch_status();
}

extern "C" int avsr_gelu_fwd(avsr_stream_t stream, int dtype, const void* x, void* y, int64_t n) {
    if (!x || !y || n <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 8192)), block(256);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(gelu_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, (long)n);
    else if (dtype == AVSR_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, (long)n);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_gelu_bwd(avsr_stream_t stream, int dtype, const void* x, const void* dy, void* dx, int64_t n) {
    if (!x || !dy || !dx || n <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 8192)), block(256);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, block, 0, s, (const float*)x, (const float*)dy, (float*)dx, (long)n);
    else if (dtype == AVSR_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, (long)n);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_gated_residual_fwd(avsr_stream_t stream, int dtype, const void* x, const void* y, const float* gate,
                                       void* out, int64_t n, float p_drop, uint64_t seed) {
    if (!x || !y || !gate || !out || n <= 0 || p_drop < 0.f || p_drop >= 1.f) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 4096)), block(256);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(gated_residual_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (const float*)y, gate, (float*)out, (long)n, p_drop, seed);
    else if (dtype == AVSR_BF16) hipLaunchKernelGGL(gated_residual_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (const bf16_t*)y, gate, (bf16_t*)out, (long)n, p_drop, seed);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_gated_residual_bwd(avsr_stream_t stream, int dtype, const void* dout, const void* y, const float* gate,
                                       void* dy, float* dgate, int64_t n, float p_drop, uint64_t seed) {
    return avsr_gated_residual_bwd_sd(stream, dtype, dout, y, gate, dy, dgate, n, p_drop, seed, nullptr);
}

extern "C" int avsr_gated_residual_bwd_sd(avsr_stream_t stream, int dtype, const void* dout, const void* y, const float* gate,
                                          void* dy, float* dgate, int64_t n, float p_drop, uint64_t seed,
                                          const uint64_t* seed_delta) {
    if (!dout || !y || !gate || !dy || !dgate || n <= 0 || p_drop < 0.f || p_drop >= 1.f) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype != AVSR_F32 && dtype != AVSR_BF16) return AVSR_EINVAL;
    const size_t es = dtype == AVSR_BF16 ? 2 : 4;
    const bool vec = n % 8 == 0 && (((uintptr_t)dout | (uintptr_t)y | (uintptr_t)dy) % (8 * es)) == 0;
    // each workgroup ends in ONE atomic on the gate gradient: 64 workgroups up to 2 M elements (the step is launch
    // bound there), then one per 32 K elements up to twice the CU count
    long blocks = (n + 32767) / 32768;
    blocks = blocks < 64 ? 64 : (blocks > 512 ? 512 : blocks);
    dim3 grid(grid_for(n, vec ? 2048 : 256, blocks)), block(256);
#define L(T, V) hipLaunchKernelGGL((gated_residual_bwd_kernel<T, V>), grid, block, 0, s, (const T*)dout, (const T*)y, gate, (T*)dy, dgate, (long)n, p_drop, seed, seed_delta)
    if (dtype == AVSR_F32) { if (vec) L(float, true); else L(float, false); }
    else { if (vec) L(bf16_t, true); else L(bf16_t, false); }
#undef L
    return avsr_launch_status();
}

extern "C" int avsr_embed_tokens(avsr_stream_t stream, int table_dtype, const int64_t* ids, const void* table, const float* pos,
                                 float* out, int64_t rows, int64_t T, int64_t past, int64_t D, int64_t V, int64_t n_pos,
                                 float scale, int* bad_id_flag) {
    if (!ids || !table || !pos || !out || rows <= 0 || T <= 0 || past < 0 || D <= 0 || V <= 0) return AVSR_EINVAL;
    if (past + T > n_pos) return AVSR_EINVAL;                  // more positions than the learned table holds
    if (D % 4 || ((uintptr_t)table | (uintptr_t)pos | (uintptr_t)out) % 16) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(rows * (D / 4), 256, 4096)), block(256);
    if (table_dtype == AVSR_F32)
        hipLaunchKernelGGL(embed_tokens_kernel<float>, grid, block, 0, s, (const long*)ids, (const float*)table, pos, out,
                           (long)rows, (int)T, (int)past, (int)D, (long)V, scale, bad_id_flag);
    else if (table_dtype == AVSR_BF16)
        hipLaunchKernelGGL(embed_tokens_kernel<bf16_t>, grid, block, 0, s, (const long*)ids, (const bf16_t*)table, pos, out,
                           (long)rows, (int)T, (int)past, (int)D, (long)V, scale, bad_id_flag);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_add3(avsr_stream_t stream, int dtype, const void* a, const void* b, const void* c, void* out, int64_t n) {
    if (!a || !b || !out || n <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 4096)), block(256);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(add3_kernel<float>, grid, block, 0, s, (const float*)a, (const float*)b, (const float*)c, (float*)out, (long)n);
    else if (dtype == AVSR_BF16) hipLaunchKernelGGL(add3_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)out, (long)n);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_add_table(avsr_stream_t stream, int dtype, const void* x, const float* table, void* out, int64_t n,
                              int64_t period) {
    if (!x || !table || !out || n <= 0 || period <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 4096)), block(256);
    if (dtype == AVSR_F32) hipLaunchKernelGGL(add_table_kernel<float>, grid, block, 0, s, (const float*)x, table, (float*)out, (long)n, (long)period);
    else if (dtype == AVSR_BF16) hipLaunchKernelGGL(add_table_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, table, (bf16_t*)out, (long)n, (long)period);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_sum_parts(avsr_stream_t stream, int out_dtype, const float* parts, int64_t part_stride, int64_t S,
                              const float* bias, int64_t ncol, void* out, int64_t n) {
    if (!parts || !out || S <= 0 || n <= 0 || (bias && ncol <= 0)) return AVSR_EINVAL;
    if (n % 4 || part_stride % 4 || ((uintptr_t)parts % 16)) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n, 1024, 4096)), block(256);
    if (out_dtype == AVSR_F32) hipLaunchKernelGGL(sum_parts_kernel<float>, grid, block, 0, s, parts, (long)part_stride, (int)S, bias, (int)ncol, (float*)out, (long)n);
    else if (out_dtype == AVSR_BF16) hipLaunchKernelGGL(sum_parts_kernel<bf16_t>, grid, block, 0, s, parts, (long)part_stride, (int)S, bias, (int)ncol, (bf16_t*)out, (long)n);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

// lengths[b] = min(lengths[b], T) in place; mask[b, t] = t >= lengths[b].  A thread may read a length before or after
// thread t = 0 has clamped it: t >= len and t >= min(len, T) agree for every t < T.
__global__ void length_mask_kernel(int64_t* __restrict__ lengths, int B, int T, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    const int64_t len = lengths[b];
    mask[i] = (uint8_t)(t >= len);
    if (t == 0 && len > T) lengths[b] = T;
}

extern "C" int avsr_length_mask(avsr_stream_t stream, int64_t* lengths, int64_t B, int64_t T, uint8_t* mask) {
    if (!lengths || !mask || B <= 0 || T <= 0 || B * T > 0x7fffffff) return AVSR_EINVAL;
    hipLaunchKernelGGL(length_mask_kernel, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lengths,
                       (int)B, (int)T, mask);
    return avsr_launch_status();
}

extern "C" int avsr_sumsq_accumulate(avsr_stream_t stream, const float* x, int64_t n, float* out) {
    if (!x || !out || n <= 0) return AVSR_EINVAL;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 2048, 2048)), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
    return avsr_launch_status();
}

extern "C" int avsr_clip_scale(avsr_stream_t stream, const float* sumsq, float max_norm, float pre_scale, float* scale,
                               float* norm_out) {
    if (!sumsq || !scale || !(pre_scale > 0.f)) return AVSR_EINVAL;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, pre_scale, scale,
                       norm_out);
    return avsr_launch_status();
}

extern "C" int avsr_adamw_step(avsr_stream_t stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                               void* bf16_shadow, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, const float* grad_scale, int clear_grad) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return AVSR_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (bf16_t*)bf16_shadow, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clear_grad,
                       (const float*)nullptr, (float*)nullptr);
    return avsr_launch_status();
}

struct StepRecord { unsigned long long seed_delta; float lr, bc1, bc2; unsigned int pad[3]; };
static_assert(sizeof(StepRecord) == 32, "the step record is 32 bytes");
__global__ void step_params_kernel(StepRecord* rec, StepRecord v) { *rec = v; }

__global__ void store_u64_kernel(unsigned long long* dst, unsigned long long v, unsigned long long* cdst,
                                 const unsigned long long* csrc, int n) {
    if (threadIdx.x == 0) *dst = v;
    for (int i = threadIdx.x; i < n; i += blockDim.x) cdst[i] = csrc[i];
}

// One 8-byte word written in stream order (the value travels in the launch packet): e.g. the address a recorded
// launch reads its input from (avsr_stem3d_fwd_indirect).  copy_dst / copy_src / n_words (optional): a few more words
// copied device to device by the same launch (the other small inputs of that recording).
extern "C" int avsr_store_u64(avsr_stream_t stream, uint64_t* dst, uint64_t value, uint64_t* copy_dst, const uint64_t* copy_src,
                              int64_t n_words) {
    if (!dst || (uintptr_t)dst % 8 || n_words < 0 || n_words > (1 << 20)) return AVSR_EINVAL;
    if (n_words > 0 && (!copy_dst || !copy_src || ((uintptr_t)copy_dst | (uintptr_t)copy_src) % 8)) return AVSR_EINVAL;
    hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(n_words > 64 ? 256 : 64), 0, (hipStream_t)stream, (unsigned long long*)dst,
                       (unsigned long long)value, (unsigned long long*)copy_dst, (const unsigned long long*)copy_src, (int)n_words);
    return avsr_launch_status();
}

extern "C" int avsr_step_params(avsr_stream_t stream, void* record, uint64_t seed_delta, float lr, float beta1, float beta2,
                                int64_t step) {
    if (!record || step <= 0 || ((uintptr_t)record % 8) != 0) return AVSR_EINVAL;
    StepRecord v{};
    v.seed_delta = seed_delta; v.lr = lr;
    v.bc1 = 1.f - powf(beta1, (float)step); v.bc2 = 1.f - powf(beta2, (float)step);      // as avsr_adamw_step computes them
    hipLaunchKernelGGL(step_params_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (StepRecord*)record, v);
    return avsr_launch_status();
}

extern "C" int avsr_adamw_step_dev(avsr_stream_t stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                   void* bf16_shadow, int64_t n, const float* hyper, float beta1, float beta2, float eps,
                                   float weight_decay, const float* grad_scale, int clear_grad, float* clip_scale_reset) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return AVSR_EINVAL;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (bf16_t*)bf16_shadow, (long)n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale,
                       clear_grad, hyper, clip_scale_reset);
    return avsr_launch_status();
}

extern "C" const char* avsr_version(void) { return "avsr_hip 0.1 (gfx950)"; }
