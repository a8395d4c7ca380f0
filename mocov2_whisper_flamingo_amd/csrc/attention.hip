// Attention kernels, head_dim 64.
//
//  * attn_fwd_mfma_kernel: flash-style forward for bf16 activations.  Scores
//    are computed TRANSPOSED (S^T = K.Q^T with v_mfma_f32_32x32x16_bf16) so a
//    lane owns one query column: the online-softmax statistics are per lane,
//    and the 32x32 accumulator is, register for register, the B operand of
//    the O^T = V^T.P^T product (no LDS round trip for P).  V is read from its
//    row-major LDS tile with the transposing ds_read_b64_tr_b16.
//  * attn_*_valu_kernel: exact fp32-math forward / backward (dQ and dK,dV)
//    with key-padding mask, attention dropout and saved log-sum-exp; used for
//    the fp32 parity path and for the trainable fusion blocks.
//
// Layout of every operand: element (b, t, h, d) at ptr + (b*T + t)*ld + h*64 + d.
#include "common.h"

#define HD 64
#define LOG2E 1.4426950408889634f

// ---------------------------------------------------------------- MFMA ----
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                            long ldk, const bf16_t* __restrict__ V, long ldv,
                                                            bf16_t* __restrict__ O, long ldo, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ lse, int Tq, int Tk, int H, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * 64];
    const int b = blockIdx.z, h = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int q = blockIdx.x * 128 + wave * 32 + r;          // this lane's query row
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 qf[4];
    {
        const bf16_t* qp = Q + ((long)b * Tq + (q < Tq ? q : 0)) * ldq + h * HD + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = q < Tq ? *(const bf16x8*)(qp + ks * 16) : zero8;
    }
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float mrun = -INFINITY, lsum = 0.f;                       // running max (raw score units), partial row sum
    const float sc2 = scale * LOG2E;
    const int lc = t & 7, lr = t >> 3;                        // staging: chunk, row (+32)
    const bf16_t* Kb = K + (long)b * Tk * ldk + h * HD;
    const bf16_t* Vb = V + (long)b * Tk * ldv + h * HD;
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;

    // register prefetch of the next K/V tile: issued before the current tile's math, written to LDS
    // after it, so the global-load latency hides under the MFMA / softmax work
    bf16x8 kreg[2], vreg[2];
    auto prefetch = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kv0 + lr + 32 * i;
            const bool ok = key < Tk;
            kreg[i] = ok ? *(const bf16x8*)(Kb + (long)key * ldk + lc * 8) : zero8;
            vreg[i] = ok ? *(const bf16x8*)(Vb + (long)key * ldv + lc * 8) : zero8;
        }
    };
    prefetch(0);
    for (int kv0 = 0; kv0 < Tk; kv0 += 64) {
        __syncthreads();                                      // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(bf16x8*)&Ks[k_off(lr + 32 * i, lc)] = kreg[i];
            *(bf16x8*)&Vs[v_off(lr + 32 * i, lc)] = vreg[i];
        }
        __syncthreads();
        if (kv0 + 64 < Tk) prefetch(kv0 + 64);

        f32x16 sacc[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[st][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)&Ks[k_off(32 * st + r, 2 * ks + hf)];
                sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[st], 0, 0, 0);
            }
        }
        // sacc[st][reg] = S[q][key = kv0 + 32 st + (reg&3) + 8 (reg>>2) + 4 hf]   (raw, unscaled)
        if (mb || kv0 + 64 > Tk) {                            // block-uniform: only masked / tail tiles pay for this
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * hf;
                    const bool dead = key >= Tk || (mb && mb[key < Tk ? key : 0]);
                    if (dead) sacc[st][i] = -INFINITY;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sacc[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[1][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__any(mx > mrun)) {                               // some row's maximum moved: rescale what is accumulated
            const float mnew = fmaxf(mrun, mx);
            const float alpha = mnew > mrun ? exp2f((mrun - mnew) * sc2) : 1.0f;
            mrun = mnew;
            lsum *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        }
        const float msc = mrun == -INFINITY ? 0.f : mrun * sc2;
        float ps = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[st][i], sc2, -msc)); sacc[st][i] = pv; ps += pv; }
        lsum += ps;

#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)sacc[st][8 * s2 + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    // V^T fragment: element j <-> key 32 st + 16 s2 + 8 (j>>2) + 4 hf + (j&3), d = 32 db + r
                    const int i16 = lane & 15;
                    const int col = 32 * db + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
                    bf16x8 vf;
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        const int row = 32 * st + 16 * s2 + 8 * jh + 4 * hf + (i16 >> 2);
                        const bf16_t* ap = &Vs[v_off(row, col >> 3) + (col & 7)];
                        const s16x4 tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(ap));
                        const bf16x4 tb = __builtin_bit_cast(bf16x4, tr);
                        vf[4 * jh + 0] = tb[0]; vf[4 * jh + 1] = tb[1]; vf[4 * jh + 2] = tb[2]; vf[4 * jh + 3] = tb[3];
                    }
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    if (q < Tq) {
        bf16_t* op = O + ((long)b * Tq + q) * ldo + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(oacc[db][4 * g + e] * inv);
                *(bf16x4*)(op + 32 * db + 8 * g + 4 * hf) = o;
            }
        if (lse && hf == 0)
            lse[((long)b * H + h) * Tq + q] = lsum > 0.f ? mrun * scale + logf(lsum) : INFINITY;
    }
}

// ---------------------------------------------------------------- VALU ----
// Block = 4 waves = 32 query rows (8 per wave, processed one at a time); key
// tiles of 64: lane j of a wave owns key j of the tile for the score, and
// column d = lane of the output row for the P.V accumulation.
#define KLD 65

template <typename T>
__device__ __forceinline__ void stage_rows_f32(float* dst, const T* src, long ld, int row0, int nrows, int limit) {
    // dst[row][KLD] <- src rows row0..row0+nrows-1 (64 columns), zero past `limit`
    for (int i = threadIdx.x; i < nrows * 64; i += blockDim.x) {
        const int rr = i >> 6, d = i & 63;
        const int g = row0 + rr;
        dst[rr * KLD + d] = g < limit ? to_f32<T>(src[(long)g * ld + d]) : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_valu_kernel(const T* __restrict__ Q, long ldq, const T* __restrict__ K, long ldk,
                                                            const T* __restrict__ V, long ldv, T* __restrict__ O, long ldo,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ lse,
                                                            int Tq, int Tk, int H, float scale, float pdrop, uint64_t seed) {
    __shared__ float Ks[64 * KLD], Vs[64 * KLD], Qs[32 * KLD];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows_f32<T>(Qs, Q + (long)b * Tq * ldq + h * HD, ldq, q0, 32, Tq);
    float m[8], l[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = -INFINITY; l[i] = 0.f; o[i] = 0.f; }
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;
    const float keep_scale = pdrop > 0.f ? 1.0f / (1.0f - pdrop) : 1.0f;
    for (int kv0 = 0; kv0 < Tk; kv0 += 64) {
        __syncthreads();
        stage_rows_f32<T>(Ks, K + (long)b * Tk * ldk + h * HD, ldk, kv0, 64, Tk);
        stage_rows_f32<T>(Vs, V + (long)b * Tk * ldv + h * HD, ldv, kv0, 64, Tk);
        __syncthreads();
        const int key = kv0 + lane;
        const bool dead = key >= Tk || (mb && mb[key < Tk ? key : 0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qr = wave * 8 + i, qg = q0 + qr;
            float s = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) s = fmaf(Qs[qr * KLD + d], Ks[lane * KLD + d], s);
            s = dead ? -INFINITY : s * scale;
            const float mnew = fmaxf(m[i], wave_max(s));
            const float msafe = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = __expf(m[i] - msafe);
            float p = __expf(s - msafe);
            m[i] = mnew;
            l[i] = l[i] * alpha + wave_sum(p);
            if (pdrop > 0.f) {
                const uint64_t idx = (((uint64_t)b * H + h) * Tq + qg) * (uint64_t)Tk + key;
                p = (!dead && qg < Tq && avsr_keep(seed, idx, pdrop)) ? p * keep_scale : 0.f;
            }
            float acc = o[i] * alpha;
#pragma unroll 16
            for (int j = 0; j < 64; ++j) acc = fmaf(__shfl(p, j, 64), Vs[j * KLD + lane], acc);
            o[i] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int qg = q0 + wave * 8 + i;
        if (qg < Tq) {
            const float inv = l[i] > 0.f ? 1.0f / l[i] : 0.f;
            O[((long)b * Tq + qg) * ldo + h * HD + lane] = from_f32<T>(o[i] * inv);
            if (lse && lane == 0) lse[((long)b * H + h) * Tq + qg] = l[i] > 0.f ? m[i] + __logf(l[i]) : INFINITY;
        }
    }
}

// dQ (and the per-row D = dO.O the dK/dV kernel needs).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_valu_kernel(const T* __restrict__ Q, long ldq, const T* __restrict__ K, long ldk,
                                                               const T* __restrict__ V, long ldv, const T* __restrict__ O, long ldo,
                                                               const T* __restrict__ dO, long lddo, T* __restrict__ dQ, long lddq,
                                                               const uint8_t* __restrict__ mask, const float* __restrict__ lse,
                                                               float* __restrict__ Dout, int Tq, int Tk, int H, float scale,
                                                               float pdrop, uint64_t seed) {
    __shared__ float Ks[64 * KLD], Vs[64 * KLD], Qs[32 * KLD], dOs[32 * KLD];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows_f32<T>(Qs, Q + (long)b * Tq * ldq + h * HD, ldq, q0, 32, Tq);
    stage_rows_f32<T>(dOs, dO + (long)b * Tq * lddo + h * HD, lddo, q0, 32, Tq);
    float Dr[8], L[8], dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int qg = q0 + wave * 8 + i;
        float d = 0.f;
        if (qg < Tq) d = to_f32<T>(O[((long)b * Tq + qg) * ldo + h * HD + lane]) *
                         to_f32<T>(dO[((long)b * Tq + qg) * lddo + h * HD + lane]);
        Dr[i] = wave_sum(d);
        L[i] = qg < Tq ? lse[((long)b * H + h) * Tq + qg] : INFINITY;
        dq[i] = 0.f;
        if (qg < Tq && lane == 0) Dout[((long)b * H + h) * Tq + qg] = Dr[i];
    }
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;
    const float keep_scale = pdrop > 0.f ? 1.0f / (1.0f - pdrop) : 1.0f;
    for (int kv0 = 0; kv0 < Tk; kv0 += 64) {
        __syncthreads();
        stage_rows_f32<T>(Ks, K + (long)b * Tk * ldk + h * HD, ldk, kv0, 64, Tk);
        stage_rows_f32<T>(Vs, V + (long)b * Tk * ldv + h * HD, ldv, kv0, 64, Tk);
        __syncthreads();
        const int key = kv0 + lane;
        const bool dead = key >= Tk || (mb && mb[key < Tk ? key : 0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qr = wave * 8 + i, qg = q0 + qr;
            float s = 0.f, dp = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) {
                s = fmaf(Qs[qr * KLD + d], Ks[lane * KLD + d], s);
                dp = fmaf(dOs[qr * KLD + d], Vs[lane * KLD + d], dp);
            }
            const float p = dead ? 0.f : __expf(s * scale - L[i]);
            if (pdrop > 0.f) {
                const uint64_t idx = (((uint64_t)b * H + h) * Tq + qg) * (uint64_t)Tk + key;
                dp = (!dead && qg < Tq && avsr_keep(seed, idx, pdrop)) ? dp * keep_scale : 0.f;
            }
            const float ds = p * (dp - Dr[i]) * scale;
            float acc = dq[i];
#pragma unroll 16
            for (int j = 0; j < 64; ++j) acc = fmaf(__shfl(ds, j, 64), Ks[j * KLD + lane], acc);
            dq[i] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int qg = q0 + wave * 8 + i;
        if (qg < Tq) dQ[((long)b * Tq + qg) * lddq + h * HD + lane] = from_f32<T>(dq[i]);
    }
}

// dK, dV for one tile of 64 keys: lane j owns key j; the four waves split the
// query rows and their partial sums are combined through LDS.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_valu_kernel(const T* __restrict__ Q, long ldq, const T* __restrict__ K, long ldk,
                                                                const T* __restrict__ V, long ldv, const T* __restrict__ dO, long lddo,
                                                                T* __restrict__ dK, long lddk, T* __restrict__ dV, long lddv,
                                                                const uint8_t* __restrict__ mask, const float* __restrict__ lse,
                                                                const float* __restrict__ Dsum, int Tq, int Tk, int H, float scale,
                                                                float pdrop, uint64_t seed) {
    __shared__ float Ks[64 * KLD], Vs[64 * KLD], Qs[32 * KLD], dOs[32 * KLD];
    const int b = blockIdx.z, h = blockIdx.y, kv0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows_f32<T>(Ks, K + (long)b * Tk * ldk + h * HD, ldk, kv0, 64, Tk);
    stage_rows_f32<T>(Vs, V + (long)b * Tk * ldv + h * HD, ldv, kv0, 64, Tk);
    const int key = kv0 + lane;
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;
    const bool dead = key >= Tk || (mb && mb[key < Tk ? key : 0]);
    const float keep_scale = pdrop > 0.f ? 1.0f / (1.0f - pdrop) : 1.0f;
    float dk[64], dv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int q0 = 0; q0 < Tq; q0 += 32) {
        __syncthreads();
        stage_rows_f32<T>(Qs, Q + (long)b * Tq * ldq + h * HD, ldq, q0, 32, Tq);
        stage_rows_f32<T>(dOs, dO + (long)b * Tq * lddo + h * HD, lddo, q0, 32, Tq);
        __syncthreads();
        for (int i = 0; i < 8; ++i) {
            const int qr = wave * 8 + i, qg = q0 + qr;
            if (qg >= Tq) break;
            float s = 0.f, dp = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) {
                s = fmaf(Qs[qr * KLD + d], Ks[lane * KLD + d], s);
                dp = fmaf(dOs[qr * KLD + d], Vs[lane * KLD + d], dp);
            }
            const float p = dead ? 0.f : __expf(s * scale - lse[((long)b * H + h) * Tq + qg]);
            float pd = p;
            if (pdrop > 0.f) {
                const uint64_t idx = (((uint64_t)b * H + h) * Tq + qg) * (uint64_t)Tk + key;
                const bool keep = !dead && avsr_keep(seed, idx, pdrop);
                pd = keep ? p * keep_scale : 0.f;
                dp = keep ? dp * keep_scale : 0.f;
            }
            const float ds = p * (dp - Dsum[((long)b * H + h) * Tq + qg]) * scale;
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                dv[d] = fmaf(pd, dOs[qr * KLD + d], dv[d]);
                dk[d] = fmaf(ds, Qs[qr * KLD + d], dk[d]);
            }
        }
    }
    // combine the four waves' partials: reuse Ks / Vs as [64][KLD] accumulators
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                Ks[lane * KLD + d] = (w == 0 ? 0.f : Ks[lane * KLD + d]) + dk[d];
                Vs[lane * KLD + d] = (w == 0 ? 0.f : Vs[lane * KLD + d]) + dv[d];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int rr = i >> 6, d = i & 63, g = kv0 + rr;
        if (g < Tk) {
            dK[((long)b * Tk + g) * lddk + h * HD + d] = from_f32<T>(Ks[rr * KLD + d]);
            dV[((long)b * Tk + g) * lddv + h * HD + d] = from_f32<T>(Vs[rr * KLD + d]);
        }
    }
}

// ---------------------------------------------------------------- host ----
extern "C" int avsr_attn_fwd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                             int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                             float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                             float p_drop, uint64_t seed) {
    if (!Q || !K || !V || !O || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return AVSR_EINVAL;
    if (head_dim != HD) return AVSR_EUNSUPPORTED;
    if (B > 65535 || H > 65535) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma) {
        if (dtype != AVSR_BF16 || p_drop > 0.f) return AVSR_EUNSUPPORTED;
        if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return AVSR_EUNSUPPORTED;
        if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) % 16 || (uintptr_t)O % 8) return AVSR_EUNSUPPORTED;
        dim3 grid((unsigned)((Tq + 127) / 128), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL(attn_fwd_mfma_kernel, grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq, (const bf16_t*)K,
                           (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse, (int)Tq, (int)Tk,
                           (int)H, scale);
        return avsr_launch_status();
    }
    dim3 grid((unsigned)((Tq + 31) / 32), (unsigned)H, (unsigned)B);
    if (dtype == AVSR_F32)
        hipLaunchKernelGGL(attn_fwd_valu_kernel<float>, grid, dim3(256), 0, s, (const float*)Q, (long)ldq, (const float*)K,
                           (long)ldk, (const float*)V, (long)ldv, (float*)O, (long)ldo, key_mask, lse, (int)Tq, (int)Tk,
                           (int)H, scale, p_drop, seed);
    else if (dtype == AVSR_BF16)
        hipLaunchKernelGGL(attn_fwd_valu_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq,
                           (const bf16_t*)K, (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse,
                           (int)Tq, (int)Tk, (int)H, scale, p_drop, seed);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_attn_bwd(avsr_stream_t stream, int dtype, const void* Q, int64_t ldq, const void* K, int64_t ldk,
                             const void* V, int64_t ldv, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                             void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv,
                             const uint8_t* key_mask, const float* lse, float* d_ws, int64_t B, int64_t H, int64_t Tq,
                             int64_t Tk, int64_t head_dim, float scale, float p_drop, uint64_t seed) {
    if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !lse || !d_ws) return AVSR_EINVAL;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return AVSR_EINVAL;
    if (head_dim != HD || B > 65535 || H > 65535) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 gq((unsigned)((Tq + 31) / 32), (unsigned)H, (unsigned)B), gk((unsigned)((Tk + 63) / 64), (unsigned)H, (unsigned)B);
#define RUN(T)                                                                                                          \
    hipLaunchKernelGGL(attn_bwd_dq_valu_kernel<T>, gq, dim3(256), 0, s, (const T*)Q, (long)ldq, (const T*)K, (long)ldk, \
                       (const T*)V, (long)ldv, (const T*)O, (long)ldo, (const T*)dO, (long)lddo, (T*)dQ, (long)lddq,    \
                       key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed);                             \
    hipLaunchKernelGGL(attn_bwd_dkv_valu_kernel<T>, gk, dim3(256), 0, s, (const T*)Q, (long)ldq, (const T*)K, (long)ldk,\
                       (const T*)V, (long)ldv, (const T*)dO, (long)lddo, (T*)dK, (long)lddk, (T*)dV, (long)lddv,        \
                       key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed)
    if (dtype == AVSR_F32) { RUN(float); }
    else if (dtype == AVSR_BF16) { RUN(bf16_t); }
    else return AVSR_EINVAL;
#undef RUN
    return avsr_launch_status();
}
