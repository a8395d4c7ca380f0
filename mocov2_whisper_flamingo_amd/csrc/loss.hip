// Head losses over the [rows, V] logits: row log-sum-exp, label-smoothed cross
// entropy, CTC (alpha/beta recursions in log space) and one fused pass that
// writes d(loss)/d(logits) for CTC + CE, plus the greedy arg-max.
// All of it is HBM bound on the logits tensor; every kernel reads a logits
// row once with one 256-thread block per row.
#include "common.h"

// lse[row] = log sum_v exp(x[row, v]); also sumx[row] = sum_v x[row, v].  One 1024-thread workgroup per row,
// ONE pass with 16-byte loads: every thread keeps a running (max, sum of exp) pair that is rescaled when its
// maximum moves, and the pairs are merged across the workgroup.  Rows must be 16-byte aligned (ld % 4 == 0).
__global__ __launch_bounds__(1024) void row_lse_kernel(const float* __restrict__ x, long ld, int V, float* __restrict__ lse,
                                                       float* __restrict__ sumx) {
    __shared__ float red[16];
    const float* xr = x + (long)blockIdx.x * ld;
    float m = -INFINITY, s = 0.f, sx = 0.f;
    const int V4 = ((ld % 4 == 0) && ((uintptr_t)x % 16 == 0)) ? (V >> 2) : 0;
    for (int i = threadIdx.x; i < V4; i += 1024) {
        const f32x4 a = *(const f32x4*)(xr + 4 * i);
        const float m4 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        if (m4 > m) { s *= __expf(m - m4); m = m4; }
        s += (__expf(a[0] - m) + __expf(a[1] - m)) + (__expf(a[2] - m) + __expf(a[3] - m));
        sx += (a[0] + a[1]) + (a[2] + a[3]);
    }
    for (int v = 4 * V4 + threadIdx.x; v < V; v += 1024) {
        const float a = xr[v];
        if (a > m) { s *= __expf(m - a); m = a; }
        s += __expf(a - m);
        sx += a;
    }
    const float mx = block_max(m, red);
    s = block_sum(m == -INFINITY ? 0.f : s * __expf(m - mx), red);
    sx = block_sum(sx, red);
    if (threadIdx.x == 0) { lse[blockIdx.x] = mx + __logf(s); if (sumx) sumx[blockIdx.x] = sx; }
}

// The same two row quantities from per-tile partials {max, sum of exp(v - max), sum of v} that the product which made the
// logits left behind (part[(tile * rows + row) * 4 + ..]): a wave per row, no pass over the logits.
__global__ __launch_bounds__(256) void row_lse_merge_kernel(const float* __restrict__ part, int nparts, int rows,
                                                            float* __restrict__ lse, float* __restrict__ sumx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float m = -INFINITY, s = 0.f, sx = 0.f;
    for (int t = lane; t < nparts; t += 64) {
        const f32x4 a = *(const f32x4*)(part + ((long)t * rows + row) * 4);
        if (a[0] > m) { s *= __expf(m - a[0]); m = a[0]; }
        s += a[1] * __expf(a[0] - m);
        sx += a[2];
    }
    const float mx = wave_max(m);
    s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - mx));
    sx = wave_sum(sx);
    if (lane == 0) { lse[row] = mx + __logf(s); if (sumx) sumx[row] = sx; }
}

// Label-smoothed CE, mean over non-ignored targets (train.py:103-128).  Rows are
// (b, t) with t < t_ce; target of row (b,t) is targets[b*ldt + t].  One block.
// out[0] = loss, out[1] = number of non-ignored targets.
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ x, long ld, int V, const float* __restrict__ lse,
                                                        const float* __restrict__ sumx, const int64_t* __restrict__ targets,
                                                        long ldt, int B, int T, int t_ce, float eps, int ignore_index,
                                                        float* __restrict__ out) {
    __shared__ float red[16];
    float acc = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < B * t_ce; i += 256) {
        const int b = i / t_ce, t = i - b * t_ce;
        const long tg = targets[(long)b * ldt + t];
        if (tg == ignore_index) continue;
        const long row = (long)b * T + t;
        const float nll = lse[row] - x[row * ld + tg];
        const float smooth = lse[row] - sumx[row] / (float)V;
        acc += (1.f - eps) * nll + eps * smooth;
        cnt += 1.f;
    }
    acc = block_sum(acc, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) { out[0] = cnt > 0.f ? acc / cnt : 0.f; out[1] = cnt; }
}

// ------------------------------------------------------------------ CTC ----
__device__ __forceinline__ float lse2(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) { return lse2(lse2(a, b), c); }

// Two workgroups per batch item (grid = 2B): workgroup b runs the forward (alpha) recursion and writes nll[b] =
// -log p(y|x), workgroup B + b the backward (beta) one.  Extended label sequence l' = (blank, y1, blank, ..., yL,
// blank), S = 2L+1 states.  The previous time step lives in LDS (two rows), and when the whole [Tb, S] table of
// log-probabilities fits there too it is gathered up front in parallel, so a time step is LDS reads and one
// barrier instead of dependent global gathers; every row is also written to the [B, T, Smax] global arrays for the
// gradient kernel.  With zero_infinity an infeasible alignment (nll = inf) is reported as inf here and zeroed by
// the consumers.
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(const float* __restrict__ x, long ld, const float* __restrict__ lse,
                                                             const int64_t* __restrict__ targets, long ldt,
                                                             const int64_t* __restrict__ in_len, const int64_t* __restrict__ tg_len,
                                                             float* __restrict__ alpha, float* __restrict__ beta,
                                                             float* __restrict__ nll, int B, int T, int Smax, int blank,
                                                             int table_in_lds) {
    extern __shared__ float ctc_lds[];                   // [2][Smax] state rows, then (optionally) the [Tb][S] table
    const bool is_beta = (int)blockIdx.x >= B;
    const int b = is_beta ? blockIdx.x - B : blockIdx.x;
    const int Tb = (int)min((long)in_len[b], (long)T);
    const int L = (int)tg_len[b];
    const int S = 2 * L + 1;
    float* out = (is_beta ? beta : alpha) + (long)b * T * Smax;
    const int64_t* tg = targets + (long)b * ldt;
    auto label = [&](int s) -> long { return (s & 1) ? tg[s >> 1] : (long)blank; };
    auto lp = [&](int t, long c) -> float { const long row = (long)b * T + t; return x[row * ld + c] - lse[row]; };
    if (Tb <= 0 || S > Smax) { if (threadIdx.x == 0 && !is_beta) nll[b] = INFINITY; return; }
    float* st = ctc_lds;
    float* tab = ctc_lds + 2 * Smax;
    if (table_in_lds) {
        for (int i = threadIdx.x; i < Tb * S; i += 256) { const int t = i / S, sidx = i - t * S; tab[i] = lp(t, label(sidx)); }
        __syncthreads();
    }
    auto LP = [&](int t, int sidx, long c) -> float { return table_in_lds ? tab[t * S + sidx] : lp(t, c); };
    if (!is_beta) {
        for (int s0 = threadIdx.x; s0 < S; s0 += 256) {
            const float v = (s0 == 0 || s0 == 1) ? LP(0, s0, label(s0)) : -INFINITY;
            st[s0] = v; out[s0] = v;
        }
        __syncthreads();
        for (int t = 1; t < Tb; ++t) {
            const float* prev = st + ((t - 1) & 1) * Smax;
            float* cur = st + (t & 1) * Smax;
            for (int s0 = threadIdx.x; s0 < S; s0 += 256) {
                const long c = label(s0);
                const float a0 = prev[s0];
                const float a1 = s0 >= 1 ? prev[s0 - 1] : -INFINITY;
                const float a2 = (s0 >= 2 && c != blank && label(s0 - 2) != c) ? prev[s0 - 2] : -INFINITY;
                const float v = lse3(a0, a1, a2) + LP(t, s0, c);
                cur[s0] = v; out[(long)t * Smax + s0] = v;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float* last = st + ((Tb - 1) & 1) * Smax;
            const float e0 = last[S - 1];
            const float e1 = S >= 2 ? last[S - 2] : -INFINITY;
            nll[b] = -lse2(e0, e1);
        }
    } else {
        for (int s0 = threadIdx.x; s0 < S; s0 += 256) {
            const float v = (s0 == S - 1 || s0 == S - 2) ? LP(Tb - 1, s0, label(s0)) : -INFINITY;
            st[((Tb - 1) & 1) * Smax + s0] = v; out[(long)(Tb - 1) * Smax + s0] = v;
        }
        __syncthreads();
        for (int t = Tb - 2; t >= 0; --t) {
            const float* prev = st + ((t + 1) & 1) * Smax;
            float* cur = st + (t & 1) * Smax;
            for (int s0 = threadIdx.x; s0 < S; s0 += 256) {
                const long c = label(s0);
                const float b0 = prev[s0];
                const float b1 = s0 + 1 < S ? prev[s0 + 1] : -INFINITY;
                const float b2 = (s0 + 2 < S && c != blank && label(s0 + 2) != c) ? prev[s0 + 2] : -INFINITY;
                const float v = lse3(b0, b1, b2) + LP(t, s0, c);
                cur[s0] = v; out[(long)t * Smax + s0] = v;
            }
            __syncthreads();
        }
    }
}

// loss_ctc = mean_b( zero_inf(nll[b]) / max(tg_len[b], 1) )   (reduction='mean')
// out4 = {ctc, ce, number of CE targets kept, ctc + ce}: the cross-entropy pair was left in `ce` by ce_reduce_kernel
__global__ void ctc_reduce_kernel(const float* __restrict__ nll, const int64_t* __restrict__ tg_len, int B,
                                  const float* __restrict__ ce, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = nll[b];
            if (v != INFINITY && v == v) s += v / (float)max((long)tg_len[b], 1L);
        }
        const float ctc = s / (float)B;
        out[0] = ctc; out[1] = ce[0]; out[2] = ce[1]; out[3] = ctc + ce[0];
    }
}

// d(w_ctc*ctc + w_ce*ce)/d(logits) for row (b,t): dense part in one sweep.
//   CTC: g_ctc * (p_v - occ_tv)              for t < in_len[b], feasible b
//   CE : g_ce  * (p_v - eps/V - (1-eps)[v==target])   for t < t_ce, target kept
// The sparse CTC occupancy term is subtracted afterwards by ctc_grad_sparse.
template <typename TG>
__global__ __launch_bounds__(256) void head_grad_dense_kernel(const float* __restrict__ x, long ld, int V,
                                                              const float* __restrict__ lse, const int64_t* __restrict__ targets,
                                                              long ldt, const int64_t* __restrict__ in_len,
                                                              const int64_t* __restrict__ tg_len, const float* __restrict__ nll,
                                                              const float* __restrict__ ce_out, int B, int T, int t_ce, float eps,
                                                              int ignore_index, float w_ctc, float w_ce,
                                                              const float* __restrict__ up_ctc, const float* __restrict__ up_ce,
                                                              TG* __restrict__ dx, long lddx) {
    const long row = blockIdx.x;
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const float upc = up_ctc ? up_ctc[0] : 1.f, upe = up_ce ? up_ce[0] : 1.f;
    float gctc = 0.f;
    if (t < in_len[b] && nll[b] != INFINITY && nll[b] == nll[b])
        gctc = upc * w_ctc / ((float)B * (float)max((long)tg_len[b], 1L));
    float gce = 0.f; long tg = -1;
    if (t < t_ce) {
        tg = targets[(long)b * ldt + t];
        if (tg != ignore_index && ce_out[1] > 0.f) gce = upe * w_ce / ce_out[1]; else tg = -1;
    }
    const float* xr = x + row * ld;
    TG* dr = dx + row * lddx;
    const float L = lse[row];
    const float gs = gctc + gce, c0 = gce * eps / (float)V, c1 = gce * (1.f - eps);
    const bool vec = (ld % 4 == 0) && (lddx % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0);
    const int V4 = vec ? (V >> 2) : 0;
    for (int i = threadIdx.x; i < V4; i += 256) {               // 16-byte loads, 8- or 16-byte stores
        const f32x4 a = *(const f32x4*)(xr + 4 * i);
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = gs * __expf(a[e] - L) - c0;
            if (4 * i + e == tg) g[e] -= c1;
        }
        if constexpr (sizeof(TG) == 4) *(f32x4*)(dr + 4 * i) = (f32x4){g[0], g[1], g[2], g[3]};
        else *(bf16x4*)(dr + 4 * i) = (bf16x4){(bf16_t)g[0], (bf16_t)g[1], (bf16_t)g[2], (bf16_t)g[3]};
    }
    for (int v = 4 * V4 + threadIdx.x; v < V; v += 256) {
        float g = gs * __expf(xr[v] - L) - c0;
        if (v == tg) g -= c1;
        dr[v] = from_f32<TG>(g);
    }
    for (long v = V + threadIdx.x; v < lddx; v += 256) dr[v] = from_f32<TG>(0.f);      // the row's padding columns
}

// Sparse CTC term: dx[b,t,l'_s] -= g_ctc * exp(alpha + beta + nll - lp).  Several
// states can carry the same label, so one thread per (b,t) walks the states
// serially (S <= 2*448+1) - no atomics, deterministic.
template <typename TG>
__global__ __launch_bounds__(64) void ctc_grad_sparse_kernel(const float* __restrict__ x, long ld, const float* __restrict__ lse,
                                                            const int64_t* __restrict__ targets, long ldt,
                                                            const int64_t* __restrict__ in_len, const int64_t* __restrict__ tg_len,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            const float* __restrict__ nll, int B, int T, int Smax, int blank,
                                                            float w_ctc, const float* __restrict__ up_ctc, TG* __restrict__ dx,
                                                            long lddx) {
    const long row = (long)blockIdx.x * 64 + threadIdx.x;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    if (t >= in_len[b] || t >= T) return;
    const float nl = nll[b];
    if (nl == INFINITY || nl != nl) return;
    const int L = (int)tg_len[b], S = 2 * L + 1;
    const float up = up_ctc ? up_ctc[0] : 1.f;
    const float g = up * w_ctc / ((float)B * (float)max((long)L, 1L));
    const int64_t* tg = targets + (long)b * ldt;
    const float* al = alpha + ((long)b * T + t) * Smax;
    const float* be = beta + ((long)b * T + t) * Smax;
    const float Lr = lse[row];
    TG* dr = dx + row * lddx;
    for (int s = 0; s < S; ++s) {
        const long c = (s & 1) ? tg[s >> 1] : (long)blank;
        const float ab = al[s] + be[s];
        if (ab == -INFINITY) continue;
        const float lpv = x[row * ld + c] - Lr;
        const float occ = __expf(ab + nl - lpv);
        dr[c] = from_f32<TG>(to_f32<TG>(dr[c]) - g * occ);
    }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld, int V, int64_t* __restrict__ out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* xr = x + (long)blockIdx.x * ld;
    float best = -INFINITY; int idx = 0;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float a = xr[v];
        if (a > best || (a != a && best == best)) { best = a; idx = v; }   // first max wins; NaN is a max, as in torch
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float a = bv[threadIdx.x + o]; const int ia = bi[threadIdx.x + o];
            const float c = bv[threadIdx.x]; const int ic = bi[threadIdx.x];
            const bool a_nan = a != a, c_nan = c != c;
            const bool take = (a_nan && !c_nan) || (a_nan == c_nan && (a > c || (a == c && ia < ic)) ) ||
                              (a_nan && c_nan && ia < ic);
            if (take) { bv[threadIdx.x] = a; bi[threadIdx.x] = ia; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}

// --------------------------------------------------------------- host ------
// Workspace layout (floats): lse[rows] | sumx[rows] | nll[B] | ce[2] | alpha[B*T*Smax] | beta[B*T*Smax]
extern "C" int64_t avsr_head_loss_workspace_floats(int64_t B, int64_t T, int64_t max_target_len) {
    const int64_t Smax = 2 * max_target_len + 1;
    return 2 * B * T + B + 2 + 2 * B * T * Smax;
}

extern "C" int avsr_head_loss_fwd(avsr_stream_t stream, const float* logits, int64_t ld, int64_t B, int64_t T, int64_t V,
                                  const int64_t* targets, int64_t ldt, int64_t max_target_len, const int64_t* input_lengths,
                                  const int64_t* target_lengths, float label_smoothing, int64_t ignore_index,
                                  int64_t blank, float* ws, float* out4, const float* row_part, int64_t n_parts) {
    if (!logits || !targets || !input_lengths || !target_lengths || !ws || !out4) return AVSR_EINVAL;
    if (B <= 0 || T <= 0 || V <= 0 || max_target_len <= 0 || max_target_len > ldt) return AVSR_EINVAL;
    if (row_part && (n_parts != (V + 127) / 128 || (uintptr_t)row_part % 16)) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = B * T, Smax = 2 * max_target_len + 1;
    float* lse = ws; float* sumx = lse + rows; float* nll = sumx + rows; float* ce = nll + B;
    float* alpha = ce + 2; float* beta = alpha + rows * Smax;
    const int t_ce = (int)(T < max_target_len ? T : max_target_len);
    if (row_part)
        hipLaunchKernelGGL(row_lse_merge_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, row_part, (int)n_parts,
                           (int)rows, lse, sumx);
    else
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)rows), dim3(1024), 0, s, logits, (long)ld, (int)V, lse, sumx);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, logits, (long)ld, (int)V, lse, sumx, targets, (long)ldt,
                       (int)B, (int)T, t_ce, label_smoothing, (int)ignore_index, ce);
    const size_t tab_bytes = (size_t)T * Smax * sizeof(float), st_bytes = 2 * (size_t)Smax * sizeof(float);
    const int in_lds = tab_bytes + st_bytes <= 60 * 1024;
    hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3((unsigned)(2 * B)), dim3(256), st_bytes + (in_lds ? tab_bytes : 0), s, logits,
                       (long)ld, lse, targets, (long)ldt, input_lengths, target_lengths, alpha, beta, nll, (int)B, (int)T,
                       (int)Smax, (int)blank, in_lds);
    hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(64), 0, s, nll, target_lengths, (int)B, ce, out4);
    return avsr_launch_status();
}

extern "C" int avsr_head_loss_bwd(avsr_stream_t stream, int grad_dtype, const float* logits, int64_t ld, int64_t B,
                                  int64_t T, int64_t V, const int64_t* targets, int64_t ldt, int64_t max_target_len,
                                  const int64_t* input_lengths, const int64_t* target_lengths, float label_smoothing,
                                  int64_t ignore_index, int64_t blank, const float* ws, float w_ctc, float w_ce,
                                  const float* up_ctc, const float* up_ce, void* dlogits, int64_t lddx) {
    if (!logits || !targets || !input_lengths || !target_lengths || !ws || !dlogits) return AVSR_EINVAL;
    if (B <= 0 || T <= 0 || V <= 0 || max_target_len <= 0 || max_target_len > ldt) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = B * T, Smax = 2 * max_target_len + 1;
    const float* lse = ws; const float* nll = lse + 2 * rows; const float* ce = nll + B;
    const float* alpha = ce + 2; const float* beta = alpha + rows * Smax;
    const int t_ce = (int)(T < max_target_len ? T : max_target_len);
#define RUN(TG)                                                                                                          \
    hipLaunchKernelGGL(head_grad_dense_kernel<TG>, dim3((unsigned)rows), dim3(256), 0, s, logits, (long)ld, (int)V, lse, \
                       targets, (long)ldt, input_lengths, target_lengths, nll, ce, (int)B, (int)T, t_ce, label_smoothing,\
                       (int)ignore_index, w_ctc, w_ce, up_ctc, up_ce, (TG*)dlogits, (long)lddx);                              \
    hipLaunchKernelGGL(ctc_grad_sparse_kernel<TG>, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, logits, (long)ld, \
                       lse, targets, (long)ldt, input_lengths, target_lengths, alpha, beta, nll, (int)B, (int)T,         \
                       (int)Smax, (int)blank, w_ctc, up_ctc, (TG*)dlogits, (long)lddx)
    if (grad_dtype == AVSR_F32) { RUN(float); }
    else if (grad_dtype == AVSR_BF16) { RUN(bf16_t); }
    else return AVSR_EINVAL;
#undef RUN
    return avsr_launch_status();
}

extern "C" int avsr_argmax_rows(avsr_stream_t stream, const float* x, int64_t ld, int64_t rows, int64_t V, int64_t* out) {
    if (!x || !out || rows <= 0 || V <= 0) return AVSR_EINVAL;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (long)ld, (int)V, out);
    return avsr_launch_status();
}
