// NT GEMM with fused epilogue and an implicit-GEMM (NHWC conv) A-operand
// loader.  Two kernels:
//   * bf16 inputs: 128x128x32 (or 128x64x32) tile, 4 waves, v_mfma_f32_16x16x32_bf16,
//     operands DMA'd global->LDS into a 4-stage ring with a counted vmcnt.
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
    int stat_parts;                         // rows of the csum / csq slabs (>= tiles_m; rows past tiles_m are zero-filled)
    int H, W, Cc, KH, KW, stride, pad_h, pad_w, OH, OW;
    int tiles_m, tiles_n, vec_ok;
    // extended epilogue (dense products only): second output, dropout, multiply by gelu'(aux)
    void* pre; long ldp; const void* aux; long ldaux; float p_drop; unsigned long long seed; int ext;
    const unsigned long long* seed_delta;
    float* row_part;                        // [tiles of 128 columns][M][4]: per-row {max, sum exp, sum} of a tile (EXT)
    // LayerNorm folded into the products on either side of it (EXT == 2 reads, EXT == 3 writes)
    const float* ln_in; int ln_parts; float ln_eps;     // [ln_parts][M][2]: per-row {sum, sum of squares} of A's row, by 128-column tile
    float* ln_out; void* copy; long ldcopy;             // the same for the stored rows of C, and a bf16 copy of them
    void* copy_lo;                                      // bf16(value - copy): copy + copy_lo carry the fp32 row to 2^-18
    const void* res_hi; const void* res_lo; long ldrp;  // a residual kept as such a pair of bf16 planes
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
    if (p.ext) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nb + r;
            if (n >= p.N) continue;
            if (p.pre) ((OutT*)p.pre)[cbase + (long)m * p.ldp + n] = from_f32<OutT>(pre[r]);
            if (p.aux) v[r] *= gelu_erf_grad(to_f32<OutT>(((const OutT*)p.aux)[(long)m * p.ldaux + n]));
            if (p.p_drop > 0.f)
                v[r] = avsr_keep(avsr_run_seed(p.seed, (const uint64_t*)p.seed_delta), (uint64_t)((long)m * p.N + n), p.p_drop)
                           ? v[r] * (1.0f / (1.0f - p.p_drop)) : 0.f;
        }
    }
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
// sub-tile, stored (no atomics, deterministic) into row `part` of the [parts, N] slabs.
__device__ __forceinline__ void stats_quad(const GemmP& p, f32x4 s, f32x4 s2, int nb, int lane, int part) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = s[r], b = s2[r];
        a = row16_sum(a); b = row16_sum(b);
        if ((lane & 15) == 0 && nb + r < p.N) {
            p.csum[(long)part * p.N + nb + r] = a;
            p.csq[(long)part * p.N + nb + r] = b;
        }
    }
}

// Epilogue of the bf16 kernel: the accumulators go through an LDS tile (two passes of 64 rows)
// so that every thread owns 8 consecutive columns of a row: scale / bias are loaded once per
// thread, residual reads and output stores are 16-byte (bf16) or 2 x 16-byte (fp32) and a row
// segment of the tile is written by consecutive lanes.  Column statistics are reduced per block
// and stored into row `tm` of the [tiles_m, N] slabs.
// The activation is a template parameter of the body (a run-time `act` test inside the per-element loop
// costs several scalar branches per element) and the row loop is NOT unrolled: the epilogue runs once per
// tile, so its instructions are fetched cold, and straight-line code four times as long measured 17-30 %
// slower on the K = 768 encoder GEMMs (instruction-cache misses, not arithmetic).
// RESPF (products with a residual operand): the residual rows of a pass are requested before the pass's LDS staging and
// used from registers - in the rolled row loop every iteration would wait for its own residual load, eight to ten exposed
// memory latencies per tile; the row loop of these instantiations is unrolled instead (2 or 4 iterations).
// EXT: 0 = plain, 1 = extended (second output, dropout, GELU gradient, row log-sum-exp partials),
// 2 = the A operand is the un-normalised input of a LayerNorm whose gain is folded into B: a row's mean and 1 / std come
//     from per-tile partial sums (lnrow: {rstd, -rstd * mean} per tile row, staged in LDS at kernel entry) and
//     out = rstd * acc - rstd * mean * colsum(B)[n] + bias'[n]  (the `scale` slot holds colsum(B), `bias` holds bias'),
// 3 = fp32 output with residual that is the input of the NEXT LayerNorm: also stores a bf16 copy of the rows and their
//     per-tile {sum, sum of squares} (what an EXT == 2 product reads).
// WN: waves side by side in N (2: four waves per workgroup; 4: eight waves, each half as wide - for grids of about one
// workgroup per CU, where four waves leave the matrix pipe idle while they wait for their fragments).
template <int BN_, typename OutT, int ACT, bool CONV, int MI, int EXT = 0, bool RESPF = false, int WN = 2>
__device__ __forceinline__ void epilogue_body(const GemmP& p, f32x4 (&acc)[MI][BN_ / (16 * WN)], char* lds, const float* sb,
                                              int m0, int n0, int tm, long cbase, int lane, int wave,
                                              const float* lnrow = nullptr) {
    constexpr int NW = 2 * WN, NT = 64 * NW;
    constexpr int NJ = BN_ / (16 * WN), LDT = BN_ + 4, CG = BN_ / 8, RS = NT / CG;
    // a pass moves SUBS 16-row sub-tiles of each wave (2 x SUBS x 16 tile rows) through the LDS tile
    constexpr int SUBS = (MI % 2 == 0) ? 2 : 1, NPASS = MI / SUBS, PROWS = 2 * SUBS * 16;
    float* Ct = (float*)lds;
    float* red = Ct + 64 * LDT;                 // [NW][2][BN_] per-wave column partials
    const int wm = wave / WN, wn = wave % WN;
    const int t = threadIdx.x;
    const int cg = t % CG, r0 = t / CG;
    const int n = n0 + cg * 8;
    const float g = p.gate ? tanhf(*p.gate) : 1.0f;
    float sc[8], bi[8], s1[8], s2[8];
    {                                           // alpha * scale and bias of this thread's 8 columns, staged at kernel entry
        const f32x4 a0 = *(const f32x4*)(sb + cg * 8), a1 = *(const f32x4*)(sb + cg * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(sb + BN_ + cg * 8), b1 = *(const f32x4*)(sb + BN_ + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; bi[e] = b0[e]; bi[4 + e] = b1[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    }
    const bool full = p.vec_ok && n + 7 < p.N;
    // 16-byte accesses to the second output and the saved GELU inputs: the host checked their alignment
    const bool ext_vec = EXT == 1 && p.ext == 2;
    const bool has_res = p.res != nullptr;
    const bool stats = p.csum != nullptr;
    uint64_t run_seed = 0;
    if constexpr (EXT == 1) run_seed = avsr_run_seed(p.seed, (const uint64_t*)p.seed_delta);
    constexpr int NIT = (PROWS + RS - 1) / RS, UNR = RESPF ? NIT : 1;
    constexpr int RW = sizeof(OutT) == 4 ? 2 : 1;       // 16-byte words of residual per thread and row
    f32x4 rpf[RESPF ? NIT * RW : 1];
#pragma unroll
    for (int h = 0; h < NPASS; ++h) {
        if constexpr (RESPF) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int lr = r0 + it * RS;
                const int m = m0 + (lr / (SUBS * 16)) * (MI * 16) + h * (SUBS * 16) + (lr % (SUBS * 16));
                if ((PROWS % RS == 0 || lr < PROWS) && m < p.M && full) {
                    if (EXT == 3 && !has_res) {         // the residual as two bf16 planes
                        rpf[it * RW] = *(const f32x4*)((const bf16_t*)p.res_hi + (long)m * p.ldrp + n);
                        rpf[it * RW + RW - 1] = *(const f32x4*)((const bf16_t*)p.res_lo + (long)m * p.ldrp + n);
                    } else {
                        const OutT* rp = (const OutT*)p.res + (long)(p.res_period > 0 ? m % p.res_period : m) * p.ldr + n;
#pragma unroll
                        for (int w = 0; w < RW; ++w) rpf[it * RW + w] = *(const f32x4*)((const char*)rp + 16 * w);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < SUBS; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *(f32x4*)&Ct[(wm * SUBS * 16 + ii * 16 + (lane & 15)) * LDT + (wn * NJ + j) * 16 + (lane >> 4) * 4] =
                    acc[SUBS * h + ii][j];
        __syncthreads();
#pragma unroll UNR
        for (int it = 0; it < NIT; ++it) {
            const int lr = r0 + it * RS;
            if (PROWS % RS != 0 && lr >= PROWS) continue;
            const int m = m0 + (lr / (SUBS * 16)) * (MI * 16) + h * (SUBS * 16) + (lr % (SUBS * 16));
            if (m >= p.M) continue;
            const f32x4 c0 = *(const f32x4*)&Ct[lr * LDT + cg * 8], c1 = *(const f32x4*)&Ct[lr * LDT + cg * 8 + 4];
            float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            float pv[8], av[8];                 // EXT: the second output's values / the saved GELU inputs
            uint32_t draws[8];                  // EXT: the eight elements' dropout draws (two generator calls)
            bool draws_ok = false;
            if constexpr (EXT == 1) {
                const uint64_t i0 = (uint64_t)((long)m * p.N + n);
                if (p.p_drop > 0.f && (i0 & 3) == 0) {
                    avsr_rand4(run_seed, i0 >> 2, draws);
                    avsr_rand4(run_seed, (i0 >> 2) + 1, draws + 4);
                    draws_ok = true;
                }
            }
            if constexpr (EXT == 1) {
                if (p.aux) {
                    const OutT* ap = (const OutT*)p.aux + (long)m * p.ldaux + n;
                    if (full && ext_vec) {
                        if constexpr (sizeof(OutT) == 4) {
                            const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { av[e] = a0[e]; av[4 + e] = a1[e]; }
                        } else {
                            const bf16x8 a0 = *(const bf16x8*)ap;
#pragma unroll
                            for (int e = 0; e < 8; ++e) av[e] = (float)a0[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) av[e] = n + e < p.N ? to_f32<OutT>(ap[e]) : 0.f;
                    }
                }
            }
            float ln_r = 0.f, ln_nrm = 0.f;     // EXT == 2: this row's 1 / std and -mean / std
            if constexpr (EXT == 2) {
                const float2 rs = ((const float2*)lnrow)[m - m0];
                ln_r = rs.x; ln_nrm = rs.y;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x;
                if constexpr (EXT == 2) x = fmaf(v[e], ln_r, fmaf(ln_nrm, sc[e], bi[e]));
                else x = fmaf(v[e], sc[e], bi[e]);
                if (stats) { s1[e] += x; s2[e] = fmaf(x, x, s2[e]); }
                if constexpr (EXT == 1) pv[e] = x;   // second output: the value before the activation / gate
                if (ACT == AVSR_ACT_GELU) x = gelu_erf_fast(x);
                else if (ACT == AVSR_ACT_RELU) x = fmaxf(x, 0.f);
                if constexpr (EXT == 1) {
                    if (p.aux) x *= gelu_erf_grad_fast(av[e]);
                    if (p.p_drop > 0.f) {
                        const bool keep = draws_ok ? avsr_bits_keep(draws[e], p.p_drop)
                                                   : avsr_keep(run_seed, (uint64_t)((long)m * p.N + n + e), p.p_drop);
                        x = keep ? x * (1.0f / (1.0f - p.p_drop)) : 0.f;
                    }
                }
                v[e] = x;
            }
            if constexpr (EXT == 1 && BN_ == 128) {
                if (p.row_part) {               // log-sum-exp partials of this row over the tile's valid columns
                    float mx = -INFINITY, sx = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) { mx = fmaxf(mx, pv[e]); sx += pv[e]; }
                    mx = row16_max(mx);         // the 16 lanes of a row hold its 128 columns
                    float se = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) se += __expf(pv[e] - mx);
                    se = row16_sum(se);
                    sx = row16_sum(sx);
                    if (cg == 0)
                        *(f32x4*)(p.row_part + ((long)(n0 / BN_) * p.M + m) * 4) = (f32x4){mx, se, sx, 0.f};
                }
            }
            if constexpr (EXT == 1) {
                if (p.pre) {
                    OutT* pp = (OutT*)p.pre + cbase + (long)m * p.ldp + n;
                    if (full && ext_vec) {
                        if constexpr (sizeof(OutT) == 4) {
                            *(f32x4*)pp = (f32x4){pv[0], pv[1], pv[2], pv[3]};
                            *(f32x4*)(pp + 4) = (f32x4){pv[4], pv[5], pv[6], pv[7]};
                        } else {
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)pv[e];
                            *(bf16x8*)pp = o;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (n + e < p.N) pp[e] = from_f32<OutT>(pv[e]);
                    }
                }
            }
            if constexpr (EXT == 1) {
                if (p.C == nullptr) continue;   // statistics-only pass: the column sums are all this launch produces
            }
            OutT* cp = (OutT*)p.C + cbase + (long)m * p.ldc + n;
            const OutT* rp = has_res ? (const OutT*)p.res + (long)(p.res_period > 0 ? m % p.res_period : m) * p.ldr + n : nullptr;
            if (full) {
                if (has_res) {
                    if constexpr (sizeof(OutT) == 4) {
                        f32x4 ra, rb;
                        if constexpr (RESPF) { ra = rpf[it * RW]; rb = rpf[it * RW + RW - 1]; }
                        else { ra = *(const f32x4*)rp; rb = *(const f32x4*)(rp + 4); }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = fmaf(g, v[e], ra[e]); v[4 + e] = fmaf(g, v[4 + e], rb[e]); }
                    } else {
                        bf16x8 rv;
                        if constexpr (RESPF) rv = __builtin_bit_cast(bf16x8, rpf[it * RW]);
                        else rv = *(const bf16x8*)rp;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaf(g, v[e], (float)rv[e]);
                    }
                } else if constexpr (EXT == 3) {
                    const bf16x8 hi = __builtin_bit_cast(bf16x8, rpf[it * RW]), lo = __builtin_bit_cast(bf16x8, rpf[it * RW + RW - 1]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)hi[e] + (float)lo[e];
                }
                if (ACT == AVSR_ACT_RELU_AFTER_RESIDUAL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (EXT == 3) {
                    if (p.C) {
                        *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
                        *(f32x4*)(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    }
                    if (p.copy) {               // the rows as the next product's operand (+ what the rounding dropped)
                        bf16x8 o, l;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { o[e] = (bf16_t)v[e]; l[e] = (bf16_t)(v[e] - (float)o[e]); }
                        *(bf16x8*)((bf16_t*)p.copy + (long)m * p.ldcopy + n) = o;
                        if (p.copy_lo) *(bf16x8*)((bf16_t*)p.copy_lo + (long)m * p.ldcopy + n) = l;
                    }
                } else if constexpr (sizeof(OutT) == 4) {
                    *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *(bf16x8*)cp = o;
                }
            } else if constexpr (EXT == 3) {    // (the host admits whole 8-column groups only: nothing of this row here)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e < p.N) {
                        float x = v[e];
                        if (has_res) x = fmaf(g, x, to_f32<OutT>(rp[e]));
                        if (ACT == AVSR_ACT_RELU_AFTER_RESIDUAL) x = fmaxf(x, 0.f);
                        cp[e] = from_f32<OutT>(x);
                    }
                }
            }
            if (EXT == 3 && BN_ == 128 && p.ln_out) {    // the 16 lanes of a row hold its 128 columns of this tile
                float sx = 0.f, sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { sx += v[e]; sq = fmaf(v[e], v[e], sq); }
                sx = row16_sum(sx); sq = row16_sum(sq);
                if (cg == 0) ((float2*)p.ln_out)[(long)(n0 / BN_) * p.M + m] = make_float2(sx, sq);
            }
        }
    }
    if (stats) {
        // lanes with equal (lane % CG) own the same columns: fold them, then the 4 waves through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (CG <= 8) { s1[e] = xor_add<8>(s1[e]); s2[e] = xor_add<8>(s2[e]); }
            s1[e] = xor_add<16>(s1[e]); s2[e] = xor_add<16>(s2[e]);
            s1[e] = xor_add<32>(s1[e]); s2[e] = xor_add<32>(s2[e]);
        }
        if (lane < CG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * BN_ + lane * 8 + e] = s1[e];
                red[(wave * 2 + 1) * BN_ + lane * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        if (t < BN_ && n0 + t < p.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[(w * 2 + 0) * BN_ + t]; b += red[(w * 2 + 1) * BN_ + t]; }
            p.csum[(long)tm * p.N + n0 + t] = a;
            p.csq[(long)tm * p.N + n0 + t] = b;
            // the slabs are sized for 128-row tiles: with taller tiles the last row of tiles clears the rows nobody owns
            if (tm == p.tiles_m - 1)
                for (int r = p.tiles_m; r < p.stat_parts; ++r) {
                    p.csum[(long)r * p.N + n0 + t] = 0.f;
                    p.csq[(long)r * p.N + n0 + t] = 0.f;
                }
        }
    }
}

template <int BN_, typename OutT, bool CONV, int MI, int EXT = 0, int WN = 2>
__device__ __forceinline__ void epilogue_tile(const GemmP& p, f32x4 (&acc)[MI][BN_ / (16 * WN)], char* lds, const float* sb,
                                              int m0, int n0, int tm, long cbase, int lane, int wave,
                                              const float* lnrow = nullptr) {
    if constexpr (EXT == 1) {                   // instantiations of their own: the plain kernels' code is untouched
        if (p.act == AVSR_ACT_GELU)
            epilogue_body<BN_, OutT, AVSR_ACT_GELU, CONV, MI, 1>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
        else
            epilogue_body<BN_, OutT, AVSR_ACT_NONE, CONV, MI, 1>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
        return;
    }
    if constexpr (EXT == 2) {
        if (p.act == AVSR_ACT_GELU)
            epilogue_body<BN_, OutT, AVSR_ACT_GELU, CONV, MI, 2>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave, lnrow);
        else
            epilogue_body<BN_, OutT, AVSR_ACT_NONE, CONV, MI, 2>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave, lnrow);
        return;
    }
    if constexpr (EXT == 3) {
        epilogue_body<BN_, OutT, AVSR_ACT_NONE, CONV, MI, 3, true>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
        return;
    }
#ifndef G_NO_RESPF
    if (p.res != nullptr) {                     // block-uniform: residual rows requested ahead of each pass
        if (p.act == AVSR_ACT_NONE) {
            epilogue_body<BN_, OutT, AVSR_ACT_NONE, CONV, MI, 0, true, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
            return;
        }
        if (p.act == AVSR_ACT_RELU_AFTER_RESIDUAL) {
            epilogue_body<BN_, OutT, AVSR_ACT_RELU_AFTER_RESIDUAL, CONV, MI, 0, true, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
            return;
        }
    }
#endif
    switch (p.act) {                            // block-uniform
        case AVSR_ACT_GELU: epilogue_body<BN_, OutT, AVSR_ACT_GELU, CONV, MI, 0, false, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave); break;
        case AVSR_ACT_RELU: epilogue_body<BN_, OutT, AVSR_ACT_RELU, CONV, MI, 0, false, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave); break;
        case AVSR_ACT_RELU_AFTER_RESIDUAL:
            epilogue_body<BN_, OutT, AVSR_ACT_RELU_AFTER_RESIDUAL, CONV, MI, 0, false, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave); break;
        default: epilogue_body<BN_, OutT, AVSR_ACT_NONE, CONV, MI, 0, false, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave); break;
    }
}

// ---------------------------------------------------------------- bf16 ----
// 128 x BN x 64 tiles, 4 waves, v_mfma_f32_16x16x32_bf16.  Operand tiles go
// global -> LDS directly (global_load_lds_dwordx4): one wave instruction moves
// 8 rows x 128 B, i.e. whole cache lines, into a ring of G_NST stages; with more
// than one stage the loads of the following K-steps stay in flight across the
// raw s_barrier behind a counted s_waitcnt vmcnt.
//
// LDS image: row-major [rows][128 B]; the 16-byte chunk c of row r sits at chunk
// position c ^ ((r >> 1) & 7).  The DMA writes lane-linear, so the XOR is applied
// to the per-lane SOURCE address (lane i of a piece fetches chunk (i&7) ^ swz of
// row i>>3) and again on the ds_read_b128 side: conflict-free for the 16-row
// fragment reads of the 16x16x32 MFMA.
// Out-of-range rows, conv padding taps and the K tail read a 16-byte zero page.
#define G_BM 128
#define G_BK 64
// Two instantiations, picked per launch by the number of tiles (measured, interleaved A/B on MI355X):
//   NST = 1: one 32-KiB stage, two barriers per K-step, 4 workgroups per CU - latency is hidden by
//            the other resident workgroups; fastest whenever the grid can fill the chip (about 500 tiles or more).
//   NST = 2: double-buffered, 2 workgroups per CU - better when there are too few tiles for that.
//   NST = 4: grids of at most one workgroup per CU (the small GEMMs of the trainable tail): nothing else hides
//            the ~1300-cycle DMA latency there, so three K-steps stay in flight (0.55 -> 0.25 us per K-step).
#ifndef G_T96_MIN_TILES
#define G_T96_MIN_TILES 1000
#endif
#ifndef G_FILL_RATIO
#define G_FILL_RATIO 1.12
#endif
#ifndef G_NST_SWITCH_TILES
#define G_NST_SWITCH_TILES 1000
#endif
#define G_NUM_CU 256
#define G_LN_MAXP 10                        // 128-column tiles of a folded LayerNorm's rows (d_model <= 1280)
#ifndef G_BM160
#define G_BM160 1
#endif
#ifndef G_BM160_NST
#define G_BM160_NST 2
#endif
#ifndef G_ONE160_NST
#define G_ONE160_NST 3
#endif
#ifndef G_SMALL_TILES
#define G_SMALL_TILES 128
#endif
#ifndef G_DEEP_TILES
#define G_DEEP_TILES 256
#endif

__device__ __attribute__((aligned(16))) unsigned int avsr_zero_page[4] = {0u, 0u, 0u, 0u};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "unsupported wait count");
}

// Build with -DGEMM_STAMPS (scripts/build_variants.sh) to record 100-MHz timestamps of workgroup 0's phases.
#ifdef GEMM_STAMPS
__device__ unsigned long long avsr_stamps[16];
extern "C" int avsr_debug_read_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(avsr_stamps), sizeof(avsr_stamps)) == hipSuccess ? 0 : -3;
}
#define STAMP(i) do { stamp[i] = wall_clock64(); cyc[i] = clock64(); } while (0)
#else
#define STAMP(i)
#endif

template <int BN_, int G_NST, int BM_ = G_BM, int EXT = 0, int WN = 2> constexpr int gemm_lds_bytes() {
    constexpr int stage = (BM_ / 8 + BN_ / 8) * 1024;
    constexpr int epi = (64 * (BN_ + 4) + 4 * WN * BN_) * 4;
    // + the tile's scale / bias columns (+ the tile rows' LayerNorm factors)
    return (G_NST * stage > epi ? G_NST * stage : epi) + 2 * BN_ * 4 + (EXT == 2 ? BM_ * 8 : 0);
}

// CONV: 0 = dense operands, 1 = implicit-GEMM convolution with per-lane (tap, channel) cursors (any channel count),
// 2 = convolution whose channel count is a multiple of the K-step: a K-step lies inside ONE filter tap, so the tap is
// a scalar, a lane keeps one byte offset per piece (its pixel) and a bit per tap (inside the image or padding), and
// the K loop carries no per-lane index arithmetic beyond one 64-bit add and the zero-page select.
template <int BN_, int G_NST, int CONV, typename OutT, int BM_ = G_BM, int EXT = 0, int WN = 2>
// Single-stage instantiations: dense operands at three workgroups per CU (168 registers, no spills) - four (128
// registers) is faster for some products run alone, but with two or three streams sharing the chip the step is
// 2.5 % faster at three; the convolution form stays at four (no difference in the step, faster alone).
#ifndef G_OCC1
#define G_OCC1 3
#endif
#ifndef G_OCC1C
#define G_OCC1C 4
#endif
__global__ __launch_bounds__(128 * WN, (G_NST == 1 && BM_ <= 128 ? (CONV ? G_OCC1C : G_OCC1) : G_NST == 1 && BM_ == 160 ? 3 : 2)) void gemm_bf16_kernel(GemmP p) {
    static_assert(WN == 2 || (EXT == 0 && CONV != 1), "the eight-wave form exists for the plain and scalar-tap kernels");
    constexpr int NW = 2 * WN;                                // waves per workgroup
    static_assert(BM_ % 32 == 0 && (BM_ == G_BM || CONV != 1), "row tile: 4 x whole pieces per wave; the cursor form assumes 128 rows");
    constexpr int MI = BM_ / 32;                              // 16-row sub-tiles per wave (two waves stacked in M)
    constexpr int APIECES = BM_ / 8, BPIECES = BN_ / 8;       // 1-KiB pieces (8 rows x 128 B)
    constexpr int STAGE_BYTES = (APIECES + BPIECES) * 1024;
    constexpr int NJ = BN_ / (16 * WN);     // 16-column sub-tiles per wave
    // pieces each wave loads per K-step; where the A pieces do not divide over the waves the last waves load a piece a
    // second time (same bytes to the same place), so that every wave issues the same number of loads
    constexpr int NAL = (APIECES + NW - 1) / NW, NBL = BPIECES / NW;
    static_assert(BPIECES % NW == 0 && NBL % 2 == 0, "B pieces per wave: whole and even (the swizzle follows q & 1)");
    constexpr int LPT = NAL + NBL;          // DMA instructions per wave per K-step
    extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef GEMM_STAMPS
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int bz = blockIdx.z;
    const bf16_t* A = (const bf16_t*)p.A + (long)bz * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long)bz * p.sB;
    const long cbase = (long)bz * p.sC;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // alpha * scale[n] and bias[n] of the tile's columns go to LDS now, so the epilogue does not wait on them
    float* sb = (float*)(lds + gemm_lds_bytes<BN_, G_NST, BM_, 0, WN>() - 2 * BN_ * 4);
    if (t < BN_) {
        const int nc = min(n0 + t, p.N - 1);
        sb[t] = p.scale ? p.scale[nc] * p.alpha : p.alpha;
        sb[BN_ + t] = p.bias ? p.bias[nc] : 0.f;
    }
    float* lnrow = sb + 2 * BN_;
    // EXT == 2: the partial sums of the tile's rows are requested now, all at once, and folded after the K loop
    float2 lnp[EXT == 2 ? G_LN_MAXP : 1];
    if constexpr (EXT == 2) {
        const int m = min(m0 + min(t, BM_ - 1), p.M - 1);
#pragma unroll
        for (int i = 0; i < G_LN_MAXP; ++i)
            lnp[i] = i < p.ln_parts ? ((const float2*)p.ln_in)[(long)i * p.M + m] : make_float2(0.f, 0.f);
    }

    // ---- DMA sources ----------------------------------------------------------
    // Piece q of this wave covers rows 8*(NAL*wave+q) .. +7; lane i fetches row i>>3, source chunk
    // (i&7) ^ swz with swz = ((row >> 1) & 7) = 4*(piece&1) + (i>>4): two distinct k offsets per lane.
    // Dense operands: a uniform tile base (scalar registers) plus a 32-bit per-lane byte offset that never
    // changes - a K-step only moves the scalar base, so the loop carries no 64-bit vector address arithmetic.
    // Rows past M / N are clamped onto the last valid row (their products are never stored).  Only the K tail
    // of the last step, and the conv taps that fall into the padding, read a 16-byte zero page; there the
    // address is selected with integer arithmetic (zero + ((real - zero) & mask)): a ?: on the pointer lets
    // the compiler branch around the DMA, and every wave must issue exactly LPT DMA instructions per step
    // for the counted wait to mean anything.
    const uintptr_t zaddr = (uintptr_t)avsr_zero_page;
    const int kofs[2] = {(((lane & 7) ^ (lane >> 4)) * 8), (((lane & 7) ^ (4 + (lane >> 4))) * 8)};
    const char* tileA = (const char*)(A + (long)m0 * p.lda);
    const char* tileB = (const char*)(B + (long)n0 * p.ldb);
    unsigned offA[NAL], offB[NBL];
    ConvRow crow[NAL];
    int kh[2] = {0, 0}, kw[2] = {0, 0}, cc[2] = {0, 0};
    int pixoff[NAL];                        // CONV == 2: byte offset of the lane's pixel (+ swizzled chunk) from A
    unsigned tapmask[NAL];                  // CONV == 2: bit t set <=> filter tap t of that pixel is inside the image
    int s_tap = 0, s_kh = 0, s_kw = 0, s_c0 = 0;      // CONV == 2: the (uniform) tap / channel cursor of the next step
    auto apiece = [&](int q) {               // the A piece this wave's q-th load moves
        if constexpr (APIECES % NW == 0) return NAL * wave + q;
        else return min(NAL * wave + q, APIECES - 1);
    };
#pragma unroll
    for (int q = 0; q < NAL; ++q) {
        const int r = apiece(q) * 8 + (lane >> 3);
        if constexpr (CONV == 2) {
            const ConvRow cr = conv_row(p, m0 + r);
            const int ko = (apiece(q) & 1) ? kofs[1] : kofs[0];
            pixoff[q] = ((cr.base + cr.ih0 * p.W + cr.iw0) * p.Cc + ko) * (int)sizeof(bf16_t);
            unsigned bits = 0;
            for (int t2 = 0, a = 0; a < p.KH; ++a)
                for (int b2 = 0; b2 < p.KW; ++b2, ++t2) {
                    const int ih = cr.ih0 + a, iw = cr.iw0 + b2;
                    if (cr.ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) bits |= 1u << t2;
                }
            tapmask[q] = bits;
            offA[q] = 0;
        } else if constexpr (CONV == 1) { crow[q] = conv_row(p, m0 + r); offA[q] = 0; }
        else {
            const int ko = (apiece(q) & 1) ? kofs[1] : kofs[0];      // swizzle follows the piece's row parity
            offA[q] = (unsigned)(((long)min(r, p.M - 1 - m0) * p.lda + ko) * (long)sizeof(bf16_t));
        }
    }
    if constexpr (CONV == 1) {
        conv_col(p, kofs[0], kh[0], kw[0], cc[0]);
        conv_col(p, kofs[1], kh[1], kw[1], cc[1]);
    }
#pragma unroll
    for (int q = 0; q < NBL; ++q) {
        const int r = (NBL * wave + q) * 8 + (lane >> 3);
        offB[q] = (unsigned)(((long)min(r, p.N - 1 - n0) * p.ldb + kofs[q & 1]) * (long)sizeof(bf16_t));
    }
    const int nk = (p.K + G_BK - 1) / G_BK;
    const bool ktail = (p.K % G_BK) != 0;

    // Steps past the last one re-load the last real step into a stage nobody reads any more (keeps the wait
    // count constant).
    auto issue = [&](int kt) {              // DMA the operands of K-step kt into stage kt % G_NST
        char* st = lds + (kt % G_NST) * STAGE_BYTES;
        const int ks = min(kt, nk - 1);
        const bool tail = ktail && ks == nk - 1;                            // workgroup-uniform
        const long kbytes = (long)ks * (G_BK * (long)sizeof(bf16_t));
        if constexpr (CONV == 2) {
            const char* tb = (const char*)A + ((long)(s_kh * p.W + s_kw) * p.Cc + s_c0) * (long)sizeof(bf16_t);
#pragma unroll
            for (int q = 0; q < NAL; ++q) {
                const uintptr_t real = (uintptr_t)(tb + (long)pixoff[q]);
                const uintptr_t a = zaddr + ((real - zaddr) & (uintptr_t)(-(long)((tapmask[q] >> s_tap) & 1u)));
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(st + apiece(q) * 1024), 16, 0, 0);
            }
            if (kt < nk - 1) {              // the next step: 64 channels further, or the next tap
                s_c0 += G_BK;
                if (s_c0 >= p.Cc) { s_c0 = 0; ++s_tap; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
            }
        }
        if constexpr (CONV == 1) {
            uintptr_t kin[2] = {~(uintptr_t)0, ~(uintptr_t)0};
            if (ktail) {
                kin[0] = (uintptr_t)(-(long)(kt * G_BK + kofs[0] < p.K));
                kin[1] = (uintptr_t)(-(long)(kt * G_BK + kofs[1] < p.K));
            }
#pragma unroll
            for (int q = 0; q < NAL; ++q) {
                const int ih = crow[q].ih0 + kh[q & 1], iw = crow[q].iw0 + kw[q & 1];
                const bool ok = crow[q].ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                const uintptr_t real = (uintptr_t)(A + ((long)(crow[q].base + ih * p.W + iw)) * p.Cc + cc[q & 1]);
                const uintptr_t a = zaddr + ((real - zaddr) & kin[q & 1] & (uintptr_t)(-(long)ok));
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(st + apiece(q) * 1024), 16, 0, 0);
            }
            if (ktail || kt < nk - 1) {     // advance the two (tap, channel) cursors by one K-step
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    cc[e] += G_BK;
                    while (cc[e] >= p.Cc) { cc[e] -= p.Cc; if (++kw[e] == p.KW) { kw[e] = 0; ++kh[e]; } }
                }
            }
        }
        if (!tail) {
            const char* a0 = tileA + kbytes;
            const char* b0 = tileB + kbytes;
            if constexpr (!CONV) {
#pragma unroll
                for (int q = 0; q < NAL; ++q) dma_piece(a0, offA[q], st + apiece(q) * 1024);
            }
#pragma unroll
            for (int q = 0; q < NBL; ++q) dma_piece(b0, offB[q], st + (APIECES + NBL * wave + q) * 1024);
        } else {                            // last step of a K that is not a multiple of 64: chunks past K read zeros
            const uintptr_t kin[2] = {(uintptr_t)(-(long)(ks * G_BK + kofs[0] < p.K)),
                                      (uintptr_t)(-(long)(ks * G_BK + kofs[1] < p.K))};
            if constexpr (!CONV) {
#pragma unroll
                for (int q = 0; q < NAL; ++q) {
                    unsigned o = offA[q];
                    asm volatile("" : "+v"(o));      // keeps the 64-bit per-lane sum out of the loop's live registers
                    const uintptr_t real = (uintptr_t)(tileA + kbytes + o);
                    const uintptr_t a = zaddr + ((real - zaddr) & kin[apiece(q) & 1]);
                    __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(st + apiece(q) * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < NBL; ++q) {
                unsigned o = offB[q];
                asm volatile("" : "+v"(o));
                const uintptr_t real = (uintptr_t)(tileB + kbytes + o);
                const uintptr_t a = zaddr + ((real - zaddr) & kin[q & 1]);
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(st + (APIECES + NBL * wave + q) * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row 16*s + (lane&15), chunk position (4*ks + (lane>>4)) ^ ((lane&15)>>1)
    const int frow = (lane & 15) * 128;
    const int fpos[2] = {(((lane >> 4)) ^ ((lane & 15) >> 1)) * 16, ((4 + (lane >> 4)) ^ ((lane & 15) >> 1)) * 16};

    auto compute = [&](const char* st, int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *(const bf16x8*)(st + (wm * MI + i) * 2048 + frow + fpos[ks]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *(const bf16x8*)(st + APIECES * 1024 + (wn * NJ + j) * 2048 + frow + fpos[ks]);
            if (G_NST > 1 && ks == 0) issue(kt + G_NST - 1);   // refill the stage the previous step used
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    };

    STAMP(1);
    if constexpr (G_NST == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();       // the previous step is fully read
            issue(kt);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();       // every wave's pieces have landed
#ifdef GEMM_STAMPS
            if (kt == 0) STAMP(2);
#endif
            compute(lds, kt);
        }
    } else {
#pragma unroll
        for (int s = 0; s < G_NST - 1; ++s) issue(s);
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<LPT * (G_NST - 2)>();    // step kt has landed for this wave ...
            __builtin_amdgcn_s_barrier();       // ... and for every wave; step kt-1 is fully read
#ifdef GEMM_STAMPS
            if (kt == 0) STAMP(2);
#endif
            compute(lds + (kt % G_NST) * STAGE_BYTES, kt);
        }
        wait_vmcnt<0>();                        // drain the re-loaded steps before the LDS is reused
    }
    STAMP(3);
    if constexpr (EXT == 2) {
        // LayerNorm factors of the tile's rows, partial sums folded in a fixed order (the epilogue's first barrier
        // comes before its first read of them)
        if (t < BM_) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < G_LN_MAXP; ++i) { s += lnp[i].x; q += lnp[i].y; }
            const float inv = 1.0f / (float)p.K;
            const float mean = s * inv;
            const float r = rsqrtf(fmaxf(fmaf(-mean, mean, q * inv), 0.f) + p.ln_eps);
            ((float2*)lnrow)[t] = make_float2(r, -r * mean);
        }
    }
    epilogue_tile<BN_, OutT, CONV != 0, MI, EXT, WN>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave, lnrow);
#ifdef GEMM_STAMPS
    STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
#ifdef GEMM_STAMPS_TWICE
    __syncthreads();
    epilogue_tile<BN_, OutT, CONV != 0, MI>(p, acc, lds, sb, m0, n0, tm, cbase, lane, wave);
    STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && t == 0)
        for (int i = 0; i < 8; ++i) { avsr_stamps[i] = stamp[i]; avsr_stamps[8 + i] = cyc[i]; }
#endif
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
        if (p.csum) stats_quad(p, s, s2, nb, lane, tm * 2 + wm);
    }
}

// ---------------------------------------------------------------- host ----
static bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

extern "C" int64_t avsr_gemm_stats_parts(int in_dtype, int64_t M) {
    return in_dtype == AVSR_BF16 ? (M + G_BM - 1) / G_BM : 2 * ((M + FM - 1) / FM);
}

extern "C" int avsr_gemm_nt(avsr_stream_t stream, const avsr_gemm_t* g) {
    if (!g || !g->A || !g->B) return AVSR_EINVAL;
    // C == NULL: a statistics-only pass (column sums without the product itself; dense bf16 products only)
    if (!g->C && !(g->col_sum && g->in_dtype == AVSR_BF16 && !g->pre && !g->residual) && !(g->copy_bf16 && g->copy_lo))
        return AVSR_EINVAL;
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
    bool conv = g->conv.enabled != 0;
    int64_t lda = g->lda;
    if (conv) {
        const auto& c = g->conv;
        if (c.C % chunk || c.H <= 0 || c.W <= 0 || c.KH <= 0 || c.KW <= 0 || c.stride <= 0) return AVSR_EINVAL;
        if ((int64_t)c.KH * c.KW * c.C != g->K) return AVSR_EINVAL;
        if (c.OH != (c.H + 2 * c.pad_h - c.KH) / c.stride + 1) return AVSR_EINVAL;
        if (c.OW != (c.W + 2 * c.pad_w - c.KW) / c.stride + 1) return AVSR_EINVAL;
        if (g->M % ((int64_t)c.OH * c.OW)) return AVSR_EINVAL;
        // a 1x1 / stride 1 / unpadded convolution over NHWC IS a dense product with lda = C: take the dense path
        // (scalar-base operand addressing, no per-pixel cursors)
        if (c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad_h == 0 && c.pad_w == 0) { conv = false; lda = c.C; }
    } else if (g->lda % chunk) return AVSR_EUNSUPPORTED;
    // per-lane operand offsets inside a tile are 32-bit byte offsets
    if (bf && (lda > (1 << 23) || g->ldb > (1 << 23))) return AVSR_EUNSUPPORTED;

    GemmP p;
    p.A = g->A; p.B = g->B; p.C = g->C;
    p.M = (int)g->M; p.N = (int)g->N; p.K = (int)g->K;
    p.lda = lda; p.ldb = g->ldb; p.ldc = g->ldc;
    p.sA = g->strideA; p.sB = g->strideB; p.sC = g->strideC;
    p.alpha = g->alpha; p.scale = g->scale; p.bias = g->bias; p.act = g->act;
    p.res = g->residual; p.ldr = g->ldr; p.res_period = (int)g->res_period; p.gate = g->gate; p.csum = g->col_sum; p.csq = g->col_sumsq;
    p.H = g->conv.H; p.W = g->conv.W; p.Cc = g->conv.C; p.KH = g->conv.KH; p.KW = g->conv.KW;
    p.stride = g->conv.stride; p.pad_h = g->conv.pad_h; p.pad_w = g->conv.pad_w; p.OH = g->conv.OH; p.OW = g->conv.OW;
    p.pre = g->pre; p.ldp = g->ldp; p.aux = g->mul_gelu_grad_of; p.ldaux = g->ld_aux;
    p.p_drop = g->p_drop; p.seed = g->seed; p.seed_delta = (const unsigned long long*)g->seed_delta;
    p.row_part = g->row_part;
    if (g->row_part && (!bf || g->act != AVSR_ACT_NONE || !aligned_to(g->row_part, 16))) return AVSR_EUNSUPPORTED;
    p.ext = (g->pre || g->mul_gelu_grad_of || g->p_drop > 0.f || (!g->C && !g->copy_lo) || g->row_part) ? 1 : 0;
    // LayerNorm folded into the products around it: 2 = this product's A rows are normalised in the epilogue,
    // 3 = this product's stored rows leave their statistics and a bf16 copy for such a product
    const int lnmode = g->ln_in ? 2 : ((g->ln_out || g->copy_bf16 || g->residual_hi) ? 3 : 0);
    p.ln_in = g->ln_in; p.ln_parts = (int)g->ln_parts; p.ln_eps = g->ln_eps;
    p.ln_out = g->ln_out; p.copy = g->copy_bf16; p.ldcopy = g->ld_copy; p.copy_lo = g->copy_lo;
    p.res_hi = g->residual_hi; p.res_lo = g->residual_lo; p.ldrp = g->ld_res_planes;
    if (lnmode) {
        if (g->ln_in && lnmode != 2) return AVSR_EUNSUPPORTED;
        if (g->ln_in && (g->ln_out || g->copy_bf16 || g->residual_hi)) return AVSR_EUNSUPPORTED;
        if (!bf || conv || g->batch != 1 || p.ext || g->col_sum || g->gate || g->N % 8 || g->alpha != 1.0f)
            return AVSR_EUNSUPPORTED;
        if (lnmode == 2) {
            // out = rstd * (A B^T) - rstd * mean * ln_colsum + bias: the column sums ride in the scale slot
            if (!g->ln_colsum || g->scale || g->residual || g->out_dtype != AVSR_BF16 || g->ln_parts <= 0 ||
                g->ln_parts > G_LN_MAXP || (g->act != AVSR_ACT_NONE && g->act != AVSR_ACT_GELU) || !aligned_to(g->ln_in, 8))
                return AVSR_EUNSUPPORTED;
            p.scale = g->ln_colsum;
        } else {
            // exactly one residual (fp32, or a pair of bf16 planes); C and / or the bf16 planes as outputs
            if ((g->residual != nullptr) == (g->residual_hi != nullptr) || (g->residual_hi && !g->residual_lo) ||
                g->res_period > 0 || g->out_dtype != AVSR_F32 || g->act != AVSR_ACT_NONE || g->scale)
                return AVSR_EUNSUPPORTED;
            if (g->residual_hi && (!aligned_to(g->residual_hi, 16) || !aligned_to(g->residual_lo, 16) ||
                                   g->ld_res_planes % 8 || g->ld_res_planes < g->N)) return AVSR_EUNSUPPORTED;
            if (g->copy_bf16 && (!aligned_to(g->copy_bf16, 16) || !aligned_to(g->copy_lo, 16) || g->ld_copy % 8 ||
                                 g->ld_copy < g->N)) return AVSR_EUNSUPPORTED;
            if ((g->ln_out && !g->copy_bf16) || (g->copy_lo && !g->copy_bf16) || !aligned_to(g->ln_out, 8) ||
                (!g->C && !(g->copy_bf16 && g->copy_lo))) return AVSR_EUNSUPPORTED;
        }
    }
    if (p.ext) {
        const size_t osz_ = g->out_dtype == AVSR_BF16 ? 2 : 4;
        if ((!g->pre || (aligned_to(g->pre, 16) && (g->ldp * osz_) % 16 == 0)) &&
            (!g->mul_gelu_grad_of || (aligned_to(g->mul_gelu_grad_of, 16) && (g->ld_aux * osz_) % 16 == 0))) p.ext = 2;
        // the extended epilogue exists for dense, unbatched products with no activation or GELU
        if (conv || g->batch != 1 || (g->act != AVSR_ACT_NONE && g->act != AVSR_ACT_GELU)) return AVSR_EUNSUPPORTED;
        if (g->p_drop < 0.f || g->p_drop >= 1.f) return AVSR_EINVAL;
        if ((g->pre && g->ldp < g->N) || (g->mul_gelu_grad_of && g->ld_aux < g->N)) return AVSR_EINVAL;
    }
    const size_t osz = g->out_dtype == AVSR_BF16 ? 2 : 4;
    // vector path: every row start of C (and of the residual) 16-byte aligned
    p.vec_ok = ((g->ldc * osz) % 16 == 0) && aligned_to(g->C, 16) && ((g->strideC * osz) % 16 == 0) &&
               (!g->residual || (((g->ldr * osz) % 16 == 0) && aligned_to(g->residual, 16)));
    // batched launches: a residual only as a table shared by every batch entry (row m % res_period), no statistics
    if (g->batch > 1 && (g->col_sum || (g->residual && g->res_period <= 0))) return AVSR_EUNSUPPORTED;
    int bn = (bf && g->N <= 64 && !p.ext) ? 64 : 128;
    // A grid of a few dozen 128 x 128 tiles leaves most CUs idle while each busy one grinds through its tile's whole
    // K loop at one CU's MFMA rate: 64 x 64 tiles put four times as many CUs to work on a quarter of the tile each.
    const bool small = bf && !conv && !g->col_sum && g->batch == 1 && bn == 128 && !g->row_part && !lnmode &&  // row_part, ln: 128-wide tiles
                       ((g->M + 127) / 128) * ((g->N + 127) / 128) <= G_SMALL_TILES;
    if (small) bn = 64;
    // Row tile: 128, or 96 where that fills the CUs' last round much better (M = 12000, N = 768: 564 tiles leave a
    // quarter of the chip idle while 52 CUs run a third tile; 750 tiles of 96 rows are 2.93 per CU).
    int bm = G_BM;
    if (small) bm = 64;
    if (bf && !conv && !g->col_sum && bn == 128 && !p.ext && lnmode != 2) {
        // batched products (the Whisper stem convolutions, one batch per clip) count every batch's tiles
        const long tn = ((p.N + 127) / 128) * std::max<long>(1, g->batch);
        const long t128 = ((p.M + 127) / 128) * tn, t96 = ((p.M + 95) / 96) * tn;
        auto fill = [](long t, long rows, long M, long tnn) {      // useful rows / rows the busiest CU's rounds pay for
            const long rounds = (t + G_NUM_CU - 1) / G_NUM_CU;
            return (double)M * tnn / ((double)rounds * G_NUM_CU * rows);
        };
        if (!lnmode && t96 >= G_T96_MIN_TILES && fill(t96, 96, p.M, tn) > G_FILL_RATIO * fill(t128, 128, p.M, tn)) bm = 96;
#if G_BM160
        // 160-row tiles where they save a whole round over 128-row ones (M = 12000, N = 768: 450 tiles in two rounds
        // instead of 564 in three; double-buffered, two workgroups per CU: alone 88 -> 68 us at K = 3072, 35 -> 31 us at
        // K = 768, and 1.2 % of the whole step with the other encoder's kernels filling the idle CUs either way)
        const long t160 = ((p.M + 159) / 160) * tn;
        if (bm == G_BM && t160 > G_NUM_CU && fill(t160, 160, p.M, tn) > G_FILL_RATIO * fill(t128, 128, p.M, tn)) bm = 160;
#endif
    }
    // One round instead of two: a grid of a little more than one 128-row tile per CU (M = 9600, N = 512, the layer4
    // convolutions: 300 tiles) makes 44 CUs run two tiles while the rest wait; 240 tiles of 160 rows run one per CU,
    // behind a three-stage ring.  Dense products and scalar-tap convolutions, with or without column statistics.
    // Measured on the five such launches of the visual trunk: -28 us together (128 x 64 tiles instead: -18 us; a lone
    // workgroup per CU runs at a fifth of the CU's matrix rate either way, which is what bounds these launches).
    bool one_round160 = false;
#ifndef G_NO_ONE_ROUND160
    if (bf && bn == 128 && !p.ext && g->batch == 1 && !small && !lnmode) {
        const long tn = (p.N + 127) / 128;
        const long t128 = ((p.M + 127) / 128) * tn, t160 = ((p.M + 159) / 160) * tn;
        const bool fast_ok = conv && g->conv.C % G_BK == 0 && g->conv.KH * g->conv.KW <= 32 &&
                             (g->M / ((int64_t)g->conv.OH * g->conv.OW)) * g->conv.H * g->conv.W * g->conv.C * 2 < (1ll << 31);
        if (t128 > G_NUM_CU && t160 <= G_NUM_CU && p.K >= 8 * G_BK && (!conv || fast_ok)) { one_round160 = true; bm = 160; }
    }
#endif
    // The same one level up: between two and four 128-row tiles per CU (M = 101400, N = 128, layer2's convolutions: 793
    // tiles on the 512 slots of the double-buffered form, a second round at 55 %) against at most three 160-row tiles per
    // CU - one round of the single-stage form at three workgroups per CU (634 tiles): the 3 x 3 convolutions 55 -> 46.7 us,
    // the 1 x 1 ones 31 -> 29.5 us, -0.04 ms per step.  (Double-buffered at two per CU the 160-row tiles lose.)
    bool one_round160x3 = false;
    if (bf && bn == 128 && !p.ext && g->batch == 1 && !small && !lnmode && !one_round160 && bm == G_BM) {
        const long tn = (p.N + 127) / 128;
        const long t128 = ((p.M + 127) / 128) * tn, t160 = ((p.M + 159) / 160) * tn;
        const bool fast_ok = conv && g->conv.C % G_BK == 0 && g->conv.KH * g->conv.KW <= 32 &&
                             (g->M / ((int64_t)g->conv.OH * g->conv.OW)) * g->conv.H * g->conv.W * g->conv.C * 2 < (1ll << 31);
        if (t128 > 2 * G_NUM_CU && t128 <= 4 * G_NUM_CU && t160 <= 3 * G_NUM_CU && (!conv || fast_ok)) { one_round160x3 = true; bm = 160; }
    }
    // 256-row tiles (a wave owns 128 x 64: a quarter less LDS traffic per MFMA) pay off where K is long and there are
    // at least two such tiles per CU: 4096^3 918 -> 1140 TFLOP/s, 8192^3 1066 -> 1161; a loss on every product of this
    // network (M = 12000, K = 768: 846 tiles on 512 resident slots, and prologue + epilogue of a 12-step tile weigh more).
    if (bf && !conv && !g->col_sum && bn == 128 && g->batch == 1 && bm == G_BM && !p.ext && !lnmode &&
        ((p.M + 255) / 256) * ((p.N + 127) / 128) >= 2 * G_NUM_CU && p.K >= 2048) bm = 256;
#ifdef G_TUNE_ENV
    if (bf && !conv && !g->col_sum && bn == 128 && !p.ext && !small && !lnmode) {          // tuning builds only
        if (const char* e = getenv("AVSR_FORCE_BM")) { bm = atoi(e); one_round160 = false; one_round160x3 = false; }
    }
#endif
    const int tmS = bf ? bm : FM, tnS = bf ? bn : FN;
    p.tiles_m = (p.M + tmS - 1) / tmS;
    p.tiles_n = (p.N + tnS - 1) / tnS;
    p.stat_parts = (int)std::max<int64_t>(p.tiles_m, bf ? (p.M + G_BM - 1) / G_BM : 0);
    const long ntiles = (long)p.tiles_m * p.tiles_n;
    if (ntiles > 0x7fffffff || g->batch > 65535) return AVSR_EUNSUPPORTED;
    dim3 grid((unsigned)ntiles, 1, (unsigned)g->batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool of = g->out_dtype == AVSR_F32;
#define LAUNCH(K, CV, OT) hipLaunchKernelGGL((K<CV, OT>), grid, block, 0, s, p)
#define LAUNCHB(BNV, NSTV, CV, OT)                                                                             \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<BNV, NSTV>();                                                 \
        auto kfn = gemm_bf16_kernel<BNV, NSTV, CV, OT>;                                                        \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, p);                                                 \
    } while (0)
#define LAUNCHBM(BNV, NSTV, OT, BMV)                                                                           \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<BNV, NSTV, BMV>();                                            \
        auto kfn = gemm_bf16_kernel<BNV, NSTV, 0, OT, BMV>;                                                \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, p);                                                 \
    } while (0)
#define LAUNCHBMC(BNV, NSTV, OT, BMV)                                                                          \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<BNV, NSTV, BMV>();                                            \
        auto kfn = gemm_bf16_kernel<BNV, NSTV, 2, OT, BMV>;                                                    \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, p);                                                 \
    } while (0)
#define LAUNCHX(BNV, NSTV, OT, BMV)                                                                            \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<BNV, NSTV, BMV>();                                            \
        auto kfn = gemm_bf16_kernel<BNV, NSTV, 0, OT, BMV, 1>;                                                 \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, p);                                                 \
    } while (0)
#define LAUNCHLN(NSTV, OT, BMV, MODE)                                                                          \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<128, NSTV, BMV, MODE>();                                      \
        auto kfn = gemm_bf16_kernel<128, NSTV, 0, OT, BMV, MODE>;                                              \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, p);                                                 \
    } while (0)
#define LAUNCH_CV_OT(BNV, NSTV)                                                                                \
    do {                                                                                                       \
        if (conv) { if (of) LAUNCHB(BNV, NSTV, 1, float); else LAUNCHB(BNV, NSTV, 1, bf16_t); }                \
        else      { if (of) LAUNCHB(BNV, NSTV, 0, float); else LAUNCHB(BNV, NSTV, 0, bf16_t); }                \
    } while (0)
#define LAUNCH_FAST_OT(BNV, NSTV) do { if (of) LAUNCHB(BNV, NSTV, 2, float); else LAUNCHB(BNV, NSTV, 2, bf16_t); } while (0)
    // whole K-steps inside one filter tap, taps in one 32-bit mask, pixel byte offsets in 31 bits.  Used with the
    // double-buffered instantiations only (grids below the single-stage switch): under the 128-register cap of the
    // single-stage ones its extra per-lane state spills inside the K loop (measured: half the rate).
    const bool conv_fast = bf && conv && g->conv.C % G_BK == 0 && g->conv.KH * g->conv.KW <= 32 &&
                           (g->M / ((int64_t)g->conv.OH * g->conv.OW)) * g->conv.H * g->conv.W * g->conv.C * 2 < (1ll << 31);
#define LAUNCHW8(NSTV, CV, OT, BMV)                                                                            \
    do {                                                                                                       \
        static avsr_dev_flags attr_set;                                                                        \
        constexpr int lds_bytes = gemm_lds_bytes<128, NSTV, BMV, 0, 4>();                                      \
        auto kfn = gemm_bf16_kernel<128, NSTV, CV, OT, BMV, 0, 4>;                                             \
        if (!avsr_allow_lds((const void*)kfn, lds_bytes, attr_set)) return AVSR_EHIP;                          \
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds_bytes, s, p);                                             \
    } while (0)
    if (lnmode == 2) {
        // the tile shapes the encoder's QKV / fc1 products take without the fold (single stage on full grids)
        if (ntiles >= G_NST_SWITCH_TILES) LAUNCHLN(1, bf16_t, 128, 2); else LAUNCHLN(2, bf16_t, 128, 2);
    } else if (lnmode == 3) {
        if (!p.vec_ok) return AVSR_EUNSUPPORTED;
        if (bm == 160) LAUNCHLN(G_BM160_NST, float, 160, 3);
        else if (ntiles >= G_NST_SWITCH_TILES) LAUNCHLN(1, float, 128, 3);
        else LAUNCHLN(2, float, 128, 3);
    } else if (bf && p.ext) {
        // extended epilogue: a second set of instantiations, restricted to the tile shapes the trainable tail uses
        const bool single = ntiles >= G_NST_SWITCH_TILES;
        if (small) { if (of) LAUNCHX(64, 4, float, 64); else LAUNCHX(64, 4, bf16_t, 64); }
        else if (ntiles <= G_DEEP_TILES) { if (of) LAUNCHX(128, 4, float, 128); else LAUNCHX(128, 4, bf16_t, 128); }
        else if (single) { if (of) LAUNCHX(128, 1, float, 128); else LAUNCHX(128, 1, bf16_t, 128); }
        else { if (of) LAUNCHX(128, 2, float, 128); else LAUNCHX(128, 2, bf16_t, 128); }
    } else if (bf) {
        const bool single = ntiles * g->batch >= G_NST_SWITCH_TILES;
        if (small) {
            if (of) LAUNCHBM(64, 4, float, 64); else LAUNCHBM(64, 4, bf16_t, 64);
        } else if (bm == 256) {
            if (of) LAUNCHBM(128, 1, float, 256); else LAUNCHBM(128, 1, bf16_t, 256);
        } else if (one_round160) {
            // one workgroup per CU: a deeper ring hides the DMA latency, and EIGHT waves (2 x 4, 80 x 32 each) keep the
            // matrix pipe busier than four while a wave waits for its fragments (layer4's 3 x 3 convolutions 68 -> 60 us,
            // its 1 x 1 ones 30.6 -> 28 us; -0.03 ms per step).  The same eight waves on the two-workgroups-per-CU grids
            // of layer3 gain 4 % alone and nothing in the step: a 512-thread workgroup finds room later beside the other
            // streams' kernels.
            if (conv) { if (of) LAUNCHW8(G_ONE160_NST, 2, float, 160); else LAUNCHW8(G_ONE160_NST, 2, bf16_t, 160); }
            else      { if (of) LAUNCHW8(G_ONE160_NST, 0, float, 160); else LAUNCHW8(G_ONE160_NST, 0, bf16_t, 160); }
        } else if (one_round160x3) {
            if (conv) { if (of) LAUNCHBMC(128, 1, float, 160); else LAUNCHBMC(128, 1, bf16_t, 160); }
            else      { if (of) LAUNCHBM(128, 1, float, 160); else LAUNCHBM(128, 1, bf16_t, 160); }
        } else if (bm == 160) {
            if (of) LAUNCHBM(128, G_BM160_NST, float, 160); else LAUNCHBM(128, G_BM160_NST, bf16_t, 160);
        } else if (bm == 96) {
            if (of) LAUNCHBM(128, 1, float, 96); else LAUNCHBM(128, 1, bf16_t, 96);
        } else if (!conv && bn == 128 && ntiles * g->batch <= G_DEEP_TILES) {     // at most one workgroup per CU: 3 K-steps in flight
            if (of) LAUNCHB(128, 4, 0, float); else LAUNCHB(128, 4, 0, bf16_t);
        } else if (bn == 64) { if (single) LAUNCH_CV_OT(64, 1); else if (conv_fast) LAUNCH_FAST_OT(64, 2); else LAUNCH_CV_OT(64, 2); }
        else          { if (single) LAUNCH_CV_OT(128, 1); else if (conv_fast) LAUNCH_FAST_OT(128, 2); else LAUNCH_CV_OT(128, 2); }
    } else {
        if (conv) { if (of) LAUNCH(gemm_f32_kernel, true, float); else LAUNCH(gemm_f32_kernel, true, bf16_t); }
        else      { if (of) LAUNCH(gemm_f32_kernel, false, float); else LAUNCH(gemm_f32_kernel, false, bf16_t); }
    }
#undef LAUNCH
#undef LAUNCHB
#undef LAUNCHBM
#undef LAUNCHBMC
#undef LAUNCHX
#undef LAUNCHLN
#undef LAUNCHW8
#undef LAUNCH_CV_OT
#undef LAUNCH_FAST_OT
    return avsr_launch_status();
}
