// Weight-gradient product: C[N x K] (fp32) (+)= A^T . B with A [M x N] and B [M x K] bf16, both ROW-major, so
// the contraction runs over the rows of both operands (dW = dY^T X of every nn.Linear of the trainable tail:
// M = B*T rows, a few hundred to a few thousand).  The operand tiles go global -> LDS as they lie in memory
// (LDS-DMA, 8 rows x 128 B per wave instruction) and the MFMA fragments are read with the transposing
// ds_read_b64_tr_b16, so neither operand is transposed in memory first; the bias gradient (column sums of A)
// comes out of the same pass.  One 64 x 64 tile of C per workgroup, four waves of one v_mfma_f32_32x32x16_bf16
// accumulator each.
#include "common.h"
#include <algorithm>

namespace {

__device__ __attribute__((aligned(16))) unsigned int tn_zero_page[4] = {0u, 0u, 0u, 0u};

// 16-byte chunk swizzle of a [64 rows][64 cols] bf16 tile: conflict-free for the transposing reads below
__device__ __forceinline__ int t_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }

// Operand fragment of X^T for v_mfma_f32_32x32x16_bf16 from a row-major LDS tile X: matrix row / column
// (= X column) 32*cb + (lane & 31), eight contraction indices (= X rows) out of rows 16*ks .. 16*ks + 15 in an
// order that depends only on the lane - the same for both operands, which is all a dot product needs.
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* X, int ks, int cb, int lane) {
    const int i16 = lane & 15, hf = lane >> 5;
    const int col = 32 * cb + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
    bf16x8 f;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        const int row = 16 * ks + 8 * jh + 4 * hf + (i16 >> 2);
        const bf16_t* ap = &X[t_off(row, col >> 3) + (col & 7)];
        const s16x4 tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
        const bf16x4 tb = __builtin_bit_cast(bf16x4, tr);
        f[4 * jh + 0] = tb[0]; f[4 * jh + 1] = tb[1]; f[4 * jh + 2] = tb[2]; f[4 * jh + 3] = tb[3];
    }
    return f;
}

// 1-D grid: workgroup ids go round-robin over the 8 XCDs, and all K tiles of one N tile are given to ONE XCD (ids
// 8 apart), so a column block of A - the large operand of the vocabulary head, 600 x 51865 - crosses the fabric
// once, not once per K tile; B (600 x 512) stays resident in every L2.
// TB: the C tile is 64*TB square (TB = 2 for outputs of thousands of tiles: half the operand re-reads; TB = 1
// otherwise: more workgroups).  NST-deep ring of 64-row stages: the contraction is short (M/64 = 10 steps for the
// fusion's 600 rows), so with one stage in flight a step costs a whole memory latency; with NST - 1 in flight only
// the first does.
template <int TB, int NST, bool GRAM = false> constexpr int tn_main_bytes() {
    constexpr int stages = (GRAM ? 1 : 2) * NST * TB * 64 * 64 * 2, epi = 4 * (32 * TB) * (32 * TB + 4) * 4;
    return stages > epi ? stages : epi;
}
template <int TB, int NST, bool GRAM = false> constexpr int tn_lds_bytes() { return tn_main_bytes<TB, NST, GRAM>() + 4 * 64 * TB * 4; }

// GRAM: the Gram matrix A^T A of ONE tall operand with at most 64*TB columns (B is A: one set of stages, both fragments
// read from it).  The rows are cut into gridDim.x shares, one workgroup each, on every XCD; share p stores its partial
// product at C + p * part_stride and its column sums at colsum + p * N with plain stores (no atomics: the caller adds
// the shares in a fixed order).
// APPLY (with GRAM): the operand in memory is a RAW convolution output whose batch-norm affine + ReLU has not been
// applied yet: every lane normalises the chunks it DMA'd, in LDS, before the stage is published, and stores them to Y
// (which may be A itself) - the Gram statistics are those of the normalised tensor, and the separate apply pass over it
// (a read and a launch) is gone.  Rows past M stay zero.
template <int TB, int NST, bool GRAM = false, bool APPLY = false>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16_t* A, long lda, const bf16_t* B,
                                                           long ldb, float* __restrict__ C, long ldc, int M, int N, int K,
                                                           int accumulate, float* __restrict__ colsum, int accumulate_colsum,
                                                           int ntiles, int ktiles, int nsplit, long part_stride,
                                                           const float* __restrict__ ascale = nullptr,
                                                           const float* __restrict__ ashift = nullptr, bf16_t* Y = nullptr,
                                                           long ldy = 0) {
    static_assert(!APPLY || GRAM, "the in-flight normalisation exists for the Gram form only");
    constexpr int TILE = 64 * TB, SUB = 64 * 64;              // an operand stage is TB sub-tiles of [64 rows][64 cols]
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    bf16_t* As = (bf16_t*)lds_raw;                            // [NST][TB][SUB]
    bf16_t* Bs = GRAM ? As : As + NST * TB * SUB;
    float* csum = (float*)(lds_raw + tn_main_bytes<TB, NST, GRAM>());   // [4][TILE], behind the stages / the epilogue blocks
    // nsplit > 1 (long contractions into few tiles): the rows are cut into nsplit shares, one workgroup each, and
    // the shares meet in C and colsum by fp32 atomic adds (accumulating calls only; summation order then varies)
    const int xcd = blockIdx.x & 7, j = GRAM ? 0 : (blockIdx.x >> 3) / nsplit;
    const int part = GRAM ? (int)blockIdx.x : (int)(blockIdx.x >> 3) - j * nsplit;
    const int kt = GRAM ? 0 : j % ktiles, nt = GRAM ? 0 : (j / ktiles) * 8 + xcd;
    if (nt >= ntiles) return;
    if constexpr (GRAM) { C += (long)part * part_stride; colsum += (long)part * N; }
    const int n0 = nt * TILE, k0 = kt * TILE;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int prow = lane >> 3, pch = lane & 7;
    const uintptr_t zaddr = (uintptr_t)tn_zero_page;

    // A wave moves pieces 2*wave and 2*wave + 1 (8 rows each) of every sub-tile: 4*TB instructions per wave and stage.
    // Interior stages (all 64 rows < M, the tile's columns inside both matrices) use the scalar-base form of the
    // instruction through inline asm: the compiler does not know these loads, so it puts no wait between them and
    // the fragment reads of the stage being computed (behind the builtin it waits for vmcnt(0) at once, which turns
    // the ring into load - wait - compute).  Edge stages go through the builtin with a per-lane address, where a chunk
    // outside the matrix (row >= M, or its 8 columns not inside the row pitch) reads the zero page instead.
    unsigned offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (2 * wave + i) * 8 + prow;
        const int ch = pch ^ (((rr >> 1) & 1) << 2);
        offA[i] = (unsigned)(((long)rr * lda + ch * 8) * 2);
        offB[i] = (unsigned)(((long)rr * ldb + ch * 8) * 2);
    }
    const bool cols_inside = n0 + TILE <= N && k0 + TILE <= K;
    auto dma_tiles = [&](int m0, int buf) {
        if (cols_inside && m0 + 64 <= M) {
            const char* ab = (const char*)(A + (long)m0 * lda + n0);
            const char* bb = (const char*)(B + (long)m0 * ldb + k0);
#pragma unroll
            for (int sub = 0; sub < TB; ++sub)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    dma_piece(ab + 128 * sub, offA[i], (const char*)&As[(buf * TB + sub) * SUB + (2 * wave + i) * 512]);
                    if constexpr (!GRAM)
                        dma_piece(bb + 128 * sub, offB[i], (const char*)&Bs[(buf * TB + sub) * SUB + (2 * wave + i) * 512]);
                }
            return;
        }
#pragma unroll
        for (int sub = 0; sub < TB; ++sub)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rr = (2 * wave + i) * 8 + prow;
                const int ch = pch ^ (((rr >> 1) & 1) << 2);
                const long m = m0 + rr;
                const long ca = n0 + 64 * sub + ch * 8, cb = k0 + 64 * sub + ch * 8;
                const uintptr_t ra = (uintptr_t)(A + m * lda + ca), rb = (uintptr_t)(B + m * ldb + cb);
                const bool oka = m < M && ca < N && ca + 8 <= lda, okb = m < M && cb < K && cb + 8 <= ldb;
                const uintptr_t pa = zaddr + ((ra - zaddr) & (uintptr_t)(-(long)oka));
                const uintptr_t pb = zaddr + ((rb - zaddr) & (uintptr_t)(-(long)okb));
                if constexpr (APPLY) dma_lane(pa, (const char*)&As[(buf * TB + sub) * SUB + (2 * wave + i) * 512]);
                else __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)&As[(buf * TB + sub) * SUB + (2 * wave + i) * 512], 16, 0, 0);
                if constexpr (!GRAM)
                    __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)&Bs[(buf * TB + sub) * SUB + (2 * wave + i) * 512], 16, 0, 0);
            }
    };
    auto wait_oldest = [&](bool younger_in_flight) {         // all but the youngest stage's 4*TB loads have landed
        if (NST > 2 && younger_in_flight) {
            if constexpr (TB == 1 && GRAM) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (TB == 1 || GRAM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    f32x16 acc[TB][TB];
#pragma unroll
    for (int a = 0; a < TB; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float cs[TB];
#pragma unroll
    for (int a = 0; a < TB; ++a) cs[a] = 0.f;
    // wave -> a TB x TB group of 32 x 32 blocks: rows (N) 32*TB*(wave & 1) .., columns (K) 32*TB*(wave >> 1) ..
    const int nb0 = TB * (wave & 1), kb0 = TB * (wave >> 1);
    const bool sums = colsum != nullptr && kt == 0;
    const int allsteps = (M + 63) >> 6;
    const int st0 = (int)((long)part * allsteps / nsplit), nsteps = (int)((long)(part + 1) * allsteps / nsplit) - st0;
    const int mbase = 64 * st0;
    // APPLY: the affine of this lane's eight channels per sub-tile (the swizzled chunk it moves is the same in both of
    // its pieces), and the pass over the stage that has just landed
    float asc[APPLY ? TB : 1][8], ash[APPLY ? TB : 1][8];
    const int ach = pch ^ (((prow >> 1) & 1) << 2);
    if constexpr (APPLY) {
#pragma unroll
        for (int sub = 0; sub < TB; ++sub)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = min(64 * sub + ach * 8 + e, N - 1);
                asc[sub][e] = ascale[c]; ash[sub][e] = ashift[c];
            }
    }
    auto normalise = [&](int m0, int buf) {
        if constexpr (APPLY) {
#pragma unroll
            for (int sub = 0; sub < TB; ++sub)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rr = (2 * wave + i) * 8 + prow;
                    const long m = m0 + rr;
                    const int col = 64 * sub + ach * 8;
                    bf16_t* lp = &As[(buf * TB + sub) * SUB + (2 * wave + i) * 512 + lane * 8];
                    const bf16x8 v = *(const bf16x8*)lp;
                    const bool ok = m < M && col < N;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(ok ? fmaxf(fmaf((float)v[e], asc[sub][e], ash[sub][e]), 0.f) : 0.f);
                    *(bf16x8*)lp = o;
                    if (ok) *(bf16x8*)(Y + m * ldy + col) = o;
                }
        }
    };
    dma_tiles(mbase, 0);
    if (NST > 2 && nsteps > 1) dma_tiles(mbase + 64, 1);
    wait_oldest(nsteps > 1);
    normalise(mbase, 0);
    __syncthreads();
    for (int st = 0, cur = 0; st < nsteps; ++st, cur = (cur + 1 == NST ? 0 : cur + 1)) {
        const int ahead = st + NST - 1;                       // its buffer was last read one barrier ago
        if (ahead < nsteps) dma_tiles(mbase + 64 * ahead, cur == 0 ? NST - 1 : cur - 1);
        const bf16_t* as = As + cur * TB * SUB;
        const bf16_t* bs = Bs + cur * TB * SUB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[TB], bfr[TB];
#pragma unroll
            for (int a = 0; a < TB; ++a) {
                af[a] = tr_frag(as + ((nb0 + a) >> 1) * SUB, ks, (nb0 + a) & 1, lane);
                bfr[a] = tr_frag(bs + ((kb0 + a) >> 1) * SUB, ks, (kb0 + a) & 1, lane);
            }
#pragma unroll
            for (int a = 0; a < TB; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (sums) {                                           // column lane of A, a quarter of the stage's rows per wave
#pragma unroll
            for (int a = 0; a < TB; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) cs[a] += (float)as[a * SUB + t_off(16 * wave + r, lane >> 3) + (lane & 7)];
        }
        // the next stage is in LDS (a younger one may be in flight).  APPLY: the normalised rows stored two stages ago
        // are waited for as well - loads return in order, so "at most the youngest stage's loads outstanding" still
        // means the stage before it has landed, whatever the stores do
        wait_oldest(st + NST - 1 < nsteps);
        if (st + 1 < nsteps) normalise(mbase + 64 * (st + 1), cur + 1 == NST ? 0 : cur + 1);
        __syncthreads();
    }
    // accumulator: lane = column (K index), register i <-> row (i & 3) + 8 (i >> 2) + 4 hf.  A wave passes its
    // 32*TB x 32*TB block through LDS (the operand stages are free now; a region per wave, so no barrier) and
    // writes - or reads, adds and writes - whole 128*TB-byte row segments with 16-byte accesses: straight from the
    // registers a wave instruction touches two rows of 128 bytes, and the read-modify-write of the vocabulary
    // head's 106 MB gradient ran at half the rate.
    const int r = lane & 31, hf = lane >> 5;
    constexpr int BW = 32 * TB, LDT = BW + 4;                 // block width, padded LDS row
    float* ct = (float*)lds_raw + wave * (BW * LDT);
#pragma unroll
    for (int a = 0; a < TB; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) ct[(32 * a + (i & 3) + 8 * (i >> 2) + 4 * hf) * LDT + 32 * b + r] = acc[a][b][i];
    const bool vec = ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0);
    constexpr int QPR = BW / 4, RPI = 64 / QPR;               // float4 per row, rows per wave instruction
    const int qc = (lane % QPR) * 4, qr = lane / QPR;
    const int kq = k0 + BW * (wave >> 1) + qc;
#pragma unroll
    for (int it = 0; it < BW / RPI; ++it) {
        const int row = it * RPI + qr;
        const int n = n0 + BW * (wave & 1) + row;
        if (n >= N || kq >= K) continue;
        const f32x4 v = *(const f32x4*)&ct[row * LDT + qc];
        float* cp = C + (long)n * ldc + kq;
        if (nsplit > 1 && !GRAM) {
            for (int e = 0; e < 4 && kq + e < K; ++e) unsafeAtomicAdd(cp + e, v[e]);      // the hardware fp32 add, no CAS loop
        } else if (vec && kq + 4 <= K) {
            f32x4 o = v;
            if (accumulate) { const f32x4 c = *(const f32x4*)cp; o[0] += c[0]; o[1] += c[1]; o[2] += c[2]; o[3] += c[3]; }
            *(f32x4*)cp = o;
        } else {
            for (int e = 0; e < 4 && kq + e < K; ++e) cp[e] = accumulate ? cp[e] + v[e] : v[e];
        }
    }
    if (sums) {
#pragma unroll
        for (int a = 0; a < TB; ++a) csum[wave * TILE + 64 * a + lane] = cs[a];
        __syncthreads();
        if (t < TILE && n0 + t < N) {
            const float s = csum[t] + csum[TILE + t] + csum[2 * TILE + t] + csum[3 * TILE + t];
            if (nsplit > 1 && !GRAM) unsafeAtomicAdd(&colsum[n0 + t], s);
            else colsum[n0 + t] = accumulate_colsum ? colsum[n0 + t] + s : s;
        }
    }
}


template <int TB, int NST>
int launch_tn(hipStream_t s, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
              int64_t N, int64_t K, int accumulate, float* colsum, int accumulate_colsum) {
    constexpr int lds = tn_lds_bytes<TB, NST>();
    static avsr_dev_flags attr_set;                           // per device; the attribute is idempotent
    if (lds > 48 * 1024 && !avsr_allow_lds((const void*)gemm_tn_bf16_kernel<TB, NST>, lds, attr_set)) return AVSR_EHIP;
    const int64_t ntiles = (N + 64 * TB - 1) / (64 * TB), ktiles = (K + 64 * TB - 1) / (64 * TB);
    // Few tiles and a very long contraction (thousands of rows): several workgroups per tile, each at least 32
    // 64-row steps, meeting in C by atomic adds (1.7x at M = 12000 into 768 x 768; at 3200 rows the atomics cost what
    // the shorter loops save).  The tail's own shapes (hundreds of rows) never split: their sums stay order-exact.
    int64_t nsplit = 1;
    if (accumulate && (!colsum || accumulate_colsum))
        nsplit = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((M + 63) / 64 / 32, 512 / (ntiles * ktiles)), 16));
    const int64_t grid = 8 * ((ntiles + 7) / 8) * ktiles * nsplit;
    if (grid > 0x7fffffff) return AVSR_EUNSUPPORTED;
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<TB, NST>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)A, (long)lda,
                       (const bf16_t*)B, (long)ldb, C, (long)ldc, (int)M, (int)N, (int)K, accumulate, colsum,
                       accumulate_colsum, (int)ntiles, (int)ktiles, (int)nsplit, 0L);
    return avsr_launch_status();
}

template <int TB, int NST, bool APPLY = false>
int launch_gram(hipStream_t s, const void* X, int64_t ldx, int64_t M, int64_t K, int64_t parts, float* G, float* colsum,
                const float* ascale = nullptr, const float* ashift = nullptr, void* Y = nullptr, int64_t ldy = 0) {
    constexpr int lds = tn_lds_bytes<TB, NST, true>();
    static avsr_dev_flags attr_set;
    if (lds > 48 * 1024 && !avsr_allow_lds((const void*)gemm_tn_bf16_kernel<TB, NST, true, APPLY>, lds, attr_set)) return AVSR_EHIP;
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<TB, NST, true, APPLY>), dim3((unsigned)parts), dim3(256), lds, s, (const bf16_t*)X,
                       (long)ldx, (const bf16_t*)X, (long)ldx, G, (long)K, (int)M, (int)K, (int)K, 0, colsum, 0, 1, 1,
                       (int)parts, (long)(K * K), ascale, ashift, (bf16_t*)Y, (long)ldy);
    return avsr_launch_status();
}

}  // namespace

#ifndef GRAM_CAP64
#define GRAM_CAP64 512
#endif
#ifndef GRAM_CAP128
#define GRAM_CAP128 256
#endif
#ifndef GRAM_MIN_STEPS
#define GRAM_MIN_STEPS 8
#endif
extern "C" int64_t avsr_gram_parts(int64_t M, int64_t K) {
    if (M <= 0 || (K != 64 && K != 128)) return 0;
    // shares of at least eight 64-row steps, at most two (K = 64) or one (K = 128) per CU
    const int64_t steps = (M + 63) / 64, cap = K == 64 ? GRAM_CAP64 : GRAM_CAP128;
    return std::max<int64_t>(1, std::min<int64_t>(cap, steps / GRAM_MIN_STEPS));
}

extern "C" int avsr_gram_stats(avsr_stream_t stream, const void* X, int64_t ldx, int64_t M, int64_t K, float* gram_parts,
                               float* colsum_parts) {
    if (!X || !gram_parts || !colsum_parts || M <= 0) return AVSR_EINVAL;
    if (K != 64 && K != 128) return AVSR_EUNSUPPORTED;
    if (ldx < K || ldx % 8 || (uintptr_t)X % 16 || M > 0x7fffffff || 64 * ldx * 2 > 0x7fffffffLL) return AVSR_EUNSUPPORTED;
    const int64_t parts = avsr_gram_parts(M, K);
    if (K == 64) return launch_gram<1, 3>((hipStream_t)stream, X, ldx, M, K, parts, gram_parts, colsum_parts);
    return launch_gram<2, 3>((hipStream_t)stream, X, ldx, M, K, parts, gram_parts, colsum_parts);
}

extern "C" int avsr_gram_stats_bn(avsr_stream_t stream, const void* X, int64_t ldx, int64_t M, int64_t K, const float* scale,
                                  const float* shift, void* Y, int64_t ldy, float* gram_parts, float* colsum_parts) {
    if (!X || !Y || !scale || !shift || !gram_parts || !colsum_parts || M <= 0) return AVSR_EINVAL;
    if (K != 64 && K != 128) return AVSR_EUNSUPPORTED;
    if (ldx < K || ldx % 8 || (uintptr_t)X % 16 || M > 0x7fffffff || 64 * ldx * 2 > 0x7fffffffLL) return AVSR_EUNSUPPORTED;
    if (ldy < K || ldy % 8 || (uintptr_t)Y % 16) return AVSR_EUNSUPPORTED;
    const int64_t parts = avsr_gram_parts(M, K);
    if (K == 64) return launch_gram<1, 3, true>((hipStream_t)stream, X, ldx, M, K, parts, gram_parts, colsum_parts, scale, shift, Y, ldy);
    return launch_gram<2, 3, true>((hipStream_t)stream, X, ldx, M, K, parts, gram_parts, colsum_parts, scale, shift, Y, ldy);
}

extern "C" int avsr_gemm_tn(avsr_stream_t stream, const void* A, int64_t lda, const void* B, int64_t ldb, float* C,
                            int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, float* colsum,
                            int accumulate_colsum) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return AVSR_EINVAL;
    if (lda < N || ldb < K || ldc < K) return AVSR_EINVAL;
    if (lda % 8 || ldb % 8 || ((uintptr_t)A | (uintptr_t)B) % 16) return AVSR_EUNSUPPORTED;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return AVSR_EUNSUPPORTED;
    if (64 * std::max(lda, ldb) * 2 > 0x7fffffffLL) return AVSR_EUNSUPPORTED;       // 32-bit byte offsets inside a stage
    hipStream_t s = (hipStream_t)stream;
    // thousands of 64 x 64 tiles (the vocabulary head): 128 x 128 tiles halve the operand traffic out of L2; the same
    // for contractions over thousands of rows, where the 64 x 64 form (one 32 x 32 block per wave, two fragment reads
    // per MFMA) is bound by its LDS reads: 146 -> see profiles for the 25 600-row weight gradients
    if (((N + 63) / 64) * ((K + 63) / 64) >= 2048 || (M >= 4096 && N >= 128 && K >= 128))
        return launch_tn<2, 2>(s, A, lda, B, ldb, C, ldc, M, N, K, accumulate, colsum, accumulate_colsum);
    return launch_tn<1, 3>(s, A, lda, B, ldb, C, ldc, M, N, K, accumulate, colsum, accumulate_colsum);
}
