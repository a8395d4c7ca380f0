// Attention kernels, head_dim 64.
//
//  * attn_fwd_mfma_kernel: flash-style forward for bf16 activations.  Scores
//    are computed TRANSPOSED (S^T = K.Q^T with v_mfma_f32_32x32x16_bf16) so a
//    lane owns one query column: the online-softmax statistics are per lane,
//    and the 32x32 accumulator is, register for register, the B operand of
//    the O^T = V^T.P^T product (no LDS round trip for P).  V is read from its
//    row-major LDS tile with the transposing ds_read_b64_tr_b16.
//    Key-padding mask, attention dropout (four keys per generator call) and
//    the saved log-sum-exp make it the core of the gated cross-attention too.
//  * attn_bwd_dq_mfma_kernel / attn_bwd_dkv_mfma_kernel: its backward on the
//    same transposed scheme, no atomics (each recomputes S and dP).
//  * attn_*_valu_kernel: exact fp32-math forward / backward (dQ and dK,dV)
//    with the same mask / dropout / log-sum-exp semantics: the fp32 parity path.
//
// Layout of every operand: element (b, t, h, d) at ptr + (b*T + t)*ld + h*64 + d.
#include "common.h"
#include <algorithm>
#include <type_traits>

#define HD 64
#define LOG2E 1.4426950408889634f

// ---------------------------------------------------------------- MFMA ----
// v_mfma_f32_32x32x16_bf16 fragments (lane = 32*hf + r): A row r / B column r,
// k = 8*hf + j (j = 0..7); C column r, register i <-> row (i&3) + 8*(i>>2) + 4*hf.
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }

// A-operand fragment of X^T for a row-major [64 rows][64 cols] LDS tile X stored with v_off: output row
// (= X column) 32*db + r, contraction index (= X row) follows the C-register order of rows
// 32*st + 16*s2 .. +15, so a 32x32 accumulator converted to bf16 is directly the matching B operand.
__device__ __forceinline__ bf16x8 read_tr_frag(const bf16_t* X, int st, int s2, int db, int lane) {
    const int i16 = lane & 15, hf = lane >> 5;
    const int col = 32 * db + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
    bf16x8 f;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        const int row = 32 * st + 16 * s2 + 8 * jh + 4 * hf + (i16 >> 2);
        const bf16_t* ap = &X[v_off(row, col >> 3) + (col & 7)];
        const s16x4 tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
        const bf16x4 tb = __builtin_bit_cast(bf16x4, tr);
        f[4 * jh + 0] = tb[0]; f[4 * jh + 1] = tb[1]; f[4 * jh + 2] = tb[2]; f[4 * jh + 3] = tb[3];
    }
    return f;
}
__device__ __forceinline__ bf16x8 acc_to_bf16(const f32x16& a, int s2) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)a[8 * s2 + j];
    return f;
}
// max of a value with its partner 32 lanes away: one v_permlane32_swap (register file only) instead of a
// shuffle through the LDS crossbar on the loop's serial chain
__device__ __forceinline__ float max_across_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// bit (key - kv0) set <=> key of this 64-key tile is padding (>= Tk) or masked; wave-uniform
__device__ __forceinline__ uint64_t dead_keys(const uint8_t* mb, int kv0, int Tk, int lane) {
    const int key = kv0 + lane;
    return __ballot(key >= Tk || (mb && mb[key < Tk ? key : 0]));
}

template <bool DROP, bool PIPE>
#ifndef ATT_OCC
#define ATT_OCC 3
#endif
#ifndef ATT_UNROLL
#define ATT_UNROLL 1
#endif
#ifndef ATT_PACKED
#define ATT_PACKED 0
#endif
#ifndef ATT_MAX3
#define ATT_MAX3 1
#endif
#ifndef ATT_LAZY_LOG2
#define ATT_LAZY_LOG2 8.0f
#endif
#ifndef ATT_NBUF
#define ATT_NBUF 3
#endif
__global__ __launch_bounds__(256, ATT_OCC) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                            long ldk, const bf16_t* __restrict__ V, long ldv,
                                                            bf16_t* __restrict__ O, long ldo, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ lse, int Tq, int Tk, int H, float scale,
                                                            float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta, int B, int causal,
                                                            int whole_per_xcd, int nsplit, float* __restrict__ part_ws,
                                                            int* __restrict__ tickets) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
    __shared__ __attribute__((aligned(16))) bf16_t Kbuf[ATT_NBUF][64 * 64];      // ring: one barrier per key tile
    __shared__ __attribute__((aligned(16))) bf16_t Vbuf[ATT_NBUF][64 * 64];
    __shared__ int last_part;
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own L2.  All
    // query tiles of one (batch, head) are given to ONE XCD so its K/V are fetched from the fabric once, not 8 times.
    // An item is one (batch, head, 128-query tile).  The first whole_per_xcd items of an XCD are one workgroup each;
    // each later item is nsplit workgroups, each over a contiguous share of the key tiles (the host picks the numbers
    // so that the last round of workgroups fills the chip: 1.5 rounds of equal items otherwise take two).  The
    // shares of an item leave their unnormalised sums in part_ws, and whichever finishes last adds them up.
    const int qtiles = (Tq + 127) >> 7, npairs = B * H;
    const int per_xcd = (npairs + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int item = slot, part = 0, nparts = 1;
    if (slot >= whole_per_xcd) {
        const int e = slot - whole_per_xcd;
        item = whole_per_xcd + e / nsplit; part = e - (e / nsplit) * nsplit; nparts = nsplit;
    }
    const int pair = xcd * per_xcd + item / qtiles, qt = item - (item / qtiles) * qtiles;
    if (pair >= npairs || item >= per_xcd * qtiles) return;
    const int b = pair / H, h = pair - b * H;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int q = qt * 128 + wave * 32 + r;                   // this lane's query row
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
    const float lazy = ATT_LAZY_LOG2 / sc2;                   // in raw score units
    const bf16_t* Kb = K + (long)b * Tk * ldk + h * HD;
    const bf16_t* Vb = V + (long)b * Tk * ldv + h * HD;
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;
    const float keep_scale = DROP ? 1.0f / (1.0f - pdrop) : 1.0f;
    const int tk4 = (Tk + 3) >> 2;
    const uint64_t row = ((uint64_t)b * H + h) * Tq + q;

    // K / V tiles go global -> LDS directly (LDS-DMA): a wave instruction moves 8 key rows x 128 B into 1 KiB of
    // lane-linear LDS, so the XOR swizzle of k_off / v_off is applied to the chunk each lane FETCHES.  Each wave
    // moves pieces 2*wave and 2*wave+1 of both tiles: four instructions per 64-key tile, no staging registers, no
    // ds_write.  The per-lane byte offsets are constants of the kernel (the tile base advances in scalar registers);
    // only a tile that reaches past Tk recomputes them with its rows clamped onto the last key (those keys are
    // masked to -inf further down, and a clamped V row is finite, so 0 * V stays 0).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int prow = lane >> 3, pch = lane & 7;
    auto piece_off = [&](int i, int kv0, unsigned& ko, unsigned& vo) {
        const int rr = (2 * wave_u + i) * 8 + prow;
        const int kc = pch ^ ((rr >> 1) & 7), vc = pch ^ (((rr >> 1) & 1) << 2);
        const long key = min(rr, Tk - 1 - kv0);
        ko = (unsigned)((key * ldk + kc * 8) * (long)sizeof(bf16_t));
        vo = (unsigned)((key * ldv + vc * 8) * (long)sizeof(bf16_t));
    };
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) piece_off(i, 0, koff[i], voff[i]);
    auto dma_tile = [&](int kv0, int buf) {
        const char* kb = (const char*)(Kb + (long)kv0 * ldk);
        const char* vb = (const char*)(Vb + (long)kv0 * ldv);
        if (kv0 + 64 <= Tk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                dma_piece(kb, koff[i], (const char*)&Kbuf[buf][(2 * wave_u + i) * 512]);
                dma_piece(vb, voff[i], (const char*)&Vbuf[buf][(2 * wave_u + i) * 512]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned ko, vo;
                piece_off(i, kv0, ko, vo);
                dma_piece(kb, ko, (const char*)&Kbuf[buf][(2 * wave_u + i) * 512]);
                dma_piece(vb, vo, (const char*)&Vbuf[buf][(2 * wave_u + i) * 512]);
            }
        }
    };
    // causal: query i sees keys <= i + (Tk - Tq) (the last Tq positions of a Tk-long sequence), so this
    // workgroup's key loop ends at the diagonal of its last query row
    const int coff = Tk - Tq;
    int kbeg = 0, kend = causal ? min(Tk, qt * 128 + 128 + coff) : Tk;
    if (nparts > 1) {                                         // this workgroup's share of the key tiles
        const int nt = (kend + 63) >> 6;
        kbeg = 64 * (part * nt / nparts);
        kend = min(kend, 64 * ((part + 1) * nt / nparts));
    }
    dma_tile(kbeg, 0);
    if (ATT_NBUF == 3 && kbeg + 64 < kend) { dma_tile(kbeg + 64, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int kv0 = kbeg, cur = 0;
    // PIPE (no key mask, no causal mask, no dropout: the Whisper encoder): the two matrix products of neighbouring
    // 32-key sub-tiles are software-pipelined inside the wave.  While the softmax of sub-tile j runs on the vector
    // ALU, the matrix pipe works on S(j+1) = K.Q^T, issued just before it, and then on the P.V of j: one
    // instruction stream with an MFMA every few vector instructions (the order is pinned chunk by chunk), instead of
    // a product, then its softmax, then the next product, each waiting for the one before.  Costs a second score
    // accumulator (16 registers) and a third K/V buffer: the next tile must be resident before this one is finished.
    if constexpr (PIPE) {
        static_assert(!PIPE || (ATT_NBUF == 3 && !DROP), "the pipelined loop needs the three-buffer ring");
        const int L = (kend - kbeg + 63) >> 6;               // tiles; only the last can reach past Tk
        float ps0 = 0.f, ps1 = 0.f;
        // the reference maximum in the two forms the loop uses, updated only when the reference moves: the softmax
        // subtracts nmsc = -mrun*scale*log2(e) inside the exponent's fma, and a new maximum is compared with thr
        float nmsc = 0.f, thr = -INFINITY;
        auto pipe_step = [&](f32x16& sc, f32x16& sn, const bf16_t* Kn, int hn, const bf16_t* Vc, int hc, int tile0) {
            // loads first: the K fragments of the next sub-tile and the first two V fragments of this one land
            // while the row maximum is found (past the last tile the scores of a stale buffer are computed and dropped)
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)&Kn[k_off(32 * hn + r, 2 * ks + hf)];
            bf16x8 vf0 = read_tr_frag(Vc, hc, 0, 0, lane), vf1 = read_tr_frag(Vc, hc, 0, 1, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (tile0 + 64 > Tk) {                            // block-uniform: the tile that reaches past Tk
                const uint32_t dead = (uint32_t)(dead_keys(nullptr, tile0, Tk, lane) >> (32 * hc));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((dead >> ((i & 3) + 8 * (i >> 2) + 4 * hf)) & 1) sc[i] = -INFINITY;
            }
            float mx = vmax3(sc[0], sc[1], sc[2]);
            float my = vmax3(sc[3], sc[4], sc[5]);
            mx = vmax3(mx, sc[6], sc[7]);
            my = vmax3(my, sc[8], sc[9]);
            mx = vmax3(mx, sc[10], sc[11]);
            my = vmax3(my, sc[12], sc[13]);
            mx = vmax3(mx, sc[14], sc[15]);
            {   // ... and across the wave's halves: one v_permlane32_swap, one v_max3 (no canonicalising v_max)
                const float m2 = vmax3(mx, my, my);
                const unsigned u = __builtin_bit_cast(unsigned, m2);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const float a0 = __builtin_bit_cast(float, (unsigned)sw[0]), a1 = __builtin_bit_cast(float, (unsigned)sw[1]);
                mx = vmax3(a0, a1, a1);
            }
            if (__any(mx > thr)) {                            // rare after the first tiles (lazy reference point)
                const float mnew = fmaxf(mrun, mx);
                const float alpha = mnew > mrun ? exp2f((mrun - mnew) * sc2) : 1.0f;
                mrun = mnew;
                thr = mrun + lazy;
                nmsc = mrun == -INFINITY ? 0.f : -(mrun * sc2);
                lsum *= alpha; ps0 *= alpha; ps1 *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sn[i] = 0.f;
#define ATT_E(i) sc[i] = __builtin_amdgcn_exp2f(fmaf(sc[i], sc2, nmsc))
// row sums as opaque single adds: left to itself the compiler packs the two chains into one serial chain of
// v_pk_add_f32 and sinks it behind the last MFMA, where nothing overlaps it
#define ATT_ADD(acc, i) asm("v_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(sc[i]))
#define ATT_QK(ks) sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], sn, 0, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            ATT_QK(0); ATT_E(0); ATT_E(1); ATT_E(2); ATT_E(3);
            __builtin_amdgcn_sched_barrier(0);
            ATT_QK(1); ATT_E(4); ATT_E(5); ATT_E(6); ATT_E(7);
            __builtin_amdgcn_sched_barrier(0);
            ATT_QK(2);
            const bf16x8 pf0 = acc_to_bf16(sc, 0);
            ATT_E(8); ATT_E(9);
            __builtin_amdgcn_sched_barrier(0);
            ATT_QK(3); ATT_E(10); ATT_E(11); ATT_E(12); ATT_E(13);
            __builtin_amdgcn_sched_barrier(0);
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf0, oacc[0], 0, 0, 0);
            ATT_E(14); ATT_E(15);
            ATT_ADD(ps0, 0); ATT_ADD(ps1, 1); ATT_ADD(ps0, 2); ATT_ADD(ps1, 3);
            __builtin_amdgcn_sched_barrier(0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf0, oacc[1], 0, 0, 0);
            const bf16x8 pf1 = acc_to_bf16(sc, 1);
            vf0 = read_tr_frag(Vc, hc, 1, 0, lane); vf1 = read_tr_frag(Vc, hc, 1, 1, lane);
            ATT_ADD(ps0, 4); ATT_ADD(ps1, 5); ATT_ADD(ps0, 6); ATT_ADD(ps1, 7);
            __builtin_amdgcn_sched_barrier(0);
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf1, oacc[0], 0, 0, 0);
            ATT_ADD(ps0, 8); ATT_ADD(ps1, 9); ATT_ADD(ps0, 10); ATT_ADD(ps1, 11);
            __builtin_amdgcn_sched_barrier(0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf1, oacc[1], 0, 0, 0);
            ATT_ADD(ps0, 12); ATT_ADD(ps1, 13); ATT_ADD(ps0, 14); ATT_ADD(ps1, 15);
            __builtin_amdgcn_sched_barrier(0);
#undef ATT_E
#undef ATT_QK
#undef ATT_ADD
        };
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&Kbuf[0][k_off(r, 2 * ks + hf)], qf[ks], s0, 0, 0, 0);
        // one tile: buffer indices are compile-time constants (the ring is walked by an unrolled-by-three loop), so
        // every LDS address of the fragment reads is a loop-invariant register plus an immediate offset
#define ATT_PIPE_TILE(CUR)                                                                                     \
        {                                                                                                          \
            pipe_step(s0, s1, Kbuf[CUR], 1, Vbuf[CUR], 0, kv0);                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the next tile (issued a tile ago) is in LDS */      \
            __syncthreads();                    /* ... for every wave, and the tile before this one is free */     \
            if (kv0 + 128 < kend) dma_tile(kv0 + 128, ((CUR) + 2) % 3);                                            \
            pipe_step(s1, s0, Kbuf[((CUR) + 1) % 3], 0, Vbuf[CUR], 1, kv0);                                        \
            kv0 += 64;                                                                                             \
        }
        for (int t = 0; t < L; t += 3) {
            ATT_PIPE_TILE(0)
            if (t + 1 < L) ATT_PIPE_TILE(1)
            if (t + 2 < L) ATT_PIPE_TILE(2)
        }
#undef ATT_PIPE_TILE
        lsum += ps0 + ps1;
    }
    if constexpr (!PIPE)
    for (; kv0 < kend; kv0 += 64, cur = (cur + 1 == ATT_NBUF ? 0 : cur + 1)) {
        const bf16_t* Ks = Kbuf[cur];
        const bf16_t* Vs = Vbuf[cur];
        // a later tile lands in the buffer that was last read one barrier ago while this one is computed: with
        // three buffers the tile after next, so a load has two tiles of arithmetic to arrive in
        const int ahead = 64 * (ATT_NBUF - 1);
        const bool issue = kv0 + ahead < kend;
        if (issue) dma_tile(kv0 + ahead, cur == 0 ? ATT_NBUF - 1 : cur - 1);

        // 32 keys at a time (rolled): half the score registers and a quarter of the K / V fragments are live at
        // once, which is what lets four waves share a SIMD (the 64-key form needed ~250 VGPRs: two waves).
#pragma unroll ATT_UNROLL
        for (int st = 0; st < 2; ++st) {
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)&Ks[k_off(32 * st + r, 2 * ks + hf)];
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
            }
            // sacc[reg] = S[q][key = kv0 + 32 st + (reg&3) + 8 (reg>>2) + 4 hf]   (raw, unscaled)
            if (mb || kv0 + 64 > Tk) {                        // block-uniform: only masked / tail tiles pay for this
                const uint32_t dead = (uint32_t)(dead_keys(mb, kv0, Tk, lane) >> (32 * st));
                if (dead) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((dead >> ((i & 3) + 8 * (i >> 2) + 4 * hf)) & 1) sacc[i] = -INFINITY;
                }
            }
            if (causal && kv0 + 63 > qt * 128 + coff) {      // block-uniform: only tiles that touch the diagonal
                const int last = q + coff - kv0 - 32 * st;    // last visible key of this lane's row, sub-tile-relative
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) + 4 * hf > last) sacc[i] = -INFINITY;
            }
#if ATT_MAX3
            // MFMA results are never signalling NaNs: skip the canonicalising v_max the compiler puts before fmaxf
            float mx = vmax3(sacc[0], sacc[1], sacc[2]);
            float my = vmax3(sacc[3], sacc[4], sacc[5]);
            mx = vmax3(mx, sacc[6], sacc[7]);
            my = vmax3(my, sacc[8], sacc[9]);
            mx = vmax3(mx, sacc[10], sacc[11]);
            my = vmax3(my, sacc[12], sacc[13]);
            mx = vmax3(mx, sacc[14], sacc[15]);
            mx = vmax3(mx, my, my);
#else
            float mx = sacc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sacc[i]);
#endif
            mx = max_across_halves(mx);
            // Lazy reference point: softmax does not care which constant is subtracted, so the accumulated sums are
            // only rescaled when some row's maximum has risen by more than ATT_LAZY_LOG2 binades over its reference
            // (probabilities then reach at most 2^ATT_LAZY_LOG2 before normalisation: nothing for fp32 sums or
            // bf16's exponent range).  After the first few tiles this branch is rarely taken.
            if (__any(mx > mrun + lazy)) {
                const float mnew = fmaxf(mrun, mx);
                const float alpha = mnew > mrun ? exp2f((mrun - mnew) * sc2) : 1.0f;
                mrun = mnew;
                lsum *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
            }
            const float msc = mrun == -INFINITY ? 0.f : mrun * sc2;
#if ATT_PACKED
            const f32x2 sc2v = {sc2, sc2}, mscv = {-msc, -msc};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 arg = __builtin_elementwise_fma((f32x2){sacc[i], sacc[i + 1]}, sc2v, mscv);
                const f32x2 pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                sacc[i] = pv[0]; sacc[i + 1] = pv[1];
                ps2 += pv;
            }
            lsum += ps2[0] + ps2[1];                          // the normaliser is that of the un-dropped softmax
#else
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                sacc[i] = __builtin_amdgcn_exp2f(fmaf(sacc[i], sc2, -msc));
                sacc[i + 1] = __builtin_amdgcn_exp2f(fmaf(sacc[i + 1], sc2, -msc));
                ps0 += sacc[i]; ps1 += sacc[i + 1];
            }
            lsum += ps0 + ps1;
#endif
            if (DROP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t bits[4];
                    avsr_rand4(seed, avsr_attn_counter(row, tk4, kv0 + 32 * st + 8 * g + 4 * hf), bits);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sacc[4 * g + e] = avsr_bits_keep(bits[e], pdrop) ? sacc[4 * g + e] * keep_scale : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = acc_to_bf16(sacc, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr_frag(Vs, st, s2, db, lane), pf, oacc[db], 0, 0, 0);
            }
        }
        // this wave's pieces of the NEXT tile are in LDS (the four of the tile after it may still be in flight)
        if (ATT_NBUF == 3 && issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (nparts > 1) {
        // A wave's partial result is its 32 accumulator registers, its reference maximum and its partial row sum, one
        // 256-byte row of the workspace per register.  The shares are written THROUGH the caches (system-scope
        // stores) and read back past them (system-scope loads), so publishing them takes no write-back or
        // invalidation of the XCD's whole L2, which a device-scope fence per workgroup would (measured: 2.4x the
        // kernel time).  Order: stores acknowledged (vmcnt 0) -> workgroup barrier -> one ticket per workgroup; the
        // workgroup that draws the last ticket of its item sums all shares and carries on to the epilogue.
        const int sidx = xcd * (per_xcd * qtiles - whole_per_xcd) + (item - whole_per_xcd);
        float* wp = part_ws + (((long)sidx * nparts + part) * 4 + wave) * (34 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __hip_atomic_store(&wp[i * 64], oacc[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&wp[(16 + i) * 64], oacc[1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&wp[32 * 64], mrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&wp[33 * 64], lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) last_part = __hip_atomic_fetch_add(&tickets[sidx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nparts - 1;
        __syncthreads();
        if (!last_part) return;
        const float* rp = part_ws + (((long)sidx * nparts) * 4 + wave) * (34 * 64) + lane;
        float mall = -INFINITY;
        for (int p = 0; p < nparts; ++p)
            mall = fmaxf(mall, __hip_atomic_load(&rp[(long)p * 4 * 34 * 64 + 32 * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
        lsum = 0.f;
        for (int p = 0; p < nparts; ++p) {
            const float* pp = rp + (long)p * 4 * 34 * 64;
            const float mp = __hip_atomic_load(&pp[32 * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float a = mp == -INFINITY ? 0.f : exp2f((mp - mall) * sc2);
            lsum = fmaf(a, __hip_atomic_load(&pp[33 * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), lsum);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oacc[0][i] = fmaf(a, __hip_atomic_load(&pp[i * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), oacc[0][i]);
                oacc[1][i] = fmaf(a, __hip_atomic_load(&pp[(16 + i) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), oacc[1][i]);
            }
        }
        mrun = mall;
        if (t == 0) tickets[sidx] = 0;                        // the next launch on this workspace starts from zero
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

// ---------------------------------------------------------------- fp8 ----
// Forward with both products on v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3 on gfx950): Q, K, V arrive as bf16 and are
// rounded to fp8 on the way into registers / LDS; P is rounded to fp8 after the fp32 softmax (scaled by 128 so
// that probabilities down to 2^-16 of the row maximum survive e4m3's 2^-9 subnormal floor; 128 <= 448 = max
// finite).  Same transposed scheme as the bf16 kernel: S^T = K.Q^T puts a query in a lane, and the S^T accumulator
// converted in place is the B operand of O^T = V^T.P^T.  V is staged TRANSPOSED in LDS ([d][key] bytes, row
// stride 68) so the A operand's eight keys - in the accumulator's register order - are two aligned 4-byte reads.
#define K8LD 72
#define V8LD 68
__device__ __forceinline__ long pack_fp8x8(const float* v) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long bf16x8_to_fp8(const bf16x8& x) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    return pack_fp8x8(v);
}

__global__ __launch_bounds__(256) void attn_fwd_fp8_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                           long ldk, const bf16_t* __restrict__ V, long ldv,
                                                           bf16_t* __restrict__ O, long ldo, const uint8_t* __restrict__ mask,
                                                           float* __restrict__ lse, int Tq, int Tk, int H, float scale, int B,
                                                           int causal) {
    __shared__ __attribute__((aligned(16))) uint8_t K8[2][64 * K8LD];
    __shared__ __attribute__((aligned(16))) uint8_t Vt8[2][64 * V8LD];
    const int qtiles = (Tq + 127) >> 7, npairs = B * H;
    const int per_xcd = (npairs + 7) >> 3;                         // same XCD-aware 1-D grid as the bf16 kernel
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = xcd * per_xcd + slot / qtiles, qt = slot - (slot / qtiles) * qtiles;
    if (pair >= npairs || slot >= per_xcd * qtiles) return;
    const int b = pair / H, h = pair - b * H;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int q = qt * 128 + wave * 32 + r;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    long qf[4];
    {
        const bf16_t* qp = Q + ((long)b * Tq + (q < Tq ? q : 0)) * ldq + h * HD + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = bf16x8_to_fp8(q < Tq ? *(const bf16x8*)(qp + ks * 16) : zero8);
    }
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float mrun = -INFINITY, lsum = 0.f;
    const float sc2 = scale * LOG2E;
    const int lc = t & 7, lr = t >> 3;
    const bf16_t* Kb = K + (long)b * Tk * ldk + h * HD;
    const bf16_t* Vb = V + (long)b * Tk * ldv + h * HD;
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;

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
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = lr + 32 * i;
            *(long*)&K8[buf][row * K8LD + lc * 8] = bf16x8_to_fp8(kreg[i]);
            const unsigned long vb = (unsigned long)bf16x8_to_fp8(vreg[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt8[buf][(8 * lc + e) * V8LD + row] = (uint8_t)(vb >> (8 * e));
        }
    };
    const int coff = Tk - Tq;
    const int kend = causal ? min(Tk, qt * 128 + 128 + coff) : Tk;
    prefetch(0);
    stage(0);
    __syncthreads();
    for (int kv0 = 0, cur = 0; kv0 < kend; kv0 += 64, cur ^= 1) {
        const uint8_t* Ks = K8[cur];
        const uint8_t* Vs = Vt8[cur];
        const bool more = kv0 + 64 < kend;
        if (more) prefetch(kv0 + 64);

        f32x16 sacc[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[st][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long kf = *(const long*)&Ks[(32 * st + r) * K8LD + (2 * ks + hf) * 8];
                sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], sacc[st], 0, 0, 0);
            }
        }
        if (mb || kv0 + 64 > Tk) {
            const uint64_t dead = dead_keys(mb, kv0, Tk, lane);
            if (dead) {
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((dead >> (32 * st + (i & 3) + 8 * (i >> 2) + 4 * hf)) & 1) sacc[st][i] = -INFINITY;
            }
        }
        if (causal && kv0 + 63 > qt * 128 + coff) {
            const int last = q + coff - kv0;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (32 * st + (i & 3) + 8 * (i >> 2) + 4 * hf > last) sacc[st][i] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sacc[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[1][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__any(mx > mrun)) {
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
            for (int i = 0; i < 16; ++i) { const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[st][i], sc2, -msc)); sacc[st][i] = pv * 128.f; ps += pv; }
        lsum += ps;

#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = sacc[st][8 * s2 + j];
                const long pf = pack_fp8x8(pv);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const uint8_t* vp = &Vs[(32 * db + r) * V8LD + 32 * st + 16 * s2 + 4 * hf];
                    const unsigned lo = *(const unsigned*)vp, hi = *(const unsigned*)(vp + 8);
                    const long vf = (long)(((unsigned long)hi << 32) | lo);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        if (more) stage(cur ^ 1);
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / (128.f * lsum) : 0.f;
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

// dQ for 128 query rows per workgroup (one per lane pair), looping over 64-key tiles; also writes the
// per-row D = dO.O the dK/dV kernel needs.  Transposed like the forward: S^T = K.Q^T and dP^T = V.dO^T put
// a query in a lane, and the dS^T accumulator is the B operand of dQ^T = K^T.dS^T.
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(
    const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K, long ldk, const bf16_t* __restrict__ V, long ldv,
    const bf16_t* __restrict__ O, long ldo, const bf16_t* __restrict__ dO, long lddo, bf16_t* __restrict__ dQ, long lddq,
    const uint8_t* __restrict__ mask, const float* __restrict__ lse, float* __restrict__ Dout, int Tq, int Tk, int H,
    float scale, float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * 64];
    const int b = blockIdx.z, h = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int q = blockIdx.x * 128 + wave * 32 + r;
    const bool qv = q < Tq;
    const long bh = (long)b * H + h;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 qf[4], dof[4];
    float dsum = 0.f;
    {
        const long qrow = (long)b * Tq + (qv ? q : 0);
        const bf16_t* qp = Q + qrow * ldq + h * HD + hf * 8;
        const bf16_t* dp = dO + qrow * lddo + h * HD + hf * 8;
        const bf16_t* op = O + qrow * ldo + h * HD + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = qv ? *(const bf16x8*)(qp + ks * 16) : zero8;
            dof[ks] = qv ? *(const bf16x8*)(dp + ks * 16) : zero8;
            const bf16x8 of = qv ? *(const bf16x8*)(op + ks * 16) : zero8;
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum = fmaf((float)dof[ks][j], (float)of[j], dsum);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float l2 = (qv ? lse[bh * Tq + q] : INFINITY) * LOG2E;       // +inf for dead rows: every p is 0
    if (qv && hf == 0) Dout[bh * Tq + q] = dsum;
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
    const float sc2 = scale * LOG2E;
    const float keep_scale = DROP ? 1.0f / (1.0f - pdrop) : 1.0f;
    const int tk4 = (Tk + 3) >> 2;
    const uint64_t row = (uint64_t)bh * Tq + q;
    const int lc = t & 7, lr = t >> 3;
    const bf16_t* Kb = K + (long)b * Tk * ldk + h * HD;
    const bf16_t* Vb = V + (long)b * Tk * ldv + h * HD;
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;

    for (int kv0 = 0; kv0 < Tk; kv0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kv0 + lr + 32 * i;
            const bool ok = key < Tk;
            const bf16x8 kk = ok ? *(const bf16x8*)(Kb + (long)key * ldk + lc * 8) : zero8;
            const bf16x8 vv = ok ? *(const bf16x8*)(Vb + (long)key * ldv + lc * 8) : zero8;
            *(bf16x8*)&Ks[k_off(lr + 32 * i, lc)] = kk;
            *(bf16x8*)&Kt[v_off(lr + 32 * i, lc)] = kk;
            *(bf16x8*)&Vs[k_off(lr + 32 * i, lc)] = vv;
        }
        __syncthreads();
        const uint64_t dead = dead_keys(mb, kv0, Tk, lane);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)&Ks[k_off(32 * st + r, 2 * ks + hf)];
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
                const bf16x8 vf = *(const bf16x8*)&Vs[k_off(32 * st + r, 2 * ks + hf)];
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], pacc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t bits[4];
                if (DROP) avsr_rand4(seed, avsr_attn_counter(row, tk4, kv0 + 32 * st + 8 * g + 4 * hf), bits);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e;
                    const bool dd = (dead >> (32 * st + 8 * g + 4 * hf + e)) & 1;
                    const float p = dd ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[i], sc2, -l2));
                    float dp = pacc[i];
                    if (DROP) dp = avsr_bits_keep(bits[e], pdrop) ? dp * keep_scale : 0.f;
                    sacc[i] = p * (dp - dsum) * scale;                     // dS^T
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 dsf = acc_to_bf16(sacc, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr_frag(Kt, st, s2, db, lane), dsf, dq[db], 0, 0, 0);
            }
        }
    }
    if (qv) {
        bf16_t* op = dQ + ((long)b * Tq + q) * lddq + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)dq[db][4 * g + e];
                *(bf16x4*)(op + 32 * db + 8 * g + 4 * hf) = o;
            }
    }
}

// dK, dV for 128 keys per workgroup (a key per lane pair), looping over 64-query tiles:
// S = Q.K^T and dP = dO.V^T put a key in a lane; the dropped-P and dS accumulators are the B operands
// of dV^T = dO^T.P and dK^T = Q^T.dS.
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(
    const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K, long ldk, const bf16_t* __restrict__ V, long ldv,
    const bf16_t* __restrict__ dO, long lddo, bf16_t* __restrict__ dK, long lddk, bf16_t* __restrict__ dV, long lddv,
    const uint8_t* __restrict__ mask, const float* __restrict__ lse, const float* __restrict__ Dsum, int Tq, int Tk, int H,
    float scale, float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
    __shared__ __attribute__((aligned(16))) bf16_t Qs[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Os[64 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t Ot[64 * 64];
    __shared__ float Ls[64], Ds[64];
    const int b = blockIdx.z, h = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int key = blockIdx.x * 128 + wave * 32 + r;
    const bool kvalid = key < Tk;
    const long bh = (long)b * H + h;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool dead = !kvalid || (mask && mask[(long)b * Tk + key]);

    bf16x8 kf[4], vf[4];
    {
        const long krow = (long)b * Tk + (kvalid ? key : 0);
        const bf16_t* kp = K + krow * ldk + h * HD + hf * 8;
        const bf16_t* vp = V + krow * ldv + h * HD + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = kvalid ? *(const bf16x8*)(kp + ks * 16) : zero8;
            vf[ks] = kvalid ? *(const bf16x8*)(vp + ks * 16) : zero8;
        }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    const float sc2 = scale * LOG2E;
    const float keep_scale = DROP ? 1.0f / (1.0f - pdrop) : 1.0f;
    const int tk4 = (Tk + 3) >> 2;
    const int lc = t & 7, lr = t >> 3;
    const bf16_t* Qb = Q + (long)b * Tq * ldq + h * HD;
    const bf16_t* Ob = dO + (long)b * Tq * lddo + h * HD;

    for (int q0 = 0; q0 < Tq; q0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = q0 + lr + 32 * i;
            const bool ok = qq < Tq;
            const bf16x8 a = ok ? *(const bf16x8*)(Qb + (long)qq * ldq + lc * 8) : zero8;
            const bf16x8 g = ok ? *(const bf16x8*)(Ob + (long)qq * lddo + lc * 8) : zero8;
            *(bf16x8*)&Qs[k_off(lr + 32 * i, lc)] = a;
            *(bf16x8*)&Qt[v_off(lr + 32 * i, lc)] = a;
            *(bf16x8*)&Os[k_off(lr + 32 * i, lc)] = g;
            *(bf16x8*)&Ot[v_off(lr + 32 * i, lc)] = g;
        }
        if (t < 64) {
            const int qq = q0 + t;
            Ls[t] = (qq < Tq ? lse[bh * Tq + qq] : INFINITY) * LOG2E;
            Ds[t] = qq < Tq ? Dsum[bh * Tq + qq] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *(const bf16x8*)&Qs[k_off(32 * st + r, 2 * ks + hf)];
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sacc, 0, 0, 0);
                const bf16x8 ga = *(const bf16x8*)&Os[k_off(32 * st + r, 2 * ks + hf)];
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[ks], pacc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ql = 32 * st + (i & 3) + 8 * (i >> 2) + 4 * hf;
                const float p = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[i], sc2, -Ls[ql]));
                float pd = p, dp = pacc[i];
                if (DROP) {
                    const bool keep = avsr_attn_keep(seed, (uint64_t)bh * Tq + q0 + ql, tk4, key, pdrop);
                    pd = keep ? p * keep_scale : 0.f;
                    dp = keep ? dp * keep_scale : 0.f;
                }
                sacc[i] = pd;                                               // dropped P
                pacc[i] = p * (dp - Ds[ql]) * scale;                        // dS
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = acc_to_bf16(sacc, s2), dsf = acc_to_bf16(pacc, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr_frag(Ot, st, s2, db, lane), pf, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr_frag(Qt, st, s2, db, lane), dsf, dk[db], 0, 0, 0);
                }
            }
        }
    }
    if (kvalid) {
        bf16_t* kp = dK + ((long)b * Tk + key) * lddk + h * HD;
        bf16_t* vp = dV + ((long)b * Tk + key) * lddv + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 a, c;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (bf16_t)dk[db][4 * g + e]; c[e] = (bf16_t)dv[db][4 * g + e]; }
                *(bf16x4*)(kp + 32 * db + 8 * g + 4 * hf) = a;
                *(bf16x4*)(vp + 32 * db + 8 * g + 4 * hf) = c;
            }
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
                                                            int Tq, int Tk, int H, float scale, float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta,
                                                            int causal) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
    __shared__ float Ks[64 * KLD], Vs[64 * KLD], Qs[32 * KLD];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_rows_f32<T>(Qs, Q + (long)b * Tq * ldq + h * HD, ldq, q0, 32, Tq);
    const int coff = Tk - Tq;                                 // causal: query i sees keys <= i + coff
    const int kend = causal ? min(Tk, q0 + 32 + coff) : Tk;
    float m[8], l[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = -INFINITY; l[i] = 0.f; o[i] = 0.f; }
    const uint8_t* mb = mask ? mask + (long)b * Tk : nullptr;
    const float keep_scale = pdrop > 0.f ? 1.0f / (1.0f - pdrop) : 1.0f;
    for (int kv0 = 0; kv0 < kend; kv0 += 64) {
        __syncthreads();
        stage_rows_f32<T>(Ks, K + (long)b * Tk * ldk + h * HD, ldk, kv0, 64, Tk);
        stage_rows_f32<T>(Vs, V + (long)b * Tk * ldv + h * HD, ldv, kv0, 64, Tk);
        __syncthreads();
        const int key = kv0 + lane;
        const bool masked = key >= Tk || (mb && mb[key < Tk ? key : 0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qr = wave * 8 + i, qg = q0 + qr;
            const bool dead = masked || (causal && key > qg + coff);
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
                const uint64_t row = ((uint64_t)b * H + h) * Tq + qg;
                p = (!dead && qg < Tq && avsr_attn_keep(seed, row, (Tk + 3) >> 2, key, pdrop)) ? p * keep_scale : 0.f;
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
                                                               float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
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
                const uint64_t row = ((uint64_t)b * H + h) * Tq + qg;
                dp = (!dead && qg < Tq && avsr_attn_keep(seed, row, (Tk + 3) >> 2, key, pdrop)) ? dp * keep_scale : 0.f;
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
                                                                float pdrop, uint64_t seed_in, const uint64_t* __restrict__ seed_delta) {
    const uint64_t seed = avsr_run_seed(seed_in, seed_delta);   // by-value seed + the optional device word
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
                const uint64_t row = ((uint64_t)b * H + h) * Tq + qg;
                const bool keep = !dead && avsr_attn_keep(seed, row, (Tk + 3) >> 2, key, pdrop);
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
// How the forward grid is cut (see the kernel): per XCD, `whole` items run as one workgroup and the remaining `rest`
// as `nsplit` workgroups each.  Splitting pays when the last round of equal workgroups would leave at least half of
// the chip's resident-workgroup slots empty; it needs the caller's workspace, no causal mask, and at least four key
// tiles per share.
struct AttnSplit { int whole, rest, nsplit; };
// Invariant the merge relies on for speed, not for correctness of a single share: every share of an item has the same
// blockIdx & 7, i.e. (under the round-robin placement of workgroups over 8 XCDs) the same XCD and L2.  The shares are
// exchanged with system-scope stores and loads, which go through to memory on any placement; the plan below is only
// offered on the 256-CU / 8-XCD layout the id mapping was measured on, any other device or partition runs unsplit.
// A stream created with a CU mask places workgroups on a subset of the CUs: ids with equal blockIdx & 7 need not share
// an XCD then, so such a stream runs unsplit.
static bool stream_sees_every_cu(hipStream_t s) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // 256 CUs
    if (hipExtStreamGetCUMask(s, 8, mask) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (int i = 0; i < 8; ++i)
        if (mask[i] != 0xffffffffu) return false;
    return true;
}

static AttnSplit attn_split_plan(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int causal, bool dropout) {
    const int per_xcd = (int)((B * H + 7) / 8), qtiles = (int)((Tq + 127) / 128), items = per_xcd * qtiles;
    AttnSplit pl{items, 0, 1};
    constexpr int MAXDEV = 64;
    static int cus_of[MAXDEV];                                // per device: CU count and resident workgroups per CU
    static int occ_of[MAXDEV][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return pl;
    if (!cus_of[dev]) {
        int n = 0;
        cus_of[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
    }
    if (!occ_of[dev][dropout]) {
        int n = 0;
        const hipError_t e = dropout ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<true, false>, 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<false, true>, 256, 0);
        occ_of[dev][dropout] = (e == hipSuccess && n > 0) ? n : ATT_OCC;
    }
    const int cus = cus_of[dev];
    if (cus != 256 || causal) return pl;                      // not the 8-XCD layout (or a partition of it): no split
    const int slots = (cus / 8) * occ_of[dev][dropout];       // resident workgroups per XCD
    if (slots <= 0) return pl;
    const int rem = items % slots;
    if (rem == 0) return pl;
    const int ns = std::min<int>(std::min<int>(4, slots / rem), (int)((Tk + 63) / 64 / 4));
    if (ns < 2) return pl;
    pl.whole = items - rem; pl.rest = rem; pl.nsplit = ns;
    return pl;
}

extern "C" int64_t avsr_attn_fwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk, float p_drop, int causal) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
    const AttnSplit pl = attn_split_plan(B, H, Tq, Tk, causal, p_drop > 0.f);
    if (pl.nsplit < 2) return 0;
    const int64_t nitems = 8LL * pl.rest;
    return ((nitems * 4 + 255) / 256) * 256 + nitems * pl.nsplit * 4 * 34 * 64 * 4;
}

extern "C" int avsr_attn_fwd_sd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                                int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                                float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                                float p_drop, uint64_t seed, const uint64_t* seed_delta, int causal, void* workspace,
                                int64_t workspace_bytes) {
    if (!Q || !K || !V || !O || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return AVSR_EINVAL;
    if (causal && Tk < Tq) return AVSR_EINVAL;
    if (use_mfma < 0 || use_mfma > 2) return AVSR_EINVAL;
    if (head_dim != HD) return AVSR_EUNSUPPORTED;
    if (B > 65535 || H > 65535) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma) {
        if (dtype != AVSR_BF16) return AVSR_EUNSUPPORTED;
        if (p_drop < 0.f || p_drop >= 1.f) return AVSR_EINVAL;
        if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return AVSR_EUNSUPPORTED;
        if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) % 16 || (uintptr_t)O % 8) return AVSR_EUNSUPPORTED;
        if (B * H * ((Tq + 127) / 128) > 0x0fffffff) return AVSR_EUNSUPPORTED;
        // the K / V tiles are addressed by 32-bit byte offsets from the tile's first key
        if (64 * std::max(ldk, ldv) * 2 > 0x7fffffffLL) return AVSR_EUNSUPPORTED;
        const int per_xcd = (int)((B * H + 7) / 8), qtiles = (int)((Tq + 127) / 128);
        dim3 grid((unsigned)(8 * per_xcd * qtiles));
        if (use_mfma == 2) {                                  // fp8 products; no dropout form
            if (p_drop != 0.f) return AVSR_EUNSUPPORTED;
            hipLaunchKernelGGL(attn_fwd_fp8_kernel, grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq, (const bf16_t*)K,
                               (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse, (int)Tq, (int)Tk,
                               (int)H, scale, (int)B, causal);
            return avsr_launch_status();
        }
        AttnSplit pl = attn_split_plan(B, H, Tq, Tk, causal, p_drop > 0.f);
        float* part_ws = nullptr;
        int* tickets = nullptr;
        if (pl.nsplit >= 2 && workspace && ((uintptr_t)workspace % 256) == 0 && stream_sees_every_cu(s) &&
            workspace_bytes >= avsr_attn_fwd_workspace_bytes(B, H, Tq, Tk, p_drop, causal)) {
            tickets = (int*)workspace;
            part_ws = (float*)((char*)workspace + ((8LL * pl.rest * 4 + 255) / 256) * 256);
            grid = dim3((unsigned)(8 * (pl.whole + pl.rest * pl.nsplit)));
        } else {
            pl = AttnSplit{per_xcd * qtiles, 0, 1};
        }
        if (p_drop > 0.f)
            hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, false>), grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq,
                               (const bf16_t*)K, (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse,
                               (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta, (int)B, causal, pl.whole, pl.nsplit, part_ws,
                               tickets);
        else if (!key_mask && !causal)                          // the software-pipelined form
            hipLaunchKernelGGL((attn_fwd_mfma_kernel<false, true>), grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq,
                               (const bf16_t*)K, (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse,
                               (int)Tq, (int)Tk, (int)H, scale, 0.f, seed, seed_delta, (int)B, causal, pl.whole, pl.nsplit, part_ws,
                               tickets);
        else
            hipLaunchKernelGGL((attn_fwd_mfma_kernel<false, false>), grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq,
                               (const bf16_t*)K, (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse,
                               (int)Tq, (int)Tk, (int)H, scale, 0.f, seed, seed_delta, (int)B, causal, pl.whole, pl.nsplit, part_ws,
                               tickets);
        return avsr_launch_status();
    }
    dim3 grid((unsigned)((Tq + 31) / 32), (unsigned)H, (unsigned)B);
    if (dtype == AVSR_F32)
        hipLaunchKernelGGL(attn_fwd_valu_kernel<float>, grid, dim3(256), 0, s, (const float*)Q, (long)ldq, (const float*)K,
                           (long)ldk, (const float*)V, (long)ldv, (float*)O, (long)ldo, key_mask, lse, (int)Tq, (int)Tk,
                           (int)H, scale, p_drop, seed, seed_delta, causal);
    else if (dtype == AVSR_BF16)
        hipLaunchKernelGGL(attn_fwd_valu_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq,
                           (const bf16_t*)K, (long)ldk, (const bf16_t*)V, (long)ldv, (bf16_t*)O, (long)ldo, key_mask, lse,
                           (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta, causal);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_attn_fwd_ws(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                                int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                                float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                                float p_drop, uint64_t seed, int causal, void* workspace, int64_t workspace_bytes) {
    return avsr_attn_fwd_sd(stream, dtype, use_mfma, Q, ldq, K, ldk, V, ldv, O, ldo, key_mask, lse, B, H, Tq, Tk, head_dim,
                            scale, p_drop, seed, nullptr, causal, workspace, workspace_bytes);
}

extern "C" int avsr_attn_fwd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                             int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                             float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                             float p_drop, uint64_t seed, int causal) {
    return avsr_attn_fwd_sd(stream, dtype, use_mfma, Q, ldq, K, ldk, V, ldv, O, ldo, key_mask, lse, B, H, Tq, Tk, head_dim,
                            scale, p_drop, seed, nullptr, causal, nullptr, 0);
}

extern "C" int avsr_attn_bwd_sd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K, int64_t ldk,
                                const void* V, int64_t ldv, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                                void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv,
                                const uint8_t* key_mask, const float* lse, float* d_ws, int64_t B, int64_t H, int64_t Tq,
                                int64_t Tk, int64_t head_dim, float scale, float p_drop, uint64_t seed,
                                const uint64_t* seed_delta) {
    if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !lse || !d_ws) return AVSR_EINVAL;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return AVSR_EINVAL;
    if (head_dim != HD || B > 65535 || H > 65535) return AVSR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma) {
        if (dtype != AVSR_BF16) return AVSR_EUNSUPPORTED;
        if (p_drop < 0.f || p_drop >= 1.f) return AVSR_EINVAL;
        if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 4 || lddk % 4 || lddv % 4) return AVSR_EUNSUPPORTED;
        if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO) % 16 ||
            ((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) % 8)
            return AVSR_EUNSUPPORTED;
        const dim3 g1((unsigned)((Tq + 127) / 128), (unsigned)H, (unsigned)B), g2((unsigned)((Tk + 127) / 128), (unsigned)H, (unsigned)B);
#define RUN_MFMA(DROP)                                                                                                   \
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<DROP>, g1, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq, (const bf16_t*)K, \
                       (long)ldk, (const bf16_t*)V, (long)ldv, (const bf16_t*)O, (long)ldo, (const bf16_t*)dO, (long)lddo, \
                       (bf16_t*)dQ, (long)lddq, key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta); \
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<DROP>, g2, dim3(256), 0, s, (const bf16_t*)Q, (long)ldq, (const bf16_t*)K, \
                       (long)ldk, (const bf16_t*)V, (long)ldv, (const bf16_t*)dO, (long)lddo, (bf16_t*)dK, (long)lddk,       \
                       (bf16_t*)dV, (long)lddv, key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta)
        if (p_drop > 0.f) { RUN_MFMA(true); } else { RUN_MFMA(false); }
#undef RUN_MFMA
        return avsr_launch_status();
    }
    dim3 gq((unsigned)((Tq + 31) / 32), (unsigned)H, (unsigned)B), gk((unsigned)((Tk + 63) / 64), (unsigned)H, (unsigned)B);
#define RUN(T)                                                                                                          \
    hipLaunchKernelGGL(attn_bwd_dq_valu_kernel<T>, gq, dim3(256), 0, s, (const T*)Q, (long)ldq, (const T*)K, (long)ldk, \
                       (const T*)V, (long)ldv, (const T*)O, (long)ldo, (const T*)dO, (long)lddo, (T*)dQ, (long)lddq,    \
                       key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta);                 \
    hipLaunchKernelGGL(attn_bwd_dkv_valu_kernel<T>, gk, dim3(256), 0, s, (const T*)Q, (long)ldq, (const T*)K, (long)ldk,\
                       (const T*)V, (long)ldv, (const T*)dO, (long)lddo, (T*)dK, (long)lddk, (T*)dV, (long)lddv,        \
                       key_mask, lse, d_ws, (int)Tq, (int)Tk, (int)H, scale, p_drop, seed, seed_delta)
    if (dtype == AVSR_F32) { RUN(float); }
    else if (dtype == AVSR_BF16) { RUN(bf16_t); }
    else return AVSR_EINVAL;
#undef RUN
    return avsr_launch_status();
}

extern "C" int avsr_attn_bwd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K, int64_t ldk,
                             const void* V, int64_t ldv, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                             void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv,
                             const uint8_t* key_mask, const float* lse, float* d_ws, int64_t B, int64_t H, int64_t Tq,
                             int64_t Tk, int64_t head_dim, float scale, float p_drop, uint64_t seed) {
    return avsr_attn_bwd_sd(stream, dtype, use_mfma, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv,
                            key_mask, lse, d_ws, B, H, Tq, Tk, head_dim, scale, p_drop, seed, nullptr);
}

// Debugging aid: resident workgroups per CU the runtime computes for the forward kernels (registers + LDS).
extern "C" int avsr_debug_attn_occupancy(int which) {
    int n = -1;
    hipError_t e = hipSuccess;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<false, true>, 256, 0);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<true, false>, 256, 0);
    else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_fp8_kernel, 256, 0);
    return e == hipSuccess ? n : -1;
}
