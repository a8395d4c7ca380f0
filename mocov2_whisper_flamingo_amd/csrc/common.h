// Shared device helpers for the gfx950 kernels: vector types, bf16 packing,
// wave64 reductions, launch checking.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/avsr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AVSR_WAVE 64

static inline int avsr_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AVSR_OK : AVSR_EHIP;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device).
// `done` is the call site's own zero-initialised table (one per kernel instantiation); a racing second thread at
// worst sets the idempotent attribute twice.
#define AVSR_MAX_DEVICES 64
struct avsr_dev_flags { bool set[AVSR_MAX_DEVICES]; };
static inline bool avsr_allow_lds(const void* kernel, int bytes, avsr_dev_flags& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool tracked = dev >= 0 && dev < AVSR_MAX_DEVICES;
    if (tracked && done.set[dev]) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (tracked) done.set[dev] = true;
    return true;
}

// Butterfly reductions over the 64 lanes without the LDS crossbar: xor 1 / 2 by quad permutes, 8- and 16-lane sums by
// the half-row and row mirrors (after the quad steps every quad is uniform, so a mirror pairs the right partners),
// xor 16 / 32 by v_permlane16_swap / v_permlane32_swap.  Every lane ends with the full result.
#define AVSR_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
#define AVSR_WAVE_REDUCE(v, OP)                                                                             \
    do {                                                                                                    \
        v = OP(v, AVSR_DPP(v, 0xB1));  /* quad_perm [1,0,3,2] */                                            \
        v = OP(v, AVSR_DPP(v, 0x4E));  /* quad_perm [2,3,0,1] */                                            \
        v = OP(v, AVSR_DPP(v, 0x141)); /* row_half_mirror */                                                \
        v = OP(v, AVSR_DPP(v, 0x140)); /* row_mirror */                                                     \
        {                                                                                                   \
            const unsigned u_ = __builtin_bit_cast(unsigned, v);                                            \
            const auto s_ = __builtin_amdgcn_permlane16_swap(u_, u_, false, false);                         \
            v = OP(__builtin_bit_cast(float, (unsigned)s_[0]), __builtin_bit_cast(float, (unsigned)s_[1])); \
        }                                                                                                   \
        {                                                                                                   \
            const unsigned u_ = __builtin_bit_cast(unsigned, v);                                            \
            const auto s_ = __builtin_amdgcn_permlane32_swap(u_, u_, false, false);                         \
            v = OP(__builtin_bit_cast(float, (unsigned)s_[0]), __builtin_bit_cast(float, (unsigned)s_[1])); \
        }                                                                                                   \
    } while (0)
__device__ __forceinline__ float avsr_addf(float a, float b) { return a + b; }
// sum over the 16 lanes of a row (lanes 16k .. 16k+15); every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
    v += AVSR_DPP(v, 0xB1);
    v += AVSR_DPP(v, 0x4E);
    v += AVSR_DPP(v, 0x141);
    v += AVSR_DPP(v, 0x140);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, AVSR_DPP(v, 0xB1));
    v = fmaxf(v, AVSR_DPP(v, 0x4E));
    v = fmaxf(v, AVSR_DPP(v, 0x141));
    v = fmaxf(v, AVSR_DPP(v, 0x140));
    return v;
}
// v + v[lane ^ O] for O = 8 (rotate the row by 8), 16 and 32 (lane-swap instructions): exact partners, no LDS
template <int O> __device__ __forceinline__ float xor_add(float v) {
    static_assert(O == 8 || O == 16 || O == 32, "xor_add: 8, 16 or 32");
    if constexpr (O == 8) {
        return v + AVSR_DPP(v, 0x128);                         // row_ror:8
    } else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        if constexpr (O == 16) {
            const auto s = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            return __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
        } else {
            const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            return __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
        }
    }
}
__device__ __forceinline__ float wave_sum(float v) {
    AVSR_WAVE_REDUCE(v, avsr_addf);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    AVSR_WAVE_REDUCE(v, fmaxf);
    return v;
}

// Block-wide sum for blocks of up to 1024 threads (<= 16 waves).  `red` is a
// 16-float LDS scratch; every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// gelu via the Abramowitz-Stegun 7.1.26 erfc polynomial (|error| <= 1.5e-7 on erf): branch-free, one rcp and
// one exp2 - for epilogues whose result is rounded to bf16 anyway
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);        // the polynomial of erfc(z) / 2
    poly = fmaf(t, poly, 0.5f * 1.421413741f);
    poly = fmaf(t, poly, 0.5f * -0.284496736f);
    poly = fmaf(t, poly, 0.5f * 0.254829592f);
    const float q = poly * t * __builtin_amdgcn_exp2f(-0.72134752044448170368f * (x * x));   // erfc(z) / 2 = Phi(-|x|)
    // x Phi(x) = max(x, 0) - |x| Phi(-|x|) for either sign: no compare / select
    return fmaf(-fabsf(x), q, fmaxf(x, 0.f));
}
// d/dx of gelu_erf
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// the same through the erfc polynomial of gelu_erf_fast (|error| <= 1.5e-7 on the cdf): for GEMM epilogues
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // exp(-x^2 / 2)
    const float q = 0.5f * poly * t * e;                                       // erfc(z) / 2
    const float cdf = x < 0.f ? q : 1.0f - q;
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// Counter-based random draws for dropout: four 16-bit uniform integers per 64-bit counter, from two 32-bit integer
// hashes of (seed, counter) - a multiply-add of the counter's low word, then the avalanche xorshift 16, * 0x7feb352d,
// xorshift 15, * 0x846ca68b, xorshift 16.  About 20 integer instructions per four draws (a Philox-4x32-10 call, which
// this replaces, costs over a hundred: the dropout attention kernels were bound by it).  The host-side restatement
// in tests/philox_ref.py replays it bit for bit.
__host__ __device__ __forceinline__ uint32_t avsr_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ void avsr_rand4(uint64_t seed, uint64_t idx, uint32_t out[4]) {
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    const uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32);
    const uint32_t base = s0 + c1 * 0x85EBCA77u;
    const uint32_t h0 = avsr_mix32((c0 * 0x9E3779B1u + base) ^ s1);
    const uint32_t h1 = avsr_mix32((c0 * 0xC2B2AE3Du + base) ^ ~s1);
    out[0] = h0 & 0xFFFFu; out[1] = h0 >> 16; out[2] = h1 & 0xFFFFu; out[3] = h1 >> 16;
}
// keep-decision of one 16-bit draw for dropout probability p: keep iff draw >= floor(65536 p)
__host__ __device__ __forceinline__ bool avsr_bits_keep(uint32_t draw, float p) {
    return draw >= (uint32_t)(p * 65536.0f);
}
// element idx of a flat tensor draws output (idx & 3) of counter idx / 4
__host__ __device__ __forceinline__ bool avsr_keep(uint64_t seed, uint64_t idx, float p) {
    uint32_t o[4];
    avsr_rand4(seed, idx >> 2, o);
    return avsr_bits_keep(o[idx & 3], p);
}
// Dropout of attention probabilities.  Element (row, key) of a [rows, Tk] matrix
// (row = (b*H + h)*Tq + q) draws output (key & 3) of counter row*tk4 + key/4 with
// tk4 = ceil(Tk/4): one generator call covers four neighbouring keys.
__host__ __device__ __forceinline__ uint64_t avsr_attn_counter(uint64_t row, int tk4, int key) {
    return row * (uint64_t)tk4 + (uint64_t)(key >> 2);
}
__host__ __device__ __forceinline__ bool avsr_attn_keep(uint64_t seed, uint64_t row, int tk4, int key, float p) {
    uint32_t o[4];
    avsr_rand4(seed, avsr_attn_counter(row, tk4, key), o);
    return avsr_bits_keep(o[key & 3], p);
}

// Internal entry points shared between the source files (exported, but not part of the declared interface): the
// public forms with one more argument, a device word that is added to the dropout seed when the kernel runs.
extern "C" int avsr_attn_fwd_sd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                                int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                                float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                                float p_drop, uint64_t seed, const uint64_t* seed_delta, int causal, void* workspace,
                                int64_t workspace_bytes);
extern "C" int avsr_attn_bwd_sd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                                int64_t ldk, const void* V, int64_t ldv, const void* O, int64_t ldo, const void* dO,
                                int64_t lddo, void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv,
                                const uint8_t* key_mask, const float* lse, float* d_ws, int64_t B, int64_t H, int64_t Tq,
                                int64_t Tk, int64_t head_dim, float scale, float p_drop, uint64_t seed,
                                const uint64_t* seed_delta);
extern "C" int avsr_gated_residual_bwd_sd(avsr_stream_t stream, int dtype, const void* dout, const void* y, const float* gate,
                                          void* dy, float* dgate, int64_t n, float p_drop, uint64_t seed,
                                          const uint64_t* seed_delta);
// the seed a kernel uses: the by-value seed plus the optional device word (one uniform load)
__device__ __forceinline__ uint64_t avsr_run_seed(uint64_t seed, const uint64_t* delta) {
    return delta ? seed + *delta : seed;
}

// ------------------------------------------------------------ LDS-DMA ----
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA piece from a UNIFORM base (scalar register pair) plus a 32-bit per-lane byte offset: the
// saddr form of the instruction, which the builtin does not select (it re-associates the sum into a 64-bit
// per-lane pointer and pays a 64-bit vector add per piece and K-step).  M0 carries the LDS destination and is
// compiler-reserved: saved and restored inside the statement.  The compiler does not count this load: every
// user waits with an explicit s_waitcnt vmcnt before the barrier that publishes the stage.
// The per-lane-address form of the same load, also hidden from the compiler (which waits for vmcnt(0) before the next
// LDS read wherever it has seen the builtin on some path to it).
__device__ __forceinline__ void dma_lane(uintptr_t gaddr, const char* lds_dst) {
    unsigned keep;
    const unsigned ldsa = (unsigned)(uintptr_t)(lptr_t)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gaddr), "s"(ldsa) : "memory");
}
__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, const char* lds_dst) {
    unsigned keep;
    const unsigned ldsa = (unsigned)(uintptr_t)(lptr_t)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsa) : "memory");
}
