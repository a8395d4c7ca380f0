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
                                                     float* __restrict__ csq, int T, int H, int W, int OH, int OW,
                                                     int parts_per_clip) {
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
        const long part0 = (long)b * parts_per_clip;
        const long part = part0 + (long)t * gridDim.x + blockIdx.x;
#pragma unroll
        for (int c = 0; c < ST_CPW; ++c) {
            const float s = wave_sum(st1[c]), s2 = wave_sum(st2[c]);
            if (lane == 0) { csum[part * ST_CO + c0 + c] = s; csq[part * ST_CO + c0 + c] = s2; }
        }
        if (t == 0 && blockIdx.x == 0)                // slab rows of this clip no workgroup owns
            for (int i = threadIdx.x; i < (parts_per_clip - T * (int)gridDim.x) * ST_CO; i += 256) {
                csum[(part0 + (long)T * gridDim.x) * ST_CO + i] = 0.f;
                csq[(part0 + (long)T * gridDim.x) * ST_CO + i] = 0.f;
            }
    }
}

// ---- MFMA form of the stem (bf16 output) --------------------------------------
// The same convolution as an implicit GEMM on v_mfma_f32_16x16x32_bf16: rows = output pixels,
// columns = 64 output channels, K = (kt, ci, kh, kw') with kw' padded 3 -> 4 (zero weight), 180 -> 192.
// A workgroup owns `rows` output rows of one frame: the (2*rows+1) x (W+8) input window of the 5 x 3
// (frame, channel) planes is converted to bf16 and staged in LDS once with its zero padding, so the
// gather needs no bounds tests: a lane's 8 consecutive k are two window rows x 4 consecutive columns,
// i.e. four aligned ds_read_b32.  The whole weight matrix lives in registers as 24 pre-packed
// fragments (avsr_stem3d_pack_frags); each wave walks 16-pixel tiles and keeps 4 accumulator tiles.
#define SM_KS 6                         // K-steps of 32
#define SM_ROWIDS 45                    // real (kt, ci, kh) rows; ids 45..47 carry zero weights
#define SM_FRAG_ELEMS (SM_KS * 4 * 64 * 8)

__global__ void stem3d_pack_frags_kernel(const float* __restrict__ wpacked, bf16_t* __restrict__ frags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= SM_FRAG_ELEMS) return;
    const int j = i & 7, lane = (i >> 3) & 63, ct = (i >> 9) & 3, ks = i >> 11;
    const int cout = 16 * ct + (lane & 15);
    const int k = 32 * ks + 8 * (lane >> 4) + j;
    const int rowid = k >> 2, kw = k & 3;
    float v = 0.f;
    if (rowid < SM_ROWIDS && kw < 3) {
        const int kh = rowid % 3, pl = rowid / 3, ci = pl % 3, kt = pl / 3;
        v = wpacked[((((ci * ST_KT + kt) * ST_KH + kh) * ST_KW) + kw) * ST_CO + cout];
    }
    frags[i] = (bf16_t)v;
}

// One work item = SM_F consecutive output frames x `rows` output rows of one clip; its window holds the
// SM_F + 4 input frames those outputs touch, so an input frame is staged (SM_F + 4) / SM_F = 2 times instead
// of 5.  Workgroups are persistent (grid = a few per CU) and walk the items, so the 24 weight fragments are
// loaded once per wave and neighbouring workgroups drift out of phase (staging of one overlaps the MFMA
// phase of the other).  Statistics of item i go to slab row clip*parts_per_clip + i; the rows of a clip no
// item owns are zeroed by the clip's first item.
#define SM_F 4
__global__ __launch_bounds__(256, 2) void stem3d_mfma_kernel(const float* __restrict__ video_arg, const bf16_t* __restrict__ frags,
                                                         bf16_t* __restrict__ out, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ csum,
                                                         float* __restrict__ csq, int B, int T, int H, int W, int OH, int OW,
                                                         int rows, int parts_per_clip, const float* const* video_slot) {
    // video_slot: the input's address is read from device memory when the kernel runs (a launch recorded in a HIP
    // graph then follows the caller's current batch without a copy of the batch into a fixed buffer)
    const float* __restrict__ video = video_slot ? *video_slot : video_arg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* win = (bf16_t*)smem;                     // [(SM_F+4)*3 planes][2*rows+1][PW]
    const int PW = W + 8, WR = 2 * rows + 1, NPL = (SM_F + 4) * 3;
    const int plane_sz = WR * PW;
    float* red = (float*)(smem + (size_t)NPL * plane_sz * sizeof(bf16_t));   // [4 waves][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    const int RB = (OH + rows - 1) / rows, NFG = (T + SM_F - 1) / SM_F;
    const int items_per_clip = NFG * RB, nitems = B * items_per_clip;

    // ---- weights: 24 fragments in registers, once per wave ----
    bf16x8 wf[SM_KS][4];
#pragma unroll
    for (int ks = 0; ks < SM_KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wf[ks][ct] = *(const bf16x8*)(frags + ((ks * 4 + ct) * 64 + lane) * 8);
    // window offsets of this lane's two rows per K-step (rowid = 8 ks + 2 q + h, clamped onto a real row)
    int roff[SM_KS][2];
#pragma unroll
    for (int ks = 0; ks < SM_KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rowid = min(8 * ks + 2 * q + h, SM_ROWIDS - 1);
            roff[ks][h] = (rowid / 3) * plane_sz + (rowid % 3) * PW;
        }
    const bool fold = scale != nullptr;
    float sc[4][4], sh[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[ct][r] = fold ? scale[16 * ct + 4 * q + r] : 1.f;
            sh[ct][r] = fold ? shift[16 * ct + 4 * q + r] : 0.f;
        }
    const int W4 = W >> 2, NI = NPL * WR * W4;
    const int dr = 256 / W4, dj = 256 - dr * W4;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / items_per_clip, ic = item - b * items_per_clip;
        const int fg = ic / RB, rbk = ic - fg * RB;
        const int t0 = fg * SM_F, nf = min(SM_F, T - t0);
        const int oh0 = rbk * rows, nrow = min(rows, OH - oh0);
        __syncthreads();                                  // the previous item's window is fully read

        // ---- stage the window: 16-byte loads, 8 in flight per thread, zero outside the image / the clip.
        // Loads are unconditional from clamped addresses and zeroed afterwards (a load under a branch gets
        // its own s_waitcnt vmcnt(0), which serialises the memory latency).  Window column = image column + 3.
        // Entry i = (window row r, 4-column group j) advances by 256 per load, tracked without divisions.
        {
            int r = tid / W4, j = tid - r * W4;
            int pl = r / WR, rr = r - pl * WR;
            for (int i0 = tid; i0 < NI; i0 += 256 * 8) {
                f32x4 v[8];
                int wofs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = i0 + 256 * u < NI;
                    const int plc = min(pl, NPL - 1);
                    const int ci = plc % 3, tt = t0 + plc / 3 - 2;
                    const int ih = 2 * oh0 - 3 + rr;
                    const bool rok = in && tt >= 0 && tt < T && ih >= 0 && ih < H;
                    const int ttc = min(max(tt, 0), T - 1), ihc = min(max(ih, 0), H - 1);
                    v[u] = *(const f32x4*)(video + ((((long)b * T + ttc) * 3 + ci) * H + ihc) * (long)W + 4 * j);
                    if (!rok) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    wofs[u] = in ? pl * plane_sz + rr * PW + 4 * j : -1;
                    j += dj; rr += dr;
                    if (j >= W4) { j -= W4; ++rr; }
                    while (rr >= WR) { rr -= WR; ++pl; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (wofs[u] >= 0) {
                        bf16_t* wp = win + wofs[u];
                        wp[3] = (bf16_t)v[u][0];
                        *(bf16x2*)(wp + 4) = (bf16x2){(bf16_t)v[u][1], (bf16_t)v[u][2]};
                        wp[6] = (bf16_t)v[u][3];
                    }
                }
            }
            for (int r2 = tid; r2 < NPL * WR; r2 += 256) {    // the 3 + 5 padding columns of every window row
                bf16_t* wp = win + r2 * PW;
                wp[0] = wp[1] = wp[2] = (bf16_t)0.f;
#pragma unroll
                for (int c = 3; c < 8; ++c) wp[W + c] = (bf16_t)0.f;
            }
        }
        __syncthreads();

        f32x4 s1[4], s2[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { s1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const int fpix = nrow * OW, npix = nf * fpix;
        const int ntile = (npix + 15) >> 4;
        for (int tile = wave; tile < ntile; tile += 4) {
            const int pix = tile * 16 + (lane & 15);
            const bool live = pix < npix;
            const int pf = live ? pix / fpix : 0, prem = live ? pix - pf * fpix : 0;
            const int pr = prem / OW, pc = prem - pr * OW;
            const int base = pf * 3 * plane_sz + 2 * pr * PW + 2 * pc;
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < SM_KS; ++ks) {
                u32x4 a;
                const unsigned int* r0 = (const unsigned int*)(win + base + roff[ks][0]);
                const unsigned int* r1 = (const unsigned int*)(win + base + roff[ks][1]);
                a[0] = r0[0]; a[1] = r0[1]; a[2] = r1[0]; a[3] = r1[1];
                const bf16x8 xf = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ct], xf, acc[ct], 0, 0, 0);
            }
            if (!live) continue;
            bf16_t* op = out + ((((long)b * T + t0 + pf) * OH + oh0 + pr) * OW + pc) * ST_CO + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[ct][r];
                    s1[ct][r] += v; s2[ct][r] = fmaf(v, v, s2[ct][r]);
                    o[r] = (bf16_t)(fold ? fmaxf(fmaf(v, sc[ct][r], sh[ct][r]), 0.f) : v);
                }
                *(bf16x4*)(op + 16 * ct) = o;
            }
        }
        if (csum) {
            // fold the 16 pixel lanes of each channel quad, then the 4 waves through LDS
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = s1[ct][r], c = s2[ct][r];
                    a = row16_sum(a); c = row16_sum(c);
                    if ((lane & 15) == 0) {
                        red[(wave * 2) * 64 + 16 * ct + 4 * q + r] = a;
                        red[(wave * 2 + 1) * 64 + 16 * ct + 4 * q + r] = c;
                    }
                }
            __syncthreads();
            const long part0 = (long)b * parts_per_clip;
            if (tid < 64) {
                float a = 0.f, c = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { a += red[(w * 2) * 64 + tid]; c += red[(w * 2 + 1) * 64 + tid]; }
                csum[(part0 + ic) * ST_CO + tid] = a;
                csq[(part0 + ic) * ST_CO + tid] = c;
            }
            if (ic == 0)                                  // slab rows of this clip no item owns
                for (int i = tid; i < (parts_per_clip - items_per_clip) * ST_CO; i += 256) {
                    csum[(part0 + items_per_clip) * ST_CO + i] = 0.f;
                    csq[(part0 + items_per_clip) * ST_CO + i] = 0.f;
                }
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

// bf16 form: a thread owns 8 channels (one 16-byte load per tap) of one output column and walks down a band of output
// rows.  Consecutive output rows share an input row (2*oh + 1 is the last row of window oh and the first of window
// oh + 1): its 3-column maximum is carried in registers, so an output costs 6 loads instead of 9 and the loop has no
// integer division.  Grid: (bands, images); a workgroup covers 256 / C8 output columns per pass.  [600, 50, 50, 64]:
// 79 us for the one-thread-per-output form, 47 us (5.1 TB/s) for this one; bands of 3 / 5 / 9 / 13 rows: 55 / 50 / 52 / 47 us.
#ifndef MP_BAND
#define MP_BAND 13
#endif
__global__ __launch_bounds__(256) void maxpool3x3s2_bf16x8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  int H, int W, int C8, int OH, int OW) {
    const int c8 = threadIdx.x % C8, col = threadIdx.x / C8, cols = 256 / C8;
    const int n = blockIdx.y;
    const int oh0 = blockIdx.x * MP_BAND, oh1 = min(OH, oh0 + MP_BAND);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale ? scale[c8 * 8 + e] : 1.f; sh[e] = scale ? shift[c8 * 8 + e] : 0.f; }
    const bf16_t* img = x + (long)n * H * W * C8 * 8 + c8 * 8;
    bf16_t* out = y + (long)n * OH * OW * C8 * 8 + c8 * 8;
    // maximum over the (up to) three columns of input row ih around output column ow; -inf for a row outside the image
    auto hmax = [&](int ih, int ow, float (&m)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        if (ih < 0 || ih >= H) return;
        const bf16_t* row = img + (long)ih * W * C8 * 8;
        bf16x8 v[3];
        bool ok[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            ok[kw] = iw >= 0 && iw < W;
            if (ok[kw]) v[kw] = *(const bf16x8*)(row + (long)iw * C8 * 8);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            if (!ok[kw]) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[kw][e];
                if (scale) f = fmaxf(fmaf(f, sc[e], sh[e]), 0.f);
                m[e] = fmaxf(m[e], f);
            }
        }
    };
    for (int ow = col; ow < OW; ow += cols) {
        float carry[8], r0[8], r1[8];
        hmax(oh0 * 2 - 1, ow, carry);
        for (int oh = oh0; oh < oh1; ++oh) {
            hmax(oh * 2, ow, r0);
            hmax(oh * 2 + 1, ow, r1);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = (bf16_t)fmaxf(fmaxf(carry[e], r0[e]), r1[e]); carry[e] = r1[e]; }
            *(bf16x8*)(out + ((long)oh * OW + ow) * C8 * 8) = o;
        }
    }
}

// Global average pool over HW pixels of an NHWC tensor [B*T, HW, C] -> [B*T, C];
// frames t >= len[b] are written as zeros (moco_visual_frontend.py:42,49).
// VideoTransform of the data path (datamodule/transforms.py:61-96) for frames already on the GPU: optional bilinear
// resize with the half-pixel centres of interpolate(align_corners=False) (:66-68), /255 (:69,94), optional
// horizontal flip per clip (:70), optional zeroing of whole frames (AdaptiveTimeMask, :27-45,73, applied before
// the normalisation as the reference orders it), then (x - mean[c]) / std[c] (:74,95).  One thread per output pixel.
template <typename TI>
__global__ __launch_bounds__(256) void video_preprocess_kernel(const TI* __restrict__ in, float* __restrict__ out, long n, int T,
                                                               int H, int W, int OH, int OW, const uint8_t* __restrict__ flip,
                                                               const uint8_t* __restrict__ frame_zero, float m0, float m1,
                                                               float m2, float i0, float i1, float i2) {
    const long stride = (long)gridDim.x * blockDim.x;
    const float sy = (float)H / OH, sx = (float)W / OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), c = (int)((i / ((long)OW * OH)) % 3);
        const long f = i / ((long)OW * OH * 3);                 // frame index b*T + t
        const int b = (int)(f / T);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), istd = c == 0 ? i0 : (c == 1 ? i1 : i2);
        float v = 0.f;
        if (!(frame_zero && frame_zero[f])) {
            const int xs = (flip && flip[b]) ? OW - 1 - ox : ox;
            const TI* src = in + (f * 3 + c) * (long)H * W;
            if (OH == H && OW == W) {
                v = (float)src[(long)oy * W + xs];
            } else {
                const float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((xs + 0.5f) * sx - 0.5f, 0.f);
                const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
                const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
                const float wy = fy - y0, wx = fx - x0;
                const float a = (float)src[(long)y0 * W + x0], bq = (float)src[(long)y0 * W + x1];
                const float cq = (float)src[(long)y1 * W + x0], d = (float)src[(long)y1 * W + x1];
                v = (1.f - wy) * ((1.f - wx) * a + wx * bq) + wy * ((1.f - wx) * cq + wx * d);
            }
            v = v / 255.0f;
        }
        out[i] = (v - mean) * istd;
    }
}

// ColorJitter + RandomGrayscale + AdaptiveTimeMask + Normalize of the training transform (datamodule/transforms.py:71-74)
// on frames in [0, 1]: one workgroup per frame, the frame's three planes in LDS.  The four jitter steps run in the
// clip's drawn order (torchvision ColorJitter.forward: 0 brightness, 1 contrast, 2 saturation, 3 hue; a negative
// entry skips the step), each as torchvision's functional_tensor ops define it: blend(img, other, f) =
// clamp(f*img + (1-f)*other, 0, 1) with other = 0 (brightness), the frame's mean grey level (contrast), the pixel's
// grey level 0.2989 r + 0.587 g + 0.114 b (saturation); hue = RGB -> HSV, h = (h + shift) mod 1, HSV -> RGB.
__device__ __forceinline__ float grey_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__global__ __launch_bounds__(256) void video_color_kernel(float* __restrict__ x, int T, int HW, const int* __restrict__ order,
                                                          const float* __restrict__ factors, const uint8_t* __restrict__ grey,
                                                          const uint8_t* __restrict__ frame_zero, float m0, float m1, float m2,
                                                          float i0, float i1, float i2) {
    extern __shared__ float plane[];            // [3][HW]
    __shared__ float red[16];
    const int f = blockIdx.x, b = f / T, tid = threadIdx.x;
    float* xf = x + (long)f * 3 * HW;
    if (frame_zero && frame_zero[f]) {          // AdaptiveTimeMask zeroes the frame after the colour steps
        for (int i = tid; i < HW; i += 256) {
            xf[i] = (0.f - m0) * i0; xf[HW + i] = (0.f - m1) * i1; xf[2 * HW + i] = (0.f - m2) * i2;
        }
        return;
    }
    for (int i = tid; i < 3 * HW; i += 256) plane[i] = xf[i];
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const int op = order ? order[b * 4 + k] : -1;
        if (op < 0 || op > 3) continue;
        const float fac = factors[b * 4 + op];
        if (op == 0) {
            for (int i = tid; i < 3 * HW; i += 256) plane[i] = clamp01(fac * plane[i]);
        } else if (op == 1) {
            float acc = 0.f;
            for (int i = tid; i < HW; i += 256) acc += grey_of(plane[i], plane[HW + i], plane[2 * HW + i]);
            const float mean = block_sum(acc, red) / (float)HW;
            for (int i = tid; i < 3 * HW; i += 256) plane[i] = clamp01(fac * plane[i] + (1.f - fac) * mean);
        } else if (op == 2) {
            for (int i = tid; i < HW; i += 256) {
                const float l = grey_of(plane[i], plane[HW + i], plane[2 * HW + i]);
#pragma unroll
                for (int c = 0; c < 3; ++c) plane[c * HW + i] = clamp01(fac * plane[c * HW + i] + (1.f - fac) * l);
            }
        } else {
            for (int i = tid; i < HW; i += 256) {
                const float r = plane[i], g = plane[HW + i], bl = plane[2 * HW + i];
                const float maxc = fmaxf(r, fmaxf(g, bl)), minc = fminf(r, fminf(g, bl));
                const bool eq = maxc == minc;
                const float cr = maxc - minc;
                const float sat = cr / (eq ? 1.f : maxc);
                const float div = eq ? 1.f : cr;
                const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - bl) / div;
                float h = 0.f;
                if (maxc == r) h = bc - gc;
                else if (maxc == g) h = 2.f + rc - bc;
                else h = 4.f + gc - rc;
                h = fmodf(h / 6.f + 1.f, 1.f);
                h = h + fac;
                h = h - floorf(h);              // python's % 1.0: result in [0, 1)
                const float h6 = h * 6.f, fi = floorf(h6), fr = h6 - fi;
                int sector = (int)fi % 6;
                if (sector < 0) sector += 6;
                const float v = maxc;
                const float pp = clamp01(v * (1.f - sat)), qq = clamp01(v * (1.f - sat * fr));
                const float tt = clamp01(v * (1.f - sat * (1.f - fr)));
                float ro, go, bo;
                switch (sector) {
                    case 0: ro = v; go = tt; bo = pp; break;
                    case 1: ro = qq; go = v; bo = pp; break;
                    case 2: ro = pp; go = v; bo = tt; break;
                    case 3: ro = pp; go = qq; bo = v; break;
                    case 4: ro = tt; go = pp; bo = v; break;
                    default: ro = v; go = pp; bo = qq; break;
                }
                plane[i] = ro; plane[HW + i] = go; plane[2 * HW + i] = bo;
            }
        }
        __syncthreads();
    }
    const bool to_grey = grey && grey[b];
    for (int i = tid; i < HW; i += 256) {
        float r = plane[i], g = plane[HW + i], bl = plane[2 * HW + i];
        if (to_grey) r = g = bl = grey_of(r, g, bl);
        xf[i] = (r - m0) * i0; xf[HW + i] = (g - m1) * i1; xf[2 * HW + i] = (bl - m2) * i2;
    }
}

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

// bf16 input, channel count a multiple of 8: 8 channels per thread as 16-byte loads, the HW positions unrolled four
// deep (the scalar form above issued one dependent 2-byte load per position: 0.7 TB/s on the [600,16,2048] trunk output)
template <typename TO>
__global__ __launch_bounds__(256) void avgpool_mask_bf16x8_kernel(const bf16_t* __restrict__ x, TO* __restrict__ y,
                                                                  const int64_t* __restrict__ lens, int Tn, int HW, int C8) {
    const int frame = blockIdx.x;
    const int cg = blockIdx.y * blockDim.x + threadIdx.x;
    if (cg >= C8) return;
    const int b = frame / Tn, t = frame - b * Tn;
    const bool dead = lens && t >= lens[b];
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (!dead) {
        const bf16_t* xp = x + ((long)frame * HW * C8 + cg) * 8;
        const long step = (long)C8 * 8;
        int p = 0;
        for (; p + 16 <= HW; p += 16) {         // the trunk's last stage is 4 x 4 pixels: all sixteen loads in flight
            bf16x8 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const bf16x8*)(xp + (p + u) * step);
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
        }
        for (; p + 4 <= HW; p += 4) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(xp + (p + u) * step);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
        }
        for (; p < HW; ++p) {
            const bf16x8 v = *(const bf16x8*)(xp + p * step);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
    const float inv = 1.0f / (float)HW;
    TO* yp = y + ((long)frame * C8 + cg) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) yp[e] = from_f32<TO>(s[e] * inv);
}

// Output rows per work item of the MFMA stem: the largest of 4, 2, 1 whose bf16 window fits in LDS twice per CU.
static int stem_mfma_rows(int64_t W, size_t* lds_bytes) {
    for (int rows = 4; rows >= 1; rows >>= 1) {
        const size_t bytes = (size_t)(SM_F + 4) * 3 * (2 * rows + 1) * (W + 8) * sizeof(bf16_t) + 4 * 2 * 64 * sizeof(float);
        if (bytes <= 72 * 1024) { *lds_bytes = bytes; return rows; }
    }
    return 0;
}
static int64_t stem_parts_per_clip(int64_t T, int64_t H, int64_t W) {
    const int64_t OH = (H + 6 - 3) / 2 + 1, OW = (W + 6 - 3) / 2 + 1;
    const int64_t valu = T * ((OH * OW + 64 * ST_ROUNDS - 1) / (64 * ST_ROUNDS));
    size_t lds;
    const int rows = stem_mfma_rows(W, &lds);
    const int64_t mfma = rows ? ((T + SM_F - 1) / SM_F) * ((OH + rows - 1) / rows) : 0;
    return valu > mfma ? valu : mfma;
}

// number of rows of the [parts, 64] statistics slabs avsr_stem3d_fwd fills (either kernel)
extern "C" int64_t avsr_stem3d_stats_parts(int64_t B, int64_t T, int64_t H, int64_t W) {
    return B * stem_parts_per_clip(T, H, W);
}

extern "C" int64_t avsr_stem3d_frag_elems(void) { return SM_FRAG_ELEMS; }

extern "C" int avsr_stem3d_pack_frags(avsr_stream_t stream, const float* weight_packed, void* weight_frags) {
    if (!weight_packed || !weight_frags) return AVSR_EINVAL;
    if ((uintptr_t)weight_frags % 16) return AVSR_EUNSUPPORTED;
    hipLaunchKernelGGL(stem3d_pack_frags_kernel, dim3((SM_FRAG_ELEMS + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       weight_packed, (bf16_t*)weight_frags);
    return avsr_launch_status();
}

static int stem3d_fwd_impl(avsr_stream_t stream, int out_dtype, const float* video, const float* const* video_slot,
                           const float* weight_packed, const void* weight_frags, void* out, const float* scale,
                           const float* shift, float* col_sum, float* col_sumsq, int64_t B, int64_t T, int64_t H, int64_t W);

extern "C" int avsr_stem3d_fwd(avsr_stream_t stream, int out_dtype, const float* video, const float* weight_packed,
                               const void* weight_frags, void* out, const float* scale, const float* shift,
                               float* col_sum, float* col_sumsq, int64_t B, int64_t T, int64_t H, int64_t W) {
    return stem3d_fwd_impl(stream, out_dtype, video, nullptr, weight_packed, weight_frags, out, scale, shift, col_sum,
                           col_sumsq, B, T, H, W);
}

extern "C" int avsr_stem3d_fwd_indirect(avsr_stream_t stream, int out_dtype, const float* const* video_slot,
                                        const float* weight_packed, const void* weight_frags, void* out,
                                        const float* scale, const float* shift, float* col_sum, float* col_sumsq,
                                        int64_t B, int64_t T, int64_t H, int64_t W) {
    if (!video_slot || (uintptr_t)video_slot % 8) return AVSR_EINVAL;
    // the host cannot see the address the kernel will read: a 16-byte-aligned stand-in passes the checks below
    return stem3d_fwd_impl(stream, out_dtype, (const float*)(uintptr_t)16, video_slot, weight_packed, weight_frags, out, scale,
                           shift, col_sum, col_sumsq, B, T, H, W);
}


static int stem3d_fwd_impl(avsr_stream_t stream, int out_dtype, const float* video, const float* const* video_slot,
                           const float* weight_packed, const void* weight_frags, void* out, const float* scale,
                           const float* shift, float* col_sum, float* col_sumsq, int64_t B, int64_t T, int64_t H, int64_t W) {
    if (!video || !weight_packed || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0) return AVSR_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return AVSR_EINVAL;
    if ((col_sum == nullptr) != (col_sumsq == nullptr)) return AVSR_EINVAL;
    if ((uintptr_t)weight_packed % 16) return AVSR_EUNSUPPORTED;
    const int OH = (int)((H + 6 - 3) / 2 + 1), OW = (int)((W + 6 - 3) / 2 + 1);
    if (B * T > 65535 || H > 8192 || W > 8192) return AVSR_EUNSUPPORTED;
    const int ppc = (int)stem_parts_per_clip(T, H, W);
    hipStream_t s = (hipStream_t)stream;
    size_t lds = 0;
    const int rows = stem_mfma_rows(W, &lds);
    if (out_dtype == AVSR_BF16 && weight_frags && rows > 0 && (W % 4) == 0 && (uintptr_t)video % 16 == 0) {
        if ((uintptr_t)weight_frags % 16 || (uintptr_t)out % 8) return AVSR_EUNSUPPORTED;
        static avsr_dev_flags attr_set;
        if (!avsr_allow_lds((const void*)stem3d_mfma_kernel, 96 * 1024, attr_set)) return AVSR_EHIP;
        const long nitems = B * ((T + SM_F - 1) / SM_F) * ((OH + rows - 1) / rows);
        const unsigned nblk = (unsigned)(nitems < 512 ? nitems : 512);      // persistent: two workgroups per CU
        hipLaunchKernelGGL(stem3d_mfma_kernel, dim3(nblk), dim3(256), lds, s, video, (const bf16_t*)weight_frags,
                           (bf16_t*)out, scale, shift, col_sum, col_sumsq, (int)B, (int)T, (int)H, (int)W, OH, OW, rows, ppc,
                           video_slot);
        return avsr_launch_status();
    }
    if (video_slot) return AVSR_EUNSUPPORTED;               // only the MFMA form reads its input through the slot
    dim3 grid((unsigned)((OH * OW + 64 * ST_ROUNDS - 1) / (64 * ST_ROUNDS)), (unsigned)(B * T));
    if (out_dtype == AVSR_F32)
        hipLaunchKernelGGL(stem3d_kernel<float>, grid, dim3(256), 0, s, video, weight_packed, (float*)out, scale, shift,
                           col_sum, col_sumsq, (int)T, (int)H, (int)W, OH, OW, ppc);
    else if (out_dtype == AVSR_BF16)
        hipLaunchKernelGGL(stem3d_kernel<bf16_t>, grid, dim3(256), 0, s, video, weight_packed, (bf16_t*)out, scale, shift,
                           col_sum, col_sumsq, (int)T, (int)H, (int)W, OH, OW, ppc);
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
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16 && C % 8 == 0 && 256 % (C / 8) == 0 &&
        (((uintptr_t)x | (uintptr_t)y) % 16) == 0 && N <= 65535) {
        hipLaunchKernelGGL(maxpool3x3s2_bf16x8_kernel, dim3((unsigned)((OH + MP_BAND - 1) / MP_BAND), (unsigned)N), dim3(256), 0, s,
                           (const bf16_t*)x, (bf16_t*)y, scale, shift, (int)H, (int)W, (int)(C / 8), OH, OW);
        return avsr_launch_status();
    }
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
#define L(TI, TO) hipLaunchKernelGGL((maxpool3x3s2_kernel<TI, TO>), dim3((unsigned)nb), dim3(256), 0, s, (const TI*)x, (TO*)y, \
                                     scale, shift, (int)N, (int)H, (int)W, (int)C, OH, OW)
    if (in_dtype == AVSR_F32 && out_dtype == AVSR_F32) L(float, float);
    else if (in_dtype == AVSR_BF16 && out_dtype == AVSR_BF16) L(bf16_t, bf16_t);
    else if (in_dtype == AVSR_F32 && out_dtype == AVSR_BF16) L(float, bf16_t);
    else return AVSR_EINVAL;
#undef L
    return avsr_launch_status();
}

extern "C" int avsr_video_preprocess(avsr_stream_t stream, int in_is_u8, const void* video, float* out, int64_t B, int64_t T,
                                     int64_t H, int64_t W, int64_t OH, int64_t OW, const uint8_t* flip,
                                     const uint8_t* frame_zero, const float* mean3, const float* std3) {
    if (!video || !out || !mean3 || !std3 || B <= 0 || T <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return AVSR_EINVAL;
    if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return AVSR_EINVAL;
    if (B * T > (1 << 30) || H > 16384 || W > 16384 || OH > 16384 || OW > 16384) return AVSR_EUNSUPPORTED;
    const long n = (long)B * T * 3 * OH * OW;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)std::min<long>((n + 255) / 256, 16384)), block(256);
    if (in_is_u8)
        hipLaunchKernelGGL(video_preprocess_kernel<uint8_t>, grid, block, 0, s, (const uint8_t*)video, out, n, (int)T, (int)H,
                           (int)W, (int)OH, (int)OW, flip, frame_zero, mean3[0], mean3[1], mean3[2], 1.f / std3[0],
                           1.f / std3[1], 1.f / std3[2]);
    else
        hipLaunchKernelGGL(video_preprocess_kernel<float>, grid, block, 0, s, (const float*)video, out, n, (int)T, (int)H,
                           (int)W, (int)OH, (int)OW, flip, frame_zero, mean3[0], mean3[1], mean3[2], 1.f / std3[0],
                           1.f / std3[1], 1.f / std3[2]);
    return avsr_launch_status();
}

extern "C" int avsr_video_color(avsr_stream_t stream, float* video, int64_t B, int64_t T, int64_t H, int64_t W,
                                const int32_t* order, const float* factors, const uint8_t* grey,
                                const uint8_t* frame_zero, const float* mean3, const float* std3) {
    if (!video || !mean3 || !std3 || B <= 0 || T <= 0 || H <= 0 || W <= 0 || (order && !factors)) return AVSR_EINVAL;
    if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return AVSR_EINVAL;
    const size_t lds = (size_t)3 * H * W * sizeof(float);
    if (lds > 150 * 1024 || B * T > 0x7fffffff) return AVSR_EUNSUPPORTED;       // a frame's three planes live in LDS
    static avsr_dev_flags attr_set;
    if (!avsr_allow_lds((const void*)video_color_kernel, 150 * 1024, attr_set)) return AVSR_EHIP;
    hipLaunchKernelGGL(video_color_kernel, dim3((unsigned)(B * T)), dim3(256), lds, (hipStream_t)stream, video, (int)T,
                       (int)(H * W), order, factors, grey, frame_zero, mean3[0], mean3[1], mean3[2], 1.f / std3[0],
                       1.f / std3[1], 1.f / std3[2]);
    return avsr_launch_status();
}

// The trunk's last BN apply (affine + shortcut + ReLU) folded into the average pool: the normalised 2048-channel
// activation is consumed by nothing else, so it is never written.  y[frame, c] = mean over the HW pixels of
// relu(x * scale[c] + shift[c] + residual), zero for frames t >= lens[b].  A thread owns 8 channels of a frame.
template <typename TO>
__global__ __launch_bounds__(256) void bn_apply_avgpool_bf16x8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      TO* __restrict__ y, const int64_t* __restrict__ lens, int Tn,
                                                                      int HW, int C8) {
    const int frame = blockIdx.x;
    const int cg = blockIdx.y * blockDim.x + threadIdx.x;
    if (cg >= C8) return;
    const int b = frame / Tn, t = frame - b * Tn;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (!(lens && t >= lens[b])) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = scale[cg * 8 + e]; sh[e] = shift[cg * 8 + e]; }
        const long base = ((long)frame * HW * C8 + cg) * 8, step = (long)C8 * 8;
        for (int p0 = 0; p0 < HW; p0 += 8) {
            bf16x8 v[8], r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long o = base + (long)min(p0 + u, HW - 1) * step;
                v[u] = *(const bf16x8*)(x + o);
                r[u] = *(const bf16x8*)(res + o);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u < HW) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] += fmaxf(fmaf((float)v[u][e], sc[e], sh[e]) + (float)r[u][e], 0.f);
                }
        }
    }
    const float inv = 1.0f / (float)HW;
    TO* yp = y + ((long)frame * C8 + cg) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) yp[e] = from_f32<TO>(s[e] * inv);
}

extern "C" int avsr_bn_apply_avgpool_mask(avsr_stream_t stream, int out_dtype, const void* x, const void* residual,
                                          const float* scale, const float* shift, void* y, const int64_t* lens, int64_t B,
                                          int64_t T, int64_t HW, int64_t C) {
    if (!x || !residual || !scale || !shift || !y || B <= 0 || T <= 0 || HW <= 0 || C <= 0) return AVSR_EINVAL;
    if (C % 8 || ((uintptr_t)x | (uintptr_t)residual) % 16 || B * T > 0x7fffffff) return AVSR_EUNSUPPORTED;
    const int C8 = (int)(C / 8);
    dim3 grid((unsigned)(B * T), (unsigned)((C8 + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == AVSR_BF16)
        hipLaunchKernelGGL(bn_apply_avgpool_bf16x8_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (const bf16_t*)residual,
                           scale, shift, (bf16_t*)y, lens, (int)T, (int)HW, C8);
    else if (out_dtype == AVSR_F32)
        hipLaunchKernelGGL(bn_apply_avgpool_bf16x8_kernel<float>, grid, block, 0, s, (const bf16_t*)x, (const bf16_t*)residual,
                           scale, shift, (float*)y, lens, (int)T, (int)HW, C8);
    else return AVSR_EINVAL;
    return avsr_launch_status();
}

extern "C" int avsr_avgpool_mask_fwd(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y,
                                     const int64_t* lens, int64_t B, int64_t T, int64_t HW, int64_t C) {
    if (!x || !y || B <= 0 || T <= 0 || HW <= 0 || C <= 0) return AVSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * T)), block(256);
    if (in_dtype == AVSR_BF16 && C % 8 == 0 && (uintptr_t)x % 16 == 0 && (out_dtype == AVSR_BF16 || out_dtype == AVSR_F32)) {
        const int C8 = (int)(C / 8);
        dim3 g8((unsigned)(B * T), (unsigned)((C8 + 255) / 256));
        if (out_dtype == AVSR_BF16)
            hipLaunchKernelGGL(avgpool_mask_bf16x8_kernel<bf16_t>, g8, block, 0, s, (const bf16_t*)x, (bf16_t*)y, lens, (int)T,
                               (int)HW, C8);
        else
            hipLaunchKernelGGL(avgpool_mask_bf16x8_kernel<float>, g8, block, 0, s, (const bf16_t*)x, (float*)y, lens, (int)T,
                               (int)HW, C8);
        return avsr_launch_status();
    }
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
