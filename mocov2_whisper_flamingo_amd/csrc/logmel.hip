// Mel front end on the GPU: framed DFT as an fp32 MFMA contraction, power,
// mel filter bank as a second MFMA contraction, then one of two normalisations.
//
//   variant AVSR_MEL_WHISPER   WhisperFeatureExtractor._torch_extract_fbank_features
//       (transformers feature_extraction_whisper.py:135-168; reference call
//       site datamodule/av_dataset.py:47-51): periodic hann(400) STFT, hop 160,
//       centre + reflect padding, |.|^2 of frames 0..n/160-1, Slaney mel bank
//       with Slaney normalisation, log10(max(.,1e-10)), floor at clip max - 8,
//       (x + 4) / 4.
//   variant AVSR_MEL_SPEC_LN   torchaudio MelSpectrogram(16 kHz, 400, 400, 160,
//       80 mels) -> pad/trim to n_frames_out -> layer_norm over the whole
//       [frames, 80] clip with eps 1e-8 (datamodule/transforms.py:115-121,
//       99-110, 131): HTK mel bank, no filter normalisation, no log.
//
// One workgroup = 32 frames of one clip.  The clip segment those frames cover
// (5360 samples) is staged once in LDS with 4 floats of padding per hop so the
// 16 frames x 4 taps a wave reads per MFMA step fall in distinct banks.  The
// DFT matrix (window folded in) is pre-packed in MFMA fragment order and
// streamed through LDS in chunks of 16 taps; each wave owns one 16-frame tile
// and half of the 13 frequency tiles, with the real and imaginary accumulators
// of a bin in the same lane so the power needs no exchange.  Power goes to LDS
// and is contracted with the (also fragment-packed) mel bank.
#include "common.h"
#include <mutex>

namespace {

constexpr int LM_NFFT = 400, LM_HOP = 160, LM_NFREQ = 201, LM_NMEL = 80;
constexpr int LM_FT = 13;                                  // frequency tiles of 16 bins (208 >= 201)
constexpr int LM_KC = 16;                                  // taps per LDS chunk = 4 MFMA k-steps
constexpr int LM_NCHUNK = LM_NFFT / LM_KC;                 // 25
constexpr int LM_CHUNK_FLOATS = 2 * LM_FT * 4 * 64;        // 6656: [tile = 2*ft + re/im][k-step][lane]
constexpr int LM_DFT_FLOATS = LM_NCHUNK * LM_CHUNK_FLOATS; // 166400
constexpr int LM_MEL_KS = 52;                              // 208 bins / 4
constexpr int LM_MEL_TILES = LM_NMEL / 16;                 // 5
constexpr int LM_MEL_FLOATS = LM_MEL_TILES * LM_MEL_KS * 64;
constexpr int LM_FR = 32;                                  // frames per workgroup
constexpr int LM_SEG = (LM_FR - 1) * LM_HOP + LM_NFFT;     // 5360 samples
constexpr int LM_SEG_LDS = LM_SEG + 4 * (LM_SEG / LM_HOP + 1);
constexpr int LM_PSTRIDE = 212;                            // power rows: 208 bins + 4 (bank spread)
constexpr int LM_FIN_FR = 64;                              // frames per finalize workgroup

__device__ double lm_hz_to_mel(int variant, double f) {
    if (variant == AVSR_MEL_WHISPER)
        return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0;
    return 2595.0 * log10(1.0 + f / 700.0);
}
__device__ double lm_mel_to_hz(int variant, double m) {
    if (variant == AVSR_MEL_WHISPER)
        return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0;
    return 700.0 * (pow(10.0, m / 2595.0) - 1.0);
}
// weight of FFT bin `f` (0..200, 40 Hz apart) in triangular filter `m` (0..79) between 0 and 8 kHz
__device__ float lm_mel_weight(int variant, int m, int f) {
    const double m_lo = lm_hz_to_mel(variant, 0.0), m_hi = lm_hz_to_mel(variant, 8000.0);
    const double step = (m_hi - m_lo) / (LM_NMEL + 1);
    const double h0 = lm_mel_to_hz(variant, m_lo + m * step);
    const double h1 = lm_mel_to_hz(variant, m_lo + (m + 1) * step);
    const double h2 = lm_mel_to_hz(variant, m_lo + (m + 2) * step);
    const double hz = 40.0 * f;
    const double down = (hz - h0) / (h1 - h0), up = (h2 - hz) / (h2 - h1);
    double w = fmax(0.0, fmin(down, up));
    if (variant == AVSR_MEL_WHISPER) w *= 2.0 / (h2 - h0);
    return (float)w;
}

__global__ void logmel_tables_kernel(int variant, float* __restrict__ dft, float* __restrict__ mel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < LM_DFT_FLOATS) {
        const int lane = i & 63, ks = (i >> 6) & 3, tile = (i >> 8) % (2 * LM_FT), c = i / LM_CHUNK_FLOATS;
        const int f = 16 * (tile >> 1) + (lane & 15), k = LM_KC * c + 4 * ks + (lane >> 4);
        float v = 0.f;
        if (f < LM_NFREQ) {
            const float win = (float)(0.5 - 0.5 * cospi(2.0 * k / LM_NFFT));   // periodic hann, fp32 like torch.hann_window
            const double a = 2.0 * ((f * k) % LM_NFFT) / LM_NFFT;
            v = (float)(((tile & 1) ? sinpi(a) : cospi(a)) * (double)win);
        }
        dft[i] = v;
    }
    if (i < LM_MEL_FLOATS) {
        const int lane = i & 63, ks = (i >> 6) % LM_MEL_KS, mt = i / (LM_MEL_KS * 64);
        const int m = 16 * mt + (lane & 15), f = 4 * ks + (lane >> 4);
        mel[i] = f < LM_NFREQ ? lm_mel_weight(variant, m, f) : 0.f;
    }
}

// raw: [B][n_frames_out][80] fp32 (log10 mel for WHISPER, mel power for SPEC_LN);
// slab: [B][gridDim.x][2] per-workgroup (max, 0) or (sum, sum of squares).
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void logmel_dft_mel_kernel(
    const float* __restrict__ wave, long wave_stride, int n_samples, int n_valid, int n_frames, int n_frames_out,
    const float* __restrict__ dft, const float* __restrict__ melw, float* __restrict__ raw, float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) float lds[LM_SEG_LDS + LM_CHUNK_FLOATS];
    __shared__ float red[8];
    float* seg = lds;
    float* bb = lds + LM_SEG_LDS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * LM_FR;
    const float* w = wave + (long)b * wave_stride;

    // stage the segment: centre + reflect padding, zeros past n_valid
    for (int i = tid; i < LM_SEG; i += 256) {
        long j = (long)t0 * LM_HOP - LM_NFFT / 2 + i;
        if (j < 0) j = -j;
        if (j >= n_samples) j = 2L * (n_samples - 1) - j;
        seg[i + 4 * (i / LM_HOP)] = (j >= 0 && j < n_valid) ? w[j] : 0.f;
    }
    const f32x4* dft4 = (const f32x4*)dft;
    constexpr int C4 = LM_CHUNK_FLOATS / 4;                 // 1664 float4 per chunk = 6.5 per thread
    f32x4 pre[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
        if (tid + 256 * i < C4) ((f32x4*)bb)[tid + 256 * i] = dft4[tid + 256 * i];
    __syncthreads();

    const int half = wv >> 1;                               // frequency half: tiles 0..6 / 7..12
    const int ft0 = half * 7, nft = half ? 6 : 7;
    const int fr = (wv & 1) * 16 + (lane & 15), kq = lane >> 4;
    f32x4 acc[7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < LM_NCHUNK; ++c) {
        if (c + 1 < LM_NCHUNK) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (tid + 256 * i < C4) pre[i] = dft4[(long)(c + 1) * C4 + tid + 256 * i];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = LM_KC * c + 4 * ks + kq;
            const float x = seg[fr * (LM_HOP + 4) + k + 4 * ((k >= LM_HOP) + (k >= 2 * LM_HOP))];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (j < nft) {
                    const float cr = bb[((2 * (ft0 + j)) * 4 + ks) * 64 + lane];
                    const float ci = bb[((2 * (ft0 + j) + 1) * 4 + ks) * 64 + lane];
                    acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cr, x, acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ci, x, acc[j][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (c + 1 < LM_NCHUNK) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (tid + 256 * i < C4) ((f32x4*)bb)[tid + 256 * i] = pre[i];
        }
        __syncthreads();
    }

    // power spectrum -> LDS rows [frame][bin]; lane holds bins 4*kq..4*kq+3 of tile ft for frame fr
    float* P = lds;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j < nft) {
            f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = acc[j][0][r] * acc[j][0][r] + acc[j][1][r] * acc[j][1][r];
            *(f32x4*)(P + fr * LM_PSTRIDE + 16 * (ft0 + j) + 4 * kq) = p;
        }
    }
    __syncthreads();

    // mel bank: out[mel][frame] = sum_bin M[mel][bin] * P[frame][bin]
    const int mt0 = half ? 3 : 0, nmt = half ? 2 : 3;
    f32x4 macc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) macc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < LM_MEL_KS; ++ks) {
        const float pv = P[fr * LM_PSTRIDE + 4 * ks + kq];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nmt)
                macc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(melw[((mt0 + j) * LM_MEL_KS + ks) * 64 + lane], pv,
                                                                macc[j], 0, 0, 0);
    }

    // lane holds mels 16*mt + 4*kq .. +3 of frame t
    const int t = t0 + fr;
    float s0 = VARIANT == AVSR_MEL_WHISPER ? -INFINITY : 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nmt && t < n_frames_out) {
            f32x4 v = macc[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (VARIANT == AVSR_MEL_WHISPER) {
                    v[r] = log10f(fmaxf(v[r], 1e-10f));
                    s0 = fmaxf(s0, v[r]);
                } else {
                    if (t >= n_frames) v[r] = 0.f;
                    s0 += v[r];
                    s1 += v[r] * v[r];
                }
            }
            *(f32x4*)(raw + ((long)b * n_frames_out + t) * LM_NMEL + 16 * (mt0 + j) + 4 * kq) = v;
        }
    }
    if (VARIANT == AVSR_MEL_WHISPER) s0 = wave_max(s0); else { s0 = wave_sum(s0); s1 = wave_sum(s1); }
    if (lane == 0) { red[wv * 2] = s0; red[wv * 2 + 1] = s1; }
    __syncthreads();
    if (tid == 0) {
        float a = red[0], q = red[1];
        for (int i = 1; i < 4; ++i) {
            a = VARIANT == AVSR_MEL_WHISPER ? fmaxf(a, red[2 * i]) : a + red[2 * i];
            q += red[2 * i + 1];
        }
        float* o = slab + ((long)b * gridDim.x + blockIdx.x) * 2;
        o[0] = a;
        o[1] = q;
    }
}

// Applies the clip-wide normalisation and writes the caller's layout / dtype.
template <int VARIANT, typename TO>
__global__ __launch_bounds__(256) void logmel_finalize_kernel(const float* __restrict__ raw,
                                                              const float* __restrict__ slab, int nparts,
                                                              int n_frames_out, TO* __restrict__ out, long sb, long st,
                                                              long sm) {
    __shared__ float tile[LM_FIN_FR * (LM_NMEL + 1)];
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y, t0 = blockIdx.x * LM_FIN_FR;
    const float* sl = slab + (long)b * nparts * 2;
    float mul, add, floorv;
    if (VARIANT == AVSR_MEL_WHISPER) {
        float mx = -INFINITY;
        for (int i = lane; i < nparts; i += 64) mx = fmaxf(mx, sl[2 * i]);
        mx = wave_max(mx);
        floorv = mx - 8.0f; mul = 0.25f; add = 1.0f;          // (max(x, floor) + 4) / 4
    } else {
        double s = 0.0, q = 0.0;
        for (int i = lane; i < nparts; i += 64) { s += sl[2 * i]; q += sl[2 * i + 1]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        const double n = (double)n_frames_out * LM_NMEL, mean = s / n;
        const double var = fmax(q / n - mean * mean, 0.0);
        const float rstd = (float)(1.0 / sqrt(var + 1e-8));
        floorv = -INFINITY; mul = rstd; add = (float)(-mean) * rstd;
    }
    const int nt = min(LM_FIN_FR, n_frames_out - t0);
    const float* src = raw + ((long)b * n_frames_out + t0) * LM_NMEL;
    TO* dst = out + (long)b * sb;
    if (sm == 1) {
        for (int i = tid; i < nt * LM_NMEL; i += 256) {
            const int tt = i / LM_NMEL, m = i - tt * LM_NMEL;
            dst[(long)(t0 + tt) * st + m] = from_f32<TO>(fmaf(fmaxf(src[i], floorv), mul, add));
        }
    } else {
        for (int i = tid; i < nt * LM_NMEL; i += 256) {
            const int tt = i / LM_NMEL, m = i - tt * LM_NMEL;
            tile[tt * (LM_NMEL + 1) + m] = fmaf(fmaxf(src[i], floorv), mul, add);
        }
        __syncthreads();
        for (int i = tid; i < LM_FIN_FR * LM_NMEL; i += 256) {
            const int m = i / LM_FIN_FR, tt = i - m * LM_FIN_FR;
            if (tt < nt) dst[(long)(t0 + tt) * st + (long)m * sm] = from_f32<TO>(tile[tt * (LM_NMEL + 1) + m]);
        }
    }
}

// ---- training-time augmentation of the mel-spectrogram variant (datamodule/transforms.py:122-132) ----
// SpecAugment (:144-179): two frequency bands of `fw` bins and two time bands of `tw` frames set to zero, then
// AddNoise (:47-59), which the reference applies to the [80, frames] mel as if its rows were audio channels:
// row m becomes x_m + scale_m * noise with scale_m = sqrt(E(x_m) / E(noise)) * 10^(-snr/20) (the
// torchaudio.functional.add_noise formula), E = sum of squares over the frames.  A row that is entirely
// zero gets no noise (its energy is 0).  The band starts, the noise segment and the SNR are drawn by the host.
__device__ __forceinline__ bool lm_banded(int x, const int* start, int width) {
    return (start[0] >= 0 && x >= start[0] && x < start[0] + width) || (start[1] >= 0 && x >= start[1] && x < start[1] + width);
}

// One workgroup of 960 threads (12 frame lanes x 80 mel bins) per clip: energies[b][0..79] = per-bin energy of
// the masked mel, energies[b][80] = energy of the noise segment.  Plain stores, fixed order: deterministic.
__global__ __launch_bounds__(960) void mel_aug_energy_kernel(const float* __restrict__ raw, const float* __restrict__ noise,
                                                             long noise_stride, const int* __restrict__ fstart, int fw,
                                                             const int* __restrict__ tstart, int tw, int T,
                                                             float* __restrict__ energies) {
    __shared__ float red[12 * 81];
    const int b = blockIdx.x, tid = threadIdx.x, m = tid % LM_NMEL, tr = tid / LM_NMEL;
    const int fs[2] = {fstart ? fstart[2 * b] : -1, fstart ? fstart[2 * b + 1] : -1};
    const int ts[2] = {tstart ? tstart[2 * b] : -1, tstart ? tstart[2 * b + 1] : -1};
    const bool mdead = lm_banded(m, fs, fw);
    const float* x = raw + (long)b * T * LM_NMEL;
    float e = 0.f, en = 0.f;
    for (int t = tr; t < T; t += 12) {
        const float v = (mdead || lm_banded(t, ts, tw)) ? 0.f : x[(long)t * LM_NMEL + m];
        e = fmaf(v, v, e);
    }
    if (m == 0)
        for (int t = tr; t < T; t += 12) { const float n = noise[(long)b * noise_stride + t]; en = fmaf(n, n, en); }
    red[tr * 81 + m] = e;
    if (m == 0) red[tr * 81 + 80] = en;
    __syncthreads();
    if (tid < 81) {
        float a = 0.f;
        for (int i = 0; i < 12; ++i) a += red[i * 81 + tid];
        energies[b * 81 + tid] = a;
    }
}

// raw <- masked(raw) + scale_m * noise[t]; per-workgroup sum / sum of squares for the clip layer norm, in the slab
// layout logmel_finalize_kernel reads (one entry per LM_FR frames).
__global__ __launch_bounds__(256) void mel_aug_apply_kernel(float* __restrict__ raw, const float* __restrict__ noise,
                                                            long noise_stride, const int* __restrict__ fstart, int fw,
                                                            const int* __restrict__ tstart, int tw, int T,
                                                            const float* __restrict__ energies, const float* __restrict__ snr_db,
                                                            float* __restrict__ slab) {
    __shared__ float scale[LM_NMEL];
    __shared__ float red[8];
    const int b = blockIdx.y, tid = threadIdx.x, t0 = blockIdx.x * LM_FR;
    const int fs[2] = {fstart ? fstart[2 * b] : -1, fstart ? fstart[2 * b + 1] : -1};
    const int ts[2] = {tstart ? tstart[2 * b] : -1, tstart ? tstart[2 * b + 1] : -1};
    if (tid < LM_NMEL) {
        const float es = energies[b * 81 + tid], en = energies[b * 81 + 80];
        scale[tid] = es > 0.f ? sqrtf(es / en) * exp10f(-snr_db[b] * 0.05f) : 0.f;
    }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    float* x = raw + ((long)b * T + t0) * LM_NMEL;
    const int n = min(LM_FR, T - t0) * LM_NMEL;
    for (int i = tid; i < n; i += 256) {
        const int tt = i / LM_NMEL, m = i - tt * LM_NMEL, t = t0 + tt;
        float v = (lm_banded(m, fs, fw) || lm_banded(t, ts, tw)) ? 0.f : x[i];
        v = fmaf(scale[m], noise[(long)b * noise_stride + t], v);
        x[i] = v;
        s0 += v;
        s1 = fmaf(v, v, s1);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s0; red[(tid >> 6) * 2 + 1] = s1; }
    __syncthreads();
    if (tid == 0) {
        float* o = slab + ((long)b * gridDim.x + blockIdx.x) * 2;
        o[0] = red[0] + red[2] + red[4] + red[6];
        o[1] = red[1] + red[3] + red[5] + red[7];
    }
}

inline int lm_nblk(int n_frames_out) { return (n_frames_out + LM_FR - 1) / LM_FR; }

}  // namespace

extern "C" int64_t avsr_logmel_dft_table_floats(void) { return LM_DFT_FLOATS; }
extern "C" int64_t avsr_logmel_mel_table_floats(void) { return LM_MEL_FLOATS; }

extern "C" int avsr_logmel_tables(avsr_stream_t stream, int variant, float* dft_table, float* mel_table) {
    if (!dft_table || !mel_table) return AVSR_EINVAL;
    if (variant != AVSR_MEL_WHISPER && variant != AVSR_MEL_SPEC_LN) return AVSR_EINVAL;
    hipLaunchKernelGGL(logmel_tables_kernel, dim3((LM_DFT_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       variant, dft_table, mel_table);
    return avsr_launch_status();
}

extern "C" int64_t avsr_logmel_workspace_floats(int64_t batch, int64_t n_frames_out) {
    if (batch <= 0 || n_frames_out <= 0) return 0;
    return batch * n_frames_out * LM_NMEL + batch * lm_nblk((int)n_frames_out) * 2;
}

namespace {
int logmel_fwd_impl(avsr_stream_t stream, int variant, int out_dtype, const float* wave,
                    int64_t wave_stride, int64_t n_samples, int64_t n_valid, int64_t batch,
                    const float* dft_table, const float* mel_table, void* out, int64_t out_stride_b,
                    int64_t out_stride_t, int64_t out_stride_m, int64_t n_frames_out, float* workspace) {
    if (!wave || !dft_table || !mel_table || !out || !workspace) return AVSR_EINVAL;
    if (batch <= 0 || n_frames_out <= 0 || n_valid < 0 || n_valid > n_samples || wave_stride < n_valid)
        return AVSR_EINVAL;
    if (out_dtype != AVSR_F32 && out_dtype != AVSR_BF16) return AVSR_EINVAL;
    if (n_samples <= LM_NFFT / 2 || n_samples > (1 << 30) || batch > 65535) return AVSR_EUNSUPPORTED;  // reflect padding needs > 200 samples
    int n_frames;
    if (variant == AVSR_MEL_WHISPER) {
        n_frames = (int)(n_samples / LM_HOP);                 // the last STFT frame is dropped (:145)
        if (n_frames_out > n_frames) return AVSR_EINVAL;
    } else if (variant == AVSR_MEL_SPEC_LN) {
        n_frames = (int)(1 + n_samples / LM_HOP);
        if (n_frames > n_frames_out) n_frames = (int)n_frames_out;
    } else return AVSR_EINVAL;
    const int nblk = lm_nblk((int)n_frames_out);
    float* raw = workspace;
    float* slab = workspace + batch * n_frames_out * LM_NMEL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblk, (unsigned)batch), fgrid(((int)n_frames_out + LM_FIN_FR - 1) / LM_FIN_FR, (unsigned)batch);
    if (variant == AVSR_MEL_WHISPER) {
        hipLaunchKernelGGL(logmel_dft_mel_kernel<AVSR_MEL_WHISPER>, grid, dim3(256), 0, s, wave, (long)wave_stride,
                           (int)n_samples, (int)n_valid, n_frames, (int)n_frames_out, dft_table, mel_table, raw, slab);
        if (out_dtype == AVSR_F32)
            hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_WHISPER, float>), fgrid, dim3(256), 0, s, raw, slab,
                               nblk, (int)n_frames_out, (float*)out, (long)out_stride_b, (long)out_stride_t,
                               (long)out_stride_m);
        else
            hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_WHISPER, bf16_t>), fgrid, dim3(256), 0, s, raw, slab,
                               nblk, (int)n_frames_out, (bf16_t*)out, (long)out_stride_b, (long)out_stride_t,
                               (long)out_stride_m);
    } else {
        hipLaunchKernelGGL(logmel_dft_mel_kernel<AVSR_MEL_SPEC_LN>, grid, dim3(256), 0, s, wave, (long)wave_stride,
                           (int)n_samples, (int)n_valid, n_frames, (int)n_frames_out, dft_table, mel_table, raw, slab);
        if (out_dtype == AVSR_F32)
            hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_SPEC_LN, float>), fgrid, dim3(256), 0, s, raw, slab,
                               nblk, (int)n_frames_out, (float*)out, (long)out_stride_b, (long)out_stride_t,
                               (long)out_stride_m);
        else
            hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_SPEC_LN, bf16_t>), fgrid, dim3(256), 0, s, raw, slab,
                               nblk, (int)n_frames_out, (bf16_t*)out, (long)out_stride_b, (long)out_stride_t,
                               (long)out_stride_m);
    }
    return avsr_launch_status();
}

// Scratch of avsr_logmel_fwd_static: one copy per GPU, held from the moment the code object is loaded on that GPU
// until the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress - the kernels
// get the per-device address that call returns.
__device__ __align__(256) float scratch[AVSR_LOGMEL_STATIC_FLOATS];
constexpr int LM_MAX_DEVICES = 64;
std::mutex scratch_mutex;
float* scratch_ptr[LM_MAX_DEVICES] = {};

// address of the current device's copy of `scratch` (looked up once per device); nullptr + *status on failure
float* logmel_static_scratch(hipStream_t stream, int* status) {
    int dev = -1;
    *status = AVSR_EHIP;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    if (dev >= LM_MAX_DEVICES) { *status = AVSR_EUNSUPPORTED; return nullptr; }
    if (stream) {                                             // the kernels run on the stream's device: it has to be this one
        int sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (sdev != dev) { *status = AVSR_EINVAL; return nullptr; }
    }
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        scratch_ptr[dev] = (float*)p;
    }
    return scratch_ptr[dev];
}
}  // namespace

extern "C" int avsr_logmel_fwd(avsr_stream_t stream, int variant, int out_dtype, const float* wave,
                               int64_t wave_stride, int64_t n_samples, int64_t n_valid, int64_t batch,
                               const float* dft_table, const float* mel_table, void* out, int64_t out_stride_b,
                               int64_t out_stride_t, int64_t out_stride_m, int64_t n_frames_out, float* workspace) {
    return logmel_fwd_impl(stream, variant, out_dtype, wave, wave_stride, n_samples, n_valid, batch, dft_table,
                           mel_table, out, out_stride_b, out_stride_t, out_stride_m, n_frames_out, workspace);
}

extern "C" int avsr_logmel_fwd_static(avsr_stream_t stream, int variant, int out_dtype, const float* wave,
                                      int64_t wave_stride, int64_t n_samples, int64_t n_valid, int64_t batch,
                                      const float* dft_table, const float* mel_table, void* out,
                                      int64_t out_stride_b, int64_t out_stride_t, int64_t out_stride_m,
                                      int64_t n_frames_out) {
    if (batch <= 0 || n_frames_out <= 0) return AVSR_EINVAL;
    if (batch > 65535 || n_frames_out > AVSR_LOGMEL_STATIC_FLOATS) return AVSR_EUNSUPPORTED;
    if (avsr_logmel_workspace_floats(batch, n_frames_out) > AVSR_LOGMEL_STATIC_FLOATS) return AVSR_EUNSUPPORTED;
    int status = AVSR_EHIP;
    float* ws = logmel_static_scratch((hipStream_t)stream, &status);
    if (!ws) return status;
    // every word of the workspace the finalize kernel reads is written by the first kernel of the same call,
    // so nothing of the previous call's contents has to be cleared
    return logmel_fwd_impl(stream, variant, out_dtype, wave, wave_stride, n_samples, n_valid, batch, dft_table,
                           mel_table, out, out_stride_b, out_stride_t, out_stride_m, n_frames_out, ws);
}

// Debugging aid (not part of the declared interface): overwrite the current device's static scratch with 0xFF bytes
// (every float a NaN), so a test can show that a call reads nothing an earlier call left there.
extern "C" int avsr_debug_logmel_static_poison(avsr_stream_t stream) {
    int status = AVSR_EHIP;
    float* ws = logmel_static_scratch((hipStream_t)stream, &status);
    if (!ws) return status;
    return hipMemsetAsync(ws, 0xFF, sizeof(float) * (size_t)AVSR_LOGMEL_STATIC_FLOATS, (hipStream_t)stream) == hipSuccess
               ? AVSR_OK : AVSR_EHIP;
}

extern "C" int64_t avsr_logmel_augment_workspace_floats(int64_t batch, int64_t n_frames_out) {
    if (batch <= 0 || n_frames_out <= 0) return 0;
    return avsr_logmel_workspace_floats(batch, n_frames_out) + batch * 81;
}

extern "C" int avsr_logmel_augment_fwd(avsr_stream_t stream, int out_dtype, const float* wave, int64_t wave_stride,
                                       int64_t n_samples, int64_t n_valid, int64_t batch, const float* dft_table,
                                       const float* mel_table, void* out, int64_t out_stride_b, int64_t out_stride_t,
                                       int64_t out_stride_m, int64_t n_frames_out, float* workspace,
                                       const int32_t* freq_start, int64_t freq_width, const int32_t* time_start,
                                       int64_t time_width, const float* noise, int64_t noise_stride, const float* snr_db) {
    if (!wave || !dft_table || !mel_table || !out || !workspace || !noise || !snr_db) return AVSR_EINVAL;
    if (batch <= 0 || n_frames_out <= 0 || n_valid < 0 || n_valid > n_samples || wave_stride < n_valid) return AVSR_EINVAL;
    if (freq_width < 0 || time_width < 0 || (noise_stride != 0 && noise_stride < n_frames_out)) return AVSR_EINVAL;
    if (out_dtype != AVSR_F32 && out_dtype != AVSR_BF16) return AVSR_EINVAL;
    if (n_samples <= LM_NFFT / 2 || n_samples > (1 << 30) || batch > 65535) return AVSR_EUNSUPPORTED;
    int n_frames = (int)(1 + n_samples / LM_HOP);
    if (n_frames > n_frames_out) n_frames = (int)n_frames_out;
    const int nblk = lm_nblk((int)n_frames_out), T = (int)n_frames_out;
    float* raw = workspace;
    float* slab = workspace + batch * n_frames_out * LM_NMEL;
    float* energies = slab + batch * nblk * 2;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblk, (unsigned)batch), fgrid((T + LM_FIN_FR - 1) / LM_FIN_FR, (unsigned)batch);
    hipLaunchKernelGGL(logmel_dft_mel_kernel<AVSR_MEL_SPEC_LN>, grid, dim3(256), 0, s, wave, (long)wave_stride,
                       (int)n_samples, (int)n_valid, n_frames, T, dft_table, mel_table, raw, slab);
    hipLaunchKernelGGL(mel_aug_energy_kernel, dim3((unsigned)batch), dim3(960), 0, s, raw, noise, (long)noise_stride,
                       freq_start, (int)freq_width, time_start, (int)time_width, T, energies);
    hipLaunchKernelGGL(mel_aug_apply_kernel, grid, dim3(256), 0, s, raw, noise, (long)noise_stride, freq_start,
                       (int)freq_width, time_start, (int)time_width, T, energies, snr_db, slab);
    if (out_dtype == AVSR_F32)
        hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_SPEC_LN, float>), fgrid, dim3(256), 0, s, raw, slab, nblk, T,
                           (float*)out, (long)out_stride_b, (long)out_stride_t, (long)out_stride_m);
    else
        hipLaunchKernelGGL((logmel_finalize_kernel<AVSR_MEL_SPEC_LN, bf16_t>), fgrid, dim3(256), 0, s, raw, slab, nblk, T,
                           (bf16_t*)out, (long)out_stride_b, (long)out_stride_t, (long)out_stride_m);
    return avsr_launch_status();
}
