/*
 * avsr_hip.h - C ABI of libavsr_hip.so: hand-written HIP (gfx950 / CDNA4)
 * kernels for the audio-visual forward/backward of the reference's
 * models/av_net.py.
 *
 * The reference has no FFI, plugin table or native code: every entry below
 * replaces a *library call* the reference's Python makes (PyTorch / cuDNN /
 * cuBLAS through nn.Module calls).  The reference call site each entry stands
 * in for is cited as file:line relative to the reference checkout;
 * "hf:" = transformers/models/whisper/modeling_whisper.py.
 *
 * Conventions
 *  - raw device pointers, explicit sizes / leading dimensions (in ELEMENTS),
 *    a hipStream_t; the caller allocates every output and workspace;
 *  - returns 0 on success, a negative AVSR_E* code on a bad argument or an
 *    unsupported shape, never throws, never aborts; HIP launch errors are
 *    returned as AVSR_EHIP;
 *  - stream ordered, no implicit synchronisation, no hidden global state,
 *    no allocation;
 *  - dtype codes: AVSR_F32 = 0, AVSR_BF16 = 1.  "act" is the activation
 *    storage type; accumulation is always fp32.
 */
#ifndef AVSR_HIP_H
#define AVSR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* avsr_stream_t; /* hipStream_t */

enum { AVSR_F32 = 0, AVSR_BF16 = 1 };
/* RELU_AFTER_RESIDUAL: relu(residual + v), the tail of a ResNet bottleneck */
enum { AVSR_ACT_NONE = 0, AVSR_ACT_GELU = 1, AVSR_ACT_RELU = 2, AVSR_ACT_RELU_AFTER_RESIDUAL = 3 };
enum { AVSR_OK = 0, AVSR_EINVAL = -1, AVSR_EUNSUPPORTED = -2, AVSR_EHIP = -3 };

/* library / build identification: returns a static string */
const char* avsr_version(void);

/* ------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[m,n] = epi( sum_k A[m,k] * B[n,k] )   ("NT")
 * Replaces every nn.Linear / F.linear on the path (models/av_net.py:35,48,
 * models/gate_cross_attention.py:16,23-25,56-57, models/utils.py:34,
 * hf:277-281,374-376) and, in conv mode, the im2col GEMM of nn.Conv1d /
 * nn.Conv2d (hf:566-567, torchvision Bottleneck convs instantiated at
 * models/moco_visual_frontend.py:20).
 *
 * epilogue, in this order:  v = acc * alpha;  v *= scale[n];  v += bias[n];
 *   v = act(v);  (optional extended steps, see the end of the struct;)
 *   v = residual[m,n] + g * v  (g = tanh(*gate) or 1);
 *   (AVSR_ACT_RELU_AFTER_RESIDUAL applies the ReLU after the residual add;)
 *   optional per-column partial sums of the PRE-activation value into the
 *   col_sum / col_sumsq slabs for batch-norm statistics.
 *
 * conv mode (conv.enabled): A is an NHWC image [n_img,H,W,C]; row m of the
 * implicit A matrix is output pixel (img,oh,ow), column k = (kh*KW+kw)*C + c;
 * B is [N][KH*KW*C]; C is the NHWC output [n_img*OH*OW, N].  M must equal
 * n_img*OH*OW and K must equal KH*KW*C.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t in_dtype;   /* dtype of A and B */
    int32_t out_dtype;  /* dtype of C and residual */
    int64_t M, N, K;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc;
    int64_t batch;      /* >= 1 */
    int64_t strideA, strideB, strideC; /* elements between batch items */
    float alpha;
    const float* scale;   /* [N] or NULL */
    const float* bias;    /* [N] or NULL */
    int32_t act;
    const void* residual; /* [M,N] (ld = ldr) of out_dtype, or NULL; batch > 1 only as a shared table (res_period > 0) */
    int64_t ldr;
    int64_t res_period;   /* > 0: residual row = m % res_period (a [period,N] table, e.g. positions) */
    const float* gate;    /* device scalar or NULL */
    float* col_sum;       /* [avsr_gemm_stats_parts(), N] partial sums, or NULL */
    float* col_sumsq;     /* same shape, partial sums of squares */
    struct {
        int32_t enabled;
        int32_t H, W, C, KH, KW, stride, pad_h, pad_w, OH, OW;
    } conv;
    /* extended epilogue (dense products, batch 1, act NONE or GELU; all optional, zero = off), applied in this
     * order after scale / bias:  pre[m,n] = v (a second output of out_dtype: the value BEFORE the activation and
     * the gate);  v = act(v);  v *= gelu'(mul_gelu_grad_of[m,n]) (backward through a GELU whose input was saved);
     * v = dropout(v) with keep decision generator(seed, m*N + n), kept values scaled by 1/(1-p_drop);
     * then the residual / gate step above.
     * C == NULL (bf16 inputs, col_sum / col_sumsq given, no residual): a statistics-only pass - the launch
     * fills the column-sum slabs and stores no product (batch statistics of a convolution whose output is
     * cheaper to recompute with the normalisation folded in than to store raw and re-read). */
    void* pre; int64_t ldp;
    const void* mul_gelu_grad_of; int64_t ld_aux;
    float p_drop;
    uint64_t seed;
    /* optional device word added to `seed` when the kernel runs (NULL = none): lets a launch recorded in a HIP
     * graph draw a different dropout mask on every replay (see avsr_step_params) */
    const uint64_t* seed_delta;
    /* optional (bf16 inputs, act NONE, dense, batch 1): per row and 128-column tile of the product, the partial
     * log-sum-exp statistics of the values `pre` would hold - row_part[(tile * M + m) * 4 + {0, 1, 2}] = {max, sum of
     * exp(v - max), sum of v} over the tile's valid columns, tile = n / 128, ceil(N / 128) tiles.  Lets the consumer of a
     * very wide product (the vocabulary head) get every row's log-sum-exp without reading the product back:
     * avsr_head_loss_fwd(..., row_part, n_parts). */
    float* row_part;
    /* A LayerNorm folded into the two products around it (HF modeling_whisper.py WhisperEncoderLayer.forward:
     * self_attn_layer_norm -> q/k/v_proj, final_layer_norm -> fc1; dense unbatched bf16 products, N % 8 == 0,
     * alpha == 1, all NULL / 0 = off).  The product that WRITES the LayerNorm's input (fp32 C with a residual,
     * no activation) sets ln_out and copy_bf16: besides C it stores a bf16 copy of the rows (row stride ld_copy)
     * and ln_out[(tile * M + m) * 2 + {0, 1}] = {sum, sum of squares} of row m over the columns of 128-column
     * tile `tile` (ceil(N / 128) tiles).  The product that READS the normalised rows takes that copy as A, a weight
     * whose columns are pre-multiplied by the LayerNorm's gain as B, and sets ln_in (= the other launch's ln_out),
     * ln_parts (its tile count, at most 10), ln_eps and ln_colsum[n] = sum_k B[n, k]; `bias` must already hold
     * bias[n] + sum_k beta[k] W[n, k].  Its epilogue computes, before the activation,
     *   rstd[m] * acc[m, n] - rstd[m] * mean[m] * ln_colsum[n] + bias[n],
     * which is LayerNorm(A)[m, :] . W[n, :] + bias[n] with mean / rstd over A's K columns: the normalised tensor is
     * never written or read.
     * The fp32 rows themselves can live as TWO bf16 planes instead of an fp32 matrix: copy_bf16 = bf16(v) and
     * copy_lo = bf16(v - copy_bf16) (together v to 2^-18 relative; same bytes as fp32, and the first plane IS the next
     * product's operand, so the writing product stores nothing extra).  With copy_lo given C may be NULL; the
     * residual of such a product is either `residual` (fp32) or residual_hi + residual_lo (two planes, row stride
     * ld_res_planes; may alias the output planes: an in-place update).  ln_out NULL with planes / C given: no
     * statistics (the last product of a chain, whose reader is not a folded LayerNorm). */
    const float* ln_in; int64_t ln_parts; float ln_eps; const float* ln_colsum;
    float* ln_out; void* copy_bf16; int64_t ld_copy;
    void* copy_lo; const void* residual_hi; const void* residual_lo; int64_t ld_res_planes;
} avsr_gemm_t;

int avsr_gemm_nt(avsr_stream_t stream, const avsr_gemm_t* g);
/* rows of the [parts, N] col_sum / col_sumsq slabs a launch with M rows fills
 * (one row per producing workgroup / wave: plain stores, deterministic) */
int64_t avsr_gemm_stats_parts(int in_dtype, int64_t M);

/* ------------------------------------------------------------------------
 * LayerNorm over the last dimension, one wave per row, fp32 statistics.
 * Replaces nn.LayerNorm / the fp32-upcasting LayerNorm subclass
 * (models/gate_cross_attention.py:8-10; hf:371,377,573).  `add` (optional)
 * is a [add_period, D] fp32 table added after the affine, row r uses
 * add[r % add_period] - the sinusoidal PositionalEncoding of
 * models/utils.py:21-28 fused behind audio_ln / video_ln
 * (models/av_net.py:83-84,88-89).  mean / rstd (optional) are saved for the
 * backward.  D % 4 == 0, D <= 2048.
 * ---------------------------------------------------------------------- */
int avsr_layernorm_fwd(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx,
                       void* y, int64_t ldy, const float* gamma, const float* beta, const float* add,
                       int64_t add_period, float* mean, float* rstd, int64_t rows, int64_t D, float eps);
/* dgamma / dbeta are ACCUMULATED into (fp32 atomics); zero them first. */
int avsr_layernorm_bwd(avsr_stream_t stream, int x_dtype, int g_dtype, const void* dy, int64_t lddy,
                       const void* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                       void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int64_t D);

/* the same with the gradient that reaches x past the norm added in: dx = LN'(dy) + dres (dres of g_dtype,
 * may be NULL) - the residual connection around a pre-norm branch, models/gate_cross_attention.py:43,47 */
int avsr_layernorm_bwd_res(avsr_stream_t stream, int x_dtype, int g_dtype, const void* dy, int64_t lddy,
                           const void* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                           void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                           const void* dres, int64_t lddr);

/* ------------------------------------------------------------------------
 * Attention, head_dim 64.  Element (b,t,h,d) of Q/K/V/O lives at
 * ptr + (b*T + t)*ld + h*64 + d.  key_mask: uint8 [B,Tk], 1 = ignore key
 * (key_padding_mask of nn.MultiheadAttention, models/gate_cross_attention.py:40);
 * a query row whose keys are all masked yields zeros.  lse (optional for the
 * forward, required by the backward): fp32 [B,H,Tq].
 * use_mfma = 1: bf16 flash kernels on v_mfma_f32_32x32x16_bf16 (the Whisper
 *   encoder self-attention, hf:284-356 / 215-238, and the core of the gated
 *   cross-attention, models/gate_cross_attention.py:40, forward + backward).
 * use_mfma = 2 (forward only, no dropout): the same flash scheme with both
 *   products on v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3).  Operands stay bf16 in
 *   memory; Q/K/V are rounded to fp8 on the way in, the softmax runs in fp32 and
 *   P is rounded to fp8 (scaled by 128) for the P.V product.
 * use_mfma = 0: exact fp32-math kernels, any dtype.
 * Attention dropout (p_drop > 0) drops softmax probabilities with a
 * counter-based generator: element (row = (b*H+h)*Tq + q, key) uses output
 * (key & 3) of counter row*ceil(Tk/4) + key/4 under `seed`; forward and
 * backward, MFMA and fp32 kernels all draw the same mask.
 * causal = 1 (forward only; the Whisper decoder self-attention,
 * hf modeling_whisper.py:416-505, is_causal): query i sees keys
 * <= i + (Tk - Tq), i.e. the queries are the last Tq positions of the
 * Tk-long key sequence; needs Tk >= Tq; combines with key_mask.
 * ---------------------------------------------------------------------- */
int avsr_attn_fwd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                  int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                  float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                  float p_drop, uint64_t seed, int causal);
/* The same forward with a workspace that lets the bf16 MFMA kernel even out its last round of workgroups: when the
 * number of (batch, head, 128-query tile) items leaves the chip's last round at most half full (Whisper-small at
 * batch 8: 1152 items on 768 resident workgroups), the items of that round are computed by 2-4 workgroups each, every
 * one over its share of the keys, and the last to finish sums the shares (same result up to fp32 summation order).
 * workspace: 256-byte aligned, at least avsr_attn_fwd_workspace_bytes(...) bytes (0 = this shape is not split);
 * its first ceil(4*items/256)*256 bytes are counters that must be ZERO before the first call and are left zero by
 * every call, so one buffer zeroed once serves a whole stream of calls; calls that share a workspace must not
 * overlap.  workspace == NULL or too small: exactly avsr_attn_fwd.
 * The split is only offered on the full 256-CU / 8-XCD device the workgroup-id mapping was measured on (all shares
 * of an item have equal blockIdx & 7, i.e. share one XCD's L2 under round-robin placement; the shares themselves
 * travel by system-scope stores and loads, so the sum does not depend on that placement): on any other device or
 * CU partition avsr_attn_fwd_workspace_bytes returns 0 and the call runs unsplit, and so does a call on a stream
 * created with a CU mask (hipExtStreamCreateWithCUMask), whatever the workspace.  The current device must be the
 * stream's device. */
int64_t avsr_attn_fwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk, float p_drop, int causal);
int avsr_attn_fwd_ws(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                     int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, const uint8_t* key_mask,
                     float* lse, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, float scale,
                     float p_drop, uint64_t seed, int causal, void* workspace, int64_t workspace_bytes);
/* d_ws: fp32 [B,H,Tq] scratch (row sums of dO*O).  Not for a causal forward. */
int avsr_attn_bwd(avsr_stream_t stream, int dtype, int use_mfma, const void* Q, int64_t ldq, const void* K,
                  int64_t ldk, const void* V, int64_t ldv, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                  void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv,
                  const uint8_t* key_mask, const float* lse, float* d_ws, int64_t B, int64_t H, int64_t Tq,
                  int64_t Tk, int64_t head_dim, float scale, float p_drop, uint64_t seed);

/* ------------------------------------------------------------------------
 * One tanh-gated cross-attention block, forward and backward, as one call each:
 * GatedCrossAttentionBlock.forward (models/gate_cross_attention.py:28-50) with its nn.MultiheadAttention
 * (:16,40), the two fp32 LayerNorms (:8-10,17-18), the GELU feed-forward (:23-26) and the tanh gates (:19-20):
 *   x1  = x  + tanh(gate_attn) * MHA(LN1(x), xa, xa, key_padding_mask)        (skipped when xa == NULL)
 *   out = x1 + tanh(gate_ff)   * Dropout(W2 gelu(W1 LN2(x1) + b1) + b2)
 * x, out: [B*Tq, d]; xa: [B*Tk, d]; all row-major and dense, of `dtype` (AVSR_F32 or AVSR_BF16), d = 64*H.
 * Weights are the caller's copies in `dtype`: W as nn.Linear stores it ([out, in]) for the forward, W^T
 * ([in, ld >= out]) for the input gradients of the backward (NULL is fine for a forward-only call); biases,
 * LayerNorm parameters and the two gates are the fp32 parameters themselves (gates: device scalars).
 * The call queues the block's kernel chain on `stream`: LayerNorm, projection GEMMs, the flash attention of
 * avsr_attn_fwd (key mask, attention dropout with seed_attn), and GEMMs whose epilogues apply bias, GELU,
 * feed-forward dropout (element (row, col) keeps iff generator(seed_ff, row*d + col) says so), tanh gate and
 * residual.  `saved` (avsr_gated_xattn_fwd_workspace_bytes) receives the activations the backward needs.
 * Backward: dx = d(loss)/dx; d(loss)/d(xa) is written to dxa, or added to it when accumulate_dxa != 0 (xa
 * feeds every block of the fusion); parameter gradients are ACCUMULATED into the fp32 buffers of
 * avsr_xattn_grads_t (shapes of the parameters; the gate gradients are single floats).
 * `workspace`: avsr_gated_xattn_bwd_workspace_bytes bytes of scratch.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;
    int64_t B, Tq, Tk, d, H, d_ff;
    float p_drop;                    /* 0 in eval mode */
    uint64_t seed_attn, seed_ff;
    const uint64_t* seed_delta;      /* optional device word added to both seeds at run time (NULL = none): graph replays */
    const void* w_in;  const void* w_inT;  int64_t ld_inT;  const float* b_in;    /* attn.in_proj_weight [3d, d], ^T [d, ld] */
    const void* w_out; const void* w_outT; int64_t ld_outT; const float* b_out;   /* attn.out_proj       [d, d]            */
    const void* w_fc1; const void* w_fc1T; int64_t ld_fc1T; const float* b_fc1;   /* ff.0                [d_ff, d]         */
    const void* w_fc2; const void* w_fc2T; int64_t ld_fc2T; const float* b_fc2;   /* ff.2                [d, d_ff]         */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;                                   /* attn_ln, ff_ln                        */
    const float *gate_attn, *gate_ff;
} avsr_xattn_block_t;
typedef struct {
    float *dw_in, *db_in, *dw_out, *db_out, *dw_fc1, *db_fc1, *dw_fc2, *db_fc2;
    float *dln1_g, *dln1_b, *dln2_g, *dln2_b, *dgate_attn, *dgate_ff;
} avsr_xattn_grads_t;
int64_t avsr_gated_xattn_fwd_workspace_bytes(const avsr_xattn_block_t* blk);
int64_t avsr_gated_xattn_bwd_workspace_bytes(const avsr_xattn_block_t* blk);
int avsr_gated_xattn_fwd(avsr_stream_t stream, const avsr_xattn_block_t* blk, const void* x, const void* xa,
                         const uint8_t* key_mask, void* out, void* saved);
int avsr_gated_xattn_bwd(avsr_stream_t stream, const avsr_xattn_block_t* blk, const void* x, const void* xa,
                         const uint8_t* key_mask, const void* saved, const void* dout, void* dx, void* dxa,
                         int accumulate_dxa, const avsr_xattn_grads_t* grads, void* workspace);

/* ------------------------------------------------------------------------
 * Mel front end: waveform [batch, n_samples] fp32 -> 80-bin mel features.
 *  AVSR_MEL_WHISPER  WhisperFeatureExtractor._torch_extract_fbank_features
 *    (transformers feature_extraction_whisper.py:135-168; reference call site
 *    datamodule/av_dataset.py:47-51): periodic-hann STFT 400/160 with centre +
 *    reflect padding, power of frames 0..n_samples/160-1, Slaney mel bank,
 *    log10(max(.,1e-10)), floor at (clip max - 8), (x+4)/4.
 *  AVSR_MEL_SPEC_LN  torchaudio MelSpectrogram(16000, 400, 400, 160, n_mels 80)
 *    + pad/trim to n_frames_out + layer_norm over the whole [frames, 80] clip,
 *    eps 1e-8 (datamodule/transforms.py:115-121,99-110,131).
 * Samples [n_valid, n_samples) read as zero (the extractor's zero padding to
 * 30 s); reflection happens at n_samples.  out element (b, t, m) is written to
 * out[b*out_stride_b + t*out_stride_t + m*out_stride_m]: strides (80*T, 80, 1)
 * give the channels-last [B,T,80] the encoder consumes, (80*T, 1, T) the
 * reference's [B,80,T].  The two constant tables (DFT matrix with the window
 * folded in, mel bank; both in MFMA fragment order) are filled once by
 * avsr_logmel_tables; workspace: avsr_logmel_workspace_floats() floats.
 * ---------------------------------------------------------------------- */
enum { AVSR_MEL_WHISPER = 0, AVSR_MEL_SPEC_LN = 1 };
int64_t avsr_logmel_dft_table_floats(void);
int64_t avsr_logmel_mel_table_floats(void);
int avsr_logmel_tables(avsr_stream_t stream, int variant, float* dft_table, float* mel_table);
int64_t avsr_logmel_workspace_floats(int64_t batch, int64_t n_frames_out);
int avsr_logmel_fwd(avsr_stream_t stream, int variant, int out_dtype, const float* wave, int64_t wave_stride,
                    int64_t n_samples, int64_t n_valid, int64_t batch, const float* dft_table,
                    const float* mel_table, void* out, int64_t out_stride_b, int64_t out_stride_t,
                    int64_t out_stride_m, int64_t n_frames_out, float* workspace);
/* avsr_logmel_fwd without a workspace argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The workspace is a fixed array of AVSR_LOGMEL_STATIC_FLOATS floats (8 MiB) inside the library's code object:
 * every GPU the library runs on holds one copy until the process ends.  The call allocates nothing; it returns
 * AVSR_EUNSUPPORTED when avsr_logmel_workspace_floats(batch, n_frames_out) exceeds AVSR_LOGMEL_STATIC_FLOATS
 * (8 clips of 30 s fit).  All calls on one GPU share that GPU's array: calls that may overlap on one GPU (two
 * streams, two threads) race - order them on one stream.  Results are bit-identical to avsr_logmel_fwd.
 * The array used is the CURRENT device's, so the current device must be the one `stream` belongs to: a stream
 * of another device is refused with AVSR_EINVAL (checked with hipStreamGetDevice; the NULL stream is the current
 * device's).  Devices numbered 64 and up are not served (AVSR_EUNSUPPORTED). */
#define AVSR_LOGMEL_STATIC_FLOATS (2 * 1024 * 1024)
int avsr_logmel_fwd_static(avsr_stream_t stream, int variant, int out_dtype, const float* wave, int64_t wave_stride,
                           int64_t n_samples, int64_t n_valid, int64_t batch, const float* dft_table,
                           const float* mel_table, void* out, int64_t out_stride_b, int64_t out_stride_t,
                           int64_t out_stride_m, int64_t n_frames_out);
/* Training-time form of the AVSR_MEL_SPEC_LN variant (datamodule/transforms.py:122-132): between the mel
 * bank and the clip layer norm it applies SpecAugment (transforms.py:144-179: per clip two frequency bands
 * of freq_width bins and two time bands of time_width frames set to zero; starts in freq_start / time_start,
 * device int32 [batch,2], a negative start = band not drawn, either pointer may be NULL; the reference calls the
 * function on the [80, frames] tensor, transforms.py:127 before the transpose at :129, which makes its bands
 * freq_width = 80/8 = 10 mel bins and time_width = 48 frames) and AddNoise
 * (transforms.py:47-59 with torchaudio.functional.add_noise: mel row m += sqrt(E_m / E_noise) * 10^(-snr/20)
 * * noise, E = sum of squares over the frames; noise: fp32, n_frames_out values per clip at
 * noise + clip * noise_stride (stride 0 = one segment for all clips); snr_db: device fp32 [batch]).
 * The random draws are the host's.  workspace: avsr_logmel_augment_workspace_floats() floats. */
int64_t avsr_logmel_augment_workspace_floats(int64_t batch, int64_t n_frames_out);
int avsr_logmel_augment_fwd(avsr_stream_t stream, int out_dtype, const float* wave, int64_t wave_stride,
                            int64_t n_samples, int64_t n_valid, int64_t batch, const float* dft_table,
                            const float* mel_table, void* out, int64_t out_stride_b, int64_t out_stride_t,
                            int64_t out_stride_m, int64_t n_frames_out, float* workspace,
                            const int32_t* freq_start, int64_t freq_width, const int32_t* time_start,
                            int64_t time_width, const float* noise, int64_t noise_stride, const float* snr_db);

/* ------------------------------------------------------------------------
 * Visual frontend pieces (models/moco_visual_frontend.py).
 * avsr_stem3d_fwd: Conv3d(3,64,(5,3,3),s(1,2,2),p(2,3,3),bias=False) (:15) on
 *   video [B,T,3,H,W] fp32 -> NHWC [B*T,OH,OW,64]; weight_packed is the
 *   weight transposed to [135 taps][64]; weight_frags (optional) is its bf16
 *   fragment copy from avsr_stem3d_pack_frags: when given with a bf16 output
 *   the convolution runs as an implicit GEMM on MFMA over a bf16 copy of the
 *   input window, otherwise as a direct fp32 convolution.  With scale/shift the eval-mode
 *   BatchNorm3d + ReLU (:16-17) are folded in; otherwise the raw convolution
 *   is stored and per-workgroup channel sums / sums of squares go to the
 *   [avsr_stem3d_stats_parts(), 64] slabs (training-mode batch statistics).
 * avsr_maxpool3x3s2: MaxPool3d((1,3,3),(1,2,2),(0,1,1)) (:18) over NHWC, with
 *   an optional per-channel affine + ReLU applied to each element first.
 * avsr_avgpool_mask_fwd: AdaptiveAvgPool2d(1) of the trunk + masked_fill of
 *   frames t >= lens[b] (:42,45,49).
 * avsr_bn_*: BatchNorm2d/3d in training mode (batch statistics -> affine,
 *   running-stat update with the unbiased variance) and eval mode (fold).
 * ---------------------------------------------------------------------- */
int avsr_stem3d_fwd(avsr_stream_t stream, int out_dtype, const float* video, const float* weight_packed,
                    const void* weight_frags, void* out, const float* scale, const float* shift, float* col_sum,
                    float* col_sumsq, int64_t B, int64_t T, int64_t H, int64_t W);
/* avsr_stem3d_fwd whose input address is read from the 8-byte device word *video_slot when the kernel RUNS (no
 * counterpart in the reference: a launch recorded in a HIP graph then follows the caller's current batch -
 * models/av_net.py:240 hands a fresh videoBatch every step - without a copy into a fixed buffer).  bf16 MFMA form
 * only (weight_frags required, W % 4 == 0); the stored address must be 16-byte aligned and stay valid until the
 * launch has finished.  avsr_store_u64 writes such a word in stream order (and, optionally, copies n_words more
 * 8-byte words from copy_src to copy_dst in the same launch: the other small inputs of such a recording). */
int avsr_stem3d_fwd_indirect(avsr_stream_t stream, int out_dtype, const float* const* video_slot,
                             const float* weight_packed, const void* weight_frags, void* out, const float* scale,
                             const float* shift, float* col_sum, float* col_sumsq, int64_t B, int64_t T, int64_t H,
                             int64_t W);
int avsr_store_u64(avsr_stream_t stream, uint64_t* dst, uint64_t value, uint64_t* copy_dst, const uint64_t* copy_src,
                   int64_t n_words);
/* n <= 4 strided regions (rows[i] x row_bytes[i], row pitch ld_bytes[i]; all multiples of 4 bytes) set to zero by one
 * launch: the zero frames stored before / after each clip so that the padded convolutions of the Whisper stem (HF
 * modeling_whisper.py WhisperEncoder.forward: conv1, conv2 with padding=1) run as plain products. */
int avsr_zero_regions(avsr_stream_t stream, int64_t n, void* const* ptr, const int64_t* ld_bytes, const int64_t* rows,
                      const int64_t* row_bytes);
int64_t avsr_stem3d_stats_parts(int64_t B, int64_t T, int64_t H, int64_t W);
/* bf16 MFMA-fragment copy of the stem weight (avsr_stem3d_frag_elems() bf16 values) from weight_packed */
int64_t avsr_stem3d_frag_elems(void);
int avsr_stem3d_pack_frags(avsr_stream_t stream, const float* weight_packed, void* weight_frags);
int avsr_maxpool3x3s2(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y,
                      const float* scale, const float* shift, int64_t N, int64_t H, int64_t W, int64_t C);
/* The two steps before it in training mode as one launch: the last Bottleneck's bn3 apply + shortcut + ReLU (torchvision
 * Bottleneck.forward, instantiated at moco_visual_frontend.py:27-31) and the pool + mask (:41-51) - the normalised
 * activation nobody else reads is never written.  x, residual: bf16 [B*T, HW, C]; y [B*T, C] bf16 or fp32. */
/* avsr_bn_finalize followed by avsr_bn_apply (bf16 in and out, residual without an affine of its own) as ONE launch, for
 * statistics slabs of at most 256 rows and C a multiple of 64 (the deep, small feature maps of the trunk, where the
 * separate finalize launch costs as much as the apply pass): every workgroup folds the slab rows of its 64 channels
 * itself.  scale_out / shift_out (optional) receive the affine; the running statistics are updated once.
 * torch nn.BatchNorm2d in training mode, as avsr_bn_finalize (moco_visual_frontend.py:27-31 via torchvision). */
int avsr_bn_finalize_apply(avsr_stream_t stream, const void* x, const void* residual, void* y, const float* sum,
                           const float* sumsq, int64_t nparts, double count, const float* weight, const float* bias,
                           float* running_mean, float* running_var, float* scale_out, float* shift_out, int64_t rows,
                           int64_t C, float eps, float momentum, int relu);
int avsr_bn_apply_avgpool_mask(avsr_stream_t stream, int out_dtype, const void* x, const void* residual,
                               const float* scale, const float* shift, void* y, const int64_t* lens, int64_t B,
                               int64_t T, int64_t HW, int64_t C);
int avsr_avgpool_mask_fwd(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y,
                          const int64_t* lens, int64_t B, int64_t T, int64_t HW, int64_t C);
/* VideoTransform of the data path (datamodule/transforms.py:61-96) for frames already in device memory:
 * video [B,T,3,H,W] uint8 (in_is_u8 = 1) or fp32 with values 0..255 -> out fp32 [B,T,3,OH,OW]:
 * bilinear resize (interpolate align_corners=False, :66-68; OH=H, OW=W copies), /255, horizontal flip of the
 * clips with flip[b] != 0 (:70), frames with frame_zero[b*T+t] != 0 zeroed (AdaptiveTimeMask, :27-45,73), then
 * (x - mean[c]) / std[c] (:74,95).  flip / frame_zero: device uint8, may be NULL; mean3 / std3: HOST float[3].
 * The random draws are the host's; ColorJitter / RandomGrayscale (:71-72) are avsr_video_color. */
int avsr_video_preprocess(avsr_stream_t stream, int in_is_u8, const void* video, float* out, int64_t B, int64_t T,
                          int64_t H, int64_t W, int64_t OH, int64_t OW, const uint8_t* flip,
                          const uint8_t* frame_zero, const float* mean3, const float* std3);
/* The colour steps of the training VideoTransform (datamodule/transforms.py:71-74) on frames already scaled to
 * [0, 1] (avsr_video_preprocess with mean 0 / std 1), in place on video fp32 [B,T,3,H,W]: torchvision ColorJitter
 * (per clip: order int32 [B,4] = the drawn permutation of 0 brightness, 1 contrast, 2 saturation, 3 hue, a negative
 * entry skips a step, order == NULL skips all; factors fp32 [B,4] = brightness, contrast, saturation factors and the
 * hue shift), RandomGrayscale (grey uint8 [B], may be NULL), AdaptiveTimeMask (frame_zero uint8 [B*T], may be NULL)
 * and Normalize (mean3 / std3: HOST float[3]).  The random draws are the host's.  3*H*W*4 bytes <= 150 KiB. */
int avsr_video_color(avsr_stream_t stream, float* video, int64_t B, int64_t T, int64_t H, int64_t W,
                     const int32_t* order, const float* factors, const uint8_t* grey, const uint8_t* frame_zero,
                     const float* mean3, const float* std3);
/* sum / sumsq: [nparts, C] partial-sum slabs; ws: avsr_bn_finalize_workspace_floats(C) floats, needed when
 * nparts > 8192.  Optional for 512 <= nparts <= 8192: with ws AND tickets (device int32, ceil(C/64) values, all
 * zero on entry; the kernel leaves them zero) the slab rows are split over up to 16 workgroups per 64 channels
 * and combined in a fixed order by the last one to finish (same result whichever that is).  Calls that share a
 * tickets buffer must be ordered on one stream. */
int64_t avsr_bn_finalize_workspace_floats(int64_t C);
int avsr_bn_finalize(avsr_stream_t stream, const float* sum, const float* sumsq, int64_t nparts, double count,
                     const float* weight, const float* bias, float* running_mean, float* running_var,
                     float* scale, float* shift, int64_t C, float eps, float momentum, float* ws, int32_t* tickets);
/* Batch statistics of a 1x1 convolution's OUTPUT from the second moments of its INPUT: for y = W h per pixel,
 * mean(y_c) = w_c . mean(h) and mean(y_c^2) = w_c^T mean(h h^T) w_c.  With them known before the convolution runs,
 * the expanding convolution of a torchvision Bottleneck (conv3 + bn3 in training mode + residual + ReLU,
 * models/moco_visual_frontend.py:20,45) applies the normalisation in its own epilogue (avsr_gemm_nt with scale /
 * bias / residual / AVSR_ACT_RELU_AFTER_RESIDUAL) instead of storing the raw product for a second pass.
 * avsr_gram_stats: X bf16 [M, K] (row pitch ldx), K = 64 or 128 -> gram_parts fp32 [parts, K, K] and colsum_parts
 * fp32 [parts, K], parts = avsr_gram_parts(M, K): per-share partial sums, plain stores, added in share order later.
 * avsr_bn_gram_finalize: W bf16 [N, K] (the convolution's weight as the product reads it), count = M; writes scale /
 * shift for the N output channels and updates the running statistics exactly as avsr_bn_finalize does.
 * workspace: avsr_bn_gram_workspace_doubles(K) doubles; it is left holding the reduced statistics of the input, and a
 * later call with gram_parts == colsum_parts == NULL and the same workspace (another convolution over the same
 * input) uses them as they are instead of reducing the partials again. */
int64_t avsr_gram_parts(int64_t M, int64_t K);
int avsr_gram_stats(avsr_stream_t stream, const void* X, int64_t ldx, int64_t M, int64_t K, float* gram_parts,
                    float* colsum_parts);
/* avsr_gram_stats of relu(X * scale + shift) - X being the RAW output of the convolution before (conv2 of the
 * Bottleneck, its bn2 affine in scale / shift) - which also stores that normalised tensor to Y (row pitch ldy; Y may
 * be X): bn2's apply pass rides in the statistics pass that reads the tensor anyway. */
int avsr_gram_stats_bn(avsr_stream_t stream, const void* X, int64_t ldx, int64_t M, int64_t K, const float* scale,
                       const float* shift, void* Y, int64_t ldy, float* gram_parts, float* colsum_parts);
int64_t avsr_bn_gram_workspace_doubles(int64_t K);
int avsr_bn_gram_finalize(avsr_stream_t stream, const float* gram_parts, const float* colsum_parts, int64_t parts,
                          int64_t K, double count, const void* W_bf16, int64_t ldw, int64_t N, const float* weight,
                          const float* bias, float* running_mean, float* running_var, float* scale, float* shift,
                          float eps, float momentum, double* workspace);
int avsr_bn_fold(avsr_stream_t stream, const float* weight, const float* bias, const float* running_mean,
                 const float* running_var, float* scale, float* shift, int64_t C, float eps);
/* y = act(x*scale + shift (+ residual)); x may be fp32 while y / residual are bf16.  res_scale / res_shift
 * (both or neither; need a residual): the residual is a raw convolution output with a BN of its own and no
 * activation - the downsample branch of torchvision's Bottleneck - applied here: + residual*res_scale + res_shift. */
int avsr_bn_apply(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, const void* residual, void* y,
                  const float* scale, const float* shift, const float* res_scale, const float* res_shift,
                  int64_t rows, int64_t C, int relu);

/* ------------------------------------------------------------------------
 * Head losses on fp32 logits [B*T, V] (train.py:95-128,137): row
 * log-sum-exp, CTC(blank, mean, zero_infinity) on log_softmax(logits) and
 * label-smoothed cross entropy over the first min(T, target_width) frames.
 * targets: int64 [B, ldt]; target_width = targets.shape[1].
 * out4 = {ctc, ce, number of CE targets kept, ctc + ce (the loss of train.py:165)}.  The workspace (size from
 * avsr_head_loss_workspace_floats) carries lse / alpha / beta to the backward,
 * which writes d(w_ctc*up_ctc[0]*ctc + w_ce*up_ce[0]*ce)/d(logits) (a NULL
 * device scalar counts as 1) into rows of pitch lddx >= V and sets the columns [V, lddx) of every row to zero
 * (the padded contraction length of the input-gradient product).
 * row_part / n_parts (optional, NULL / 0 = off): the per-tile log-sum-exp partials the product that made the logits
 * wrote (avsr_gemm_t.row_part, n_parts = ceil(V / 128)); the forward then merges them instead of reading the
 * logits row by row (it still gathers the target columns the two losses need).
 * ---------------------------------------------------------------------- */
int64_t avsr_head_loss_workspace_floats(int64_t B, int64_t T, int64_t target_width);
int avsr_head_loss_fwd(avsr_stream_t stream, const float* logits, int64_t ld, int64_t B, int64_t T, int64_t V,
                       const int64_t* targets, int64_t ldt, int64_t target_width, const int64_t* input_lengths,
                       const int64_t* target_lengths, float label_smoothing, int64_t ignore_index,
                       int64_t blank, float* ws, float* out4, const float* row_part, int64_t n_parts);
int avsr_head_loss_bwd(avsr_stream_t stream, int grad_dtype, const float* logits, int64_t ld, int64_t B,
                       int64_t T, int64_t V, const int64_t* targets, int64_t ldt, int64_t target_width,
                       const int64_t* input_lengths, const int64_t* target_lengths, float label_smoothing,
                       int64_t ignore_index, int64_t blank, const float* ws, float w_ctc, float w_ce,
                       const float* up_ctc, const float* up_ce, void* dlogits, int64_t lddx);
/* torch.argmax(logits, -1) (train.py:137): first maximum wins. */
int avsr_argmax_rows(avsr_stream_t stream, const float* x, int64_t ld, int64_t rows, int64_t V, int64_t* out);

/* ------------------------------------------------------------------------
 * Element-wise helpers around the GEMMs.
 * ---------------------------------------------------------------------- */
int avsr_cast(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, void* y, int64_t n, float scale);
/* the same for [rows, cols] operands with row strides ldx / ldy >= cols (cols a multiple of 4): a cast into or out of
 * a padded buffer, e.g. the mel input with one zero frame before and after each clip (the padding=1 of the Whisper
 * stem convolutions, hf modeling_whisper.py:570-571, made part of the buffer) */
int avsr_cast_rows(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx, void* y, int64_t ldy,
                   int64_t rows, int64_t cols, float scale);
/* y[b][c][r] = x[b][r][c] * scale; copy (optional, batch 1): copy[r*ldc + c] = x[r][c] * scale in out_dtype */
int avsr_transpose(avsr_stream_t stream, int in_dtype, int out_dtype, const void* x, int64_t ldx, int64_t sx,
                   void* y, int64_t ldy, int64_t sy, int64_t batch, int64_t R, int64_t C, float scale,
                   void* copy, int64_t ldc);
/* Weight and bias gradient of an nn.Linear in one pass (autograd of F.linear as models/utils.py:42 and
 * models/gate_cross_attention.py:16,23-26 call it): C[N,K] (+)= A^T B and colsum[n] (+)= sum_m A[m][n], with
 * A = dY [M,N] and B = X [M,K] bf16, row-major as the forward and backward passes leave them (no transposed
 * copies); C, colsum fp32.  lda, ldb multiples of 8 elements, A and B 16-byte aligned; colsum may be NULL. */
int avsr_gemm_tn(avsr_stream_t stream, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, int accumulate, float* colsum, int accumulate_colsum);
/* out[n] (+)= sum_m x[m][n]  (bias gradients) */
int avsr_colsum(avsr_stream_t stream, int dtype, const void* x, int64_t ld, int64_t M, int64_t N, float* out,
                int accumulate);
/* exact (erf) GELU, models/gate_cross_attention.py:24 */
int avsr_gelu_fwd(avsr_stream_t stream, int dtype, const void* x, void* y, int64_t n);
int avsr_gelu_bwd(avsr_stream_t stream, int dtype, const void* x, const void* dy, void* dx, int64_t n);
/* out = x + tanh(*gate) * dropout(y)  (models/gate_cross_attention.py:26,43,47);
 * keep decision of element i = generator(seed, i).  Backward: dy =
 * tanh(*gate)*dropout(dout), *dgate += (1-tanh^2) * sum(dout*dropout(y)). */
int avsr_gated_residual_fwd(avsr_stream_t stream, int dtype, const void* x, const void* y, const float* gate,
                            void* out, int64_t n, float p_drop, uint64_t seed);
int avsr_gated_residual_bwd(avsr_stream_t stream, int dtype, const void* dout, const void* y, const float* gate,
                            void* dy, float* dgate, int64_t n, float p_drop, uint64_t seed);
/* out[i] = sum_s parts[s*part_stride + i] (+ bias[i % ncol]): combines the fp32 partial products of a
 * GEMM whose K dimension was split over the batch index (few tiles, very long K: the head's dX) */
int avsr_sum_parts(avsr_stream_t stream, int out_dtype, const float* parts, int64_t part_stride, int64_t S,
                   const float* bias, int64_t ncol, void* out, int64_t n);
/* AVNet.make_padding_mask after the clamp of models/av_net.py:96,111-114 in one launch: lengths[b] =
 * min(lengths[b], T) in place, mask[b, t] = (t >= lengths[b]) as bytes (1 = padded key). */
int avsr_length_mask(avsr_stream_t stream, int64_t* lengths, int64_t B, int64_t T, uint8_t* mask);
/* out = a + b + c (c may be NULL)  (models/av_net.py:106) */
int avsr_add3(avsr_stream_t stream, int dtype, const void* a, const void* b, const void* c, void* out, int64_t n);
/* out[i] = x[i] + table[i % period]: PositionalEncoding.forward (models/utils.py:21-28) called on its own, x + pe[:, :T]
 * for x [B,T,d] with table = the fp32 [T,d] slice of the sinusoid buffer and period = T*d */
int avsr_add_table(avsr_stream_t stream, int dtype, const void* x, const float* table, void* out, int64_t n,
                   int64_t period);
/* Whisper decoder input (hf modeling_whisper.py:649-797: embed_tokens(ids) * embed_scale +
 * embed_positions.weight[past : past + T]): out[row, :] = table[ids[row], :] * scale + pos[past + row % T, :],
 * fp32 [rows, D] with rows = batch * T.  table: [V, D] fp32 or bf16; pos: fp32 [n_pos, D]; past + T <= n_pos.
 * An id outside [0, V) contributes a zero embedding and sets *bad_id_flag (device int, may be NULL). */
int avsr_embed_tokens(avsr_stream_t stream, int table_dtype, const int64_t* ids, const void* table, const float* pos,
                      float* out, int64_t rows, int64_t T, int64_t past, int64_t D, int64_t V, int64_t n_pos,
                      float scale, int* bad_id_flag);

/* ------------------------------------------------------------------------
 * Optimizer step of train.py:218-225,321: global-norm clipping and AdamW over
 * flat fp32 buffers.  *out += sum(x^2);  norm = pre_scale*sqrt(sumsq);
 * *scale = pre_scale * min(1, max_norm/(norm+1e-6)): clipping of pre_scale * gradient
 * (pre_scale = 1/world folds DDP's mean over ranks, train.py:319, into the step).
 * clear_grad != 0: the step leaves grad zeroed (optimizer.zero_grad of the next step).
 * ---------------------------------------------------------------------- */
int avsr_sumsq_accumulate(avsr_stream_t stream, const float* x, int64_t n, float* out);
int avsr_clip_scale(avsr_stream_t stream, const float* sumsq, float max_norm, float pre_scale, float* scale,
                    float* norm_out);
int avsr_adamw_step(avsr_stream_t stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                    void* bf16_shadow, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, const float* grad_scale, int clear_grad);
/* The per-step scalars of a training step that is replayed from a HIP graph (train.py:218-246: the OneCycleLR
 * learning rate and AdamW's step count change every step, and so must every dropout mask): one launch, outside the
 * graph, writes them into a 32-byte device record the recorded kernels read -
 *   +0  uint64  seed_delta   (what avsr_gemm_t.seed_delta / avsr_xattn_block_t.seed_delta point at)
 *   +8  float   lr, +12 float 1 - beta1^step, +16 float 1 - beta2^step   (what avsr_adamw_step_dev reads).
 * The values travel as kernel arguments, so the host may run any number of steps ahead of the GPU.
 * avsr_adamw_step_dev is avsr_adamw_step with lr and the two bias corrections taken from `hyper` = record + 8;
 * clip_scale_reset != NULL: the word it points at (the sum of squares the clip factor was computed from) is set
 * to zero afterwards, ready for the next step's avsr_sumsq_accumulate. */
int avsr_step_params(avsr_stream_t stream, void* record, uint64_t seed_delta, float lr, float beta1, float beta2,
                     int64_t step);
int avsr_adamw_step_dev(avsr_stream_t stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                        void* bf16_shadow, int64_t n, const float* hyper, float beta1, float beta2, float eps,
                        float weight_decay, const float* grad_scale, int clear_grad, float* clip_scale_reset);

#ifdef __cplusplus
}
#endif
#endif /* AVSR_HIP_H */
