// One Flamingo-style tanh-gated cross-attention block (models/gate_cross_attention.py:28-50) as ONE entry
// point per direction: avsr_gated_xattn_fwd / avsr_gated_xattn_bwd queue the block's whole kernel chain
// on the caller's stream from a single host call, with every element-wise step folded into a producing
// kernel's epilogue:
//
//   forward   h1 = LN1(x);  q = h1 Wq^T + bq                                        [layernorm, GEMM]
//             [k | v] = xa Wkv^T + bkv                                              [GEMM]
//             softmax(QK^T + key mask) V with dropout, per head                     [flash attention]
//             x1 = x + tanh(g_a) * (ao Wo^T + b)      (gate + residual in the GEMM epilogue, the pre-gate
//                                                      value stored beside it for the gate gradient)
//             LN2 -> u = LN2(x1) W1^T + b, f = gelu(u) (both stored by one epilogue)
//             out = x1 + tanh(g_f) * dropout(f W2^T + b) (dropout + gate + residual in the epilogue)
//   backward  the chain in reverse; gelu' is applied in the epilogue of the dX product, the residual
//             gradients are added inside the LayerNorm backward kernels, parameter gradients are
//             accumulated straight into the caller's fp32 buffers, and the gradient of xa is
//             accumulated in place (it receives a contribution from every block of the fusion).
//
// Why a chain of launches and not one persistent kernel: at d = 512 a block is 6.9 MFLOP and 5 MB of bf16
// weights per query row.  A workgroup that keeps a 64-row tile on chip through the whole block has to pull
// those 5 MB through its CU's L2 port for 442 MFLOP of matrix work - 64 B/clk at full MFMA rate against the
// ~56 B/clk a CU gets from L2 - and B*T = 600 ... 3200 rows give 10 ... 50 such workgroups on 256 CUs.
// Splitting every product over its N dimension as separate launches is what fills the chip at these sizes.
#include "common.h"

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t mean1, rstd1, mean2, rstd2, lse, h1, q, kv, ao, y1, x1, h2, u, f, y2, total;
};

// Activations the forward leaves behind for the backward (also its scratch), in one buffer.
Layout fwd_layout(const avsr_xattn_block_t* b) {
    const size_t es = b->dtype == AVSR_BF16 ? 2 : 4;
    const size_t M = (size_t)(b->B * b->Tq), Mk = (size_t)(b->B * b->Tk), d = (size_t)b->d, dff = (size_t)b->d_ff;
    Layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
    L.mean1 = take(M * 4); L.rstd1 = take(M * 4); L.mean2 = take(M * 4); L.rstd2 = take(M * 4);
    L.lse = take((size_t)(b->B * b->H * b->Tq) * 4);
    L.h1 = take(M * d * es); L.q = take(M * d * es); L.kv = take(Mk * 2 * d * es); L.ao = take(M * d * es);
    L.y1 = take(M * d * es); L.x1 = take(M * d * es); L.h2 = take(M * d * es);
    L.u = take(M * dff * es); L.f = take(M * dff * es); L.y2 = take(M * d * es);
    L.total = o;
    return L;
}

struct BwdLayout {
    size_t dy, dh, dx1, dao, dq, dkv, du, dws, tA, tB, total;
};

BwdLayout bwd_layout(const avsr_xattn_block_t* b) {
    const size_t es = b->dtype == AVSR_BF16 ? 2 : 4;
    const size_t M = (size_t)(b->B * b->Tq), Mk = (size_t)(b->B * b->Tk), d = (size_t)b->d, dff = (size_t)b->d_ff;
    const size_t Mp = (M + 7) / 8 * 8, Mkp = (Mk + 7) / 8 * 8, Rp = Mp > Mkp ? Mp : Mkp;
    BwdLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
    L.dy = take(M * d * es); L.dh = take(M * d * es); L.dx1 = take(M * d * es); L.dao = take(M * d * es);
    L.dq = take(M * d * es); L.dkv = take(Mk * 2 * d * es); L.du = take(M * dff * es);
    L.dws = take((size_t)(b->B * b->H * b->Tq) * 4);
    // transposed operands of the dW products: the widest one is dff rows (fc1 / fc2) or 2*d rows (the K|V projection)
    const size_t tw = dff > 2 * d ? dff : 2 * d;
    L.tA = take(tw * Rp * es); L.tB = take(tw * Rp * es);
    L.total = o;
    return L;
}

int check_block(const avsr_xattn_block_t* b) {
    if (!b) return AVSR_EINVAL;
    if (b->dtype != AVSR_F32 && b->dtype != AVSR_BF16) return AVSR_EINVAL;
    if (b->B <= 0 || b->Tq <= 0 || b->Tk <= 0 || b->H <= 0) return AVSR_EINVAL;
    if (b->d != b->H * 64 || b->d_ff <= 0 || b->d_ff % 8 || b->d % 8) return AVSR_EUNSUPPORTED;   // head_dim 64 kernels
    if (b->p_drop < 0.f || b->p_drop >= 1.f) return AVSR_EINVAL;
    if (!b->w_in || !b->b_in || !b->w_out || !b->b_out || !b->w_fc1 || !b->b_fc1 || !b->w_fc2 || !b->b_fc2) return AVSR_EINVAL;
    if (!b->ln1_g || !b->ln1_b || !b->ln2_g || !b->ln2_b || !b->gate_attn || !b->gate_ff) return AVSR_EINVAL;
    return AVSR_OK;
}

// C[M,N] = epi(A[M,K] B[N,K]^T) through avsr_gemm_nt
struct G {
    avsr_gemm_t g;
    G(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
      int out_dtype = -1) {
        g = avsr_gemm_t{};
        g.in_dtype = dtype; g.out_dtype = out_dtype < 0 ? dtype : out_dtype;
        g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
        g.batch = 1; g.alpha = 1.0f; g.act = AVSR_ACT_NONE;
    }
    int run(avsr_stream_t s) { return avsr_gemm_nt(s, &g); }
};

#define TRY(expr) do { const int rc_ = (expr); if (rc_ != AVSR_OK) return rc_; } while (0)

// dW (+)= dY^T X and db += column sums of dY, with dY [M,N] and X [M,K]: both operands are transposed into
// scratch ([N,Mp] and [K,Mp], Mp = M rounded up to 8, padding zeroed) so the product is the NT form again.
int param_grads(avsr_stream_t s, int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, int64_t M, int64_t N,
                int64_t K, float* dw, int64_t lddw, float* db, char* tA, char* tB) {
    // bf16: one pass over dY and X as they lie in memory (contraction over their rows, bias gradient included)
    if (dtype == AVSR_BF16 && lddy % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)dy | (uintptr_t)x) % 16 == 0)
        return avsr_gemm_tn(s, dy, lddy, x, ldx, dw, lddw, M, N, K, 1, db, 1);
    const size_t es = dtype == AVSR_BF16 ? 2 : 4;
    const int64_t Mp = (M + 7) / 8 * 8;
    if (Mp != M) {
        if (hipMemsetAsync(tA, 0, (size_t)(N * Mp) * es, (hipStream_t)s) != hipSuccess) return AVSR_EHIP;
        if (hipMemsetAsync(tB, 0, (size_t)(K * Mp) * es, (hipStream_t)s) != hipSuccess) return AVSR_EHIP;
    }
    TRY(avsr_transpose(s, dtype, dtype, dy, lddy, 0, tA, Mp, 0, 1, M, N, 1.0f, nullptr, 0));
    TRY(avsr_transpose(s, dtype, dtype, x, ldx, 0, tB, Mp, 0, 1, M, K, 1.0f, nullptr, 0));
    G g(dtype, N, K, Mp, tA, Mp, tB, Mp, dw, lddw, AVSR_F32);
    g.g.residual = dw; g.g.ldr = lddw;                       // accumulate in place
    TRY(g.run(s));
    if (db) TRY(avsr_colsum(s, dtype, dy, lddy, M, N, db, 1));
    return AVSR_OK;
}

}  // namespace

extern "C" int64_t avsr_gated_xattn_fwd_workspace_bytes(const avsr_xattn_block_t* b) {
    if (check_block(b) != AVSR_OK) return 0;
    return (int64_t)fwd_layout(b).total;
}

extern "C" int64_t avsr_gated_xattn_bwd_workspace_bytes(const avsr_xattn_block_t* b) {
    if (check_block(b) != AVSR_OK) return 0;
    return (int64_t)bwd_layout(b).total;
}

extern "C" int avsr_gated_xattn_fwd(avsr_stream_t s, const avsr_xattn_block_t* b, const void* x, const void* xa,
                                    const uint8_t* key_mask, void* out, void* saved) {
    TRY(check_block(b));
    if (!x || !out || !saved) return AVSR_EINVAL;
    const int dt = b->dtype;
    const size_t es = dt == AVSR_BF16 ? 2 : 4;
    const int64_t M = b->B * b->Tq, Mk = b->B * b->Tk, d = b->d, dff = b->d_ff;
    const Layout L = fwd_layout(b);
    char* w = (char*)saved;
    const void* x1 = x;
    if (xa) {
        // ---- attention half ------------------------------------------------------------------------------
        TRY(avsr_layernorm_fwd(s, dt, dt, x, d, w + L.h1, d, b->ln1_g, b->ln1_b, nullptr, 0, (float*)(w + L.mean1),
                               (float*)(w + L.rstd1), M, d, 1e-5f));
        {   // q = h1 Wq^T + bq   (in_proj rows [0, d))
            G g(dt, M, d, d, w + L.h1, d, b->w_in, d, w + L.q, d);
            g.g.bias = b->b_in;
            TRY(g.run(s));
        }
        {   // [k | v] = xa Wkv^T + bkv   (in_proj rows [d, 3d): keys and values come from the un-normalised xa)
            G g(dt, Mk, 2 * d, d, xa, d, (const char*)b->w_in + (size_t)(d * d) * es, d, w + L.kv, 2 * d);
            g.g.bias = b->b_in + d;
            TRY(g.run(s));
        }
        TRY(avsr_attn_fwd_sd(s, dt, dt == AVSR_BF16 ? 1 : 0, w + L.q, d, w + L.kv, 2 * d, w + L.kv + (size_t)d * es, 2 * d,
                             w + L.ao, d, key_mask, (float*)(w + L.lse), b->B, b->H, b->Tq, b->Tk, 64, 0.125f, b->p_drop,
                             b->seed_attn, b->seed_delta, 0, nullptr, 0));
        {   // x1 = x + tanh(g_a) * y1,  y1 = ao Wo^T + bo
            G g(dt, M, d, d, w + L.ao, d, b->w_out, d, w + L.x1, d);
            g.g.bias = b->b_out; g.g.residual = x; g.g.ldr = d; g.g.gate = b->gate_attn;
            g.g.pre = w + L.y1; g.g.ldp = d;
            TRY(g.run(s));
        }
        x1 = w + L.x1;
    }
    // ---- feed-forward half -----------------------------------------------------------------------------------
    TRY(avsr_layernorm_fwd(s, dt, dt, x1, d, w + L.h2, d, b->ln2_g, b->ln2_b, nullptr, 0, (float*)(w + L.mean2),
                           (float*)(w + L.rstd2), M, d, 1e-5f));
    {   // u = h2 W1^T + b1 (kept for gelu'), f = gelu(u)
        G g(dt, M, dff, d, w + L.h2, d, b->w_fc1, d, w + L.f, dff);
        g.g.bias = b->b_fc1; g.g.act = AVSR_ACT_GELU; g.g.pre = w + L.u; g.g.ldp = dff;
        TRY(g.run(s));
    }
    {   // out = x1 + tanh(g_f) * dropout(y2),  y2 = f W2^T + b2
        G g(dt, M, d, dff, w + L.f, dff, b->w_fc2, dff, out, d);
        g.g.bias = b->b_fc2; g.g.residual = x1; g.g.ldr = d; g.g.gate = b->gate_ff;
        g.g.pre = w + L.y2; g.g.ldp = d; g.g.p_drop = b->p_drop; g.g.seed = b->seed_ff;
        g.g.seed_delta = b->seed_delta;
        TRY(g.run(s));
    }
    return AVSR_OK;
}

extern "C" int avsr_gated_xattn_bwd(avsr_stream_t s, const avsr_xattn_block_t* b, const void* x, const void* xa,
                                    const uint8_t* key_mask, const void* saved, const void* dout, void* dx, void* dxa,
                                    int accumulate_dxa, const avsr_xattn_grads_t* gr, void* workspace) {
    TRY(check_block(b));
    if (!x || !saved || !dout || !dx || !gr || !workspace) return AVSR_EINVAL;
    if (xa && !dxa) return AVSR_EINVAL;
    if (!b->w_inT || !b->w_outT || !b->w_fc1T || !b->w_fc2T) return AVSR_EINVAL;
    if (!gr->dw_in || !gr->db_in || !gr->dw_out || !gr->db_out || !gr->dw_fc1 || !gr->db_fc1 || !gr->dw_fc2 ||
        !gr->db_fc2 || !gr->dln1_g || !gr->dln1_b || !gr->dln2_g || !gr->dln2_b || !gr->dgate_attn || !gr->dgate_ff)
        return AVSR_EINVAL;
    const int dt = b->dtype;
    const size_t es = dt == AVSR_BF16 ? 2 : 4;
    const int64_t M = b->B * b->Tq, Mk = b->B * b->Tk, d = b->d, dff = b->d_ff;
    const Layout L = fwd_layout(b);
    const BwdLayout W = bwd_layout(b);
    const char* sv = (const char*)saved;
    char* w = (char*)workspace;
    const void* x1 = xa ? (const void*)(sv + L.x1) : x;

    // ---- feed-forward half -----------------------------------------------------------------------------------
    // dy2 = tanh(g_f) * dropout(dout);  dg_f += (1 - tanh^2) * sum(dout * dropout(y2))
    TRY(avsr_gated_residual_bwd_sd(s, dt, dout, sv + L.y2, b->gate_ff, w + W.dy, gr->dgate_ff, M * d, b->p_drop, b->seed_ff,
                                   b->seed_delta));
    {   // du = (dy2 W2) * gelu'(u)
        G g(dt, M, dff, d, w + W.dy, d, b->w_fc2T, b->ld_fc2T, w + W.du, dff);
        g.g.mul_gelu_grad_of = sv + L.u; g.g.ld_aux = dff;
        TRY(g.run(s));
    }
    TRY(param_grads(s, dt, w + W.dy, d, sv + L.f, dff, M, d, dff, gr->dw_fc2, dff, gr->db_fc2, w + W.tA, w + W.tB));
    {   // dh2 = du W1
        G g(dt, M, d, dff, w + W.du, dff, b->w_fc1T, b->ld_fc1T, w + W.dh, d);
        TRY(g.run(s));
    }
    TRY(param_grads(s, dt, w + W.du, dff, sv + L.h2, d, M, dff, d, gr->dw_fc1, d, gr->db_fc1, w + W.tA, w + W.tB));
    // dx1 = dout + LN2'(dh2)
    void* dx1 = xa ? (void*)(w + W.dx1) : dx;
    TRY(avsr_layernorm_bwd_res(s, dt, dt, w + W.dh, d, x1, d, b->ln2_g, (const float*)(sv + L.mean2),
                               (const float*)(sv + L.rstd2), dx1, d, gr->dln2_g, gr->dln2_b, M, d, dout, d));
    if (!xa) return AVSR_OK;

    // ---- attention half --------------------------------------------------------------------------------------
    TRY(avsr_gated_residual_bwd(s, dt, dx1, sv + L.y1, b->gate_attn, w + W.dy, gr->dgate_attn, M * d, 0.f, 0));
    {   // dao = dy1 Wo
        G g(dt, M, d, d, w + W.dy, d, b->w_outT, b->ld_outT, w + W.dao, d);
        TRY(g.run(s));
    }
    TRY(param_grads(s, dt, w + W.dy, d, sv + L.ao, d, M, d, d, gr->dw_out, d, gr->db_out, w + W.tA, w + W.tB));
    TRY(avsr_attn_bwd_sd(s, dt, dt == AVSR_BF16 ? 1 : 0, sv + L.q, d, sv + L.kv, 2 * d, sv + L.kv + (size_t)d * es, 2 * d,
                         sv + L.ao, d, w + W.dao, d, w + W.dq, d, w + W.dkv, 2 * d, w + W.dkv + (size_t)d * es, 2 * d, key_mask,
                         (const float*)(sv + L.lse), (float*)(w + W.dws), b->B, b->H, b->Tq, b->Tk, 64, 0.125f, b->p_drop,
                         b->seed_attn, b->seed_delta));
    {   // dh1 = dq Wq   (W^T columns [0, d))
        G g(dt, M, d, d, w + W.dq, d, b->w_inT, b->ld_inT, w + W.dh, d);
        TRY(g.run(s));
    }
    TRY(param_grads(s, dt, w + W.dq, d, sv + L.h1, d, M, d, d, gr->dw_in, d, gr->db_in, w + W.tA, w + W.tB));
    {   // dxa (+)= [dk | dv] Wkv   (W^T columns [d, 3d))
        G g(dt, Mk, d, 2 * d, w + W.dkv, 2 * d, (const char*)b->w_inT + (size_t)d * es, b->ld_inT, dxa, d);
        if (accumulate_dxa) { g.g.residual = dxa; g.g.ldr = d; }
        TRY(g.run(s));
    }
    TRY(param_grads(s, dt, w + W.dkv, 2 * d, xa, d, Mk, 2 * d, d, gr->dw_in + d * d, d, gr->db_in + d, w + W.tA, w + W.tB));
    // dx = dx1 + LN1'(dh1)
    TRY(avsr_layernorm_bwd_res(s, dt, dt, w + W.dh, d, x, d, b->ln1_g, (const float*)(sv + L.mean1),
                               (const float*)(sv + L.rstd1), dx, d, gr->dln1_g, gr->dln1_b, M, d, dx1, d));
    return AVSR_OK;
}
