// Stage-level composites of the ViT path: every function only validates, then enqueues the kernels of
// one stage on the caller's stream (no allocation, no synchronisation, graph-capturable).
#include "../../include/signal_hip.h"
#include "sig_common.h"
#include "sig_kernels.h"

#define RUN(expr)              \
    do {                       \
        int rc__ = (expr);     \
        if (rc__) return rc__; \
    } while (0)

static inline int pad128(int m) { return (m + 127) & ~127; }

static int check_dims(const SigVitDims* d, const char* who) {
    SIG_CHECK_ARG(d, "%s: dims missing", who);
    SIG_CHECK_ARG(d->S > 0 && d->B > 0 && d->S % d->B == 0 && d->L > 1 && d->L <= 144, "%s: bad S/B/L (%d,%d,%d)", who, d->S, d->B, d->L);
    SIG_CHECK_ARG(d->D == d->H * 64 && (d->D & 127) == 0, "%s: width %d must be heads*64 and a multiple of 128", who, d->D);
    SIG_CHECK_ARG(d->F > 0 && (d->F & 127) == 0 && d->out_dim > 0 && (d->out_dim & 127) == 0,
                  "%s: F=%d / out_dim=%d must be multiples of 128", who, d->F, d->out_dim);
    SIG_CHECK_DT(d->dtype, who);
    return 0;
}

static SigGemmNT nt(int dt, const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K, void* out, int ldo,
                    const float* bias = nullptr, const float* res = nullptr, int ldr = 0, void* aux = nullptr, int ldaux = 0) {
    SigGemmNT p;
    p.A = A; p.lda = lda; p.Bt = Bt; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo;
    p.bias = bias; p.res = res; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux; p.band = 0; p.colsum = nullptr; p.dt = dt;
    return p;
}
static SigGemmNT with_colsum(SigGemmNT p, float* colsum) {
    p.colsum = colsum;
    return p;
}
static SigGemmTN tn(int dt, const bf16_t* P, int ldp, const bf16_t* Q, int ldq, int Mr, int I, int J, float* out, int ldo) {
    SigGemmTN p;
    p.P = P; p.ldp = ldp; p.Q = Q; p.ldq = ldq; p.Mr = Mr; p.I = I; p.J = J; p.out = out; p.ldo = ldo;
    p.split = 0; p.m_chunk = 0; p.ws = nullptr; p.dt = dt;
    return p;
}

extern "C" {

// ------------------------------------------------------------------------------------------------
// embed: im2col -> conv1 as GEMM -> +CLS/camera/positional -> ln_pre
// ------------------------------------------------------------------------------------------------
int sig_embed_fwd(const SigVitDims* d, const SigEmbedParams* p, const SigEmbedActs* a, const float* const* img_parts, int n_parts,
                  const int64_t* cam_label, int img_h, int img_w, int patch, void* stream) {
    RUN(check_dims(d, "embed_fwd"));
    SIG_CHECK_ARG(p && a && img_parts && n_parts > 0 && d->S % n_parts == 0, "embed_fwd: null argument / image parts must divide S");
    for (int i = 0; i < n_parts; ++i) SIG_CHECK_ARG(img_parts[i], "embed_fwd: image part %d is null", i);
    SIG_CHECK_ARG(patch > 0 && (img_h / patch) * (img_w / patch) == d->L - 1, "embed_fwd: %dx%d / %d does not give %d patches",
                  img_h, img_w, patch, d->L - 1);
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int K = 3 * patch * patch, Mt = d->S * (d->L - 1);
    SIG_CHECK_ARG((K & 63) == 0, "embed_fwd: 3*P*P must be a multiple of 64");
    // one gather per modality straight from the caller's tensors (the reference's three backbone calls, make_model.py:181-183,
    // become row ranges of ONE patch matrix): no staging copy of the 75 MB of f32 images
    const int per = d->S / n_parts;
    for (int i = 0; i < n_parts; ++i)
        RUN(sig_launch_im2col(img_parts[i], a->patches + (size_t)i * per * (d->L - 1) * K, per, img_h, img_w, patch, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->patches, K, p->w_conv, K, Mt, d->D, K, a->tok, d->D), SIG_EPI_F32, st));
    RUN(sig_launch_embed_assemble(a->tok, p->class_embedding, p->positional_embedding, p->cv_embed, cam_label, p->sie_coe,
                                  p->ln_w, p->ln_b, a->x0, a->pre_ln, a->mean, a->rstd, d->S, d->B, d->L, d->D, 1e-5f, st));
    return 0;
}

int sig_embed_bwd(const SigVitDims* d, const SigEmbedParams* p, const SigEmbedActs* a, const SigEmbedGrads* g,
                  const float* dx0, float* scratch_dpre, uint16_t* scratch_dtok, const int64_t* cam_label, int patch,
                  void* stream) {
    RUN(check_dims(d, "embed_bwd"));
    SIG_CHECK_ARG(p && a && g && dx0 && scratch_dpre && scratch_dtok, "embed_bwd: null argument");
    SIG_CHECK_ARG(a->pre_ln && a->mean && a->rstd, "embed_bwd: forward ran without saving pre_ln / statistics");
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int M = d->S * d->L, Mt = d->S * (d->L - 1), K = 3 * patch * patch;
    SIG_CHECK_ARG((K & 127) == 0, "embed_bwd: 3*P*P must be a multiple of 128");
    RUN(sig_launch_layernorm_bwd(dx0, 0, a->pre_ln, p->ln_w, a->mean, a->rstd, nullptr, scratch_dpre, nullptr, g->ln_w,
                                 g->ln_b, M, d->D, dt, st));
    RUN(sig_launch_embed_bwd(scratch_dpre, nullptr, scratch_dtok, g->class_embedding, g->positional_embedding,
                             p->cv_embed ? g->cv_embed : nullptr, cam_label, p->sie_coe, d->S, d->B, d->L, d->D, dt, st));
    // conv1.weight gradient: dW[D, 3PP] += dtok^T patches   (pad rows of both operands are zero)
    RUN(sig_launch_gemm_tn(tn(dt, scratch_dtok, d->D, a->patches, K, pad128(Mt), d->D, K, g->w_conv, K), st));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// one residual attention block
// ------------------------------------------------------------------------------------------------
int sig_block_fwd(const SigVitDims* d, const SigBlockParams* p, const SigBlockActs* a, void* stream) {
    RUN(check_dims(d, "block_fwd"));
    SIG_CHECK_ARG(p && a, "block_fwd: null argument");
    SIG_CHECK_ARG(a->x_in && a->h1 && a->qkv && a->attn && a->x_mid && a->h2 && a->g && a->x_out, "block_fwd: activation buffer missing");
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int M = d->S * d->L, D = d->D, F = d->F;
    RUN(sig_launch_layernorm_fwd(a->x_in, p->ln1_w, p->ln1_b, a->h1, nullptr, a->mean1, a->rstd1, M, D, 1e-5f, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->h1, D, p->w_in, D, M, 3 * D, D, a->qkv, 3 * D, p->b_in), SIG_EPI_BIAS_BF16, st));
    RUN(sig_launch_attn_fwd(a->qkv, a->attn, a->lse, d->S, d->L, d->H, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->attn, D, p->w_out, D, M, D, D, a->x_mid, D, p->b_out, a->x_in, D), SIG_EPI_BIAS_RES_F32, st));
    RUN(sig_launch_layernorm_fwd(a->x_mid, p->ln2_w, p->ln2_b, a->h2, nullptr, a->mean2, a->rstd2, M, D, 1e-5f, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->h2, D, p->w_fc, D, M, F, D, a->g, F, p->b_fc, nullptr, 0, a->u, F), SIG_EPI_BIAS_GELU_BF16, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->g, F, p->w_proj, F, M, D, F, a->x_out, D, p->b_proj, a->x_mid, D), SIG_EPI_BIAS_RES_F32, st));
    return 0;
}

int sig_block_bwd(const SigVitDims* d, const SigBlockParams* p, const SigBlockActs* a, const SigBlockGrads* g,
                  const SigBlockScratch* s, const float* dx_out, const uint16_t* dx_out_b, float* dx_in, uint16_t* dx_in_b,
                  float* dx_in_colsum, int b_proj_done, void* stream) {
    RUN(check_dims(d, "block_bwd"));
    SIG_CHECK_ARG(p && a && g && s && dx_out && dx_out_b && dx_in && dx_in_b, "block_bwd: null argument");
    SIG_CHECK_ARG(p->wt_in && p->wt_out && p->wt_fc && p->wt_proj, "block_bwd: transposed weights missing");
    SIG_CHECK_ARG(a->u && a->lse && a->mean1 && a->rstd1 && a->mean2 && a->rstd2, "block_bwd: forward ran without saving for backward");
    SIG_CHECK_ARG(s->du && s->dh && s->dqkv && s->dx_mid && s->dx_mid_b, "block_bwd: scratch missing");
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int M = d->S * d->L, Mp = pad128(M), D = d->D, F = d->F;
    // Bias gradients are by-products of kernels that already hold the data: c_fc bias from the dGELU epilogue, out_proj
    // bias from LN2's backward (column sums of dx_mid), and c_proj bias =
    // column sums of dx_out, which the stage ABOVE accumulates when it writes dx_out (b_proj_done) -- else a pass here.
    // ---- MLP ----
    RUN(sig_launch_gemm_nt(with_colsum(nt(dt, dx_out_b, D, p->wt_proj, D, M, F, D, s->du, F, nullptr, nullptr, 0, a->u, F), g->b_fc),
                           SIG_EPI_DGELU_BF16, st));                                              // du = (dx_out W_proj) * QuickGELU'(pre-act), saved by c_fc
    if (!b_proj_done) RUN(sig_launch_colsum_bf16(dx_out_b, D, M, D, g->b_proj, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, s->du, F, p->wt_fc, F, M, D, F, s->dh, D), SIG_EPI_BF16, st));     // dh2 = du W_fc
    // dx_mid = dx_out + LN2'(dh2)
    RUN(sig_launch_layernorm_bwd(s->dh, 1, a->x_mid, p->ln2_w, a->mean2, a->rstd2, dx_out, s->dx_mid, s->dx_mid_b, g->ln2_w,
                                 g->ln2_b, M, D, dt, st, g->b_out));
    // ---- attention ----
    RUN(sig_launch_gemm_nt(nt(dt, s->dx_mid_b, D, p->wt_out, D, M, D, D, s->dh, D), SIG_EPI_BF16, st));  // d attn
    // (in_proj bias = column sums of dqkv: by-product of the grouped weight-gradient launch below, which streams every dqkv
    //  tile through its MFMA fragments anyway; inside the attention backward it cost 48 registers at the 256 cap)
    RUN(sig_launch_attn_bwd(a->qkv, a->attn, s->dh, a->lse, s->dqkv, d->S, d->L, d->H, dt, st));
    // ---- the block's four weight gradients: every (dY, X) pair is still in place here (dx_out_b is overwritten by the last
    // LayerNorm backward below), so they go as ONE stream-K launch + one reduce (gemm_tn_grouped.hip) ----
    {
        const SigTnJob jobs[4] = {
            // (in_proj: its bias gradient = column sums of dqkv over the M VALID rows -- a caller's reused scratch may leave stale pad rows,
            //  which the GEMM is protected from by h1's zero pad rows but a column sum is not)
            {s->dqkv, a->h1, g->w_in, 3 * D, D, D, 3 * D, D, g->b_in, M}, // attn.in_proj_weight [3D, D] + in_proj_bias
            {s->dx_mid_b, a->attn, g->w_out, D, D, D, D, D, nullptr, 0},  // attn.out_proj.weight [D, D]
            {s->du, a->h2, g->w_fc, F, D, D, F, D, nullptr, 0},           // mlp.c_fc.weight      [F, D]
            {dx_out_b, a->g, g->w_proj, D, F, F, D, F, nullptr, 0},       // mlp.c_proj.weight    [D, F]
        };
        RUN(sig_launch_gemm_tn_grouped(jobs, 4, Mp, dt, st));
    }
    RUN(sig_launch_gemm_nt(nt(dt, s->dqkv, 3 * D, p->wt_in, 3 * D, M, D, 3 * D, s->dh, D), SIG_EPI_BF16, st));  // dh1
    // dx_in = dx_mid + LN1'(dh1)
    RUN(sig_launch_layernorm_bwd(s->dh, 1, a->x_in, p->ln1_w, a->mean1, a->rstd1, s->dx_mid, dx_in, dx_in_b, g->ln1_w, g->ln1_b,
                                 M, D, dt, st, dx_in_colsum));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ln_post + projection of all tokens
// ------------------------------------------------------------------------------------------------
int sig_head_fwd(const SigVitDims* d, const SigHeadParams* p, const SigHeadActs* a, void* stream) {
    RUN(check_dims(d, "head_fwd"));
    SIG_CHECK_ARG(p && a && a->x && a->hp && a->tokens && p->proj_t, "head_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int M = d->S * d->L;
    RUN(sig_launch_layernorm_fwd(a->x, p->ln_w, p->ln_b, a->hp, nullptr, a->mean, a->rstd, M, d->D, 1e-5f, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->hp, d->D, p->proj_t, d->D, M, d->out_dim, d->D, a->tokens, d->out_dim), SIG_EPI_F32, st));
    return 0;
}

int sig_head_bwd(const SigVitDims* d, const SigHeadParams* p, const SigHeadActs* a, const SigHeadGrads* g,
                 const float* dtokens, uint16_t* scratch_dtok_b, uint16_t* scratch_dh, float* dx, uint16_t* dx_b,
                 float* dx_colsum, void* stream) {
    RUN(check_dims(d, "head_bwd"));
    SIG_CHECK_ARG(p && a && g && dtokens && scratch_dtok_b && scratch_dh && dx && dx_b && p->proj, "head_bwd: null argument");
    SIG_CHECK_ARG(a->mean && a->rstd, "head_bwd: forward ran without saving statistics");
    hipStream_t st = (hipStream_t)stream;
    const int dt = d->dtype;
    const int M = d->S * d->L, Mp = pad128(M), D = d->D, O = d->out_dim;
    RUN(sig_launch_cast_bf16(dtokens, scratch_dtok_b, (size_t)Mp * O, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, scratch_dtok_b, O, p->proj, O, M, D, O, scratch_dh, D), SIG_EPI_BF16, st));  // d ln_post out
    RUN(sig_launch_gemm_tn(tn(dt, a->hp, D, scratch_dtok_b, O, Mp, D, O, g->proj, O), st));                    // d proj [D,O]
    RUN(sig_launch_layernorm_bwd(scratch_dh, 1, a->x, p->ln_w, a->mean, a->rstd, nullptr, dx, dx_b, g->ln_w, g->ln_b, M, D, dt, st, dx_colsum));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// SIM
// ------------------------------------------------------------------------------------------------
int sig_sim_select(const float* tokens, int B, int L, const SigSimParams* p, const SigSimActs* a, void* stream) {
    SIG_CHECK_ARG(tokens && p && a, "sim_select: null argument");
    return sig_launch_sim_select(tokens, B, L, p->sel_wq, p->sel_bq, p->sel_wk, p->sel_bk, p->topk, p->max_keep, a->qprime, a->cconst,
                                 a->intra, a->inter, a->mask_f, a->mask_u8, (hipStream_t)stream);
}

int sig_sim_fwd(const float* tokens, int B, int L, const SigSimParams* p, const SigSimActs* a, void* stream) {
    RUN(sig_sim_select(tokens, B, L, p, a, stream));
    SIG_CHECK_ARG(a->sel && a->cls_b && a->cls_f && a->qh && a->kv && a->ao && a->y && a->z1 && a->z1_b && a->f1 && a->y2 && a->out,
                  "sim_fwd: activation buffer missing");
    hipStream_t st = (hipStream_t)stream;
    const int dt = p->dtype;
    const int Mq = 3 * B, Mk = 3 * B * (L - 1), d = 512;
    RUN(sig_launch_sim_gather(tokens, a->mask_f, B, L, a->sel, a->cls_b, a->cls_f, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->cls_b, d, p->w_q, d, Mq, d, d, a->qh, d, p->b_q), SIG_EPI_BIAS_F32, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->sel, d, p->w_kv, d, Mk, 2 * d, d, a->kv, 2 * d, p->b_kv), SIG_EPI_BIAS_BF16, st));
    RUN(sig_launch_xattn_fwd(a->qh, a->kv, B, 3 * (L - 1), a->ao, a->probs, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->ao, d, p->w_o, d, Mq, d, d, a->y, d, p->b_o, a->cls_f, d), SIG_EPI_BIAS_RES_F32, st));
    RUN(sig_launch_layernorm_fwd(a->y, p->n1_w, p->n1_b, a->z1_b, a->z1, a->mean1, a->rstd1, Mq, d, 1e-5f, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->z1_b, d, p->w_f1, d, Mq, 2 * d, d, a->f1, 2 * d, p->b_f1, nullptr, 0, a->f1_pre, 2 * d),
                           SIG_EPI_BIAS_GELUERF_BF16, st));
    RUN(sig_launch_gemm_nt(nt(dt, a->f1, 2 * d, p->w_f2, 2 * d, Mq, d, 2 * d, a->y2, d, p->b_f2, a->z1, d), SIG_EPI_BIAS_RES_F32, st));
    RUN(sig_launch_layernorm_fwd(a->y2, p->n2_w, p->n2_b, nullptr, a->out, a->mean2, a->rstd2, Mq, d, 1e-5f, dt, st));
    return 0;
}

int sig_sim_bwd(const float* dout, int B, int L, const SigSimParams* p, const SigSimActs* a, const SigSimGrads* g,
                const SigSimScratch* s, float* dtokens, void* stream) {
    SIG_CHECK_ARG(dout && p && a && g && s && dtokens, "sim_bwd: null argument");
    SIG_CHECK_ARG(p->wt_q && p->wt_kv && p->wt_o && p->wt_f1 && p->wt_f2, "sim_bwd: transposed weights missing");
    SIG_CHECK_ARG(a->probs && a->f1_pre && a->mean1 && a->rstd1 && a->mean2 && a->rstd2, "sim_bwd: forward ran without saving for backward");
    hipStream_t st = (hipStream_t)stream;
    const int dt = p->dtype;
    const int Mq = 3 * B, Mqp = pad128(Mq), Mk = 3 * B * (L - 1), Mkp = pad128(Mk), d = 512;
    // norm2
    RUN(sig_launch_layernorm_bwd(dout, 0, a->y2, p->n2_w, a->mean2, a->rstd2, nullptr, s->dy2, s->dy2_b, g->n2_w, g->n2_b, Mq, d, dt, st));
    // ffn.2 (+ residual into z1)
    RUN(sig_launch_gemm_nt(nt(dt, s->dy2_b, d, p->wt_f2, d, Mq, 2 * d, d, s->df1, 2 * d, nullptr, nullptr, 0, a->f1_pre, 2 * d), SIG_EPI_DGELUERF_BF16, st));
    RUN(sig_launch_gemm_tn(tn(dt, s->dy2_b, d, a->f1, 2 * d, Mqp, d, 2 * d, g->w_f2, 2 * d), st));
    RUN(sig_launch_colsum_f32(s->dy2, d, Mq, d, g->b_f2, st));
    // ffn.0 ; total gradient of z1 = dy2 (residual) + df1 W_f1
    RUN(sig_launch_gemm_nt(nt(dt, s->df1, 2 * d, p->wt_f1, 2 * d, Mq, d, 2 * d, s->dz1, d, nullptr, s->dy2, d), SIG_EPI_RES_F32, st));
    RUN(sig_launch_gemm_tn(tn(dt, s->df1, 2 * d, a->z1_b, d, Mqp, 2 * d, d, g->w_f1, d), st));
    RUN(sig_launch_colsum_bf16(s->df1, 2 * d, Mq, 2 * d, g->b_f1, dt, st));
    // norm1
    RUN(sig_launch_layernorm_bwd(s->dz1, 0, a->y, p->n1_w, a->mean1, a->rstd1, nullptr, s->dy, s->dy_b, g->n1_w, g->n1_b, Mq, d, dt, st));
    // out_proj
    RUN(sig_launch_gemm_nt(nt(dt, s->dy_b, d, p->wt_o, d, Mq, d, d, s->dao, d), SIG_EPI_F32, st));
    RUN(sig_launch_gemm_tn(tn(dt, s->dy_b, d, a->ao, d, Mqp, d, d, g->w_o, d), st));
    RUN(sig_launch_colsum_f32(s->dy, d, Mq, d, g->b_o, st));
    // attention core
    RUN(sig_launch_xattn_bwd(a->qh, a->kv, a->probs, s->dao, B, 3 * (L - 1), s->dqh, s->dkv, dt, st));
    // k|v projection
    RUN(sig_launch_gemm_nt(nt(dt, s->dkv, 2 * d, p->wt_kv, 2 * d, Mk, d, 2 * d, s->dsel, d), SIG_EPI_BF16, st));
    RUN(sig_launch_gemm_tn(tn(dt, s->dkv, 2 * d, a->sel, d, Mkp, 2 * d, d, g->w_kv, d), st));
    RUN(sig_launch_colsum_bf16(s->dkv, 2 * d, Mk, 2 * d, g->b_kv, dt, st));
    // q projection ; total gradient of the stacked CLS = dy (residual) + dqh W_q
    RUN(sig_launch_cast_bf16(s->dqh, s->dqh_b, (size_t)Mqp * d, dt, st));
    RUN(sig_launch_gemm_nt(nt(dt, s->dqh_b, d, p->wt_q, d, Mq, d, d, s->dcls, d, nullptr, s->dy, d), SIG_EPI_RES_F32, st));
    RUN(sig_launch_gemm_tn(tn(dt, s->dqh_b, d, a->cls_b, d, Mqp, d, d, g->w_q, d), st));
    RUN(sig_launch_colsum_f32(s->dqh, d, Mq, d, g->b_q, st));
    // scatter back to the token gradient
    RUN(sig_launch_sim_gather_bwd(s->dsel, s->dcls, a->mask_f, B, L, dtokens, dt, st));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GAM / LAM
// ------------------------------------------------------------------------------------------------
int sig_gam_fwd(const float* tokens, int B, int L, const float* contra_temp, const SigGamActs* a, void* stream) {
    SIG_CHECK_ARG(tokens && contra_temp && a, "gam_fwd: null argument");
    return sig_launch_gam_fwd(tokens, B, L, contra_temp, a->fh, a->nrm, a->lv, a->la, a->vec, a->coef, a->loss, (hipStream_t)stream);
}
int sig_gam_bwd(int B, int L, const SigGamActs* a, const float* dloss, float* dtokens, float* d_contra_temp, void* stream) {
    SIG_CHECK_ARG(a && dloss && dtokens, "gam_bwd: null argument");
    return sig_launch_gam_bwd(a->fh, a->nrm, a->coef, dloss, B, L, dtokens, d_contra_temp, (hipStream_t)stream);
}

static SigLamTailPtrs lam_ptrs(const SigDasParams* p3, const SigDasGrads* g3) {
    SigLamTailPtrs t;
    for (int m = 0; m < 3; ++m) {
        t.wd[m] = p3[m].wd; t.bd[m] = p3[m].bd; t.w4[m] = p3[m].w4;
        t.dwd[m] = g3 ? g3[m].wd : nullptr; t.dbd[m] = g3 ? g3[m].bd : nullptr; t.dw4[m] = g3 ? g3[m].w4 : nullptr;
    }
    return t;
}

int sig_lam_fwd(const float* tokens, int B, int L, int h, int w, int dtype, const SigDasParams* p3, const SigLamActs* a, void* stream) {
    SIG_CHECK_ARG(tokens && p3 && a, "lam_fwd: null argument");
    SIG_CHECK_ARG(a->xb && a->q && a->a1 && a->a1pre && a->a2pre && a->offs && a->samp && a->loss, "lam_fwd: activation buffer missing");
    hipStream_t st = (hipStream_t)stream;
    const int dt = dtype;
    const int R = B * (L - 1), Rp = pad128(R), d = 512, P = (h / 4) * (w / 4);
    RUN(sig_launch_lam_gather(tokens, B, L, a->xb, (size_t)Rp, dt, st));
    for (int m = 0; m < 3; ++m) {
        const SigDasParams* p = p3 + m;
        const size_t o = (size_t)m * Rp * d;
        RUN(sig_launch_gemm_nt(nt(dt, a->xb + o, d, p->w_q, d, R, d, d, a->q + o, d, p->b_q), SIG_EPI_BIAS_BF16, st));
        RUN(sig_launch_gemm_nt(nt(dt, a->q + o, d, p->w_0, d, R, d, d, a->a1 + o, d, p->b_0, nullptr, 0, a->a1pre + o, d),
                               SIG_EPI_BIAS_GELUERF_BF16, st));
    }
    // the three modalities' tails (depthwise conv, offsets, bilinear sampling) in ONE launch
    RUN(sig_launch_lam_tail_fwd(tokens, B, L, h, w, a->a1, (size_t)Rp * d, lam_ptrs(p3, nullptr), a->a2pre, a->offs, a->samp, dt, st));
    RUN(sig_launch_lam_loss(a->samp, (size_t)B * P * d, a->loss, st));
    return 0;
}

int sig_lam_bwd(const float* tokens, int B, int L, int h, int w, int dtype, const SigDasParams* p3, const SigDasGrads* g3,
                const SigLamActs* a, const SigLamScratch* s, const float* dloss, float* dtokens, void* stream) {
    SIG_CHECK_ARG(tokens && p3 && g3 && a && s && dloss && dtokens, "lam_bwd: null argument");
    SIG_CHECK_ARG(s->da1pre && s->dq && s->dx, "lam_bwd: scratch missing");
    hipStream_t st = (hipStream_t)stream;
    const int dt = dtype;
    const int R = B * (L - 1), Rp = pad128(R), d = 512, P = (h / 4) * (w / 4);
    const size_t nsamp = (size_t)B * P * d;
    for (int m = 0; m < 3; ++m) SIG_CHECK_ARG(p3[m].wt_q && p3[m].wt_0, "lam_bwd: transposed weights missing");
    // all three tails first (one launch + one reduce): da1pre [3][Rp,512]; the per-sample partial rows of the tail's
    // parameter gradients live in dx, which is free until the first proj_q dgrad below
    RUN(sig_launch_lam_tail_bwd(tokens, B, L, h, w, a->a1, a->a1pre, (size_t)Rp * d, lam_ptrs(p3, g3), a->a2pre, a->offs, a->samp, nsamp,
                                dloss, s->da1pre, dtokens, dt, st, (size_t)3 * B * 18 * d <= (size_t)Rp * d ? s->dx : nullptr));
    for (int m = 0; m < 3; ++m) {
        const SigDasParams* p = p3 + m;
        const SigDasGrads* g = g3 + m;
        const size_t o = (size_t)m * Rp * d;
        const bf16_t* da1 = s->da1pre + o;
        // conv_offset.0
        RUN(sig_launch_gemm_nt(nt(dt, da1, d, p->wt_0, d, R, d, d, s->dq, d), SIG_EPI_BF16, st));
        RUN(sig_launch_gemm_tn(tn(dt, da1, d, a->q + o, d, Rp, d, d, g->w_0, d), st));
        RUN(sig_launch_colsum_bf16(da1, d, R, d, g->b_0, dt, st));
        // proj_q
        RUN(sig_launch_gemm_nt(nt(dt, s->dq, d, p->wt_q, d, R, d, d, s->dx, d), SIG_EPI_F32, st));
        RUN(sig_launch_gemm_tn(tn(dt, s->dq, d, a->xb + o, d, Rp, d, d, g->w_q, d), st));
        RUN(sig_launch_colsum_bf16(s->dq, d, R, d, g->b_q, dt, st));
        RUN(sig_launch_lam_scatter_add(s->dx, m, B, L, dtokens, st));
    }
    return 0;
}

int sig_xattn_fwd(const float* q, const uint16_t* kv, int B, int NK, uint16_t* out, float* probs, int dtype, void* stream) {
    return sig_launch_xattn_fwd(q, kv, B, NK, out, probs, dtype, (hipStream_t)stream);
}
int sig_xattn_bwd(const float* q, const uint16_t* kv, const float* probs, const float* dout, int B, int NK, float* dq,
                  uint16_t* dkv, int dtype, void* stream) {
    return sig_launch_xattn_bwd(q, kv, probs, dout, B, NK, dq, dkv, dtype, (hipStream_t)stream);
}

}  // extern "C"
