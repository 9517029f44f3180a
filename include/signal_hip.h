/* signal_hip.h -- C ABI of libsignal_hip.so, the MI355X (gfx950) implementation of the Signal hot path.
 *
 * The reference (maxingan2412/Signal) has no native layer: every stage below is an ATen call reached from
 * Python.  Each entry point therefore cites the reference Python it replaces (paths relative to the
 * reference root); INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer the caller owns (PyTorch caching allocator); the library borrows
 *     it for the duration of the call, keeps no reference and never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); calls are re-entrant;
 *   - 16-bit tensors are raw uint16_t bit patterns of ONE operand type per model, chosen by `dtype`:
 *     SIG_DT_BF16 (bfloat16) or SIG_DT_F16 (IEEE half, the type the reference's CUDA autocast computes in,
 *     engine/processor.py:165); both run the same MFMA rate, accumulation / residual stream / LayerNorm / softmax
 *     are f32 either way.  Names that say "bf16" mean "the 16-bit operand type"; "rows padded" means the allocation holds
 *     ceil(rows/128)*128 rows and the pad rows are zero (the library never writes them);
 *   - return 0 on success; non-zero = argument (1) or launch (2) error, message via sig_last_error().
 */
#ifndef SIGNAL_HIP_H
#define SIGNAL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIG_ABI_VERSION 3

enum { SIG_DT_BF16 = 0, SIG_DT_F16 = 1 };

const char* sig_last_error(void);
int sig_version(void);

/* Measurement aid (bench.py roofline leg): bracket every sig_gemm_nt launch of ONE shape (epilogue, N, K)
 * with HIP events on its own stream, from sig_prof_begin until sig_prof_end; sig_prof_end waits for the
 * recorded events (host-synchronising: call it outside any timed region) and returns the summed kernel
 * time, the number of launches and their algorithmic FLOPs (2*M*N*K each).  epilogue = SIG_PROF_TN256 selects
 * the 256x256 weight-gradient kernel behind sig_gemm_tn instead (N = I, K = J; N = K = 0: every shape), and
 * SIG_PROF_TN_GROUP the grouped weight-gradient kernel (FLOPs = the sum over its jobs). */
#define SIG_PROF_TN256 100
#define SIG_PROF_TN_GROUP 101   /* gemm_tn_group_kernel behind sig_gemm_tn_grouped / sig_block_bwd (N = K = 0) */
int sig_prof_begin(int epilogue, int N, int K, int max_launches);
int sig_prof_end(double* total_ms, int* launches, double* flops);
/* Tuning / test aid: pin the row tile of sig_gemm_nt (128, 256 or 320; 0 = choose by the cost estimate) wherever that
 * kernel is legal for the shape; returns the previous setting.  Same as the environment variable SIG_GEMM_TILE. */
int sig_tune_gemm_tile(int tile);
/* Tuning / test aid: the persistent 192x256 form of sig_gemm_nt (the store tail of tile n issued under the main loop of tile
 * n+1; wide K >= 768 contractions with a 16-bit output, whole 192-row tiles): 0 = off, 1 = the plain / bias epilogues and the
 * GELU' dgrad (default), 2 = also the QuickGELU forward without a saved derivative, 3 = also with one (the training c_fc),
 * wherever legal; returns the previous setting.  Environment preset: SIG_NT_PERSIST. */
int sig_tune_nt_persist(int on);
/* Tuning / test aid: waves per workgroup of the attention backward at L = 129 (16 * 8 + 1, every shipped geometry): 8 = one
 * 16-row tile per wave and pass (default; two resident workgroups put four waves on a SIMD), 4 = two tiles per wave sharing their
 * fragment reads.  Bit-identical results; returns the previous setting.  Environment preset: SIG_ATTN_BWD_WAVES. */
int sig_tune_attn_bwd_waves(int waves);
/* Same for the attention forward at L in (128, 144]: 9 = one 16-row query tile per wave (default: nine waves' q-fragment loads and
 * softmax chains in flight at once, 70 registers), 3 = three tiles per wave, one after the other.  Bit-identical results. */
int sig_tune_attn_fwd_waves(int waves);
/* Same for the weight-gradient path: 128 = the 128x128-tile kernel with f32 atomics, one launch per weight; 256 = the 256x256
 * kernel, one launch per weight; 0 = default (a block's four weights grouped into one launch).  Environment: SIG_GEMM_TN_TILE. */
int sig_tune_tn_path(int path);
/* Host-logic probe (no GPU needed): the grouped weight-gradient launch's work plan for `tiles` 256x256 output tiles, `ks` K-steps of
 * 64 rows, `grid` free CUs and `cs_units` column-sum units.  out8 = {balanced, nsplit, per, short_group, n_long, n_short_wg,
 * workgroups, column sums inside the launch (1) or as their own pass (0)}; see gemm_tn_grouped.hip (tng_plan). */
int sig_debug_tn_plan(int tiles, int ks, int grid, int cs_units, int* out8);
/* CUs (0..192) the GEMM launchers leave to concurrent work -- the RCCL channel workgroups that all-reduce gradient buckets
 * under the backward pass (engine/processor.py:212-261 runs DDP's reducer there).  The one-block-per-CU kernels are sized
 * in rounds of the FREE CUs; returns the previous setting.  Environment preset: SIG_RESERVED_CUS. */
int sig_tune_reserved_cus(int n);

/* ---- epilogues of sig_gemm_nt ---------------------------------------------------------------- */
enum {
    SIG_GEMM_F32 = 0,           /* out f32  = acc                                   */
    SIG_GEMM_BF16 = 1,          /* out bf16 = acc                                   */
    SIG_GEMM_BIAS_F32 = 2,      /* out f32  = acc + bias                            */
    SIG_GEMM_BIAS_BF16 = 3,     /* out bf16 = acc + bias                            */
    SIG_GEMM_BIAS_RES_F32 = 4,  /* out f32  = acc + bias + res  (res may alias out) */
    SIG_GEMM_BIAS_GELU_BF16 = 5,/* out 16-bit = QuickGELU(acc + bias); aux 16-bit (if given) = QuickGELU'(acc + bias),
                                   the only thing backward needs of the pre-activation             */
    SIG_GEMM_DGELU_BF16 = 6,    /* out 16-bit = acc * aux  (aux = the derivative saved by 5)         */
    SIG_GEMM_BIAS_GELUERF_BF16 = 7, /* exact-erf GELU (nn.GELU(), useA.py:356, DAS.py:59): out = GELU(acc + bias),
                                   aux (if given) = the pre-activation acc + bias                  */
    SIG_GEMM_DGELUERF_BF16 = 8, /* out bf16 = acc * GELU_erf'(aux)                  */
    SIG_GEMM_RES_F32 = 9        /* out f32  = acc + res                             */
};

/* out[M,N] = A[M,K] * Bt[N,K]^T (+ epilogue).  Every dense layer of the path: nn.Linear / MHA in-proj /
 * out-proj / `x @ proj` (modeling/clip/model.py:174-178,225,487; AddModule/useA.py:123-124,351-358) and
 * their input gradients (Bt = the transposed weight).  A rows padded; N % 128 == 0, K % 64 == 0. */
int sig_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, int M, int N, int K, int epilogue,
                void* out, int ldo, const float* bias, const float* res, int ldr, void* aux, int ldaux,
                int dtype, void* stream);

/* out[I,J] += P[Mr,I]^T * Q[Mr,J]  (f32, atomically accumulated): weight gradients dW = dY^T X that autograd
 * computes for the same layers.  Mr % 64 == 0 with zero pad rows; I, J % 128 == 0; split = 0 lets the
 * library choose the number of row chunks. */
int sig_gemm_tn(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, int Mr, int I, int J, float* out,
                int ldo, int split, int dtype, void* stream);

/* n (1..4) weight gradients that share the row count Mr, out_k[I_k,J_k] += P_k[Mr,I_k]^T * Q_k[Mr,J_k], as ONE
 * launch: the four nn.Linear weights of a ResidualAttentionBlock (modeling/clip/model.py:172-178: attn.in_proj,
 * attn.out_proj, mlp.c_fc, mlp.c_proj) in one pass over the chip, every CU doing the same number of K-steps
 * (stream-K over the token rows) and a fixed-order reduce (deterministic; no atomics).  Same operand contract as
 * sig_gemm_tn; shapes the grouped kernel does not take (I or J not a multiple of 256, short Mr) fall back to
 * n calls of sig_gemm_tn. */
typedef struct SigTnJobDesc {
    const uint16_t* P;   /* dY [Mr, ldp] */
    const uint16_t* Q;   /* X  [Mr, ldq] */
    float* out;          /* dW [I, ldo] */
    int ldp, ldq, ldo, I, J;
    float* colsum;       /* NULL, or [I]: += column sums of P over the rows = the bias gradient that goes with dW */
} SigTnJobDesc;
int sig_gemm_tn_grouped(const SigTnJobDesc* jobs, int n, int Mr, int dtype, void* stream);

/* LayerNorm (fp32 statistics, eps as given): modeling/clip/model.py:154-160, AddModule/useA.py:414-423.
 * y_bf16 / y_f32 / mean / rstd may be NULL when not wanted. */
int sig_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y_bf16, float* y_f32,
                      float* mean, float* rstd, int M, int D, float eps, int dtype, void* stream);
/* dx = dres + LN'(dy); dgamma/dbeta are ACCUMULATED and may both be NULL. */
int sig_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx_f32, uint16_t* dx_bf16, float* dgamma,
                      float* dbeta, int M, int D, int dtype, void* stream);
/* Chained column reduce for a caller that owns a whole backward pass (the stage calls below run LayerNorm backward 25 times per
 * step): with sig_tune_ln_defer(1) a LayerNorm backward that produces column sums (dgamma / dbeta, the bias gradients that ride
 * along in sig_block_bwd) leaves its per-workgroup partial rows in library scratch and the NEXT LayerNorm backward on the same
 * stream adds them up in its first workgroups, instead of one small reduce launch each; sig_ln_flush(stream) launches the reduce
 * for whatever is still pending.  Until the flush the affected gradients are incomplete.  Returns the previous setting.  Same
 * summation order either way (bit-identical gradients). */
int sig_tune_ln_defer(int on);
/* Tuning: the grouped weight-gradient operation of sig_block_bwd / sig_gemm_tn_grouped writes dW = dY^T X instead of adding to
 * dW (the paths that accumulate by construction zero dW themselves first).  For a caller that writes every weight gradient exactly
 * once per step and therefore need not zero them (-345 MB of fills and -340 MB of reads per step at B = 64).  Default 0; returns
 * the previous setting. */
int sig_tune_tn_overwrite(int on);
int sig_ln_flush(void* stream);

/* Self-attention of nn.MultiheadAttention as called at modeling/clip/model.py:223-225 (no mask, no dropout):
 * qkv bf16 [S*L, 3*H*64] packed (q|k|v, head h = columns 64h..64h+63 of each) -> out bf16 [S*L, H*64],
 * lse f32 [S,H,L] (log-sum-exp of the scaled scores, kept for backward).  L <= 144. */
int sig_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int S, int L, int H, int dtype, void* stream);
int sig_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                 uint16_t* dqkv, int S, int L, int H, int dtype, void* stream);

/* Packing helpers: f32 -> bf16 (optionally transposed), column sums (bias gradients, accumulated). */
int sig_cast_bf16(const float* src, uint16_t* dst, size_t n, int dtype, void* stream);
int sig_transpose_cast_bf16(const float* src, uint16_t* dst, int rows, int cols, int dtype, void* stream);
/* n transposes in one launch: table[d] = {src f32*, dst bf16*, rows, cols} as int64, tile_start[d] = index of matrix d's
 * first 64x64 tile in the flattened grid, tile_start[n] = total_tiles. */
int sig_transpose_cast_multi(const int64_t* table, const int* tile_start, int n, int total_tiles, int dtype, void* stream);
/* The same for sources that already are 16-bit (the operand mirror sig_adam_step / sig_cast_bf16 maintain): table[d] = {src 16-bit*,
 * dst 16-bit*, rows, cols}, rows and cols multiples of 64; 16-B loads and stores, a pure permutation of bit patterns (the result equals
 * cast-then-transpose of the f32 master bit for bit). */
int sig_transpose16_multi(const int64_t* table, const int* tile_start, int n, int total_tiles, void* stream);
int sig_colsum_bf16(const uint16_t* a, int lda, int M, int N, float* out, int dtype, void* stream);
int sig_colsum_f32(const float* a, int lda, int M, int N, float* out, void* stream);

/* Patch embedding front end (modeling/clip/model.py:448-459; modeling/meta_arch.py:101-103).
 * sig_im2col: img f32 [nimg,3,H,W] -> bf16 [nimg*(H/P)*(W/P), 3*P*P] (column = c*P*P + dy*P + dx), the A
 * operand of the conv1-as-GEMM.  sig_embed_assemble: prepend class_embedding, add sie_coe*cv_embed[cam[b]] to
 * the CLS row, add positional_embedding, apply ln_pre.  Sequences are ordered s = modality*B + b. */
int sig_im2col(const float* img, uint16_t* out, int nimg, int H, int W, int P, int dtype, void* stream);
int sig_embed_assemble(const float* tok, const float* class_embedding, const float* positional_embedding,
                       const float* cv_embed, const int64_t* cam_label, float sie_coe, const float* ln_w,
                       const float* ln_b, float* x, float* pre_ln, float* mean, float* rstd, int S, int B, int L,
                       int D, float eps, void* stream);
int sig_embed_assemble_bwd(const float* d_pre_ln, float* dtok_f32, uint16_t* dtok_bf16, float* d_class_embedding,
                  float* d_positional_embedding, float* d_cv_embed, const int64_t* cam_label, float sie_coe,
                  int S, int B, int L, int D, int dtype, void* stream);

/* ================================================================================================
 * Stage-level entry points: one call per stage of VisionTransformer.forward (modeling/clip/model.py:447-488)
 * and of its backward.  Token tensors are [M = S*L rows padded, columns]; S = 3*B sequences ordered
 * modality-major (s = modality*B + b), so the reference's three backbone calls (make_model.py:181-183) are
 * one batched problem.  "acts" structs are written by *_fwd and read by *_bwd; "grads" are f32 and
 * ACCUMULATED (the caller zeroes them once per step).
 * ================================================================================================ */
typedef struct SigVitDims {
    int S, B, L, D, H, F, out_dim;   /* sequences, per-modality batch, tokens, width, heads, MLP width, proj columns */
    int dtype;                       /* SIG_DT_BF16 / SIG_DT_F16: type of every 16-bit activation and weight operand */
} SigVitDims;

/* --- patch embedding + CLS/camera/positional + ln_pre (clip/model.py:448-459, meta_arch.py:101-103) --- */
typedef struct SigEmbedParams {
    const uint16_t* w_conv;            /* bf16 [D, 3*P*P] = conv1.weight flattened */
    const float *class_embedding, *positional_embedding, *cv_embed /* [cams, D] or NULL */, *ln_w, *ln_b;
    float sie_coe;
} SigEmbedParams;
typedef struct SigEmbedActs {
    uint16_t* patches;                 /* bf16 [S*(L-1) padded, 3*P*P] */
    float* tok;                        /* f32  [S*(L-1) padded, D]     */
    float* pre_ln;                     /* f32  [M, D] (NULL when no backward is needed) */
    float *mean, *rstd;                /* [M] (may be NULL with pre_ln) */
    float* x0;                         /* f32  [M padded, D]: the residual stream entering block 0 */
} SigEmbedActs;
typedef struct SigEmbedGrads {
    float *w_conv, *class_embedding, *positional_embedding, *cv_embed, *ln_w, *ln_b;
} SigEmbedGrads;
/* img_parts[i]: f32 [S/n_parts, 3, img_h, img_w] (contiguous) = the images of sequences i*S/n_parts .. : the three modality
 * tensors the reference passes to its three backbone calls, read in place */
int sig_embed_fwd(const SigVitDims* d, const SigEmbedParams* p, const SigEmbedActs* a, const float* const* img_parts,
                  int n_parts, const int64_t* cam_label, int img_h, int img_w, int patch, void* stream);
/* dx0: f32 [M,D] gradient of x0; scratch_dpre f32 [M,D]; scratch_dtok bf16 [S*(L-1) padded, D] (pad rows zero) */
int sig_embed_bwd(const SigVitDims* d, const SigEmbedParams* p, const SigEmbedActs* a, const SigEmbedGrads* g,
                  const float* dx0, float* scratch_dpre, uint16_t* scratch_dtok, const int64_t* cam_label,
                  int patch, void* stream);

/* --- one ResidualAttentionBlock.forward_ori (clip/model.py:227-231) --- */
typedef struct SigBlockParams {
    const uint16_t *w_in, *w_out, *w_fc, *w_proj;      /* bf16 [3D,D] [D,D] [F,D] [D,F] (PyTorch [out,in] layout) */
    const uint16_t *wt_in, *wt_out, *wt_fc, *wt_proj;  /* their transposes, backward only (may be NULL in fwd)    */
    const float *b_in, *b_out, *b_fc, *b_proj, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
} SigBlockParams;
typedef struct SigBlockActs {
    float* x_in;                       /* f32 [Mp,D] block input (read) */
    uint16_t* h1;  float *mean1, *rstd1;
    uint16_t* qkv; float* lse;         /* bf16 [Mp,3D], f32 [S,H,L] */
    uint16_t* attn;                    /* bf16 [Mp,D] */
    float* x_mid;                      /* f32 [Mp,D] */
    uint16_t* h2;  float *mean2, *rstd2;
    uint16_t* u;                       /* 16-bit [Mp,F] QuickGELU'(c_fc pre-activation) (NULL when no backward is needed) */
    uint16_t* g;                       /* 16-bit [Mp,F] QuickGELU(c_fc pre-activation) */
    float* x_out;                      /* f32 [Mp,D] block output (written) */
} SigBlockActs;
typedef struct SigBlockGrads {
    float *w_in, *w_out, *w_fc, *w_proj, *b_in, *b_out, *b_fc, *b_proj, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
} SigBlockGrads;
typedef struct SigBlockScratch {       /* reusable across blocks; pad rows must be zero */
    uint16_t* du;                      /* bf16 [Mp,F]  */
    uint16_t* dh;                      /* bf16 [Mp,D]  */
    uint16_t* dqkv;                    /* bf16 [Mp,3D] */
    float* dx_mid;                     /* f32  [Mp,D]  */
    uint16_t* dx_mid_b;                /* bf16 [Mp,D]  */
} SigBlockScratch;
int sig_block_fwd(const SigVitDims* d, const SigBlockParams* p, const SigBlockActs* a, void* stream);
/* dx_out (f32) and dx_out_b (its bf16 copy) are the gradient of x_out; writes dx_in and dx_in_b (may alias dx_out*).
 * dx_in_colsum (nullable, [D]) accumulates the column sums of dx_in: it IS the c_proj bias gradient of the block
 * below.  b_proj_done != 0 says the stage above already did that for this block's own g->b_proj. */
int sig_block_bwd(const SigVitDims* d, const SigBlockParams* p, const SigBlockActs* a, const SigBlockGrads* g,
                  const SigBlockScratch* s, const float* dx_out, const uint16_t* dx_out_b, float* dx_in,
                  uint16_t* dx_in_b, float* dx_in_colsum, int b_proj_done, void* stream);

/* --- ln_post + x @ proj on all tokens (clip/model.py:485-488) --- */
typedef struct SigHeadParams {
    const uint16_t* proj_t;            /* bf16 [out_dim, D] = proj^T (forward operand)  */
    const uint16_t* proj;              /* bf16 [D, out_dim]          (dgrad operand)    */
    const float *ln_w, *ln_b;
} SigHeadParams;
typedef struct SigHeadActs {
    float* x;                          /* f32 [Mp,D] last block output (read) */
    uint16_t* hp; float *mean, *rstd;  /* bf16 [Mp,D] ln_post output */
    float* tokens;                     /* f32 [Mp,out_dim] */
} SigHeadActs;
typedef struct SigHeadGrads { float *proj, *ln_w, *ln_b; } SigHeadGrads;
int sig_head_fwd(const SigVitDims* d, const SigHeadParams* p, const SigHeadActs* a, void* stream);
/* dtokens f32 [Mp,out_dim] (pad rows zero); scratch_dtok_b bf16 [Mp,out_dim]; scratch_dh bf16 [Mp,D] */
int sig_head_bwd(const SigVitDims* d, const SigHeadParams* p, const SigHeadActs* a, const SigHeadGrads* g,
                 const float* dtokens, uint16_t* scratch_dtok_b, uint16_t* scratch_dh, float* dx, uint16_t* dx_b,
                 float* dx_colsum /* nullable [D]: += column sums of dx = last block's c_proj bias gradient */,
                 void* stream);

/* ================================================================================================
 * K18 -- the path's one exchange step: SUM all-reduce of gradient ranges over RCCL (xGMI) on a side HIP stream,
 * overlapped with the backward.  Replaces DistributedDataParallel's gradient reduction (engine/processor.py:100-105).
 * RCCL is bound at run time (dlopen librccl.so.1), so a process that already carries PyTorch's RCCL keeps using that copy.
 *   rank 0: sig_comm_unique_id(id) -> ship the 128 bytes to the other ranks (any channel) -> every rank: sig_comm_init
 *   per bucket: sig_comm_allreduce_async(c, grad + lo, hi - lo, compute_stream)  [right after the bucket's backward is enqueued]
 *   before the optimizer: sig_comm_wait(c, compute_stream)                        [device-side wait, no host sync]
 * The average (1 / world) is the caller's (sig_adam_step's grad_scale).
 * ================================================================================================ */
typedef struct SigComm SigComm;
int sig_comm_unique_id(void* id128 /* out: 128 bytes */);
int sig_comm_init(SigComm** comm, int rank, int world, const void* id128);
int sig_comm_allreduce_async(SigComm* comm, float* buf, size_t count, void* compute_stream);
int sig_comm_wait(SigComm* comm, void* stream);
int sig_comm_destroy(SigComm* comm);

/* ================================================================================================
 * SIM -- Select_Interactive_Module (modeling/AddModule/useA.py:426-476), on the projected tokens
 * tokens f32 [S*L padded, 512] (row s*L = CLS of sequence s = modality*B + b, rows s*L+1.. = patches).
 * ================================================================================================ */
typedef struct SigSimParams {
    /* TokenSelection.W_q / W_k (f32 [512,512] + bias); W_v is dead in the reference (useA.py:46-48) */
    const float *sel_wq, *sel_bq, *sel_wk, *sel_bk;
    /* ModalInteractive: cross_attn.in_proj (rows 0..511 = q, 512..1535 = k|v), out_proj, ffn.0, ffn.2, norm1/2 */
    const uint16_t *w_q, *w_kv, *w_o, *w_f1, *w_f2;          /* bf16 [512,512] [1024,512] [512,512] [1024,512] [512,1024] */
    const uint16_t *wt_q, *wt_kv, *wt_o, *wt_f1, *wt_f2;     /* transposes, backward only */
    const float *b_q, *b_kv, *b_o, *b_f1, *b_f2, *n1_w, *n1_b, *n2_w, *n2_b;
    int topk;                                                /* MODEL.TOPK: k1 = topk, k2 = 2*topk */
    int dtype;                                               /* SIG_DT_BF16 / SIG_DT_F16 (selection itself is f32) */
    int max_keep;                                            /* 0, or int(Lp * MODEL.KEEP_RATIO) when MODEL.FIXED_KEEP_RATIO:
                                                              * exactly that many tokens per modality (useA.py:253-316) */
} SigSimParams;
typedef struct SigSimActs {
    float *qprime, *cconst, *intra, *inter;  /* [B,3,512] [B,3] [B,3,Lp] [B,3,3Lp] : selection scores (raw, pre-softmax) */
    float* mask_f;                           /* [3,B,Lp] 0/1 : TokenSelection.last_masks */
    uint8_t* mask_u8;                        /* [3,B,Lp] or NULL */
    uint16_t* sel;                           /* bf16 [B*3*Lp padded, 512] masked tokens, rows (b, modality, j) */
    uint16_t* cls_b; float* cls_f;           /* [B*3 padded, 512] stacked CLS queries, rows (b, modality) */
    float* qh;                               /* f32  [B*3 padded, 512] projected queries */
    uint16_t* kv;                            /* bf16 [B*3*Lp padded, 1024] */
    float* probs;                            /* f32  [B,24,3Lp] */
    uint16_t* ao;                            /* bf16 [B*3 padded, 512] attention output */
    float* y;  float* z1; uint16_t* z1_b; float *mean1, *rstd1;
    uint16_t *f1_pre, *f1;                   /* bf16 [B*3 padded, 1024] */
    float* y2; float *mean2, *rstd2;
    float* out;                              /* f32 [B*3 padded, 512] == vars_total [B,1536] */
} SigSimActs;
typedef struct SigSimGrads {
    float *w_q, *w_kv, *w_o, *w_f1, *w_f2, *b_q, *b_kv, *b_o, *b_f1, *b_f2, *n1_w, *n1_b, *n2_w, *n2_b;
} SigSimGrads;
typedef struct SigSimScratch {               /* pad rows zero */
    float* dy2;  uint16_t* dy2_b;            /* [B*3 padded, 512]  */
    uint16_t* df1;                           /* [B*3 padded, 1024] */
    float* dz1;  float* dy; uint16_t* dy_b;  /* [B*3 padded, 512]  */
    float* dao;  float* dqh; uint16_t* dqh_b;/* [B*3 padded, 512]  */
    uint16_t* dkv;                           /* [B*3*Lp padded, 1024] */
    uint16_t* dsel;                          /* [B*3*Lp padded, 512]  */
    float* dcls;                             /* [B*3 padded, 512]  */
} SigSimScratch;
/* selection only (useA.py:223-251): masks from tokens */
int sig_sim_select(const float* tokens, int B, int L, const SigSimParams* p, const SigSimActs* a, void* stream);
/* selection + interaction (the whole Select_Interactive_Module.forward) */
int sig_sim_fwd(const float* tokens, int B, int L, const SigSimParams* p, const SigSimActs* a, void* stream);
/* dout f32 [B*3 padded,512]; dtokens f32 [S*L padded,512] is ACCUMULATED into */
int sig_sim_bwd(const float* dout, int B, int L, const SigSimParams* p, const SigSimActs* a, const SigSimGrads* g,
                const SigSimScratch* s, float* dtokens, void* stream);
/* attention core of the interaction block, exposed for tests */
int sig_xattn_fwd(const float* q, const uint16_t* kv, int B, int NK, uint16_t* out, float* probs, int dtype, void* stream);
int sig_xattn_bwd(const float* q, const uint16_t* kv, const float* probs, const float* dout, int B, int NK, float* dq,
                  uint16_t* dkv, int dtype, void* stream);

/* ================================================================================================
 * GAM -- AlignmentM.Cls_Align (modeling/AddModule/useB.py:76-126) with volume_computation3
 * (utils/volume.py:14-62) in closed form: mean-pool the patch tokens per modality, L2-normalise,
 * V[i,j] = sqrt|det Gram(r_i, n_j, t_j)|, loss = (CE(-V/T, diag, ls=.1) + CE(-V^T/T, diag, ls=.1)) / 2.
 * Backward is analytic; where det == 0 the reference produces NaN gradients, this uses a zero sub-gradient.
 * ================================================================================================ */
typedef struct SigGamActs {
    float *fh, *nrm;          /* [3,B,512] normalised pooled features, [3B] their pre-normalisation norms */
    float *lv, *la, *vec;     /* [B,B] r.n, [B,B] r.t, [4B] (ll | vv | va | aa) */
    float* coef;              /* [2*B*B + 4*B + 1] backward coefficients */
    float* loss;              /* [1] */
} SigGamActs;
int sig_gam_fwd(const float* tokens, int B, int L, const float* contra_temp, const SigGamActs* a, void* stream);
/* dloss: device scalar (upstream gradient); dtokens f32 [S*L,512] and d_contra_temp [1] are ACCUMULATED into */
int sig_gam_bwd(int B, int L, const SigGamActs* a, const float* dloss, float* dtokens, float* d_contra_temp, void* stream);

/* ================================================================================================
 * LAM -- AlignmentM.patch_Align (useB.py:128-167) with DA_sample (DAS.py:107-165), one parameter set per
 * modality (DAS_r, DAS_n, DAS_t).  The patch tokens of a modality, viewed as an [h, w] map of 512-vectors:
 * q = proj_q(x); o = conv_offset(q) (1x1 -> GELU -> depthwise 4x4 stride 4 -> GELU -> 1x1 to one channel);
 * p = clamp(ref + 2 tanh(o) / (n-1)) on both axes; bilinear sample of x at p (align_corners, zero padding);
 * loss = mean of the three pairwise MSEs of the sampled maps.
 * ================================================================================================ */
typedef struct SigDasParams {
    const uint16_t *w_q, *w_0;      /* bf16 [512,512] proj_q.weight, conv_offset.0.weight ([out,in,1,1])   */
    const uint16_t *wt_q, *wt_0;    /* transposes (backward only)                                          */
    const float *b_q, *b_0;         /* [512]                                                               */
    const float *wd, *bd, *w4;      /* conv_offset.2.weight [512,1,4,4], .2.bias [512], conv_offset.4.weight [1,512,1,1] */
} SigDasParams;
typedef struct SigDasGrads { float *w_q, *w_0, *b_q, *b_0, *wd, *bd, *w4; } SigDasGrads;
typedef struct SigLamActs {         /* leading [3] = modality; R = B*(L-1) rows padded to 128 */
    uint16_t *xb, *q, *a1, *a1pre;  /* bf16 [3][R,512] */
    float *a2pre, *offs, *samp;     /* f32 [3][B,P,512], [3][B,P,3] (o, p_y, p_x), [3][B,P,512] ; P = (h/4)*(w/4) <= 8 */
    float* loss;                    /* [1 + 64]: loss[0] = the LAM loss, the rest = per-block partial sums (scratch) */
} SigLamActs;
typedef struct SigLamScratch { uint16_t *da1pre, *dq; float* dx; } SigLamScratch;   /* da1pre [3][R padded,512]; dq, dx [R padded,512]; pad rows zero */
int sig_lam_fwd(const float* tokens, int B, int L, int h, int w, int dtype, const SigDasParams* p3, const SigLamActs* a,
                void* stream);
int sig_lam_bwd(const float* tokens, int B, int L, int h, int w, int dtype, const SigDasParams* p3, const SigDasGrads* g3,
                const SigLamActs* a, const SigLamScratch* s, const float* dloss, float* dtokens, void* stream);

/* ================================================================================================
 * Optimizer step of the train loop (engine/processor.py:259-261 with solver/make_optimizer.py:4-45):
 * torch.optim.Adam semantics (L2 weight decay added to the gradient, bias correction, eps outside the sqrt)
 * over a flat f32 parameter buffer cut into nseg segments (one per parameter, the reference's one param
 * group per parameter) with per-segment lr and weight decay.  seg_end[k] = end offset (elements) of segment
 * k; grad_scale multiplies the gradient first (1/world_size after a sum all-reduce); p16 (may be NULL)
 * receives the refreshed 16-bit GEMM operands of type `dtype`.
 *
 * fp16 loss scaling = the reference's amp.GradScaler (processor.py:119,259-261) kept on the device:
 * scale_state (NULL = no scaling) is a 5-float record [scale, 1/scale, found_inf, growth_tracker, applied_steps].
 *   sig_grad_check         found_inf = 1 if any gradient element is inf / NaN
 *   sig_adam_step          multiplies gradients by 1/scale, skips the WHOLE update when found_inf is set and takes
 *                          the bias-correction step from applied_steps + 1 (the `step` argument is then ignored)
 *   sig_loss_scale_update  scale *= backoff after an overflow, *= growth after `interval` clean steps; clears
 *                          found_inf; counts applied steps.  No call synchronises with the host.
 * ================================================================================================ */
int sig_adam_step(float* p, const float* g, float* m, float* v, uint16_t* p16, int dtype, const int* seg_end,
                  const float* seg_lr, const float* seg_wd, int nseg, float beta1, float beta2, float eps, int step,
                  float grad_scale, const float* scale_state, size_t n, void* stream);
int sig_grad_check(const float* g, size_t n, float* scale_state, void* stream);
/* Zero n ranges of one f32 buffer in one launch: table[k] = {offset, length} (floats), chunk_start[k] = index of range k's first
 * 4096-float chunk, chunk_start[n] = total_chunks.  With sig_tune_tn_overwrite(1) around its backward a training engine zeroes only
 * the gradients that accumulate during a step; the transformer blocks' weight gradients are overwritten by sig_block_bwd. */
int sig_zero_ranges(float* base, const int64_t* table, const int* chunk_start, int n, int total_chunks, void* stream);
int sig_loss_scale_update(float* scale_state, float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* ================================================================================================
 * ReID head.  All f32, B <= 128.
 * sig_bnneck_fwd: y = BatchNorm1d(x) in TRAINING mode (batch statistics, running stats updated with `momentum`;
 *   pass NULL running stats to skip) and logits = y W^T with the bias-free classifier W [C,F]
 *   (modeling/make_model.py:77-81,194-195,213-219).  sig_bnneck_bwd: dW += dlogits^T y, dx += BN'(dlogits W)
 *   (dx is ACCUMULATED: the triplet gradient of the same features is usually already in it), dbn_w += ...,
 *   dbn_b (may be NULL: the reference freezes it).
 * sig_reid_loss: loss = id_weight * CE_labelsmooth(logits, target; eps) + triplet_weight * batch-hard triplet(feat)
 *   (layers/softmax_loss.py:23-34, layers/triplet_loss.py:16-135, layers/make_loss.py:109-150); margin < 0 selects the
 *   soft-margin form (MODEL.NO_MARGIN).  With dlogits / dfeat non-NULL it also writes dL/dlogits and ACCUMULATES
 *   dL/dfeat, both times upstream[0] (device scalar, NULL = 1).  gram [B,B], pidx/nidx [B], coef [2B] are scratch.
 * ================================================================================================ */
int sig_bnneck_fwd(const float* x, const float* bn_w, const float* bn_b, float* running_mean, float* running_var,
                   float momentum, const float* cls_w, int B, int F, int C, float* y, float* mean, float* rstd,
                   float* logits, void* stream);
int sig_bnneck_bwd(const float* x, const float* y, const float* bn_w, const float* mean, const float* rstd,
                   const float* cls_w, const float* dlogits, int B, int F, int C, float* dy_scratch, float* dx,
                   float* dbn_w, float* dbn_b, float* dcls_w, void* stream);
int sig_reid_loss(const float* logits, const float* feat, const int64_t* target, int B, int F, int C,
                  float label_smooth_eps, float id_weight, float triplet_weight, float margin, const float* upstream,
                  float* loss, float* dlogits, float* gram, int* pidx, int* nidx, float* coef, float* dfeat,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGNAL_HIP_H */
