// SIM -- Selective Interaction Module (modeling/AddModule/useA.py).
//
// Token selection (useA.py:50-251) works on fp32 scores: integer output, bit-exact contract on tie-free
// rows.  The inter-modal scores q.(W_k p + b_k) are evaluated as (W_k^T q).p + q.b_k, so the
// [B*384,512]x[512,512] key projection of the reference (useA.py:124) is never formed.  Selection is a
// rank count in LDS (rank < k, lowest index first on ties); no Python loops, no host syncs.
//
// Interaction (useA.py:364-411): the K/V projection of the 384 masked tokens runs on the bf16 MFMA GEMM;
// this file holds the 3-query x 384-key x 8-head attention core (one workgroup per sample) and its backward.
#include "sig_common.h"
#include "sig_kernels.h"

#define SIM_D 512
#define SIM_SQRT_D 22.627416997969522f

// ------------------------------------------------------------------------------------------------
// q' = W_k^T (W_q g + b_q),  c = (W_q g + b_q).b_k        one workgroup per (sample, query modality)
// cls row of sequence s = m*B + b lives at tokens[(s*L)*d .. ]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void sim_qprime_kernel(const float* __restrict__ tokens, int L, int B,
                                                         const float* __restrict__ Wq, const float* __restrict__ bq,
                                                         const float* __restrict__ Wk, const float* __restrict__ bk,
                                                         float* __restrict__ qprime, float* __restrict__ cconst) {
    __shared__ float g[SIM_D], q[SIM_D], red[8];
    const int s = blockIdx.x;  // m*B + b
    const int m = s / B, b = s - m * B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    g[tid] = tokens[(size_t)s * L * SIM_D + tid];
    __syncthreads();
    const float4 g0 = *(const float4*)&g[lane * 8], g1 = *(const float4*)&g[lane * 8 + 4];
    // 64 outputs per wave, 8 at a time: the 16 row loads of a batch are requested before the first reduction (one by one,
    // the 64 L2 latencies were serialised).  Per output the arithmetic and its order are unchanged -- the selection
    // downstream is compared bit-exactly with the reference.
    for (int ob = wave; ob < SIM_D; ob += 64) {
        float4 w0[8], w1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = ob + u * 8;
            w0[u] = *(const float4*)(Wq + (size_t)o * SIM_D + lane * 8);
            w1[u] = *(const float4*)(Wq + (size_t)o * SIM_D + lane * 8 + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = ob + u * 8;
            float acc = w0[u].x * g0.x + w0[u].y * g0.y + w0[u].z * g0.z + w0[u].w * g0.w + w1[u].x * g1.x + w1[u].y * g1.y +
                        w1[u].z * g1.z + w1[u].w * g1.w;
            acc = wave_sum(acc);
            if (lane == 0) q[o] = acc + bq[o];
        }
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll 16
    for (int o = 0; o < SIM_D; ++o) acc += Wk[(size_t)o * SIM_D + tid] * q[o];
    qprime[((size_t)b * 3 + m) * SIM_D + tid] = acc;
    float c = wave_sum(q[tid] * bk[tid]);
    if (lane == 0) red[wave] = c;
    __syncthreads();
    if (tid == 0) cconst[b * 3 + m] = red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7];
}

// one wave per patch token: intra score with its own CLS, inter scores with the three q'
__global__ __launch_bounds__(256) void sim_scores_kernel(const float* __restrict__ tokens, int L, int B,
                                                         const float* __restrict__ qprime, const float* __restrict__ cconst,
                                                         float* __restrict__ intra, float* __restrict__ inter) {
    const int Lp = L - 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // (s, j)
    if (row >= 3 * B * Lp) return;
    const int s = row / Lp, j = row - s * Lp;
    const int m = s / B, b = s - m * B;
    const float* t = tokens + ((size_t)s * L + 1 + j) * SIM_D + lane * 8;
    const float* gcls = tokens + (size_t)s * L * SIM_D + lane * 8;
    const float4 t0 = *(const float4*)t, t1 = *(const float4*)(t + 4);
    auto dot8 = [&](const float* v) {
        const float4 a = *(const float4*)v, c = *(const float4*)(v + 4);
        return a.x * t0.x + a.y * t0.y + a.z * t0.z + a.w * t0.w + c.x * t1.x + c.y * t1.y + c.z * t1.z + c.w * t1.w;
    };
    const float di = wave_sum(dot8(gcls));
    float dq[3];
#pragma unroll
    for (int mq = 0; mq < 3; ++mq) dq[mq] = wave_sum(dot8(qprime + ((size_t)b * 3 + mq) * SIM_D + lane * 8));
    if (lane == 0) {
        intra[((size_t)b * 3 + m) * Lp + j] = di / SIM_SQRT_D;
#pragma unroll
        for (int mq = 0; mq < 3; ++mq)
            inter[((size_t)b * 3 + mq) * (3 * Lp) + m * Lp + j] = (dq[mq] + cconst[b * 3 + mq]) / SIM_SQRT_D;
    }
}

// softmax + top-k as rank count + union; one workgroup (256 threads) per sample.  Lp <= 128.
__device__ __forceinline__ float block_max256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sim_select_kernel(const float* __restrict__ intra, const float* __restrict__ inter,
                                                         int B, int Lp, int k1, int k2, int max_keep, float* __restrict__ mask_f,
                                                         unsigned char* __restrict__ mask_u8) {
    __shared__ float pi[3][128], pc[3][384], red[4];
    __shared__ int sel[3][128];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 3 * 128; i += 256) sel[i / 128][i % 128] = 0;
    // ---- softmax rows (useA.py:72-74 and :129) ----
    for (int m = 0; m < 3; ++m) {
        const float v = tid < Lp ? intra[((size_t)b * 3 + m) * Lp + tid] : -INFINITY;
        const float mx = block_max256(v, red);
        const float e = tid < Lp ? expf(v - mx) : 0.f;
        const float sm = block_sum256(e, red);
        if (tid < Lp) pi[m][tid] = e / sm;
    }
    for (int m = 0; m < 3; ++m) {
        const float* src = inter + ((size_t)b * 3 + m) * 3 * Lp;
        const float v0 = tid < 3 * Lp ? src[tid] : -INFINITY;
        const float v1 = tid + 256 < 3 * Lp ? src[tid + 256] : -INFINITY;
        const float mx = block_max256(fmaxf(v0, v1), red);
        const float e0 = tid < 3 * Lp ? expf(v0 - mx) : 0.f, e1 = tid + 256 < 3 * Lp ? expf(v1 - mx) : 0.f;
        const float sm = block_sum256(e0 + e1, red);
        if (tid < 3 * Lp) pc[m][tid] = e0 / sm;
        if (tid + 256 < 3 * Lp) pc[m][tid + 256] = e1 / sm;
    }
    __syncthreads();
    // ---- intra: top-k1 of the modality's own 128 (useA.py:79-93) ----
    for (int m = 0; m < 3; ++m) {
        if (tid < Lp) {
            const float v = pi[m][tid];
            int rank = 0;
#pragma unroll 16   // independent broadcast LDS reads: without unrolling each iteration paid the LDS latency
            for (int j = 0; j < Lp; ++j) {
                const float o = pi[m][j];
                rank += (o > v) || (o == v && j < tid);
            }
            if (rank < k1) sel[m][tid] = 1;
        }
    }
    // ---- inter: top-k2 of the two OTHER modalities' scores, marks THEIR masks (useA.py:136-218) ----
    for (int m = 0; m < 3; ++m) {
        const int oa = m == 0 ? 1 : 0, ob = m == 2 ? 1 : 2;
        if (tid < 2 * Lp) {
            const int mt = tid < Lp ? oa : ob, jt = tid < Lp ? tid : tid - Lp;
            const float v = pc[m][mt * Lp + jt];
            int rank = 0;   // candidate list = [oa's Lp scores | ob's Lp scores]; ties go to the lower list index
            const float* la = &pc[m][oa * Lp];
            const float* lb = &pc[m][ob * Lp];
#pragma unroll 16
            for (int j = 0; j < Lp; ++j) {
                const float o = la[j];
                rank += (o > v) || (o == v && j < tid);
            }
#pragma unroll 16
            for (int j = 0; j < Lp; ++j) {
                const float o = lb[j];
                rank += (o > v) || (o == v && j + Lp < tid);
            }
            if (rank < k2) sel[mt][jt] = 1;  // benign race: every writer stores 1
        }
    }
    __syncthreads();
    // ---- optional exact keep ratio (useA.py:253-316, MODEL.FIXED_KEEP_RATIO): every modality keeps exactly max_keep tokens,
    // ranked by the RAW intra-modal dot product (intra[] is that product / sqrt(d): the same order) -- more selected than
    // max_keep: the best max_keep of the SELECTED stay; fewer: the best un-selected ones are added.  Rank count among the
    // tokens of the same status, lowest index first on ties. ----
    if (max_keep > 0) {
        for (int m = 0; m < 3; ++m) {
            const int mine = tid < Lp ? sel[m][tid] : 0;
            const int cnt = (int)(block_sum256((float)mine, red) + 0.5f);
            if (tid < Lp) pi[m][tid] = intra[((size_t)b * 3 + m) * Lp + tid];      // the softmax rows are no longer needed
            __syncthreads();
            int on = mine;
            if (tid < Lp && cnt != max_keep) {
                const float v = pi[m][tid];
                int rank = 0;
#pragma unroll 16
                for (int j = 0; j < Lp; ++j) {
                    const float o = pi[m][j];
                    rank += (sel[m][j] == mine) && ((o > v) || (o == v && j < tid));
                }
                if (cnt > max_keep) on = mine && rank < max_keep;
                else on = mine || rank < max_keep - cnt;
            }
            __syncthreads();
            if (tid < Lp) sel[m][tid] = on;
        }
        __syncthreads();
    }
    for (int i = tid; i < 3 * Lp; i += 256) {
        const int m = i / Lp, j = i - m * Lp;
        const int on = sel[m][j];
        if (mask_f) mask_f[((size_t)m * B + b) * Lp + j] = on ? 1.f : 0.f;
        if (mask_u8) mask_u8[((size_t)m * B + b) * Lp + j] = (unsigned char)on;
    }
}

// sel[(b*3+m)*Lp + j][:] = mask[m][b][j] ? bf16(tokens[s=m*B+b][1+j][:]) : 0     (useA.py:318-320,383)
// also emits the stacked CLS queries cls_b[(b*3+m)][:] (bf16) and cls_f (f32) used by the interaction block
__global__ __launch_bounds__(256) void sim_gather_kernel(const float* __restrict__ tokens, const float* __restrict__ mask_f,
                                                         int L, int B, bf16_t* __restrict__ sel, bf16_t* __restrict__ cls_b,
                                                         float* __restrict__ cls_f, int dt) {
    const int Lp = L - 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nsel = 3 * B * Lp;
    if (row < nsel) {
        const int j = row % Lp, bm = row / Lp, m = bm % 3, b = bm / 3;
        const int s = m * B + b;
        const bool on = mask_f[((size_t)m * B + b) * Lp + j] != 0.f;
        uint4 o = make_uint4(0, 0, 0, 0);
        if (on) {
            const float* t = tokens + ((size_t)s * L + 1 + j) * SIM_D + lane * 8;
            const float4 a = *(const float4*)t, c = *(const float4*)(t + 4);
            o = make_uint4(pack2_16(a.x, a.y, dt), pack2_16(a.z, a.w, dt), pack2_16(c.x, c.y, dt), pack2_16(c.z, c.w, dt));
        }
        *(uint4*)(sel + (size_t)row * SIM_D + lane * 8) = o;
    } else if (row < nsel + 3 * B) {
        const int bm = row - nsel, m = bm % 3, b = bm / 3;
        const float* t = tokens + (size_t)(m * B + b) * L * SIM_D + lane * 8;
        const float4 a = *(const float4*)t, c = *(const float4*)(t + 4);
        *(uint4*)(cls_b + (size_t)bm * SIM_D + lane * 8) = make_uint4(pack2_16(a.x, a.y, dt), pack2_16(a.z, a.w, dt), pack2_16(c.x, c.y, dt), pack2_16(c.z, c.w, dt));
        *(float4*)(cls_f + (size_t)bm * SIM_D + lane * 8) = a;
        *(float4*)(cls_f + (size_t)bm * SIM_D + lane * 8 + 4) = c;
    }
}

// backward of the gather: dtokens[s][1+j][:] += mask * dsel[(b*3+m)*Lp+j][:]; dtokens[s][0][:] += dcls[(b*3+m)][:]
__global__ __launch_bounds__(256) void sim_gather_bwd_kernel(const bf16_t* __restrict__ dsel, const float* __restrict__ dcls,
                                                             const float* __restrict__ mask_f, int L, int B,
                                                             float* __restrict__ dtokens, int dt) {
    const int Lp = L - 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nsel = 3 * B * Lp;
    if (row < nsel) {
        const int j = row % Lp, bm = row / Lp, m = bm % 3, b = bm / 3;
        if (mask_f[((size_t)m * B + b) * Lp + j] == 0.f) return;
        const uint4 u = *(const uint4*)(dsel + (size_t)row * SIM_D + lane * 8);
        float* t = dtokens + ((size_t)(m * B + b) * L + 1 + j) * SIM_D + lane * 8;
        float4 a = *(float4*)t, c = *(float4*)(t + 4);
        a.x += cvt16f((bf16_t)(u.x & 0xffff), dt); a.y += cvt16f((bf16_t)(u.x >> 16), dt);
        a.z += cvt16f((bf16_t)(u.y & 0xffff), dt); a.w += cvt16f((bf16_t)(u.y >> 16), dt);
        c.x += cvt16f((bf16_t)(u.z & 0xffff), dt); c.y += cvt16f((bf16_t)(u.z >> 16), dt);
        c.z += cvt16f((bf16_t)(u.w & 0xffff), dt); c.w += cvt16f((bf16_t)(u.w >> 16), dt);
        *(float4*)t = a;
        *(float4*)(t + 4) = c;
    } else if (row < nsel + 3 * B) {
        const int bm = row - nsel, m = bm % 3, b = bm / 3;
        float* t = dtokens + (size_t)(m * B + b) * L * SIM_D + lane * 8;
        const float* d = dcls + (size_t)bm * SIM_D + lane * 8;
        float4 a = *(float4*)t, c = *(float4*)(t + 4);
        const float4 da = *(const float4*)d, dc = *(const float4*)(d + 4);
        a.x += da.x; a.y += da.y; a.z += da.z; a.w += da.w;
        c.x += dc.x; c.y += dc.y; c.z += dc.z; c.w += dc.w;
        *(float4*)t = a;
        *(float4*)(t + 4) = c;
    }
}

int sig_launch_sim_select(const float* tokens, int B, int L, const float* Wq, const float* bq, const float* Wk,
                          const float* bk, int topk, int max_keep, float* qprime, float* cconst, float* intra, float* inter,
                          float* mask_f, unsigned char* mask_u8, hipStream_t st) {
    SIG_CHECK_ARG(tokens && Wq && bq && Wk && bk && qprime && cconst && intra && inter && (mask_f || mask_u8), "sim_select: null pointer");
    SIG_CHECK_ARG(B > 0 && L > 1 && L - 1 <= 128 && topk > 0, "sim_select: needs 1 <= L-1 <= 128 patches (got %d) and topk > 0", L - 1);
    SIG_CHECK_ARG(max_keep >= 0 && max_keep <= L - 1, "sim_select: max_keep=%d outside 0..Lp (0 = no exact keep ratio)", max_keep);
    const int Lp = L - 1;
    const int k1 = topk < Lp ? topk : Lp, k2 = 2 * topk < 2 * Lp ? 2 * topk : 2 * Lp;
    hipLaunchKernelGGL(sim_qprime_kernel, dim3(3 * B), dim3(512), 0, st, tokens, L, B, Wq, bq, Wk, bk, qprime, cconst);
    SIG_CHECK_LAUNCH("sim_qprime");
    hipLaunchKernelGGL(sim_scores_kernel, dim3(sig_ceil_div(3 * B * Lp, 4)), dim3(256), 0, st, tokens, L, B, qprime, cconst, intra, inter);
    SIG_CHECK_LAUNCH("sim_scores");
    hipLaunchKernelGGL(sim_select_kernel, dim3(B), dim3(256), 0, st, intra, inter, B, Lp, k1, k2, max_keep, mask_f, mask_u8);
    SIG_CHECK_LAUNCH("sim_select");
    return 0;
}

int sig_launch_sim_gather(const float* tokens, const float* mask_f, int B, int L, bf16_t* sel, bf16_t* cls_b, float* cls_f,
                          int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "sim_gather");
    SIG_CHECK_ARG(tokens && mask_f && sel && cls_b && cls_f && B > 0 && L > 1, "sim_gather: bad arguments");
    const int rows = 3 * B * (L - 1) + 3 * B;
    hipLaunchKernelGGL(sim_gather_kernel, dim3(sig_ceil_div(rows, 4)), dim3(256), 0, st, tokens, mask_f, L, B, sel, cls_b, cls_f, dt);
    SIG_CHECK_LAUNCH("sim_gather");
    return 0;
}

int sig_launch_sim_gather_bwd(const bf16_t* dsel, const float* dcls, const float* mask_f, int B, int L, float* dtokens,
                              int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "sim_gather_bwd");
    SIG_CHECK_ARG(dsel && dcls && mask_f && dtokens && B > 0 && L > 1, "sim_gather_bwd: bad arguments");
    const int rows = 3 * B * (L - 1) + 3 * B;
    hipLaunchKernelGGL(sim_gather_bwd_kernel, dim3(sig_ceil_div(rows, 4)), dim3(256), 0, st, dsel, dcls, mask_f, L, B, dtokens, dt);
    SIG_CHECK_LAUNCH("sim_gather_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// cross-attention core: 3 queries x NK keys x 8 heads of 64 (useA.py:388), one workgroup per (sample, head).
//   q   f32  [B*3, 512]            (already projected, bias added)
//   kv  bf16 [B*NK, 1024]          (k | v, bias added)
//   out bf16 [B*3, 512]            heads re-concatenated
//   probs f32 [B, 24, NK]          kept for backward (row = head*3 + query)
// ------------------------------------------------------------------------------------------------
#define XA_H 8
#define XA_MAXK 384
// One workgroup per (sample, head): 512 blocks at B = 64 instead of 64, and the loops over the keys are split over the eight waves
// (wave w takes keys w, w + 8, ...; the eight partial sums are added in wave order: deterministic).  With one workgroup per sample
// every thread walked all NK keys alone: 48 us forward / 96 us backward for 50 MB.
__global__ __launch_bounds__(512) void xattn_fwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv, int NK,
                                                        bf16_t* __restrict__ out, float* __restrict__ probs, int dt) {
    __shared__ float sq[3][64];
    __shared__ float sp[3][XA_MAXK];
    __shared__ float part[3][8][64];
    const int b = blockIdx.x / XA_H, h = blockIdx.x - b * XA_H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 192) sq[tid >> 6][lane] = q[((size_t)b * 3 + (tid >> 6)) * SIM_D + h * 64 + lane];
    __syncthreads();
    // scores: thread = key
    if (tid < NK) {
        const bf16_t* krow = kv + ((size_t)b * NK + tid) * 1024 + h * 64;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = *(const uint4*)(krow + c * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float k0 = cvt16f((bf16_t)(w[e] & 0xffff), dt), k1 = cvt16f((bf16_t)(w[e] >> 16), dt);
                const int d = c * 8 + e * 2;
                a0 += k0 * sq[0][d] + k1 * sq[0][d + 1];
                a1 += k0 * sq[1][d] + k1 * sq[1][d + 1];
                a2 += k0 * sq[2][d] + k1 * sq[2][d + 1];
            }
        }
        sp[0][tid] = a0 * 0.125f;
        sp[1][tid] = a1 * 0.125f;
        sp[2][tid] = a2 * 0.125f;
    }
    __syncthreads();
    // softmax of the head's three query rows: one wave each
    if (wave < 3) {
        const int r = wave;
        float mx = -INFINITY;
        for (int j = lane; j < NK; j += 64) mx = fmaxf(mx, sp[r][j]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int j = lane; j < NK; j += 64) {
            const float e = __expf(sp[r][j] - mx);
            sp[r][j] = e;
            sm += e;
        }
        sm = 1.0f / wave_sum(sm);
        for (int j = lane; j < NK; j += 64) {
            const float pv = sp[r][j] * sm;
            sp[r][j] = pv;
            if (probs) probs[((size_t)b * 24 + h * 3 + r) * NK + j] = pv;
        }
    }
    __syncthreads();
    // out[qi][c] = sum_j p[qi][j] v[j][c]: lane = column of the head, wave = key group
    {
        const bf16_t* vcol = kv + (size_t)b * NK * 1024 + 512 + h * 64 + lane;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 4
        for (int j = wave; j < NK; j += 8) {
            const float v = cvt16f(vcol[(size_t)j * 1024], dt);
            a0 += sp[0][j] * v;
            a1 += sp[1][j] * v;
            a2 += sp[2][j] * v;
        }
        part[0][wave][lane] = a0; part[1][wave][lane] = a1; part[2][wave][lane] = a2;
    }
    __syncthreads();
    if (tid < 192) {
        const int qi = tid >> 6;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += part[qi][w][lane];
        out[((size_t)b * 3 + qi) * SIM_D + h * 64 + lane] = f2cvt16(t, dt);
    }
}

// backward: given dout f32 [B*3,512] -> dq f32 [B*3,512], dkv bf16 [B*NK,1024]
__global__ __launch_bounds__(512) void xattn_bwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv,
                                                        const float* __restrict__ probs, const float* __restrict__ dout, int NK,
                                                        float* __restrict__ dq, bf16_t* __restrict__ dkv, int dt) {
    __shared__ float sq[3][64], sdo[3][64];
    __shared__ float sp[3][XA_MAXK];   // probs, then dS (scaled)
    __shared__ float sdelta[3];
    __shared__ float spart[3][8];
    __shared__ float part[3][8][64];
    const int b = blockIdx.x / XA_H, h = blockIdx.x - b * XA_H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 192) {
        sq[tid >> 6][lane] = q[((size_t)b * 3 + (tid >> 6)) * SIM_D + h * 64 + lane];
        sdo[tid >> 6][lane] = dout[((size_t)b * 3 + (tid >> 6)) * SIM_D + h * 64 + lane];
    }
    for (int i = tid; i < 3 * NK; i += 512) sp[i / NK][i % NK] = probs[((size_t)b * 24 + h * 3) * NK + i];
    __syncthreads();
    // dV[j][c] = sum_qi p[qi][j] dO[qi][c]   (lane = column of the head, wave = key group) -- written straight to dkv
    {
        const float d0 = sdo[0][lane], d1 = sdo[1][lane], d2 = sdo[2][lane];
        bf16_t* dv = dkv + (size_t)b * NK * 1024 + 512 + h * 64 + lane;
#pragma unroll 4
        for (int j = wave; j < NK; j += 8) dv[(size_t)j * 1024] = f2cvt16(sp[0][j] * d0 + sp[1][j] * d1 + sp[2][j] * d2, dt);
    }
    // dP[qi][j] = dO[qi] . v[j] over the head's 64 dims; delta[qi] = sum_j p dP; dS = p (dP - delta) / 8
    float dpr[3] = {0.f, 0.f, 0.f};
    if (tid < NK) {
        const bf16_t* vrow = kv + ((size_t)b * NK + tid) * 1024 + 512 + h * 64;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = *(const uint4*)(vrow + c * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = cvt16f((bf16_t)(w[e] & 0xffff), dt), v1 = cvt16f((bf16_t)(w[e] >> 16), dt);
                const int d = c * 8 + e * 2;
                a0 += v0 * sdo[0][d] + v1 * sdo[0][d + 1];
                a1 += v0 * sdo[1][d] + v1 * sdo[1][d + 1];
                a2 += v0 * sdo[2][d] + v1 * sdo[2][d + 1];
            }
        }
        dpr[0] = a0; dpr[1] = a1; dpr[2] = a2;
    }
    // delta: per-wave partials, added in wave order (deterministic)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float v = tid < NK ? sp[r][tid] * dpr[r] : 0.f;
        v = wave_sum(v);
        if (lane == 0) spart[r][wave] = v;
    }
    __syncthreads();      // (also: every dV store above has read its probabilities)
    if (tid < 3) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += spart[tid][w];
        sdelta[tid] = t;
    }
    __syncthreads();
    if (tid < NK) {
#pragma unroll
        for (int r = 0; r < 3; ++r) sp[r][tid] = sp[r][tid] * (dpr[r] - sdelta[r]) * 0.125f;
    }
    __syncthreads();
    // dK[j][c] = sum_qi dS[qi][j] q[qi][c];  dq[qi][c] = sum_j dS[qi][j] k[j][c]  (lane = column, wave = key group)
    {
        const float q0 = sq[0][lane], q1 = sq[1][lane], q2 = sq[2][lane];
        bf16_t* dk = dkv + (size_t)b * NK * 1024 + h * 64 + lane;
        const bf16_t* kcol = kv + (size_t)b * NK * 1024 + h * 64 + lane;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 4
        for (int j = wave; j < NK; j += 8) {
            const float s0 = sp[0][j], s1 = sp[1][j], s2 = sp[2][j];
            dk[(size_t)j * 1024] = f2cvt16(s0 * q0 + s1 * q1 + s2 * q2, dt);
            const float k = cvt16f(kcol[(size_t)j * 1024], dt);
            a0 += s0 * k;
            a1 += s1 * k;
            a2 += s2 * k;
        }
        part[0][wave][lane] = a0; part[1][wave][lane] = a1; part[2][wave][lane] = a2;
    }
    __syncthreads();
    if (tid < 192) {
        const int qi = tid >> 6;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += part[qi][w][lane];
        dq[((size_t)b * 3 + qi) * SIM_D + h * 64 + lane] = t;
    }
}

int sig_launch_xattn_fwd(const float* q, const bf16_t* kv, int B, int NK, bf16_t* out, float* probs, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "xattn_fwd");
    SIG_CHECK_ARG(q && kv && out && B > 0 && NK > 0 && NK <= XA_MAXK, "xattn_fwd: bad arguments (keys %d, max %d)", NK, XA_MAXK);
    hipLaunchKernelGGL(xattn_fwd_kernel, dim3(B * XA_H), dim3(512), 0, st, q, kv, NK, out, probs, dt);
    SIG_CHECK_LAUNCH("xattn_fwd");
    return 0;
}
int sig_launch_xattn_bwd(const float* q, const bf16_t* kv, const float* probs, const float* dout, int B, int NK, float* dq,
                         bf16_t* dkv, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "xattn_bwd");
    SIG_CHECK_ARG(q && kv && probs && dout && dq && dkv && B > 0 && NK > 0 && NK <= XA_MAXK, "xattn_bwd: bad arguments");
    hipLaunchKernelGGL(xattn_bwd_kernel, dim3(B * XA_H), dim3(512), 0, st, q, kv, probs, dout, NK, dq, dkv, dt);
    SIG_CHECK_LAUNCH("xattn_bwd");
    return 0;
}
