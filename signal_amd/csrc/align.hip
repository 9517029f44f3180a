// GAM (Gramian-volume contrastive loss) and LAM (deformable-offset sampling + MSE) of AlignmentM
// (modeling/AddModule/useB.py, DAS.py, utils/volume.py).  fp32 throughout except the two 1x1-conv GEMMs of
// each DAS block, which run on the bf16 MFMA GEMM (gemm_bf16.hip) from the host composite in vit.hip.
//
// tokens f32 [S*L, 512]: row (m*B+b)*L is the CLS of modality m / sample b, rows +1.. its Lp = h*w patches.
#include <mutex>

#include "sig_common.h"
#include "sig_kernels.h"

#define AL_D 512

__device__ __forceinline__ float block_sum512(float v, float* red) {   // 512 threads = 8 waves
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// ================================================================================================
// GAM
// ================================================================================================
// mean over the patch tokens (NOT the CLS, useB.py:92-94) + F.normalize (eps 1e-12, useB.py:98-100)
__global__ __launch_bounds__(512) void gam_pool_kernel(const float* __restrict__ tokens, int L, float* __restrict__ fh,
                                                       float* __restrict__ nrm) {
    __shared__ float red[8];
    const int s = blockIdx.x, c = threadIdx.x, Lp = L - 1;
    const float* t = tokens + ((size_t)s * L + 1) * AL_D + c;
    float acc = 0.f;
    for (int j = 0; j < Lp; ++j) acc += t[(size_t)j * AL_D];
    acc /= (float)Lp;
    const float n = fmaxf(sqrtf(block_sum512(acc * acc, red)), 1e-12f);
    fh[(size_t)s * AL_D + c] = acc / n;
    if (c == 0) nrm[s] = n;
}

// pairwise dots: wave per (i,j) -> lv = r_i.n_j, la = r_i.t_j ; then per-j (vv,va,aa) and per-i ll   (volume.py:35-44)
__global__ __launch_bounds__(256) void gam_dots_kernel(const float* __restrict__ fh, int B, float* __restrict__ lv,
                                                       float* __restrict__ la, float* __restrict__ vec) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* R = fh;
    const float* N = fh + (size_t)B * AL_D;
    const float* T = fh + (size_t)2 * B * AL_D;
    auto dot = [&](const float* a, const float* b) {
        const float4 a0 = *(const float4*)(a + lane * 8), a1 = *(const float4*)(a + lane * 8 + 4);
        const float4 b0 = *(const float4*)(b + lane * 8), b1 = *(const float4*)(b + lane * 8 + 4);
        return wave_sum(a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w);
    };
    if (w < B * B) {
        const int i = w / B, j = w - i * B;
        const float d1 = dot(R + (size_t)i * AL_D, N + (size_t)j * AL_D);
        const float d2 = dot(R + (size_t)i * AL_D, T + (size_t)j * AL_D);
        if (lane == 0) { lv[w] = d1; la[w] = d2; }
    } else if (w < B * B + B) {
        const int j = w - B * B;
        const float vv = dot(N + (size_t)j * AL_D, N + (size_t)j * AL_D);
        const float va = dot(N + (size_t)j * AL_D, T + (size_t)j * AL_D);
        const float aa = dot(T + (size_t)j * AL_D, T + (size_t)j * AL_D);
        const float ll = dot(R + (size_t)j * AL_D, R + (size_t)j * AL_D);
        if (lane == 0) { vec[j] = ll; vec[B + j] = vv; vec[2 * B + j] = va; vec[3 * B + j] = aa; }
    }
}

// one workgroup: V = sqrt|det Gram|, A = -V/temp, loss = (CE_ls(A) + CE_ls(A^T))/2 with eps = 0.1 (useB.py:106-124),
// and the backward coefficients of the loss w.r.t. every Gram entry.
//   coef layout: g_lv[B*B], g_la[B*B], g_ll[B], g_vv[B], g_va[B], g_aa[B], dtemp[1]
__global__ __launch_bounds__(1024) void gam_loss_kernel(const float* __restrict__ lv, const float* __restrict__ la,
                                                        const float* __restrict__ vec, const float* __restrict__ temp, int B,
                                                        float* __restrict__ loss, float* __restrict__ coef) {
    extern __shared__ float sm[];
    float* A = sm;                 // [B*B]
    float* rlse = sm + B * B;      // [B]
    float* clse = rlse + B;        // [B]
    float* racc = clse + B;        // [6*B] accumulators g_ll, g_vv, g_va, g_aa (+ spare)
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const float tp = temp[0];
    const float eps = 0.1f;
    const float* ll = vec; const float* vv = vec + B; const float* va = vec + 2 * B; const float* aa = vec + 3 * B;
    for (int w = tid; w < B * B; w += 1024) {
        const int i = w / B, j = w - i * B;
        const float a = lv[w], b = la[w];
        const float det = ll[i] * (vv[j] * aa[j] - va[j] * va[j]) - a * (a * aa[j] - va[j] * b) + b * (a * va[j] - vv[j] * b);
        A[w] = -sqrtf(fabsf(det)) / tp;
    }
    for (int k = tid; k < 6 * B; k += 1024) racc[k] = 0.f;
    __syncthreads();
    if (tid < B) {            // row LSE
        float mx = -INFINITY;
        for (int j = 0; j < B; ++j) mx = fmaxf(mx, A[tid * B + j]);
        float s = 0.f;
        for (int j = 0; j < B; ++j) s += expf(A[tid * B + j] - mx);
        rlse[tid] = mx + logf(s);
    } else if (tid >= 512 && tid < 512 + B) {   // column LSE
        const int j = tid - 512;
        float mx = -INFINITY;
        for (int i = 0; i < B; ++i) mx = fmaxf(mx, A[i * B + j]);
        float s = 0.f;
        for (int i = 0; i < B; ++i) s += expf(A[i * B + j] - mx);
        clse[j] = mx + logf(s);
    }
    __syncthreads();
    // loss = 1/(2B) sum_i [ (1-eps)(-logp_ii) + eps * mean_k(-logp_ik) ]  for rows and for columns
    float part = 0.f, dtp = 0.f;
    for (int w = tid; w < B * B; w += 1024) {
        const int i = w / B, j = w - i * B;
        const float a = A[w];
        const float lpr = a - rlse[i], lpc = a - clse[j];
        const float tgt = (i == j ? 1.f - eps : 0.f) + eps / (float)B;
        part += -tgt * (lpr + lpc);
        // dL/dA = (1/2B) [ (softmax_row - tgt) + (softmax_col - tgt) ]
        const float dA = (expf(lpr) + expf(lpc) - 2.f * tgt) / (2.f * (float)B);
        const float V = -a * tp;
        const float dV = -dA / tp;
        dtp += dA * V / (tp * tp);
        const float x = lv[w], y = la[w];
        const float det = ll[i] * (vv[j] * aa[j] - va[j] * va[j]) - x * (x * aa[j] - va[j] * y) + y * (x * va[j] - vv[j] * y);
        // dV/ddet = sign(det) / (2 sqrt|det|); at det == 0 the reference yields NaN: zero sub-gradient here
        const float G = V > 0.f ? dV * (det >= 0.f ? 0.5f : -0.5f) / V : 0.f;
        coef[w] = G * (-2.f * x * aa[j] + 2.f * va[j] * y);             // g_lv
        coef[B * B + w] = G * (2.f * x * va[j] - 2.f * y * vv[j]);      // g_la
        A[w] = G;                                                       // A is dead from here on: keep G for the row / column sums
    }
    __syncthreads();
    // g_ll[i] = sum_j G_ij (..), g_vv / g_va / g_aa[j] = sum_i G_ij (..): one thread per output, fixed summation order
    // (LDS float atomics here made the token gradient differ from run to run in the last bit, and every 16-bit rounding of
    // the backward signal downstream amplified that to 1e-3 of the whole gradient)
    for (int k = tid; k < 4 * B; k += 1024) {
        const int which = k / B, idx = k - which * B;
        float acc = 0.f;
        for (int o = 0; o < B; ++o) {
            const int i = which == 0 ? idx : o, j = which == 0 ? o : idx;
            const float G = A[i * B + j], x = lv[i * B + j], y = la[i * B + j];
            acc += which == 0 ? G * (vv[j] * aa[j] - va[j] * va[j])
                 : which == 1 ? G * (ll[i] * aa[j] - y * y)
                 : which == 2 ? G * (-2.f * ll[i] * va[j] + 2.f * x * y)
                              : G * (ll[i] * vv[j] - x * x);
        }
        racc[k] = acc;
    }
    __syncthreads();
    part = wave_sum(part);
    dtp = wave_sum(dtp);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[k];
        loss[0] = s / (2.f * (float)B);
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = dtp;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[k];
        coef[2 * B * B + 4 * B] = s;
    }
    for (int k = tid; k < 4 * B; k += 1024) coef[2 * B * B + k] = racc[k];
}

// d fh -> d pooled (normalize backward) -> d tokens (mean backward, accumulated); one workgroup per (m,b)
__global__ __launch_bounds__(512) void gam_bwd_kernel(const float* __restrict__ fh, const float* __restrict__ nrm,
                                                      const float* __restrict__ coef, const float* __restrict__ dloss, int B,
                                                      int L, float* __restrict__ dtokens, float* __restrict__ dtemp) {
    __shared__ float red[8];
    const int s = blockIdx.x, c = threadIdx.x, m = s / B, b = s - m * B, Lp = L - 1;
    const float* R = fh; const float* N = fh + (size_t)B * AL_D; const float* T = fh + (size_t)2 * B * AL_D;
    const float* g_lv = coef; const float* g_la = coef + B * B; const float* g_ll = coef + 2 * B * B;
    const float* g_vv = g_ll + B; const float* g_va = g_ll + 2 * B; const float* g_aa = g_ll + 3 * B;
    float d = 0.f;
    if (m == 0) {
        for (int j = 0; j < B; ++j) d += g_lv[b * B + j] * N[(size_t)j * AL_D + c] + g_la[b * B + j] * T[(size_t)j * AL_D + c];
        d += 2.f * g_ll[b] * R[(size_t)b * AL_D + c];
    } else if (m == 1) {
        for (int i = 0; i < B; ++i) d += g_lv[i * B + b] * R[(size_t)i * AL_D + c];
        d += 2.f * g_vv[b] * N[(size_t)b * AL_D + c] + g_va[b] * T[(size_t)b * AL_D + c];
    } else {
        for (int i = 0; i < B; ++i) d += g_la[i * B + b] * R[(size_t)i * AL_D + c];
        d += 2.f * g_aa[b] * T[(size_t)b * AL_D + c] + g_va[b] * N[(size_t)b * AL_D + c];
    }
    const float f = fh[(size_t)s * AL_D + c];
    const float dotv = block_sum512(f * d, red);
    const float n = nrm[s];
    const float up = dloss[0];
    // normalize backward (norm above the eps clamp; below it F.normalize divides by the constant eps)
    const float df = (n > 1e-12f ? (d - f * dotv) / n : d / 1e-12f) * up / (float)Lp;
    float* t = dtokens + ((size_t)s * L + 1) * AL_D + c;
    for (int j = 0; j < Lp; ++j) t[(size_t)j * AL_D] += df;
    if (s == 0 && c == 0 && dtemp) atomicAdd(dtemp, coef[2 * B * B + 4 * B] * up);
}

int sig_launch_gam_fwd(const float* tokens, int B, int L, const float* temp, float* fh, float* nrm, float* lv, float* la,
                       float* vec, float* coef, float* loss, hipStream_t st) {
    SIG_CHECK_ARG(tokens && temp && fh && nrm && lv && la && vec && coef && loss, "gam_fwd: null pointer");
    SIG_CHECK_ARG(B > 0 && B <= 128 && L > 1, "gam_fwd: per-GPU batch %d must be in 1..128", B);
    hipLaunchKernelGGL(gam_pool_kernel, dim3(3 * B), dim3(512), 0, st, tokens, L, fh, nrm);
    SIG_CHECK_LAUNCH("gam_pool");
    hipLaunchKernelGGL(gam_dots_kernel, dim3(sig_ceil_div(B * B + B, 4)), dim3(256), 0, st, fh, B, lv, la, vec);
    SIG_CHECK_LAUNCH("gam_dots");
    const int lds = (B * B + 8 * B) * 4;
    static std::once_flag attr_done;
    std::call_once(attr_done, [] {
        (void)hipFuncSetAttribute((const void*)gam_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (128 * 128 + 8 * 128) * 4);
        });
    hipLaunchKernelGGL(gam_loss_kernel, dim3(1), dim3(1024), lds, st, lv, la, vec, temp, B, loss, coef);
    SIG_CHECK_LAUNCH("gam_loss");
    return 0;
}
int sig_launch_gam_bwd(const float* fh, const float* nrm, const float* coef, const float* dloss, int B, int L, float* dtokens,
                       float* dtemp, hipStream_t st) {
    SIG_CHECK_ARG(fh && nrm && coef && dloss && dtokens, "gam_bwd: null pointer");
    hipLaunchKernelGGL(gam_bwd_kernel, dim3(3 * B), dim3(512), 0, st, fh, nrm, coef, dloss, B, L, dtokens, dtemp);
    SIG_CHECK_LAUNCH("gam_bwd");
    return 0;
}

// ================================================================================================
// LAM / DAS
// ================================================================================================
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
}

// xb[m][b*Lp + j][:] = bf16(patch j of (m,b));  modality stride = xstride rows
__global__ __launch_bounds__(256) void lam_gather_kernel(const float* __restrict__ tokens, int L, int B, bf16_t* __restrict__ xb,
                                                         size_t xstride, int dt) {
    const int Lp = L - 1, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 3 * B * Lp) return;
    const int s = row / Lp, j = row - s * Lp, m = s / B, b = s - m * B;
    const float* t = tokens + ((size_t)s * L + 1 + j) * AL_D + lane * 8;
    const float4 a = *(const float4*)t, c = *(const float4*)(t + 4);
    *(uint4*)(xb + ((size_t)m * xstride + (size_t)b * Lp + j) * AL_D + lane * 8) =
        make_uint4(pack2_16(a.x, a.y, dt), pack2_16(a.z, a.w, dt), pack2_16(c.x, c.y, dt), pack2_16(c.z, c.w, dt));
}

// dtokens[patch rows of modality m] += src[(b*Lp + j)][:]   (f32)
__global__ __launch_bounds__(256) void lam_scatter_add_kernel(const float* __restrict__ src, int m, int L, int B,
                                                              float* __restrict__ dtokens) {
    const int Lp = L - 1, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Lp) return;
    const int b = row / Lp, j = row - b * Lp;
    float* t = dtokens + ((size_t)(m * B + b) * L + 1 + j) * AL_D + lane * 8;
    const float* sp = src + (size_t)row * AL_D + lane * 8;
    float4 a = *(float4*)t, c = *(float4*)(t + 4);
    const float4 da = *(const float4*)sp, dc = *(const float4*)(sp + 4);
    a.x += da.x; a.y += da.y; a.z += da.z; a.w += da.w;
    c.x += dc.x; c.y += dc.y; c.z += dc.z; c.w += dc.w;
    *(float4*)t = a;
    *(float4*)(t + 4) = c;
}

struct LamGeom { int h, w, Hk, Wk; };

// tail of DA_sample.forward (DAS.py:136-163) for one (sample, modality): depthwise 4x4/4 conv + GELU, 1x1 conv to the
// single offset channel, tanh * range * 2 (the same scalar drives y and x), reference points, clamp, bilinear sample.
//   a1     bf16 [B*Lp, 512]   GELU(conv_offset.0(proj_q(x)))
//   a2pre  f32  [B, P, 512]   depthwise pre-activation (kept for backward), P = Hk*Wk
//   offs   f32  [B, P, 3]     raw offset o, p_y, p_x
//   samp   f32  [B, P, 512]   sampled feature
// (grid = (B, 3): the three modalities' tails are ONE launch -- 64 blocks per launch left three quarters of the chip idle)
__global__ __launch_bounds__(512) void lam_tail_fwd_kernel(const float* __restrict__ tokens, int L, int B, LamGeom g,
                                                           const bf16_t* __restrict__ a1_, size_t a_stride, SigLamTailPtrs tp,
                                                           float* __restrict__ a2pre_, float* __restrict__ offs_,
                                                           float* __restrict__ samp_, int dt) {
    __shared__ float red[8][8];
    __shared__ float pos[8][2];
    const int b = blockIdx.x, m = blockIdx.y, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int P = g.Hk * g.Wk, Lp = L - 1;
    const bf16_t* __restrict__ a1 = a1_ + (size_t)m * a_stride;
    const float* __restrict__ wd = tp.wd[m];
    const float* __restrict__ bd = tp.bd[m];
    const float* __restrict__ w4 = tp.w4[m];
    float* __restrict__ a2pre = a2pre_ + (size_t)m * B * P * AL_D;
    float* __restrict__ offs = offs_ + (size_t)m * B * P * 3;
    float* __restrict__ samp = samp_ + (size_t)m * B * P * AL_D;
    float wdc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wdc[k] = wd[c * 16 + k];
    const float bdc = bd[c], w4c = w4[c];
    float part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        part[p] = 0.f;
        if (p < P) {
            const int hk = p / g.Wk, wk = p - hk * g.Wk;
            float acc = bdc;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int tok = (4 * hk + dy) * g.w + 4 * wk + dx;
                    acc += wdc[dy * 4 + dx] * cvt16f(a1[((size_t)b * Lp + tok) * AL_D + c], dt);
                }
            a2pre[((size_t)b * P + p) * AL_D + c] = acc;
            part[p] = w4c * gelu_erf(acc);
        }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float v = wave_sum(part[p]);
        if (lane == 0) red[wave][p] = v;
    }
    __syncthreads();
    if (c < P) {
        float o = 0.f;
        for (int k = 0; k < 8; ++k) o += red[k][c];
        const int hk = c / g.Wk, wk = c - hk * g.Wk;
        const float t = tanhf(o);
        const float ry = ((float)hk + 0.5f) / ((float)g.Hk - 1.0f) * 2.0f - 1.0f;
        const float rx = ((float)wk + 0.5f) / ((float)g.Wk - 1.0f) * 2.0f - 1.0f;
        const float py = fminf(fmaxf(t * (1.0f / ((float)g.Hk - 1.0f)) * 2.0f + ry, -1.0f), 1.0f);
        const float px = fminf(fmaxf(t * (1.0f / ((float)g.Wk - 1.0f)) * 2.0f + rx, -1.0f), 1.0f);
        pos[c][0] = py; pos[c][1] = px;
        offs[((size_t)b * P + c) * 3 + 0] = o;
        offs[((size_t)b * P + c) * 3 + 1] = py;
        offs[((size_t)b * P + c) * 3 + 2] = px;
    }
    __syncthreads();
    const float* x = tokens + ((size_t)(m * B + b) * L + 1) * AL_D + c;
    for (int p = 0; p < P; ++p) {
        const float fy = (pos[p][0] + 1.0f) * 0.5f * (float)(g.h - 1), fx = (pos[p][1] + 1.0f) * 0.5f * (float)(g.w - 1);
        const float y0f = floorf(fy), x0f = floorf(fx);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float wy1 = fy - y0f, wx1 = fx - x0f, wy0 = 1.0f - wy1, wx0 = 1.0f - wx1;
        float v = 0.f;
        if (y0 >= 0 && y0 < g.h && x0 >= 0 && x0 < g.w) v += wy0 * wx0 * x[(size_t)(y0 * g.w + x0) * AL_D];
        if (y0 >= 0 && y0 < g.h && x0 + 1 >= 0 && x0 + 1 < g.w) v += wy0 * wx1 * x[(size_t)(y0 * g.w + x0 + 1) * AL_D];
        if (y0 + 1 >= 0 && y0 + 1 < g.h && x0 >= 0 && x0 < g.w) v += wy1 * wx0 * x[(size_t)((y0 + 1) * g.w + x0) * AL_D];
        if (y0 + 1 >= 0 && y0 + 1 < g.h && x0 + 1 >= 0 && x0 + 1 < g.w) v += wy1 * wx1 * x[(size_t)((y0 + 1) * g.w + x0 + 1) * AL_D];
        samp[((size_t)b * P + p) * AL_D + c] = v;
    }
}

// loss = (MSE(n,r) + MSE(t,r) + MSE(t,n)) / 3 over B*512*P elements (useB.py:161-165); samp = [3][B*P*512]
// Deterministic two-stage sum (atomics across blocks reordered it): LAM_LOSS_BLOCKS blocks park their partial sums behind
// loss[0] (the buffer holds 1 + LAM_LOSS_BLOCKS floats), then ONE thread-block adds them in block order.
#define LAM_LOSS_BLOCKS 64
__global__ __launch_bounds__(256) void lam_loss_kernel(const float* __restrict__ samp, size_t n, float* __restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float r = samp[i], nn = samp[n + i], t = samp[2 * n + i];
        acc += (nn - r) * (nn - r) + (t - r) * (t - r) + (t - nn) * (t - nn);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void lam_loss_final_kernel(float* __restrict__ loss, size_t n) {
    float v = loss[1 + threadIdx.x];                         // LAM_LOSS_BLOCKS == 64 partials, fixed shuffle tree
    v = wave_sum(v);
    if (threadIdx.x == 0) loss[0] = v / (3.0f * (float)n);
}

// backward of the tail for one (sample, modality).  Writes da1pre bf16 [B*Lp,512] (gradient of the conv_offset.0
// pre-activation, i.e. already multiplied by GELU'), accumulates dwd[512,16], dbd[512], dw4[512] and the bilinear
// gradient w.r.t. the sampled feature map into dtokens.
__global__ __launch_bounds__(512) void lam_tail_bwd_kernel(const float* __restrict__ tokens, int L, int B, LamGeom g,
                                                           const bf16_t* __restrict__ a1_, const bf16_t* __restrict__ a1pre_, size_t a_stride,
                                                           SigLamTailPtrs tp, const float* __restrict__ a2pre_, const float* __restrict__ offs_,
                                                           const float* __restrict__ samp_all, size_t nsamp,
                                                           const float* __restrict__ dloss, bf16_t* __restrict__ da1pre_,
                                                           float* __restrict__ dtokens, float* __restrict__ part_, int dt) {
    __shared__ float red[8][16];
    __shared__ float dofs[8];
    const int b = blockIdx.x, m = blockIdx.y, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int P = g.Hk * g.Wk, Lp = L - 1;
    const bf16_t* __restrict__ a1 = a1_ + (size_t)m * a_stride;
    const bf16_t* __restrict__ a1pre = a1pre_ + (size_t)m * a_stride;
    bf16_t* __restrict__ da1pre = da1pre_ + (size_t)m * a_stride;
    const float* __restrict__ wd = tp.wd[m];
    const float* __restrict__ w4 = tp.w4[m];
    float* __restrict__ dwd = tp.dwd[m];
    float* __restrict__ dbd = tp.dbd[m];
    float* __restrict__ dw4 = tp.dw4[m];
    const float* __restrict__ a2pre = a2pre_ + (size_t)m * nsamp;
    const float* __restrict__ offs = offs_ + (size_t)m * B * P * 3;
    float* __restrict__ part = part_ ? part_ + (size_t)m * B * (18 * AL_D) : nullptr;
    const float scale = dloss[0] * 2.0f / (3.0f * (float)nsamp);
    const float* x = tokens + ((size_t)(m * B + b) * L + 1) * AL_D + c;
    float* dx = dtokens + ((size_t)(m * B + b) * L + 1) * AL_D + c;
    const int m1 = (m + 1) % 3, m2 = (m + 2) % 3;
    float gp[16];   // per position: d loss / d p_y , d loss / d p_x   (partial over this channel)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        gp[2 * p] = gp[2 * p + 1] = 0.f;
        if (p < P) {
            const size_t e = ((size_t)b * P + p) * AL_D + c;
            const float sv = samp_all[m * nsamp + e];
            const float ds = scale * ((sv - samp_all[m1 * nsamp + e]) + (sv - samp_all[m2 * nsamp + e]));
            const float py = offs[((size_t)b * P + p) * 3 + 1], px = offs[((size_t)b * P + p) * 3 + 2];
            const float fy = (py + 1.0f) * 0.5f * (float)(g.h - 1), fx = (px + 1.0f) * 0.5f * (float)(g.w - 1);
            const float y0f = floorf(fy), x0f = floorf(fx);
            const int y0 = (int)y0f, x0 = (int)x0f;
            const float wy1 = fy - y0f, wx1 = fx - x0f, wy0 = 1.0f - wy1, wx0 = 1.0f - wx1;
            float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
            const bool i00 = y0 >= 0 && y0 < g.h && x0 >= 0 && x0 < g.w, i01 = y0 >= 0 && y0 < g.h && x0 + 1 >= 0 && x0 + 1 < g.w;
            const bool i10 = y0 + 1 >= 0 && y0 + 1 < g.h && x0 >= 0 && x0 < g.w, i11 = y0 + 1 >= 0 && y0 + 1 < g.h && x0 + 1 >= 0 && x0 + 1 < g.w;
            if (i00) { v00 = x[(size_t)(y0 * g.w + x0) * AL_D]; dx[(size_t)(y0 * g.w + x0) * AL_D] += ds * wy0 * wx0; }
            if (i01) { v01 = x[(size_t)(y0 * g.w + x0 + 1) * AL_D]; dx[(size_t)(y0 * g.w + x0 + 1) * AL_D] += ds * wy0 * wx1; }
            if (i10) { v10 = x[(size_t)((y0 + 1) * g.w + x0) * AL_D]; dx[(size_t)((y0 + 1) * g.w + x0) * AL_D] += ds * wy1 * wx0; }
            if (i11) { v11 = x[(size_t)((y0 + 1) * g.w + x0 + 1) * AL_D]; dx[(size_t)((y0 + 1) * g.w + x0 + 1) * AL_D] += ds * wy1 * wx1; }
            // d sample / d fy, d fx  -> d p (align_corners=True: f = (p+1)/2*(size-1))
            const float dfy = (v10 - v00) * wx0 + (v11 - v01) * wx1;
            const float dfx = (v01 - v00) * wy0 + (v11 - v10) * wy1;
            gp[2 * p] = ds * dfy * 0.5f * (float)(g.h - 1);
            gp[2 * p + 1] = ds * dfx * 0.5f * (float)(g.w - 1);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float v = wave_sum(gp[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (c < P) {
        float gy = 0.f, gx = 0.f;
        for (int k = 0; k < 8; ++k) { gy += red[k][2 * c]; gx += red[k][2 * c + 1]; }
        const float o = offs[((size_t)b * P + c) * 3 + 0];
        const float t = tanhf(o);
        const int hk = c / g.Wk, wk = c - hk * g.Wk;
        const float ry = ((float)hk + 0.5f) / ((float)g.Hk - 1.0f) * 2.0f - 1.0f;
        const float rx = ((float)wk + 0.5f) / ((float)g.Wk - 1.0f) * 2.0f - 1.0f;
        const float sy = (1.0f / ((float)g.Hk - 1.0f)) * 2.0f, sx = (1.0f / ((float)g.Wk - 1.0f)) * 2.0f;
        const float uy = t * sy + ry, ux = t * sx + rx;
        // torch.clamp passes the gradient where min <= x <= max
        float dt = 0.f;
        if (uy >= -1.0f && uy <= 1.0f) dt += gy * sy;
        if (ux >= -1.0f && ux <= 1.0f) dt += gx * sx;
        dofs[c] = dt * (1.0f - t * t);
    }
    __syncthreads();
    // o[p] = sum_c w4[c] GELU(a2pre[c][p])
    const float w4c = w4[c];
    float dw4c = 0.f, dbdc = 0.f;
    float dwdc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dwdc[k] = 0.f;
    for (int p = 0; p < P; ++p) {
        const float pre = a2pre[((size_t)b * P + p) * AL_D + c];
        const float d_o = dofs[p];
        dw4c += d_o * gelu_erf(pre);
        const float dpre = d_o * w4c * gelu_erf_grad(pre);
        dbdc += dpre;
        const int hk = p / g.Wk, wk = p - hk * g.Wk;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dxx = 0; dxx < 4; ++dxx) {
                const size_t row = (size_t)b * Lp + (4 * hk + dy) * g.w + 4 * wk + dxx;
                dwdc[dy * 4 + dxx] += dpre * cvt16f(a1[row * AL_D + c], dt);
                const float da1 = dpre * wd[c * 16 + dy * 4 + dxx];
                da1pre[row * AL_D + c] = f2cvt16(da1 * gelu_erf_grad(cvt16f(a1pre[row * AL_D + c], dt)), dt);
            }
    }
    // Parameter gradients: one partial row per sample, [B][18 * 512] = (dwd[512*16] | dbd[512] | dw4[512]), summed by
    // lam_tail_reduce_kernel.  Atomics straight into dwd/dbd/dw4 had all 64 sample blocks hit the same 9216 addresses
    // (the 14x-slower contention case of MI355X_MICROARCH 'Global float atomics'): 106 us per modality.
    if (part) {
        float* pr = part + (size_t)b * (18 * AL_D);
#pragma unroll
        for (int k = 0; k < 16; ++k) pr[c * 16 + k] = dwdc[k];
        pr[16 * AL_D + c] = dbdc;
        pr[17 * AL_D + c] = dw4c;
    } else {
        atomicAdd(dw4 + c, dw4c);
        atomicAdd(dbd + c, dbdc);
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(dwd + c * 16 + k, dwdc[k]);
    }
}

__global__ __launch_bounds__(256) void lam_tail_reduce_kernel(const float* __restrict__ part_, int B, SigLamTailPtrs tp) {
    const int i = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (i >= 18 * AL_D) return;
    const float* __restrict__ part = part_ + (size_t)m * B * (18 * AL_D);
    float* dwd = tp.dwd[m];
    float* dbd = tp.dbd[m];
    float* dw4 = tp.dw4[m];
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += part[(size_t)b * (18 * AL_D) + i];
    float* dst = i < 16 * AL_D ? dwd + i : (i < 17 * AL_D ? dbd + (i - 16 * AL_D) : dw4 + (i - 17 * AL_D));
    *dst += a;
}

int sig_launch_lam_gather(const float* tokens, int B, int L, bf16_t* xb, size_t xstride, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "lam_gather");
    SIG_CHECK_ARG(tokens && xb && B > 0 && L > 1, "lam_gather: bad arguments");
    hipLaunchKernelGGL(lam_gather_kernel, dim3(sig_ceil_div(3 * B * (L - 1), 4)), dim3(256), 0, st, tokens, L, B, xb, xstride, dt);
    SIG_CHECK_LAUNCH("lam_gather");
    return 0;
}
int sig_launch_lam_scatter_add(const float* src, int m, int B, int L, float* dtokens, hipStream_t st) {
    SIG_CHECK_ARG(src && dtokens, "lam_scatter_add: null pointer");
    hipLaunchKernelGGL(lam_scatter_add_kernel, dim3(sig_ceil_div(B * (L - 1), 4)), dim3(256), 0, st, src, m, L, B, dtokens);
    SIG_CHECK_LAUNCH("lam_scatter_add");
    return 0;
}
static int lam_geom(int h, int w, LamGeom* g) {
    SIG_CHECK_ARG(h % 4 == 0 && w % 4 == 0 && h >= 8 && w >= 8, "lam: grid %dx%d must be multiples of 4 and >= 8 (stride-4, 4x4 depthwise conv)", h, w);
    g->h = h; g->w = w; g->Hk = h / 4; g->Wk = w / 4;
    SIG_CHECK_ARG(g->Hk * g->Wk <= 8, "lam: at most 8 sampling points per map are supported (got %d)", g->Hk * g->Wk);
    return 0;
}
int sig_launch_lam_tail_fwd(const float* tokens, int B, int L, int h, int w, const bf16_t* a1, size_t a_stride, const SigLamTailPtrs& tp,
                            float* a2pre, float* offs, float* samp, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "lam_tail_fwd");
    LamGeom g;
    if (int rc = lam_geom(h, w, &g)) return rc;
    SIG_CHECK_ARG(h * w == L - 1, "lam: grid %dx%d does not match %d patches", h, w, L - 1);
    hipLaunchKernelGGL(lam_tail_fwd_kernel, dim3(B, 3), dim3(512), 0, st, tokens, L, B, g, a1, a_stride, tp, a2pre, offs, samp, dt);
    SIG_CHECK_LAUNCH("lam_tail_fwd");
    return 0;
}
int sig_launch_lam_loss(const float* samp, size_t n, float* loss, hipStream_t st) {
    SIG_CHECK_ARG(samp && loss && n > 0, "lam_loss: bad arguments");
    static_assert(LAM_LOSS_BLOCKS == 64, "lam_loss_final_kernel sums exactly one wave of partials");
    hipLaunchKernelGGL(lam_loss_kernel, dim3(LAM_LOSS_BLOCKS), dim3(256), 0, st, samp, n, loss);
    SIG_CHECK_LAUNCH("lam_loss");
    hipLaunchKernelGGL(lam_loss_final_kernel, dim3(1), dim3(64), 0, st, loss, n);
    SIG_CHECK_LAUNCH("lam_loss");
    return 0;
}
int sig_launch_lam_tail_bwd(const float* tokens, int B, int L, int h, int w, const bf16_t* a1, const bf16_t* a1pre, size_t a_stride,
                            const SigLamTailPtrs& tp, const float* a2pre, const float* offs, const float* samp_all, size_t nsamp,
                            const float* dloss, bf16_t* da1pre, float* dtokens, int dt, hipStream_t st, float* partials) {
    SIG_CHECK_DT(dt, "lam_tail_bwd");
    LamGeom g;
    if (int rc = lam_geom(h, w, &g)) return rc;
    // partials: caller scratch of >= 3 * B * 18 * 512 floats (nullptr -> contended atomics straight into the gradients)
    hipLaunchKernelGGL(lam_tail_bwd_kernel, dim3(B, 3), dim3(512), 0, st, tokens, L, B, g, a1, a1pre, a_stride, tp, a2pre, offs, samp_all,
                       nsamp, dloss, da1pre, dtokens, partials, dt);
    SIG_CHECK_LAUNCH("lam_tail_bwd");
    if (partials) {
        hipLaunchKernelGGL(lam_tail_reduce_kernel, dim3(sig_ceil_div(18 * AL_D, 256), 3), dim3(256), 0, st, partials, B, tp);
        SIG_CHECK_LAUNCH("lam_tail_reduce");
    }
    return 0;
}
