// Row-wise, HBM-bound kernels around the contractions: LayerNorm fwd/bwd, weight packing (f32 -> bf16,
// optional transpose), patch gather (im2col), token assembly (+CLS, camera, positional, ln_pre) and its
// backward, column sums (bias gradients).  One wavefront per token row, 16-B accesses, shuffle reductions.
#include <stdlib.h>

#include "sig_common.h"
#include "sig_kernels.h"

#define LN_MAXV 4  // float4 per lane -> D <= 1024

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: x f32 [M,D] -> y (bf16 and/or f32), mean/rstd [M]      (clip/model.py:154-160)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ yb,
                                                            float* __restrict__ yf, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int M, int D, float eps, int dt) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            v[it] = *(const float4*)(xr + c);
            s += v[it].x + v[it].y + v[it].z + v[it].w;
        }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float a = v[it].x - mu, b = v[it].y - mu, cc = v[it].z - mu, d = v[it].w - mu;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            float4 o;
            o.x = (v[it].x - mu) * rs * g.x + b.x;
            o.y = (v[it].y - mu) * rs * g.y + b.y;
            o.z = (v[it].z - mu) * rs * g.z + b.z;
            o.w = (v[it].w - mu) * rs * g.w + b.w;
            if (yf) *(float4*)(yf + (size_t)row * D + c) = o;
            if (yb) *(uint2*)(yb + (size_t)row * D + c) = make_uint2(pack2_16(o.x, o.y, dt), pack2_16(o.z, o.w, dt));
        }
    }
}

int sig_launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y_bf16, float* y_f32,
                             float* mean, float* rstd, int M, int D, float eps, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "layernorm_fwd");
    SIG_CHECK_ARG(M > 0 && D > 0 && (D & 3) == 0 && D <= 256 * LN_MAXV, "layernorm_fwd: D=%d unsupported (multiple of 4, <= 1024)", D);
    SIG_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32), "layernorm_fwd: null pointer");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(sig_ceil_div(M, 4)), dim3(256), 0, st, x, gamma, beta, y_bf16, y_f32,
                       mean, rstd, M, D, eps, dt);
    SIG_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  dx = dres + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));
// dgamma += sum_rows dy*xhat, dbeta += sum_rows dy  (register partials per wave, LDS across the 4 waves,
// one atomic per column per workgroup).
// ------------------------------------------------------------------------------------------------
// One pending column reduce: out_k[c] += sum over blocks b of part[b][k][c], k = 0 (dgamma), 1 (dbeta), 2 (dx column sums)
struct LnPrevReduce {
    const float* part;       // [nblocks][3][D] partial rows of the previous launch (nullptr: nothing pending)
    float* out[3];
    int nblocks, D, groups;  // groups = 3 * ceil(D / 64): one (array k, 64-column slice) per workgroup
};
// a workgroup of 64 * WPB threads adds up one (k, 64-column) group: the 16 row chunks of ln_bwd_reduce_kernel, two per wave,
// combined through LDS in chunk order.  `sm` = at least 16 * 64 floats of the caller's LDS, free before its own rows start.
template <int WPB>
__device__ __forceinline__ void ln_reduce_group(const LnPrevReduce& pv, int group, float* sm) {
#pragma clang fp reassociate(off)       // the ORDER of these additions is the contract (bit-identical to ln_bwd_reduce_kernel); -ffast-math would re-tree them
    const int slices = (pv.D + 63) >> 6, k = group / slices, c = (group - k * slices) * 64 + (threadIdx.x & 63);
    float* out = pv.out[k];
    if (out != nullptr) {           // (uniform)
        const int per = (pv.nblocks + 15) >> 4;
        for (int chunk = threadIdx.x >> 6; chunk < 16; chunk += WPB) {
            const int b0 = chunk * per, b1 = b0 + per < pv.nblocks ? b0 + per : pv.nblocks;
            float acc = 0.f;
            if (c < pv.D) {
                // every row of the chunk requested before any is added (same order of addition as ln_bwd_reduce_kernel: row by
                // row); with one load in flight per thread the 2 x 28 rows of a wave were 13 us of latency on the critical path
                const float* src = pv.part + (size_t)k * pv.D + c;
                const size_t rs = (size_t)3 * pv.D;
                for (int b = b0; b < b1; b += 32) {      // (a chunk is at most 32 rows at the 512-block cap: one batch)
                    float v[32];
#pragma unroll
                    for (int q = 0; q < 32; ++q) v[q] = b + q < b1 ? src[(size_t)(b + q) * rs] : 0.f;
#pragma unroll
                    for (int q = 0; q < 32; ++q)
                        if (b + q < b1) acc += v[q];
                }
            }
            sm[chunk * 64 + (threadIdx.x & 63)] = acc;
        }
        __syncthreads();
        if (threadIdx.x < 64 && c < pv.D) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += sm[q * 64 + threadIdx.x];
            out[c] += t;
        }
        __syncthreads();
    }
}

#ifndef LN_BWD_WPB
#define LN_BWD_WPB 8    // waves per block of layernorm_bwd_kernel (8 x 384 blocks: half the partial rows of 4 x 768 for the same waves)
#endif
template <bool DY_BF16, int NV, bool SUMX, int WPB>
__global__ __launch_bounds__(64 * WPB) void layernorm_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dxf, bf16_t* __restrict__ dxb,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int D,
                                                            float* __restrict__ dxsum, int dt, float* __restrict__ part, LnPrevReduce prev) {
    // NV = float4 per lane (D <= 256*NV): sized to the row so the per-lane accumulators stay small (more waves per SIMD).
    // Column sums (dgamma, dbeta, optional sum of dx): every wave parks its accumulators in its own LDS slice (plain
    // b128 stores), the block adds the slices and issues one global atomic per column.  That flush runs at the
    // memory-side atomic rate and the LDS slices cap residency, so the launcher uses FEW blocks with many rows each
    // (in-model, D = 768, M = 24768, atomics flush: 2048 blocks 98 us, 512 blocks 71 us, 768 blocks 67 us; LDS float atomics
    // instead of slices were slower at every block count.  Round 2: per-block partial rows + ln_bwd_reduce_kernel instead of the
    // atomics: 4 x 768 blocks 55.9 + 9.5 us, 8 x 384 blocks 57.6 + 5.9 us = 63.5 against 68.5, and deterministic.)
    __shared__ float red[SUMX ? 3 : 2][WPB][256 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool sums = dgamma != nullptr || SUMX;
    // Chained column reduce (sig_tune_ln_defer): the PREVIOUS LayerNorm backward on this stream left its per-block partial rows in
    // scratch instead of paying a launch for ln_bwd_reduce_kernel; the first blocks of this launch add them up before their own
    // rows -- the same sixteen row chunks in the same order as that kernel, so the sums carry the same bits.
    // The launcher makes the grid a little larger than the exact fit in this mode, so the LAST workgroups walk one row per wave
    // fewer than the first ones: the reduce rides in their slack instead of on the launch's critical path.
    if (prev.part != nullptr && (int)blockIdx.x >= (int)gridDim.x - prev.groups)
        ln_reduce_group<WPB>(prev, (int)blockIdx.x - ((int)gridDim.x - prev.groups), &red[0][0][0]);
    float4 ag[NV], ab[NV], ax[SUMX ? NV : 1], gm[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        ag[it] = make_float4(0, 0, 0, 0);
        ab[it] = make_float4(0, 0, 0, 0);
        if (SUMX) ax[it] = make_float4(0, 0, 0, 0);
        const int c = lane * 4 + it * 256;
        gm[it] = c < D ? *(const float4*)(gamma + c) : make_float4(0, 0, 0, 0);
    }
    for (int row = blockIdx.x * WPB + wave; row < M; row += gridDim.x * WPB) {
        const float mu = mean[row], rs = rstd[row];
        float4 dyv[NV], xh[NV], rv[NV];
        float s1 = 0.f, s2 = 0.f;
        // every load of the row is requested before anything is consumed (the residual-stream gradient too: it used to
        // be requested only after the two row reductions)
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = lane * 4 + it * 256;
            dyv[it] = make_float4(0, 0, 0, 0);
            xh[it] = make_float4(0, 0, 0, 0);
            rv[it] = make_float4(0, 0, 0, 0);
            if (c < D) {
                if (DY_BF16) {
                    const uint2 u = *(const uint2*)((const bf16_t*)dy_ + (size_t)row * D + c);
                    dyv[it] = make_float4(cvt16f((bf16_t)(u.x & 0xffff), dt), cvt16f((bf16_t)(u.x >> 16), dt),
                                          cvt16f((bf16_t)(u.y & 0xffff), dt), cvt16f((bf16_t)(u.y >> 16), dt));
                } else {
                    dyv[it] = *(const float4*)((const float*)dy_ + (size_t)row * D + c);
                }
                xh[it] = *(const float4*)(x + (size_t)row * D + c);
                if (dres) rv[it] = *(const float4*)(dres + (size_t)row * D + c);
            }
        }
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < D) {
                const float4 g = gm[it];
                xh[it] = make_float4((xh[it].x - mu) * rs, (xh[it].y - mu) * rs, (xh[it].z - mu) * rs, (xh[it].w - mu) * rs);
                const float a0 = dyv[it].x * g.x, a1 = dyv[it].y * g.y, a2 = dyv[it].z * g.z, a3 = dyv[it].w * g.w;
                s1 += a0 + a1 + a2 + a3;
                s2 += a0 * xh[it].x + a1 * xh[it].y + a2 * xh[it].z + a3 * xh[it].w;
                ag[it].x += dyv[it].x * xh[it].x; ag[it].y += dyv[it].y * xh[it].y;
                ag[it].z += dyv[it].z * xh[it].z; ag[it].w += dyv[it].w * xh[it].w;
                ab[it].x += dyv[it].x; ab[it].y += dyv[it].y; ab[it].z += dyv[it].z; ab[it].w += dyv[it].w;
            }
        }
        const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < D) {
                const float4 g = gm[it];
                float4 o;
                o.x = rs * (dyv[it].x * g.x - m1 - xh[it].x * m2) + rv[it].x;
                o.y = rs * (dyv[it].y * g.y - m1 - xh[it].y * m2) + rv[it].y;
                o.z = rs * (dyv[it].z * g.z - m1 - xh[it].z * m2) + rv[it].z;
                o.w = rs * (dyv[it].w * g.w - m1 - xh[it].w * m2) + rv[it].w;
                if (dxf) *(float4*)(dxf + (size_t)row * D + c) = o;
                if (dxb) *(uint2*)(dxb + (size_t)row * D + c) = make_uint2(pack2_16(o.x, o.y, dt), pack2_16(o.z, o.w, dt));
                if (SUMX) { ax[it].x += o.x; ax[it].y += o.y; ax[it].z += o.z; ax[it].w += o.w; }
            }
        }
    }
    if (!sums) return;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        *(float4*)&red[0][wave][c] = ag[it];
        *(float4*)&red[1][wave][c] = ab[it];
        if (SUMX) *(float4*)&red[SUMX ? 2 : 0][wave][c] = ax[it];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * WPB) {
        float g = 0.f, bsum = 0.f, xs = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) {
            g += red[0][w][c];
            bsum += red[1][w][c];
            if (SUMX) xs += red[SUMX ? 2 : 0][w][c];
        }
        if (part) {
            // this block's column sums go to its own row of a scratch matrix with plain stores; ln_bwd_reduce_kernel adds the
            // rows in a fixed order.  (One global atomic per column and block: 768 blocks hammering the same 2304 addresses
            // cost ~24 ns per block, 19 of the kernel's 67 us, and made these gradients order-dependent.)
            float* pr = part + (size_t)blockIdx.x * 3 * D;
            pr[c] = g;
            pr[D + c] = bsum;
            pr[2 * D + c] = xs;
        } else {
            if (dgamma) {
                atomicAdd(dgamma + c, g);
                atomicAdd(dbeta + c, bsum);
            }
            if (SUMX) atomicAdd(dxsum + c, xs);
        }
    }
}

// out_k[c] += sum over blocks b of part[b][k][c], k = 0 (dgamma), 1 (dbeta), 2 (dx column sums): a 1024-thread block owns 64
// columns of one array; 16 row chunks are summed in parallel and combined through LDS in chunk order (deterministic)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int D, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dxsum) {
#pragma clang fp reassociate(off)       // fixed order of addition: see ln_reduce_group
    __shared__ float sm[16][64];
    const int k = blockIdx.y;
    float* out = k == 0 ? dgamma : (k == 1 ? dbeta : dxsum);
    if (!out) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), chunk = threadIdx.x >> 6;
    const int per = (nblocks + 15) >> 4, b0 = chunk * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float acc = 0.f;
    if (c < D)
        for (int b = b0; b < b1; ++b) acc += part[((size_t)b * 3 + k) * D + c];
    sm[chunk][threadIdx.x & 63] = acc;
    __syncthreads();
    if (chunk == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x & 63];
        out[c] += t;
    }
}

template <bool DY_BF16, int NV>
static void launch_ln_bwd(int blocks, hipStream_t st, const void* dy, const float* x, const float* gamma, const float* mean,
                          const float* rstd, const float* dres, float* dx_f32, bf16_t* dx_bf16, float* dgamma, float* dbeta, int M,
                          int D, float* dx_colsum, int dt, float* part, const LnPrevReduce& prev) {
    constexpr int WPB = LN_BWD_WPB;
    if (dx_colsum)
        hipLaunchKernelGGL((layernorm_bwd_kernel<DY_BF16, NV, true, WPB>), dim3(blocks), dim3(64 * WPB), 0, st, dy, x, gamma, mean, rstd, dres,
                           dx_f32, dx_bf16, dgamma, dbeta, M, D, dx_colsum, dt, part, prev);
    else
        hipLaunchKernelGGL((layernorm_bwd_kernel<DY_BF16, NV, false, WPB>), dim3(blocks), dim3(64 * WPB), 0, st, dy, x, gamma, mean, rstd, dres,
                           dx_f32, dx_bf16, dgamma, dbeta, M, D, dx_colsum, dt, part, prev);
}

// ---- chained column reduce: state per (device, stream) ----------------------------------------------------------------------
// sig_tune_ln_defer(1): a LayerNorm backward that produces column sums no longer launches ln_bwd_reduce_kernel behind itself; it
// leaves a record here and the NEXT LayerNorm backward on the same stream adds the rows up in its first workgroups (28 launches of
// ~7 us + their boundaries per train step otherwise).  sig_ln_flush(stream) launches the reduce for whatever is still pending.
// Until then the affected dgamma / dbeta / column sums are incomplete: the mode is for a caller that owns the whole backward (the
// training engine turns it on around vit_backward and flushes before anybody reads the gradients; with a data-parallel reducer
// hooked in it stays off, because a block's bucket is sent as soon as that block's backward is enqueued).
#include <atomic>
#include <mutex>
static std::atomic<int> g_ln_defer{0};
struct LnPending { int dev; hipStream_t st; LnPrevReduce job; int slot; };
static LnPending g_ln_pending[8];
static int g_ln_npending = 0;
static std::mutex g_ln_mu;
int sig_tune_ln_defer_impl(int on) {
    const int prev = g_ln_defer.exchange(on != 0);
    return prev;
}
static LnPending* ln_pending_for(hipStream_t st, bool create) {      // (g_ln_mu held)
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < g_ln_npending; ++i)
        if (g_ln_pending[i].st == st && g_ln_pending[i].dev == dev) return &g_ln_pending[i];
    if (!create || g_ln_npending == 8) return nullptr;
    g_ln_pending[g_ln_npending] = LnPending{dev, st, LnPrevReduce{nullptr, {nullptr, nullptr, nullptr}, 0, 0, 0}, 1};
    return &g_ln_pending[g_ln_npending++];
}
int sig_ln_flush_impl(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_ln_mu);
    LnPending* pd = ln_pending_for(st, false);
    if (pd == nullptr || pd->job.part == nullptr) return 0;
    const LnPrevReduce j = pd->job;
    pd->job.part = nullptr;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(sig_ceil_div(j.D, 64), 3), dim3(1024), 0, st, j.part, j.nblocks, j.D, j.out[0], j.out[1], j.out[2]);
    SIG_CHECK_LAUNCH("layernorm_bwd_reduce (flush)");
    return 0;
}

int sig_launch_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                             const float* rstd, const float* dres, float* dx_f32, bf16_t* dx_bf16, float* dgamma,
                             float* dbeta, int M, int D, int dt, hipStream_t st, float* dx_colsum) {
    SIG_CHECK_DT(dt, "layernorm_bwd");
    SIG_CHECK_ARG(M > 0 && D > 0 && (D & 3) == 0 && D <= 256 * LN_MAXV, "layernorm_bwd: D=%d unsupported", D);
    SIG_CHECK_ARG(dy && x && gamma && mean && rstd && (dx_f32 || dx_bf16), "layernorm_bwd: null pointer");
    SIG_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must come together");
    // rows per wave made EQUAL: with a fixed block count 24768 rows over 768 x 4 waves is 8.06 rows per wave -- 6 % of the
    // waves walk a ninth row while the chip idles.  k = rows per wave for ~cap blocks, then exactly ceil(M / (4k)) blocks.
    // cap = 512 = two resident blocks per CU: the largest grid that is still ONE round.  Measured at M = 24768 (in the train step,
    // same box, us): 344 blocks (9 rows per wave; the round-2 setting) 58.0 | 387 (8) 54.3 | 443 (7) 51.6 | 516 (6: a second
    // round starts) 70.1 | 620 (5) 63.4 | 774 (4) 57.4; the column-sum reduce behind it 6.0 -> 6.9 us.
    static int cap = 0, use_part = -1;
    if (!cap) { const char* e = getenv("SIG_LN_BWD_BLOCKS"); cap = e ? atoi(e) : 512; }
    if (use_part < 0) { const char* e = getenv("SIG_LN_BWD_ATOMICS"); use_part = e && atoi(e) ? 0 : 1; }
    const int k = sig_ceil_div(M, LN_BWD_WPB * cap);
    int blocks = sig_ceil_div(M, LN_BWD_WPB * (k > 0 ? k : 1));
    const int nv = (D + 255) / 256;
    const bool sums = dgamma != nullptr || dx_colsum != nullptr;
    // chained mode: this launch adds up the previous launch's partial rows (if any) and writes its own to the OTHER scratch slot
    LnPrevReduce prev{nullptr, {nullptr, nullptr, nullptr}, 0, 0, 0};
    LnPending* pd = nullptr;
    int slot = 1;
    const bool defer = g_ln_defer.load() != 0;
    if (blocks > 64) {
        // room for the chained reduce: up to 40 more workgroups while the rows per wave of the busiest one stay the same and the
        // grid stays one round (cap) -- then the last workgroups have a row per wave less than the first ones.  (In both modes:
        // the partial rows, and with them the bits of the column sums, depend on the grid.)
        int more = blocks + 40 < cap ? blocks + 40 : cap;
        while (more > blocks && sig_ceil_div(M, LN_BWD_WPB * more) != sig_ceil_div(M, LN_BWD_WPB * blocks)) --more;
        blocks = more;
    }
    std::unique_lock<std::mutex> lock(g_ln_mu, std::defer_lock);
    if (defer) {
        lock.lock();
        pd = ln_pending_for(st, true);
        if (pd != nullptr) {
            if (pd->job.part != nullptr && pd->job.groups <= blocks) { prev = pd->job; pd->job.part = nullptr; }
            slot = pd->slot == 1 ? 3 : 1;
        }
    }
    float* part = (sums && use_part && blocks > 16) ? sig_stream_scratch(st, (size_t)blocks * 3 * D * sizeof(float), slot) : nullptr;
    if (prev.part != nullptr && sums && part == nullptr) {
        // this launch adds its own column sums with atomics (few blocks, or no scratch): a chained reduce inside it could meet
        // them on the same addresses -- pay the previous launch's reduce as its own launch, first
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(sig_ceil_div(prev.D, 64), 3), dim3(1024), 0, st, prev.part, prev.nblocks, prev.D, prev.out[0],
                           prev.out[1], prev.out[2]);
        prev.part = nullptr;
    }
#define SIG_LN(BF, NV_) launch_ln_bwd<BF, NV_>(blocks, st, dy, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, dgamma, dbeta, M, D, dx_colsum, dt, part, prev)
    if (dy_is_bf16) {
        if (nv == 1) SIG_LN(true, 1); else if (nv == 2) SIG_LN(true, 2); else if (nv == 3) SIG_LN(true, 3); else SIG_LN(true, 4);
    } else {
        if (nv == 1) SIG_LN(false, 1); else if (nv == 2) SIG_LN(false, 2); else if (nv == 3) SIG_LN(false, 3); else SIG_LN(false, 4);
    }
#undef SIG_LN
    SIG_CHECK_LAUNCH("layernorm_bwd");
    if (part) {
        if (pd != nullptr && pd->job.part == nullptr) {      // leave the rows for the next launch on this stream (or sig_ln_flush)
            pd->job = LnPrevReduce{part, {dgamma, dbeta, dx_colsum}, blocks, D, 3 * sig_ceil_div(D, 64)};
            pd->slot = slot;
        } else {
            hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(sig_ceil_div(D, 64), 3), dim3(1024), 0, st, part, blocks, D, dgamma, dbeta, dx_colsum);
            SIG_CHECK_LAUNCH("layernorm_bwd_reduce");
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n, int dt) {
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const float4 a = *(const float4*)(src + i), b = *(const float4*)(src + i + 4);
            *(uint4*)(dst + i) = make_uint4(pack2_16(a.x, a.y, dt), pack2_16(a.z, a.w, dt), pack2_16(b.x, b.y, dt), pack2_16(b.z, b.w, dt));
        } else {
            for (size_t k = i; k < n; ++k) dst[k] = f2cvt16(src[k], dt);
        }
    }
}
int sig_launch_cast_bf16(const float* src, bf16_t* dst, size_t n, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "cast");
    SIG_CHECK_ARG(src && dst && n > 0, "cast_bf16: bad arguments");
    SIG_CHECK_ARG((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, "cast_bf16: pointers must be 16-B aligned");
    size_t blocks = (n + 2047) / 2048;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n, dt);
    SIG_CHECK_LAUNCH("cast_bf16");
    return 0;
}

// dst[c][r] = bf16(src[r][c]);  64x64 tile through LDS (65-float rows: conflict-free column reads)
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols, int dt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(size_t)(r0 + r) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) dst[(size_t)(c0 + c) * rows + r0 + tx] = f2cvt16(tile[tx][c], dt);
}
int sig_launch_transpose_cast_bf16(const float* src, bf16_t* dst, int rows, int cols, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "transpose_cast");
    SIG_CHECK_ARG(src && dst && rows > 0 && cols > 0, "transpose_cast: bad arguments");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(sig_ceil_div(cols, 64), sig_ceil_div(rows, 64)), dim3(256), 0, st, src, dst, rows, cols, dt);
    SIG_CHECK_LAUNCH("transpose_cast");
    return 0;
}

// All transposed weight copies of a step in ONE launch: table[d] = {src f32*, dst bf16*, rows, cols} (as int64),
// tile_start[d] = first 64x64 tile of matrix d in the flattened grid (tile_start[n] = total).
__global__ __launch_bounds__(256) void transpose_cast_multi_kernel(const long long* __restrict__ table, const int* __restrict__ tile_start, int n, int dt) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const float* src = (const float*)table[lo * 4 + 0];
    bf16_t* dst = (bf16_t*)table[lo * 4 + 1];
    const int rows = (int)table[lo * 4 + 2], cols = (int)table[lo * 4 + 3];
    const int t = b - tile_start[lo], tx_n = (cols + 63) >> 6;
    const int r0 = (t / tx_n) * 64, c0 = (t % tx_n) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(size_t)(r0 + r) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) dst[(size_t)(c0 + c) * rows + r0 + tx] = f2cvt16(tile[tx][c], dt);
}
// The same for sources that already ARE 16-bit (the operand mirror the fused Adam refreshes): table[d] = {src 16-bit*, dst 16-bit*,
// rows, cols}, rows and cols multiples of 64.  16-B loads and 16-B stores (the f32-source kernel above writes 2 B per lane: 3.2 TB/s);
// a pure permutation of bit patterns, so the result equals cast-then-transpose of the f32 master bit for bit.
__global__ __launch_bounds__(256) void transpose16_multi_kernel(const long long* __restrict__ table, const int* __restrict__ tile_start, int n) {
    __shared__ unsigned short tile[64][72];      // row pitch 144 B: the transposed 2-B reads of a lane group spread over the banks
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const bf16_t* src = (const bf16_t*)table[lo * 4 + 0];
    bf16_t* dst = (bf16_t*)table[lo * 4 + 1];
    const int rows = (int)table[lo * 4 + 2], cols = (int)table[lo * 4 + 3];
    const int t = b - tile_start[lo], tx_n = cols >> 6;
    const int r0 = (t / tx_n) * 64, c0 = (t % tx_n) * 64;
    const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;          // 8 chunks of 8 elements x 32 rows per sweep
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = rr + 32 * k;
        const uint4 v = *(const uint4*)(src + (size_t)(r0 + r) * cols + c0 + ch * 8);
        *(uint4*)&tile[r][ch * 8] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = rr + 32 * k;                                   // output row = source column
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = tile[ch * 8 + i][c];
        uint4 o;
        o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
        o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
        *(uint4*)(dst + (size_t)(c0 + c) * rows + r0 + ch * 8) = o;
    }
}
int sig_launch_transpose16_multi(const long long* table, const int* tile_start, int n, int total_tiles, hipStream_t st) {
    SIG_CHECK_ARG(table && tile_start && n > 0 && total_tiles > 0, "transpose16_multi: bad arguments");
    hipLaunchKernelGGL(transpose16_multi_kernel, dim3(total_tiles), dim3(256), 0, st, table, tile_start, n);
    SIG_CHECK_LAUNCH("transpose16_multi");
    return 0;
}
int sig_launch_transpose_cast_multi(const long long* table, const int* tile_start, int n, int total_tiles, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "transpose_cast_multi");
    SIG_CHECK_ARG(table && tile_start && n > 0 && total_tiles > 0, "transpose_cast_multi: bad arguments");
    hipLaunchKernelGGL(transpose_cast_multi_kernel, dim3(total_tiles), dim3(256), 0, st, table, tile_start, n, dt);
    SIG_CHECK_LAUNCH("transpose_cast_multi");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[n] += sum_m a[m][n]
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ a_, int lda, int M, int N, float* __restrict__ out, int dt) {
    constexpr int CPL = BF16 ? 8 : 4;  // columns per lane (16 B)
    __shared__ float red[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.y * 64 + lane) * CPL;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int rbeg = blockIdx.x * 128 + wave * 32;
    if (c < N) {
        // 8 rows requested before any is consumed (a row-at-a-time loop left one 16-B load per lane in flight: 3.2 TB/s)
        for (int r0 = rbeg; r0 < rbeg + 32 && r0 < M; r0 += 8) {
            uint4 u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + k < M ? r0 + k : M - 1;              // clamped rows are read but not added
                u[k] = BF16 ? *(const uint4*)((const bf16_t*)a_ + (size_t)r * lda + c) : *(const uint4*)((const float*)a_ + (size_t)r * lda + c);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (r0 + k >= M) break;
                if (BF16) {
                    acc[0] += cvt16f((bf16_t)(u[k].x & 0xffff), dt); acc[1] += cvt16f((bf16_t)(u[k].x >> 16), dt);
                    acc[2] += cvt16f((bf16_t)(u[k].y & 0xffff), dt); acc[3] += cvt16f((bf16_t)(u[k].y >> 16), dt);
                    acc[4] += cvt16f((bf16_t)(u[k].z & 0xffff), dt); acc[5] += cvt16f((bf16_t)(u[k].z >> 16), dt);
                    acc[6] += cvt16f((bf16_t)(u[k].w & 0xffff), dt); acc[7] += cvt16f((bf16_t)(u[k].w >> 16), dt);
                } else {
                    acc[0] += __uint_as_float(u[k].x); acc[1] += __uint_as_float(u[k].y);
                    acc[2] += __uint_as_float(u[k].z); acc[3] += __uint_as_float(u[k].w);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < CPL; ++e) red[wave][lane * CPL + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * CPL; i += 256) {
        const int col = blockIdx.y * 64 * CPL + i;
        if (col < N) atomicAdd(out + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}
int sig_launch_colsum_bf16(const bf16_t* a, int lda, int M, int N, float* out, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "colsum");
    SIG_CHECK_ARG(a && out && M > 0 && N > 0 && (N & 7) == 0 && (lda & 7) == 0, "colsum_bf16: N and lda must be multiples of 8");
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(sig_ceil_div(M, 128), sig_ceil_div(N, 512)), dim3(256), 0, st, a, lda, M, N, out, dt);
    SIG_CHECK_LAUNCH("colsum_bf16");
    return 0;
}
int sig_launch_colsum_f32(const float* a, int lda, int M, int N, float* out, hipStream_t st) {
    SIG_CHECK_ARG(a && out && M > 0 && N > 0 && (N & 3) == 0 && (lda & 3) == 0, "colsum_f32: N and lda must be multiples of 4");
    hipLaunchKernelGGL(colsum_kernel<false>, dim3(sig_ceil_div(M, 128), sig_ceil_div(N, 256)), dim3(256), 0, st, a, lda, M, N, out, 0);
    SIG_CHECK_LAUNCH("colsum_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// patch gather: img f32 [nimg,3,H,W] -> bf16 [nimg*h*w, 3*P*P], column = c*P*P + dy*P + dx
// (the im2col of the stride-16 conv at clip/model.py:433,448-450).  Thread = 8 consecutive pixels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int nimg, int H, int W, int P, int dt) {
    const int w8 = W >> 3;
    const size_t total = (size_t)nimg * 3 * H * w8;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int x8 = (int)(t % w8);
        size_t r = t / w8;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % 3);
        const int n = (int)(r / 3);
        const float* src = img + (((size_t)n * 3 + c) * H + y) * W + x8 * 8;
        const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
        const int x = x8 * 8, pi = y / P, pj = x / P, dy = y - pi * P, dx = x - pj * P;
        const int wg = W / P, hg = H / P;
        const size_t row = ((size_t)n * hg + pi) * wg + pj;
        bf16_t* dst = out + row * (size_t)(3 * P * P) + c * P * P + dy * P + dx;
        *(uint4*)dst = make_uint4(pack2_16(a.x, a.y, dt), pack2_16(a.z, a.w, dt), pack2_16(b.x, b.y, dt), pack2_16(b.z, b.w, dt));
    }
}
int sig_launch_im2col(const float* img, bf16_t* out, int nimg, int H, int W, int P, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "im2col");
    SIG_CHECK_ARG(img && out && nimg > 0, "im2col: bad arguments");
    SIG_CHECK_ARG(P % 8 == 0 && H % P == 0 && W % P == 0, "im2col: patch %d must be a multiple of 8 and divide %dx%d", P, H, W);
    const size_t total = (size_t)nimg * 3 * H * (W >> 3);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, out, nimg, H, W, P, dt);
    SIG_CHECK_LAUNCH("im2col");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// token assembly + ln_pre (clip/model.py:451-459): sequences s = modality*B + b, row 0 = CLS.
//   pre[s,0]   = class_embedding + sie_coe*cv_embed[cam[b]] + pos[0]
//   pre[s,1+p] = tok[s*Lp + p] + pos[1+p]
//   x = LN(pre)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_assemble_kernel(const float* __restrict__ tok, const float* __restrict__ cls_emb,
                                                             const float* __restrict__ pos, const float* __restrict__ cv,
                                                             const int64_t* __restrict__ cam, float sie, const float* __restrict__ g,
                                                             const float* __restrict__ b, float* __restrict__ x, float* __restrict__ pre,
                                                             float* __restrict__ mean, float* __restrict__ rstd, int S, int B, int L,
                                                             int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S * L) return;
    const int s = row / L, l = row - s * L;
    float4 v[LN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            float4 t;
            if (l == 0) {
                t = *(const float4*)(cls_emb + c);
                if (cv) {
                    const float4 e = *(const float4*)(cv + (size_t)cam[s % B] * D + c);
                    t.x += sie * e.x; t.y += sie * e.y; t.z += sie * e.z; t.w += sie * e.w;
                }
            } else {
                t = *(const float4*)(tok + ((size_t)s * (L - 1) + (l - 1)) * D + c);
            }
            const float4 pe = *(const float4*)(pos + (size_t)l * D + c);
            t.x += pe.x; t.y += pe.y; t.z += pe.z; t.w += pe.w;
            v[it] = t;
            sum += t.x + t.y + t.z + t.w;
            if (pre) *(float4*)(pre + (size_t)row * D + c) = t;
        }
    }
    const float mu = wave_sum(sum) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float a0 = v[it].x - mu, a1 = v[it].y - mu, a2 = v[it].z - mu, a3 = v[it].w - mu;
            q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 gg = *(const float4*)(g + c), bb = *(const float4*)(b + c);
            float4 o;
            o.x = (v[it].x - mu) * rs * gg.x + bb.x;
            o.y = (v[it].y - mu) * rs * gg.y + bb.y;
            o.z = (v[it].z - mu) * rs * gg.z + bb.z;
            o.w = (v[it].w - mu) * rs * gg.w + bb.w;
            *(float4*)(x + (size_t)row * D + c) = o;
        }
    }
}
int sig_launch_embed_assemble(const float* tok, const float* cls_emb, const float* pos, const float* cv_embed,
                              const int64_t* cam, float sie_coe, const float* g, const float* b, float* x,
                              float* pre_ln, float* mean, float* rstd, int S, int B, int L, int D, float eps,
                              hipStream_t st) {
    SIG_CHECK_ARG(tok && cls_emb && pos && g && b && x, "embed_assemble: null pointer");
    SIG_CHECK_ARG((cv_embed == nullptr) || cam, "embed_assemble: cam labels missing");
    SIG_CHECK_ARG(S > 0 && B > 0 && S % B == 0 && L > 1 && (D & 3) == 0 && D <= 256 * LN_MAXV, "embed_assemble: bad shape");
    hipLaunchKernelGGL(embed_assemble_kernel, dim3(sig_ceil_div(S * L, 4)), dim3(256), 0, st, tok, cls_emb, pos, cv_embed,
                       cam, sie_coe, g, b, x, pre_ln, mean, rstd, S, B, L, D, eps);
    SIG_CHECK_LAUNCH("embed_assemble");
    return 0;
}

// backward of the assembly (given d pre-LN tokens): dtok (f32 and/or bf16, [S*Lp, D]), dpos[l] = sum_s,
// dcls = sum_s d[s,0], dcv[cam] += sie * d[s,0].  Block = (token position, chunk of sequences, 1024-column slab); a lane
// owns 4 consecutive columns (16-B loads/stores) and walks its chunk of sequences; one atomic per column and chunk.
// (One block per position walking all 192 sequences with 4-B accesses ran at ~1 TB/s: 114 us.)
#define EMB_BWD_CHUNKS 8
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dpre, float* __restrict__ dtokf, bf16_t* __restrict__ dtokb,
                                                        float* __restrict__ dcls, float* __restrict__ dpos, float* __restrict__ dcv,
                                                        const int64_t* __restrict__ cam, float sie, int S, int B, int L, int D, int dt) {
    const int l = blockIdx.x;
    const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
    if (c >= D) return;
    const int per = (S + EMB_BWD_CHUNKS - 1) / EMB_BWD_CHUNKS;
    const int s0 = blockIdx.y * per, s1 = s0 + per < S ? s0 + per : S;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; ++s) {
        const f32x4_t v = *(const f32x4_t*)(dpre + ((size_t)s * L + l) * D + c);
        acc += v;
        if (l == 0) {
            if (dcv) {
                float* dst = dcv + (size_t)cam[s % B] * D + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, sie * v[e]);
            }
        } else {
            const size_t o = ((size_t)s * (L - 1) + (l - 1)) * D + c;
            if (dtokf) *(f32x4_t*)(dtokf + o) = v;
            if (dtokb) *(uint2*)(dtokb + o) = make_uint2(pack2_16(v[0], v[1], dt), pack2_16(v[2], v[3], dt));
        }
    }
    if (s1 <= s0) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        atomicAdd(dpos + (size_t)l * D + c + e, acc[e]);
        if (l == 0) atomicAdd(dcls + c + e, acc[e]);
    }
}
int sig_launch_embed_bwd(const float* dx_pre, float* dtok_f32, bf16_t* dtok_bf16, float* dcls, float* dpos,
                         float* dcv, const int64_t* cam, float sie_coe, int S, int B, int L, int D, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "embed_bwd");
    SIG_CHECK_ARG(dx_pre && dcls && dpos && (dtok_f32 || dtok_bf16), "embed_bwd: null pointer");
    SIG_CHECK_ARG((dcv == nullptr) || cam, "embed_bwd: cam labels missing");
    SIG_CHECK_ARG((D & 3) == 0, "embed_bwd: D=%d must be a multiple of 4", D);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(L, EMB_BWD_CHUNKS, sig_ceil_div(D, 1024)), dim3(256), 0, st, dx_pre, dtok_f32, dtok_bf16,
                       dcls, dpos, dcv, cam, sie_coe, S, B, L, D, dt);
    SIG_CHECK_LAUNCH("embed_bwd");
    return 0;
}
