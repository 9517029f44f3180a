// Self-attention of the CLIP ViT block (clip/model.py:223-225): softmax(q k^T / 8) v per (sequence, head),
// L <= 144 tokens (129 in every shipped config), head dim 64, no mask, no dropout.
//
// One workgroup (3 waves) per (sequence, head).  K and V of the head (<= 18 KB each in bf16) live in LDS
// for the whole block; q fragments come straight from HBM.  Scores are formed TRANSPOSED (S^T = K q^T,
// key on the accumulator row, query on the lane) so that after the in-register softmax the probability
// tile already is the B operand of O^T = V^T P^T -- no LDS round trip for P.  V^T fragments are fetched with
// ds_read_b64_tr_b16 from the row-major V image.  Row max / sum are two xor-shuffles across the 4 lane
// groups that share a query.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "sig_common.h"
#include "sig_kernels.h"

#ifdef SIG_ATTN_STAMPS   // diagnostic build only (tools/attn_stamps.py)
__device__ unsigned long long g_astamps[4 * 4096];
extern "C" int sig_debug_read_attn_stamps(unsigned long long* out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_astamps), sizeof(unsigned long long) * 4 * nblocks) == hipSuccess ? 0 : 2;
}
#define ATT_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define ATT_STAMP(v)
#endif
#define ATT_KROWS 144
#define ATT_VROWS 160
#define ATT_NT 9  // max 16-row tiles

// K image: 128-B rows, physical chunk = chunk ^ ((row >> 1) & 7)          (row reads, ds_read_b128)
// V image: 128-B rows, physical chunk = chunk ^ (((row >> 1) & 3) << 1)   (transposed reads, 32-B blocks)
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

// FULL: L in (128, 144], i.e. all nine 16-row tiles exist (every shipped config: L = 129) -- the per-tile guards fold away;
// with a run-time tile count the ~50 live scalar conditions were spilled to VGPR lanes (v_writelane / v_readlane).
// NW = waves per block: 3 (three query tiles per wave, one after the other) or 9 (one tile per wave: nine waves' q-fragment loads and
// softmax chains in flight at once; FULL only)
template <bool FULL, int DT, int NW = 3>
__global__ __launch_bounds__(64 * NW, NW == 3 ? 3 : 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int S, int L, int H) {
    __shared__ __attribute__((aligned(16))) char smem[ATT_KROWS * 128 + ATT_VROWS * 128];
#ifdef SIG_ATTN_STAMPS
    unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0;
    ATT_STAMP(ta0);
#endif
    char* sK = smem;
    char* sV = smem + ATT_KROWS * 128;
    const int s = blockIdx.x / H, h = blockIdx.x - s * H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* base = qkv + (size_t)s * L * D3 + h * 64;

    // K and V of the head: every load of a thread is requested before the first LDS write (sweep by sweep the HBM
    // latencies were serialised)
    constexpr int NTH = 64 * NW;
    constexpr int KS = ATT_KROWS * 8 / NTH, VS = (ATT_VROWS * 8 + NTH - 1) / NTH;
    static_assert(ATT_KROWS * 8 % NTH == 0, "K staging sweeps");
    uint4 rk[KS], rv[VS];
#pragma unroll
    for (int it = 0; it < KS; ++it) {
        const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
        rk[it] = make_uint4(0, 0, 0, 0);
        if (r < L) rk[it] = *(const uint4*)(base + (size_t)r * D3 + Dm + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < VS; ++it) {
        const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
        rv[it] = make_uint4(0, 0, 0, 0);
        if (c < ATT_VROWS * 8 && r < L) rv[it] = *(const uint4*)(base + (size_t)r * D3 + 2 * Dm + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < KS; ++it) {
        const int c = tid + it * NTH;
        *(uint4*)(sK + k_off(c >> 3, c & 7)) = rk[it];
    }
#pragma unroll
    for (int it = 0; it < VS; ++it) {
        const int c = tid + it * NTH;
        if (c < ATT_VROWS * 8) *(uint4*)(sV + v_off(c >> 3, c & 7)) = rv[it];
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int NT = FULL ? ATT_NT : (L + 15) >> 4;
#ifdef SIG_ATTN_STAMPS
    ATT_STAMP(ta1);
#endif
    const int tq = fr >> 2, tp = fr & 3;  // transposed-read address roles inside the 16-lane group
    const float scale = 0.125f;

#pragma unroll 1   // (with a constant tile count the compiler would unroll the three tiles: 336 registers)
    for (int qt = wave; qt < NT; qt += NW) {
        const int q = qt * 16 + fr;
        const int qc = q < L ? q : L - 1;
        bf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8_t*)(base + (size_t)qc * D3 + ks * 32 + g * 8);

        // S^T tiles: lane holds keys kt*16 + 4g + reg of query fr.  Scores stay RAW (unscaled): the 1/8 is folded into the
        // exponent below; only the tail key tile needs masking (the kernel is instruction-issue bound: ~440 instructions
        // per query tile before this diet).
        f32x4_t sc[ATT_NT + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < ATT_NT; ++kt) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            if (FULL || kt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + k_off(kt * 16 + fr, (ks << 2) | g));
                    a = mfma16<DT>(kf, qf[ks], a);
                }
                if (kt * 16 + 16 > L) {   // uniform: the tile that straddles L
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt * 16 + 4 * g + e >= L) a[e] = -INFINITY;
                }
            } else {
                a = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
            mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
            sc[kt] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // p = exp(scale * (s - mx)) = 2^(s*c - mx*c), c = scale * log2(e): one FMA + one v_exp per element
        const float c2 = scale * 1.4426950408889634f, mc = -mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < ATT_NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][e], c2, mc));
                sc[kt][e] = pv;
                sum += pv;
            }
        sc[ATT_NT] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);

        // O^T = V^T P^T : k-step = key tiles (2kk, 2kk+1); hardware k index 8g+j <-> key 32kk + (j<4 ? 4g+j : 16+4g+j-4)
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < (ATT_NT + 1) / 2; ++kk) {
            if (!FULL && kk * 2 >= NT) break;
            const f32x4_t p0 = sc[2 * kk], p1 = sc[2 * kk + 1];
            union { uint32_t w[4]; bf16x8_t v; } pk;
            pk.w[0] = pack2_t<DT>(p0[0], p0[1]); pk.w[1] = pack2_t<DT>(p0[2], p0[3]);
            pk.w[2] = pack2_t<DT>(p1[0], p1[1]); pk.w[3] = pack2_t<DT>(p1[2], p1[3]);
            const bf16x8_t pf = pk.v;
            const int r0 = 32 * kk + 4 * g + tq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(sV + v_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(sV + v_off(r0 + 16, chunk) + ((tp & 1) << 3));
                const bf16x8_t vf = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[dt] = mfma16<DT>(vf, pf, o[dt]);
            }
        }
        // lane (fr, g) holds head columns dt*16 + 4g .. +3 of query fr: exchange lane pairs g / g^1 (v_permlane16_swap, as
        // in the GEMM epilogue) so that a lane owns 8 consecutive columns -> two 16-B stores instead of four 8-B ones
        {
            const float inv = 1.0f / sum;
            uint32_t w[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                w[dt][0] = pack2_t<DT>(o[dt][0] * inv, o[dt][1] * inv);
                w[dt][1] = pack2_t<DT>(o[dt][2] * inv, o[dt][3] * inv);
            }
#pragma unroll
            for (int dp = 0; dp < 2; ++dp) {
                auto r0 = __builtin_amdgcn_permlane16_swap(w[2 * dp][0], w[2 * dp + 1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane16_swap(w[2 * dp][1], w[2 * dp + 1][1], false, false);
                if (q < L) {
                    bf16_t* orow = out + ((size_t)s * L + q) * Dm + h * 64 + (2 * dp + (g & 1)) * 16 + (g >> 1) * 8;
                    *(uint4*)orow = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
            }
            if (q < L && lse && g == 0) lse[((size_t)s * H + h) * L + q] = mx * scale + __logf(sum);
        }
    }
#ifdef SIG_ATTN_STAMPS
    ATT_STAMP(ta2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP(ta3);
    if (tid == 0 && blockIdx.x < 4096) {
        g_astamps[blockIdx.x * 4 + 0] = ta0; g_astamps[blockIdx.x * 4 + 1] = ta1;
        g_astamps[blockIdx.x * 4 + 2] = ta2; g_astamps[blockIdx.x * 4 + 3] = ta3;
    }
#endif
}

// waves per block of the forward at L in (128, 144]: 9 (one query tile per wave, default) or 3 (three tiles per wave, one after the
// other); SIG_ATTN_FWD_WAVES / sig_tune_attn_fwd_waves
static std::atomic<int> g_attn_fwd_waves{-1};
static int attn_fwd_waves() {
    int v = g_attn_fwd_waves.load();
    if (v < 0) {
        const char* e = getenv("SIG_ATTN_FWD_WAVES");
        v = e && atoi(e) == 3 ? 3 : 9;
        g_attn_fwd_waves = v;
    }
    return v;
}
int sig_tune_attn_fwd_waves_impl(int waves) {
    const int prev = attn_fwd_waves();
    g_attn_fwd_waves = waves == 3 ? 3 : 9;
    return prev;
}

template <int DT>
static void launch_attn_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int S, int L, int H, hipStream_t st) {
    if (L > 16 * (ATT_NT - 1)) {
        if (attn_fwd_waves() == 9) hipLaunchKernelGGL((attn_fwd_kernel<true, DT, 9>), dim3(S * H), dim3(576), 0, st, qkv, out, lse, S, L, H);
        else hipLaunchKernelGGL((attn_fwd_kernel<true, DT>), dim3(S * H), dim3(192), 0, st, qkv, out, lse, S, L, H);
    } else {
        hipLaunchKernelGGL((attn_fwd_kernel<false, DT>), dim3(S * H), dim3(192), 0, st, qkv, out, lse, S, L, H);
    }
}
int sig_launch_attn_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int S, int L, int H, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "attn_fwd");
    SIG_CHECK_ARG(qkv && out, "attn_fwd: null pointer");
    SIG_CHECK_ARG(S > 0 && H > 0 && L > 0 && L <= ATT_KROWS, "attn_fwd: L=%d must be in 1..%d", L, ATT_KROWS);
    if (dt == SIG_DT_F16) launch_attn_fwd<SIG_DT_F16>(qkv, out, lse, S, L, H, st);
    else launch_attn_fwd<SIG_DT_BF16>(qkv, out, lse, S, L, H, st);
    SIG_CHECK_LAUNCH("attn_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward.  Recomputes P from q, k and the saved LSE.  Two passes per (sequence, head), both without
// cross-wave sums, atomics or an LDS image of P:
//   pass A (a wave owns QUERY tiles): S^T, dP^T with the key on the accumulator row -> dS^T is the B
//          operand of dQ^T = K^T dS^T (sum over keys = accumulator rows);
//   pass B (a wave owns KEY tiles):   S, dP with the query on the accumulator row -> P and dS are the B
//          operands of dV^T = dO^T P and dK^T = Q^T dS (sum over queries = accumulator rows).
// Q, K, V, dO of the head are staged once in LDS in one image each that serves both the row reads
// (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16).
// ------------------------------------------------------------------------------------------------
// dual-use image: 128-B rows (64 bf16), physical chunk = chunk ^ f(row), f = PERM[(row >> 1) & 7] with
// PERM = (0,2,4,6,5,7,1,3): found with tools/lds_bank_sim.py, conflict-free for BOTH the ds_read_b128 row
// reads (16 rows at chunk c / c+1 per lane group) and the ds_read_b64_tr_b16 reads (rows 4g+q', 32-B blocks).
__device__ __forceinline__ int d_off(int row, int chunk) {
    const int p = (row >> 1) & 7;
    const int f = (((p & 3) << 1) | (p >> 2)) ^ ((p >> 2) << 2);
    return row * 128 + ((chunk ^ f) << 4);
}

// Store a 16-row x 64-column f32 accumulator tile (lane (fr, g): columns dt*16 + 4g .. +3 of row fr) as bf16 with
// 16-B stores: lanes g / g^1 exchange halves (v_permlane16_swap, see gemm_bf16.hip) so a lane owns 8 consecutive columns.
// `row` = this lane's row base (64 columns), `live` = the row exists; every lane must call it (the swap needs all).
template <int DT>
__device__ __forceinline__ void store_rows16(const f32x4_t (&o)[4], bf16_t* row, bool live, int g) {
    uint32_t w[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        w[dt][0] = pack2_t<DT>(o[dt][0], o[dt][1]);
        w[dt][1] = pack2_t<DT>(o[dt][2], o[dt][3]);
    }
#pragma unroll
    for (int dp = 0; dp < 2; ++dp) {
        auto r0 = __builtin_amdgcn_permlane16_swap(w[2 * dp][0], w[2 * dp + 1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(w[2 * dp][1], w[2 * dp + 1][1], false, false);
        if (live) *(uint4*)(row + (2 * dp + (g & 1)) * 16 + (g >> 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

#define ATB_ROWS 144  // 9 tiles; reads past it are clamped (they only ever meet zero probabilities)
template <bool FULL, int DT>
__global__ __launch_bounds__(192, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                       const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                       bf16_t* __restrict__ dqkv, int S, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SIG_ATTN_STAMPS
    unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0;
    ATT_STAMP(ta0);
#endif
    char* sQ = smem;
    char* sK = smem + ATB_ROWS * 128;
    char* sV = smem + 2 * ATB_ROWS * 128;
    char* sG = smem + 3 * ATB_ROWS * 128;                 // dO
    float* sLse = (float*)(smem + 4 * ATB_ROWS * 128);    // [160] lse * log2(e), +inf for rows >= L
    float* sDel = sLse + 160;                             // [160] delta = rowsum(dO * O) * scale
    const int s = blockIdx.x / H, h = blockIdx.x - s * H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* base = qkv + (size_t)s * L * D3 + h * 64;
    const bf16_t* obase = out + (size_t)s * L * Dm + h * 64;
    const bf16_t* gbase = dout + (size_t)s * L * Dm + h * 64;

    // 192 threads = 24 rows x 8 chunks per sweep, 6 sweeps; delta via an 8-lane xor reduction.  All 30 loads of a thread
    // are requested before any is consumed: sweep by sweep, the six HBM latencies were serialised (22 k of the block's
    // 70 k cycles, tools/attn_stamps.py).
    static_assert(ATB_ROWS * 8 % 192 == 0, "staging sweeps");
    constexpr int SWEEPS = ATB_ROWS * 8 / 192;
    uint4 lq[SWEEPS], lk[SWEEPS], lv[SWEEPS], lg[SWEEPS], lo[SWEEPS];
    float llse[SWEEPS];
#pragma unroll
    for (int it = 0; it < SWEEPS; ++it) {
        const int c = tid + it * 192, r = c >> 3, ch = c & 7;
        lq[it] = lk[it] = lv[it] = lg[it] = lo[it] = make_uint4(0, 0, 0, 0);
        llse[it] = 0.f;
        if (r < L) {
            lq[it] = *(const uint4*)(base + (size_t)r * D3 + ch * 8);
            lk[it] = *(const uint4*)(base + (size_t)r * D3 + Dm + ch * 8);
            lv[it] = *(const uint4*)(base + (size_t)r * D3 + 2 * Dm + ch * 8);
            lg[it] = *(const uint4*)(gbase + (size_t)r * Dm + ch * 8);
            lo[it] = *(const uint4*)(obase + (size_t)r * Dm + ch * 8);
            if (ch == 0) llse[it] = lse[((size_t)s * H + h) * L + r];
        }
    }
#pragma unroll
    for (int it = 0; it < SWEEPS; ++it) {
        const int c = tid + it * 192, r = c >> 3, ch = c & 7;
        const int off = d_off(r, ch);
        *(uint4*)(sQ + off) = lq[it];
        *(uint4*)(sK + off) = lk[it];
        *(uint4*)(sV + off) = lv[it];
        *(uint4*)(sG + off) = lg[it];
        const uint32_t gw[4] = {lg[it].x, lg[it].y, lg[it].z, lg[it].w}, ow[4] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w};
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d += cvt16f_t<DT>((bf16_t)(gw[e] & 0xffff)) * cvt16f_t<DT>((bf16_t)(ow[e] & 0xffff));
            d += cvt16f_t<DT>((bf16_t)(gw[e] >> 16)) * cvt16f_t<DT>((bf16_t)(ow[e] >> 16));
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (ch == 0) {   // pre-scaled: p = 2^(s*c2 - lse*log2e), dS = p * (dP*scale - delta*scale); rows >= L give p = 0
            sDel[r] = d * 0.125f;
            sLse[r] = r < L ? llse[it] * 1.4426950408889634f : INFINITY;
        }
    }
    if (tid < 16) {       // rows 144..159: the tenth (non-existent) tile read by the paired k-steps
        sLse[ATB_ROWS + tid] = INFINITY;
        sDel[ATB_ROWS + tid] = 0.f;
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int NT = FULL ? ATT_NT : (L + 15) >> 4;
    const int tq = fr >> 2, tp = fr & 3;
    const float scale = 0.125f;

#ifdef SIG_ATTN_STAMPS
    ATT_STAMP(ta1);
#endif
    // ------------------------------ pass A: dQ (wave owns query tiles) ------------------------------
#pragma unroll 1   // (with a constant tile count the compiler would unroll the three tiles: 336 registers)
    for (int qt = wave; qt < NT; qt += 3) {
        const int q = qt * 16 + fr;
        bf16x8_t qf[2], gf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *(const bf16x8_t*)(sQ + d_off(q, (ks << 2) | g));
            gf[ks] = *(const bf16x8_t*)(sG + d_off(q, (ks << 2) | g));
        }
        const float lq2 = sLse[q], dqs = sDel[q];     // +inf / 0 for q >= L
        const float c2 = scale * 1.4426950408889634f;
        f32x4_t ds[ATT_NT + 1];
#pragma unroll
        for (int kt = 0; kt < ATT_NT; ++kt) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (FULL || kt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + d_off(kt * 16 + fr, (ks << 2) | g));
                    const bf16x8_t vf = *(const bf16x8_t*)(sV + d_off(kt * 16 + fr, (ks << 2) | g));
                    a = mfma16<DT>(kf, qf[ks], a);  // S^T
                    b = mfma16<DT>(vf, gf[ks], b);  // dP^T
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(a[e], c2, -lq2));
                a[e] = pv * __builtin_fmaf(b[e], scale, -dqs);
            }
            if (kt * 16 + 16 > L) {   // uniform: the key tile that straddles L (and the ones past it)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kt * 16 + 4 * g + e >= L) a[e] = 0.f;
            }
            ds[kt] = a;
        }
        ds[ATT_NT] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < (ATT_NT + 1) / 2; ++kk) {
            if (!FULL && kk * 2 >= NT) break;
            const f32x4_t p0 = ds[2 * kk], p1 = ds[2 * kk + 1];
            union { uint32_t w[4]; bf16x8_t v; } pk;
            pk.w[0] = pack2_t<DT>(p0[0], p0[1]); pk.w[1] = pack2_t<DT>(p0[2], p0[3]);
            pk.w[2] = pack2_t<DT>(p1[0], p1[1]); pk.w[3] = pack2_t<DT>(p1[2], p1[3]);
            const bf16x8_t pf = pk.v;
            const int r0 = 32 * kk + 4 * g + tq;
            const int r1 = r0 + 16 < ATB_ROWS ? r0 + 16 : r0;  // tile 9 does not exist: its dS is 0
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(sK + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(sK + d_off(r1, chunk) + ((tp & 1) << 3));
                const bf16x8_t kf = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[dt] = mfma16<DT>(kf, pf, o[dt]);
            }
        }
        store_rows16<DT>(o, dqkv + ((size_t)s * L + (q < L ? q : 0)) * D3 + h * 64, q < L, g);
    }

#ifdef SIG_ATTN_STAMPS
    ATT_STAMP(ta2);
#endif
    // ------------------------------ pass B: dK, dV (wave owns key tiles) ------------------------------
    // (run-time tile count here even when FULL: with the guards folded the two accumulator sets and the fully unrolled
    //  k-steps need 336 registers -- 80 spills, pass B 18 k -> 49 k cycles)
    const int NTb = (L + 15) >> 4;
#pragma unroll 1
    for (int kt = wave; kt < NTb; kt += 3) {
        const int key = kt * 16 + fr;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = *(const bf16x8_t*)(sK + d_off(key, (ks << 2) | g));
            vf[ks] = *(const bf16x8_t*)(sV + d_off(key, (ks << 2) | g));
        }
        const bool keyok = key < L;
        const float c2 = scale * 1.4426950408889634f;
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qq = 0; qq < (ATT_NT + 1) / 2; ++qq) {  // query tiles (2qq, 2qq+1) = one 32-deep k-step
            if (qq * 2 >= NTb) break;
            uint32_t pw[4], sw[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * qq + half;
                f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                if (qt < NTb) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8_t qf = *(const bf16x8_t*)(sQ + d_off(qt * 16 + fr, (ks << 2) | g));
                        const bf16x8_t gf = *(const bf16x8_t*)(sG + d_off(qt * 16 + fr, (ks << 2) | g));
                        a = mfma16<DT>(qf, kf[ks], a);  // S  [row q][col key]
                        b = mfma16<DT>(gf, vf[ks], b);  // dP [row q][col key]
                    }
                }
                // 4 consecutive queries per lane: their pre-scaled lse / delta in one b128 read each (+inf / 0 past L)
                const f32x4_t l4 = *(const f32x4_t*)(sLse + qt * 16 + 4 * g), d4 = *(const f32x4_t*)(sDel + qt * 16 + 4 * g);
                float pv[4], dsv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[e] = keyok ? __builtin_amdgcn_exp2f(__builtin_fmaf(a[e], c2, -l4[e])) : 0.f;
                    dsv[e] = pv[e] * __builtin_fmaf(b[e], scale, -d4[e]);
                }
                pw[half * 2] = pack2_t<DT>(pv[0], pv[1]); pw[half * 2 + 1] = pack2_t<DT>(pv[2], pv[3]);
                sw[half * 2] = pack2_t<DT>(dsv[0], dsv[1]); sw[half * 2 + 1] = pack2_t<DT>(dsv[2], dsv[3]);
            }
            union { uint32_t w[4]; bf16x8_t v; } pu, su;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pu.w[e] = pw[e]; su.w[e] = sw[e]; }
            const bf16x8_t pf = pu.v, sf = su.v;
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
            const int r0 = 32 * qq + 4 * g + tq;
            const int r1 = r0 + 16 < ATB_ROWS ? r0 + 16 : r0;  // query tile 9 does not exist: P = dS = 0 there
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t g0 = lds_tr16(sG + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t g1 = lds_tr16(sG + d_off(r1, chunk) + ((tp & 1) << 3));
                const bf16x4_t q0 = lds_tr16(sQ + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t q1 = lds_tr16(sQ + d_off(r1, chunk) + ((tp & 1) << 3));
                const bf16x8_t gT = (bf16x8_t){g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const bf16x8_t qT = (bf16x8_t){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                dv[dt] = mfma16<DT>(gT, pf, dv[dt]);
                dk[dt] = mfma16<DT>(qT, sf, dk[dt]);
            }
        }
        bf16_t* krow = dqkv + ((size_t)s * L + (keyok ? key : 0)) * D3 + Dm + h * 64;
        store_rows16<DT>(dk, krow, keyok, g);
        store_rows16<DT>(dv, krow + Dm, keyok, g);
    }
#ifdef SIG_ATTN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP(ta3);
#ifdef SIG_ATTN_HWID   // tools/attn_timeline.py: where the block ran instead of the pass-A stamp (HW_ID = reg 4, XCC_ID = reg 20)
    ta2 = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) |
          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
#endif
    if (tid == 0 && blockIdx.x < 4096) {
        g_astamps[blockIdx.x * 4 + 0] = ta0; g_astamps[blockIdx.x * 4 + 1] = ta1;
        g_astamps[blockIdx.x * 4 + 2] = ta2; g_astamps[blockIdx.x * 4 + 3] = ta3;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Backward for L = 16 * 8 + 1 (every shipped geometry: 128 patches + the class token).  With nine 16-row tiles the
// ninth is 94 % padding: 17 of the 81 (query tile, key tile) pairs are almost empty, and nine tasks per pass on three
// waves leave the block's time at three tasks per pass.  Here the first 128 rows are EIGHT clean tiles -- one per wave on
// eight waves (PAIR = 1, the default: the arithmetic phase of a block, which bounds the kernel, then has two waves per SIMD)
// or two per wave on four (PAIR = 2) -- and row 128, `x` below, is handled beside them:
//   column x of S / dP (every query against key x) and row x (query x against every key): 4 x 129 dot products of 64, one
//   per thread, as packed two-element dot products (v_dot2c); from them P[:,x], dS[:,x], P[x,:], dS[x,:];
//   dQ[x] = sum_j dS[x,j] K[j],  dK[x] = sum_i dS[i,x] Q[i],  dV[x] = sum_i P[i,x] dO[i]: one wave each, on the matrix cores
//   (the transposed image fragments of the passes against a B operand whose only non-zero column is the weight vector);
//   the rank-1 terms dQ[i] += dS[i,x] k_x, dK[j] += dS[x,j] q_x, dV[j] += P[x,j] dO_x are added to the MFMA accumulators of
//   pass A / pass B just before they are stored (16 FMAs per lane and tile).
// Everything else (dual-use LDS images, passes A and B, in-register dS) is the kernel above with 8 tiles.
// ------------------------------------------------------------------------------------------------
// c + a.lo * b.lo + a.hi * b.hi on two packed 16-bit pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): no conversions, one instruction
template <int DT> __device__ __forceinline__ float dot2_t(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    if (DT == SIG_DT_F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), c, false);
}
#define ATX_NT 8
#ifndef SIG_ATTN_BWD_WAVES8_DEFAULT
#define SIG_ATTN_BWD_WAVES8_DEFAULT 1
#endif
// PAIR = tiles per wave and pass: 2 = four waves (256 threads), 1 = eight waves (512 threads, <= 128 registers): two resident
// blocks then put FOUR waves on a SIMD, two of them in the arithmetic phase while the other block stages
template <int DT, int PAIR>
__global__ __launch_bounds__(512 / PAIR, 2) void attn_bwd_x1_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int S, int H) {
    constexpr int L = 16 * ATX_NT + 1, XR = 16 * ATX_NT, NW = ATX_NT / PAIR, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;
    char* sK = smem + ATB_ROWS * 128;
    char* sV = smem + 2 * ATB_ROWS * 128;
    char* sG = smem + 3 * ATB_ROWS * 128;                 // dO
    float* sLse = (float*)(smem + 4 * ATB_ROWS * 128);    // [160] lse * log2(e)
    float* sDel = sLse + 160;                             // [160] delta = rowsum(dO * O) * scale
    float* colS = sDel + 160;                             // [132] q_i . k_x  -> P[i,x]
    float* colD = colS + 132;                             // [132] dO_i . v_x -> dS[i,x]
    float* rowS = colD + 132;                             // [132] q_x . k_j  -> P[x,j]
    float* rowD = rowS + 132;                             // [132] dO_x . v_j -> dS[x,j]
    const int s = blockIdx.x / H, h = blockIdx.x - s * H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* base = qkv + (size_t)s * L * D3 + h * 64;
    const bf16_t* obase = out + (size_t)s * L * Dm + h * 64;
    const bf16_t* gbase = dout + (size_t)s * L * Dm + h * 64;
#ifdef SIG_ATTN_STAMPS
    unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0;
    ATT_STAMP(ta0);
#endif

    // ---- staging: 144 rows x 8 chunks = 1152 16-B chunks per image, NTH threads (256: 5 sweeps, the last one half empty) ----
    constexpr int SWEEPS = (ATB_ROWS * 8 + NTH - 1) / NTH;
    uint4 lq[SWEEPS], lk[SWEEPS], lv[SWEEPS], lg[SWEEPS], lo[SWEEPS];
    float llse[SWEEPS];
#pragma unroll
    for (int it = 0; it < SWEEPS; ++it) {
        const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
        lq[it] = lk[it] = lv[it] = lg[it] = lo[it] = make_uint4(0, 0, 0, 0);
        llse[it] = 0.f;
        if (r < L) {
            lq[it] = *(const uint4*)(base + (size_t)r * D3 + ch * 8);
            lk[it] = *(const uint4*)(base + (size_t)r * D3 + Dm + ch * 8);
            lv[it] = *(const uint4*)(base + (size_t)r * D3 + 2 * Dm + ch * 8);
            lg[it] = *(const uint4*)(gbase + (size_t)r * Dm + ch * 8);
            lo[it] = *(const uint4*)(obase + (size_t)r * Dm + ch * 8);
            if (ch == 0) llse[it] = lse[((size_t)s * H + h) * L + r];
        }
    }
#pragma unroll
    for (int it = 0; it < SWEEPS; ++it) {
        const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
        if (c < ATB_ROWS * 8) {
            const int off = d_off(r, ch);
            *(uint4*)(sQ + off) = lq[it];
            *(uint4*)(sK + off) = lk[it];
            *(uint4*)(sV + off) = lv[it];
            *(uint4*)(sG + off) = lg[it];
        }
        const uint32_t gw[4] = {lg[it].x, lg[it].y, lg[it].z, lg[it].w}, ow[4] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w};
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) d = dot2_t<DT>(gw[e], ow[e], d);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (ch == 0 && c < ATB_ROWS * 8) {
            sDel[r] = d * 0.125f;
            sLse[r] = r < L ? llse[it] * 1.4426950408889634f : INFINITY;
        }
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int tq = fr >> 2, tp = fr & 3;
    const float scale = 0.125f, c2 = scale * 1.4426950408889634f;
#if defined(SIG_ATTN_STAMPS) && SIG_ATTN_STAMPS != 2
    ATT_STAMP(ta1);
#endif

    // ---- row / column x, step 1: the 4 x 129 raw dot products of 64, spread over all threads (512 threads: one each) ----
    for (int idx = tid; idx < 4 * L; idx += NTH) {
        const int pr = idx / L, r = idx - pr * L;                                          // operand pair, row
        const char* rows_img = pr == 0 ? sQ : pr == 1 ? sG : pr == 2 ? sK : sV;            // row r of this image ...
        const char* vec_img = pr == 0 ? sK : pr == 1 ? sV : pr == 2 ? sQ : sG;             // ... against row x of this one
        float* dst = pr == 0 ? colS : pr == 1 ? colD : pr == 2 ? rowS : rowD;
        float a0 = 0.f, a1 = 0.f;             // (two chains; 32 packed two-element dot products in all)
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const uint4 w4 = *(const uint4*)(rows_img + d_off(r, ch)), x4 = *(const uint4*)(vec_img + d_off(XR, ch));
            a0 = dot2_t<DT>(w4.x, x4.x, a0);
            a1 = dot2_t<DT>(w4.y, x4.y, a1);
            a0 = dot2_t<DT>(w4.z, x4.z, a0);
            a1 = dot2_t<DT>(w4.w, x4.w, a1);
        }
        dst[r] = a0 + a1;
    }
    __syncthreads();
    // ---- step 2: probabilities and dS of column x (per query i) and row x (per key j), in place ----
    if (tid < L) {
        const float lx = sLse[XR], dx = sDel[XR];
        const float pc = __builtin_amdgcn_exp2f(__builtin_fmaf(colS[tid], c2, -sLse[tid]));
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(rowS[tid], c2, -lx));
        const float dc = pc * __builtin_fmaf(colD[tid], scale, -sDel[tid]);
        const float dr = pr * __builtin_fmaf(rowD[tid], scale, -dx);
        colS[tid] = pc; colD[tid] = dc; rowS[tid] = pr; rowD[tid] = dr;
    }
    __syncthreads();
    // ---- step 3: the three outputs of row x -- dQ[x] = sum_j dS[x,j] K[j], dK[x] = sum_i dS[i,x] Q[i], dV[x] = sum_i P[i,x] dO[i]
    //      (the (x, x) term is inside: column x and row x meet there) -- one per wave, on the matrix cores: the transposed image
    //      fragments of passes A / B against a B operand whose ONLY non-zero column is the weight vector (lanes fr == 0), five
    //      32-deep steps (the fifth holds row 128 alone); the weights are rounded to the operand type like every other row's.
    if (wave < 3) {
        const float* wv = wave == 0 ? rowD : wave == 1 ? colD : colS;
        const char* img = wave == 0 ? sK : wave == 1 ? sQ : sG;
        f32x4_t ox[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ox[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < ATX_NT / 2 + 1; ++kk) {
            // hardware k index 8g + j <-> row 32kk + (j < 4 ? 4g + j : 16 + 4g + j - 4)
            union { uint32_t w[4]; bf16x8_t v; } pk;
            pk.w[0] = pk.w[1] = pk.w[2] = pk.w[3] = 0u;
            if (fr == 0) {
                float lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k0 = 32 * kk + 4 * g + j, k1 = k0 + 16;
                    lo[j] = k0 < L ? wv[k0] : 0.f;
                    hi[j] = k1 < L ? wv[k1] : 0.f;
                }
                pk.w[0] = pack2_t<DT>(lo[0], lo[1]); pk.w[1] = pack2_t<DT>(lo[2], lo[3]);
                pk.w[2] = pack2_t<DT>(hi[0], hi[1]); pk.w[3] = pack2_t<DT>(hi[2], hi[3]);
            }
            const int r0 = 32 * kk + 4 * g + tq, r1 = r0 + 16 < ATB_ROWS ? r0 + 16 : ATB_ROWS - 1;      // (rows >= 129 carry weight 0)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(img + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(img + d_off(r1, chunk) + ((tp & 1) << 3));
                const bf16x8_t aT = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                ox[dt] = mfma16<DT>(aT, pk.v, ox[dt]);
            }
        }
        if (fr == 0) {      // lane (fr = 0, g) holds columns dt*16 + 4g .. +3 of the one live output column
            bf16_t* orow = dqkv + ((size_t)s * L + XR) * D3 + wave * Dm + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(orow + dt * 16) = make_uint2(pack2_t<DT>(ox[dt][0], ox[dt][1]), pack2_t<DT>(ox[dt][2], ox[dt][3]));
        }
    }

#if defined(SIG_ATTN_STAMPS) && SIG_ATTN_STAMPS == 2      // (variant: second stamp after the row-x steps instead of after staging)
    ATT_STAMP(ta1);
#endif
    // ------------------------------ pass A: dQ of rows 0..127 ------------------------------
    // A wave owns query tiles (wave, wave + 4) and walks the keys ONCE for both: every K / V fragment and every transposed K
    // fragment read from LDS feeds two MFMAs instead of one (the arithmetic phases sit on LDS reads, profiles/r03_experiments.md).
    {
        int qa[PAIR];
#pragma unroll
        for (int t = 0; t < PAIR; ++t) qa[t] = (wave + t * NW) * 16 + fr;
        bf16x8_t qf[PAIR][2], gf[PAIR][2];
        float lq2[PAIR], dqs[PAIR];
#pragma unroll
        for (int t = 0; t < PAIR; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[t][ks] = *(const bf16x8_t*)(sQ + d_off(qa[t], (ks << 2) | g));
                gf[t][ks] = *(const bf16x8_t*)(sG + d_off(qa[t], (ks << 2) | g));
            }
            lq2[t] = sLse[qa[t]];
            dqs[t] = sDel[qa[t]];
        }
        f32x4_t ds[PAIR][ATX_NT];
#pragma unroll
        for (int kt = 0; kt < ATX_NT; ++kt) {
            bf16x8_t kf[2], vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[ks] = *(const bf16x8_t*)(sK + d_off(kt * 16 + fr, (ks << 2) | g));
                vf[ks] = *(const bf16x8_t*)(sV + d_off(kt * 16 + fr, (ks << 2) | g));
            }
#pragma unroll
            for (int t = 0; t < PAIR; ++t) {
                f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    a = mfma16<DT>(kf[ks], qf[t][ks], a);  // S^T
                    b = mfma16<DT>(vf[ks], gf[t][ks], b);  // dP^T
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(a[e], c2, -lq2[t]));
                    a[e] = pv * __builtin_fmaf(b[e], scale, -dqs[t]);
                }
                ds[t][kt] = a;
            }
        }
        // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
        f32x4_t o[PAIR][4];
#pragma unroll
        for (int t = 0; t < PAIR; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < ATX_NT / 2; ++kk) {
            bf16x8_t pf[PAIR];
#pragma unroll
            for (int t = 0; t < PAIR; ++t) {
                const f32x4_t p0 = ds[t][2 * kk], p1 = ds[t][2 * kk + 1];
                union { uint32_t w[4]; bf16x8_t v; } pk;
                pk.w[0] = pack2_t<DT>(p0[0], p0[1]); pk.w[1] = pack2_t<DT>(p0[2], p0[3]);
                pk.w[2] = pack2_t<DT>(p1[0], p1[1]); pk.w[3] = pack2_t<DT>(p1[2], p1[3]);
                pf[t] = pk.v;
            }
            const int r0 = 32 * kk + 4 * g + tq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(sK + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(sK + d_off(r0 + 16, chunk) + ((tp & 1) << 3));
                const bf16x8_t kT = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int t = 0; t < PAIR; ++t) o[t][dt] = mfma16<DT>(kT, pf[t], o[t][dt]);
            }
        }
        // + dS[q, x] k_x: lane (fr, g) holds columns dt*16 + 4g .. +3 of query fr
        {
            float dsx[PAIR];
#pragma unroll
            for (int t = 0; t < PAIR; ++t) dsx[t] = colD[qa[t]];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint2 m2 = *(const uint2*)(sK + d_off(XR, 2 * dt + (g >> 1)) + ((g & 1) << 3));
                const float kx[4] = {cvt16f_t<DT>((bf16_t)(m2.x & 0xffff)), cvt16f_t<DT>((bf16_t)(m2.x >> 16)),
                                     cvt16f_t<DT>((bf16_t)(m2.y & 0xffff)), cvt16f_t<DT>((bf16_t)(m2.y >> 16))};
#pragma unroll
                for (int t = 0; t < PAIR; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[t][dt][e] = __builtin_fmaf(dsx[t], kx[e], o[t][dt][e]);
            }
        }
#pragma unroll
        for (int t = 0; t < PAIR; ++t) store_rows16<DT>(o[t], dqkv + ((size_t)s * L + qa[t]) * D3 + h * 64, true, g);
    }

#ifdef SIG_ATTN_STAMPS
    ATT_STAMP(ta2);
#endif
    // ------------------------------ pass B: dK, dV of rows 0..127 ------------------------------
    // the wave's two key tiles (wave, wave + 4) walk the queries together: Q / dO fragments, lse / delta and the transposed
    // dO / Q fragments are read once for both
    {
        int ka[PAIR];
#pragma unroll
        for (int t = 0; t < PAIR; ++t) ka[t] = (wave + t * NW) * 16 + fr;
        bf16x8_t kf[PAIR][2], vf[PAIR][2];
#pragma unroll
        for (int t = 0; t < PAIR; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[t][ks] = *(const bf16x8_t*)(sK + d_off(ka[t], (ks << 2) | g));
                vf[t][ks] = *(const bf16x8_t*)(sV + d_off(ka[t], (ks << 2) | g));
            }
        f32x4_t dk[PAIR][4], dv[PAIR][4];
#pragma unroll
        for (int t = 0; t < PAIR; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dk[t][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                dv[t][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 2      // (fully unrolled the four accumulator sets and four k-steps' fragments spill)
        for (int qq = 0; qq < ATX_NT / 2; ++qq) {  // query tiles (2qq, 2qq+1) = one 32-deep k-step
            uint32_t pw[PAIR][4], sw[PAIR][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * qq + half;
                bf16x8_t qf[2], gf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qf[ks] = *(const bf16x8_t*)(sQ + d_off(qt * 16 + fr, (ks << 2) | g));
                    gf[ks] = *(const bf16x8_t*)(sG + d_off(qt * 16 + fr, (ks << 2) | g));
                }
                const f32x4_t l4 = *(const f32x4_t*)(sLse + qt * 16 + 4 * g), d4 = *(const f32x4_t*)(sDel + qt * 16 + 4 * g);
#pragma unroll
                for (int t = 0; t < PAIR; ++t) {
                    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        a = mfma16<DT>(qf[ks], kf[t][ks], a);  // S  [row q][col key]
                        b = mfma16<DT>(gf[ks], vf[t][ks], b);  // dP [row q][col key]
                    }
                    float pv[4], dsv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[e], c2, -l4[e]));
                        dsv[e] = pv[e] * __builtin_fmaf(b[e], scale, -d4[e]);
                    }
                    pw[t][half * 2] = pack2_t<DT>(pv[0], pv[1]); pw[t][half * 2 + 1] = pack2_t<DT>(pv[2], pv[3]);
                    sw[t][half * 2] = pack2_t<DT>(dsv[0], dsv[1]); sw[t][half * 2 + 1] = pack2_t<DT>(dsv[2], dsv[3]);
                }
            }
            bf16x8_t pf[PAIR], sf[PAIR];
#pragma unroll
            for (int t = 0; t < PAIR; ++t) {
                union { uint32_t w[4]; bf16x8_t v; } pu, su;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pu.w[e] = pw[t][e]; su.w[e] = sw[t][e]; }
                pf[t] = pu.v; sf[t] = su.v;
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
            const int r0 = 32 * qq + 4 * g + tq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t g0 = lds_tr16(sG + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t g1 = lds_tr16(sG + d_off(r0 + 16, chunk) + ((tp & 1) << 3));
                const bf16x4_t q0 = lds_tr16(sQ + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t q1 = lds_tr16(sQ + d_off(r0 + 16, chunk) + ((tp & 1) << 3));
                const bf16x8_t gT = (bf16x8_t){g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const bf16x8_t qT = (bf16x8_t){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
                for (int t = 0; t < PAIR; ++t) {
                    dv[t][dt] = mfma16<DT>(gT, pf[t], dv[t][dt]);
                    dk[t][dt] = mfma16<DT>(qT, sf[t], dk[t][dt]);
                }
            }
        }
        // + dS[x, key] q_x and P[x, key] dO_x: lane (fr, g) holds columns dt*16 + 4g .. +3 of key fr
        {
            float rds[PAIR], rp[PAIR];
#pragma unroll
            for (int t = 0; t < PAIR; ++t) { rds[t] = rowD[ka[t]]; rp[t] = rowS[ka[t]]; }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int off = d_off(XR, 2 * dt + (g >> 1)) + ((g & 1) << 3);
                const uint2 q2 = *(const uint2*)(sQ + off), g2 = *(const uint2*)(sG + off);
                const float qx[4] = {cvt16f_t<DT>((bf16_t)(q2.x & 0xffff)), cvt16f_t<DT>((bf16_t)(q2.x >> 16)),
                                     cvt16f_t<DT>((bf16_t)(q2.y & 0xffff)), cvt16f_t<DT>((bf16_t)(q2.y >> 16))};
                const float gx[4] = {cvt16f_t<DT>((bf16_t)(g2.x & 0xffff)), cvt16f_t<DT>((bf16_t)(g2.x >> 16)),
                                     cvt16f_t<DT>((bf16_t)(g2.y & 0xffff)), cvt16f_t<DT>((bf16_t)(g2.y >> 16))};
#pragma unroll
                for (int t = 0; t < PAIR; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dk[t][dt][e] = __builtin_fmaf(rds[t], qx[e], dk[t][dt][e]);
                        dv[t][dt][e] = __builtin_fmaf(rp[t], gx[e], dv[t][dt][e]);
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < PAIR; ++t) {
            bf16_t* krow = dqkv + ((size_t)s * L + ka[t]) * D3 + Dm + h * 64;
            store_rows16<DT>(dk[t], krow, true, g);
            store_rows16<DT>(dv[t], krow + Dm, true, g);
        }
    }
#ifdef SIG_ATTN_STAMPS      // tools/attn_stamps.py: staging | row-x steps + pass A | pass B + store drain (wave 0's view; =2: staging + row-x steps | pass A)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP(ta3);
    if (tid == 0 && blockIdx.x < 4096) {
        g_astamps[blockIdx.x * 4 + 0] = ta0; g_astamps[blockIdx.x * 4 + 1] = ta1;
        g_astamps[blockIdx.x * 4 + 2] = ta2; g_astamps[blockIdx.x * 4 + 3] = ta3;
    }
#endif
}

// waves per block of the L = 129 backward: 8 (one tile per wave, default) or 4 (two tiles per wave); SIG_ATTN_BWD_WAVES / sig_tune_attn_bwd_waves
static std::atomic<int> g_attn_bwd_waves{-1};
static int attn_bwd_waves() {
    int v = g_attn_bwd_waves.load();
    if (v < 0) {
        const char* e = getenv("SIG_ATTN_BWD_WAVES");
        v = e ? (atoi(e) == 4 ? 4 : 8) : (SIG_ATTN_BWD_WAVES8_DEFAULT ? 8 : 4);
        g_attn_bwd_waves = v;
    }
    return v;
}
int sig_tune_attn_bwd_waves_impl(int waves) {
    const int prev = attn_bwd_waves();      // (resolves the environment preset first: restoring `prev` keeps it)
    g_attn_bwd_waves = waves == 4 ? 4 : 8;
    return prev;
}

template <int DT>
static void launch_attn_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int S, int L,
                            int H, hipStream_t st) {
    const int lds = 4 * ATB_ROWS * 128 + 2 * 160 * 4;
    static std::once_flag attr_done;
    std::call_once(attr_done, [] {
        (void)hipFuncSetAttribute((const void*)&attn_bwd_kernel<true, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)&attn_bwd_kernel<false, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
    static int x1 = -1;       // SIG_ATTN_BWD_X1=0: the nine-tile kernel also at L = 129 (A/B runs)
    if (x1 < 0) { const char* e = getenv("SIG_ATTN_BWD_X1"); x1 = e ? atoi(e) : 1; }
    if (x1 && L == 16 * ATX_NT + 1) {
        const int ldsx = lds + 4 * 132 * 4;
        static std::once_flag attr_x1;
        std::call_once(attr_x1, [ldsx] {
            (void)hipFuncSetAttribute((const void*)&attn_bwd_x1_kernel<DT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsx);
            (void)hipFuncSetAttribute((const void*)&attn_bwd_x1_kernel<DT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsx);
            });
        if (attn_bwd_waves() == 8) hipLaunchKernelGGL((attn_bwd_x1_kernel<DT, 1>), dim3(S * H), dim3(512), ldsx, st, qkv, out, dout, lse, dqkv, S, H);
        else hipLaunchKernelGGL((attn_bwd_x1_kernel<DT, 2>), dim3(S * H), dim3(256), ldsx, st, qkv, out, dout, lse, dqkv, S, H);
        return;
    }
    if (L > 16 * (ATT_NT - 1)) hipLaunchKernelGGL((attn_bwd_kernel<true, DT>), dim3(S * H), dim3(192), lds, st, qkv, out, dout, lse, dqkv, S, L, H);
    else hipLaunchKernelGGL((attn_bwd_kernel<false, DT>), dim3(S * H), dim3(192), lds, st, qkv, out, dout, lse, dqkv, S, L, H);
}
int sig_launch_attn_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv,
                        int S, int L, int H, int dt, hipStream_t st) {
    SIG_CHECK_DT(dt, "attn_bwd");
    SIG_CHECK_ARG(qkv && out && dout && lse && dqkv, "attn_bwd: null pointer");
    SIG_CHECK_ARG(S > 0 && H > 0 && L > 0 && L <= ATT_KROWS, "attn_bwd: L=%d must be in 1..%d", L, ATT_KROWS);
    if (dt == SIG_DT_F16) launch_attn_bwd<SIG_DT_F16>(qkv, out, dout, lse, dqkv, S, L, H, st);
    else launch_attn_bwd<SIG_DT_BF16>(qkv, out, dout, lse, dqkv, S, L, H, st);
    SIG_CHECK_LAUNCH("attn_bwd");
    return 0;
}
