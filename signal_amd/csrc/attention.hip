// Self-attention of the CLIP ViT block (clip/model.py:223-225): softmax(q k^T / 8) v per (sequence, head),
// L <= 144 tokens (129 in every shipped config), head dim 64, no mask, no dropout.
//
// One workgroup (3 waves) per (sequence, head).  K and V of the head (<= 18 KB each in bf16) live in LDS
// for the whole block; q fragments come straight from HBM.  Scores are formed TRANSPOSED (S^T = K q^T,
// key on the accumulator row, query on the lane) so that after the in-register softmax the probability
// tile already is the B operand of O^T = V^T P^T -- no LDS round trip for P.  V^T fragments are fetched with
// ds_read_b64_tr_b16 from the row-major V image.  Row max / sum are two xor-shuffles across the 4 lane
// groups that share a query.
#include "sig_common.h"
#include "sig_kernels.h"

#define ATT_KROWS 144
#define ATT_VROWS 160
#define ATT_NT 9  // max 16-row tiles

// K image: 128-B rows, physical chunk = chunk ^ ((row >> 1) & 7)          (row reads, ds_read_b128)
// V image: 128-B rows, physical chunk = chunk ^ (((row >> 1) & 3) << 1)   (transposed reads, 32-B blocks)
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

__global__ __launch_bounds__(192) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int S, int L, int H) {
    __shared__ __attribute__((aligned(16))) char smem[ATT_KROWS * 128 + ATT_VROWS * 128];
    char* sK = smem;
    char* sV = smem + ATT_KROWS * 128;
    const int s = blockIdx.x / H, h = blockIdx.x - s * H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* base = qkv + (size_t)s * L * D3 + h * 64;

    for (int c = tid; c < ATT_KROWS * 8; c += 192) {
        const int r = c >> 3, ch = c & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < L) v = *(const uint4*)(base + (size_t)r * D3 + Dm + ch * 8);
        *(uint4*)(sK + k_off(r, ch)) = v;
    }
    for (int c = tid; c < ATT_VROWS * 8; c += 192) {
        const int r = c >> 3, ch = c & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < L) v = *(const uint4*)(base + (size_t)r * D3 + 2 * Dm + ch * 8);
        *(uint4*)(sV + v_off(r, ch)) = v;
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int NT = (L + 15) >> 4;
    const int tq = fr >> 2, tp = fr & 3;  // transposed-read address roles inside the 16-lane group
    const float scale = 0.125f;

    for (int qt = wave; qt < NT; qt += 3) {
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
            if (kt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + k_off(kt * 16 + fr, (ks << 2) | g));
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], a, 0, 0, 0);
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
            if (kk * 2 >= NT) break;
            const f32x4_t p0 = sc[2 * kk], p1 = sc[2 * kk + 1];
            union { uint32_t w[4]; bf16x8_t v; } pk;
            pk.w[0] = pack2bf(p0[0], p0[1]); pk.w[1] = pack2bf(p0[2], p0[3]);
            pk.w[2] = pack2bf(p1[0], p1[1]); pk.w[3] = pack2bf(p1[2], p1[3]);
            const bf16x8_t pf = pk.v;
            const int r0 = 32 * kk + 4 * g + tq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(sV + v_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(sV + v_off(r0 + 16, chunk) + ((tp & 1) << 3));
                const bf16x8_t vf = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        // lane (fr, g) holds head columns dt*16 + 4g .. +3 of query fr: exchange lane pairs g / g^1 (v_permlane16_swap, as
        // in the GEMM epilogue) so that a lane owns 8 consecutive columns -> two 16-B stores instead of four 8-B ones
        {
            const float inv = 1.0f / sum;
            uint32_t w[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                w[dt][0] = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
                w[dt][1] = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
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
}

int sig_launch_attn_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int S, int L, int H, hipStream_t st) {
    SIG_CHECK_ARG(qkv && out, "attn_fwd: null pointer");
    SIG_CHECK_ARG(S > 0 && H > 0 && L > 0 && L <= ATT_KROWS, "attn_fwd: L=%d must be in 1..%d", L, ATT_KROWS);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(S * H), dim3(192), 0, st, qkv, out, lse, S, L, H);
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

#define ATB_ROWS 144  // 9 tiles; reads past it are clamped (they only ever meet zero probabilities)
__global__ __launch_bounds__(192) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                       const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                       bf16_t* __restrict__ dqkv, int S, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;
    char* sK = smem + ATB_ROWS * 128;
    char* sV = smem + 2 * ATB_ROWS * 128;
    char* sG = smem + 3 * ATB_ROWS * 128;                 // dO
    float* sLse = (float*)(smem + 4 * ATB_ROWS * 128);    // [160]
    float* sDel = sLse + ATB_ROWS;                        // [160] delta = rowsum(dO * O)
    const int s = blockIdx.x / H, h = blockIdx.x - s * H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* base = qkv + (size_t)s * L * D3 + h * 64;
    const bf16_t* obase = out + (size_t)s * L * Dm + h * 64;
    const bf16_t* gbase = dout + (size_t)s * L * Dm + h * 64;

    // 192 threads = 24 rows x 8 chunks per sweep; delta via an 8-lane xor reduction
    for (int c = tid; c < ATB_ROWS * 8; c += 192) {
        const int r = c >> 3, ch = c & 7;
        uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq, vg = vq, vo = vq;
        if (r < L) {
            vq = *(const uint4*)(base + (size_t)r * D3 + ch * 8);
            vk = *(const uint4*)(base + (size_t)r * D3 + Dm + ch * 8);
            vv = *(const uint4*)(base + (size_t)r * D3 + 2 * Dm + ch * 8);
            vg = *(const uint4*)(gbase + (size_t)r * Dm + ch * 8);
            vo = *(const uint4*)(obase + (size_t)r * Dm + ch * 8);
        }
        const int off = d_off(r, ch);
        *(uint4*)(sQ + off) = vq;
        *(uint4*)(sK + off) = vk;
        *(uint4*)(sV + off) = vv;
        *(uint4*)(sG + off) = vg;
        const uint32_t gw[4] = {vg.x, vg.y, vg.z, vg.w}, ow[4] = {vo.x, vo.y, vo.z, vo.w};
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d += bf2f((bf16_t)(gw[e] & 0xffff)) * bf2f((bf16_t)(ow[e] & 0xffff));
            d += bf2f((bf16_t)(gw[e] >> 16)) * bf2f((bf16_t)(ow[e] >> 16));
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (ch == 0) {
            sDel[r] = d;
            sLse[r] = r < L ? lse[((size_t)s * H + h) * L + r] : 0.f;
        }
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int NT = (L + 15) >> 4;
    const int tq = fr >> 2, tp = fr & 3;
    const float scale = 0.125f;

    // ------------------------------ pass A: dQ (wave owns query tiles) ------------------------------
    for (int qt = wave; qt < NT; qt += 3) {
        const int q = qt * 16 + fr;
        bf16x8_t qf[2], gf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *(const bf16x8_t*)(sQ + d_off(q, (ks << 2) | g));
            gf[ks] = *(const bf16x8_t*)(sG + d_off(q, (ks << 2) | g));
        }
        const float lq = sLse[q], dq_ = sDel[q];
        f32x4_t ds[ATT_NT + 1];
#pragma unroll
        for (int kt = 0; kt < ATT_NT; ++kt) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (kt < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + d_off(kt * 16 + fr, (ks << 2) | g));
                    const bf16x8_t vf = *(const bf16x8_t*)(sV + d_off(kt * 16 + fr, (ks << 2) | g));
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], a, 0, 0, 0);  // S^T
                    b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[ks], b, 0, 0, 0);  // dP^T
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt * 16 + 4 * g + e;
                const float pv = (key < L && q < L) ? __expf(a[e] * scale - lq) : 0.f;
                a[e] = pv * (b[e] - dq_) * scale;
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
            if (kk * 2 >= NT) break;
            const f32x4_t p0 = ds[2 * kk], p1 = ds[2 * kk + 1];
            bf16x8_t pf;
            pf[0] = (short)f2bf(p0[0]); pf[1] = (short)f2bf(p0[1]); pf[2] = (short)f2bf(p0[2]); pf[3] = (short)f2bf(p0[3]);
            pf[4] = (short)f2bf(p1[0]); pf[5] = (short)f2bf(p1[1]); pf[6] = (short)f2bf(p1[2]); pf[7] = (short)f2bf(p1[3]);
            const int r0 = 32 * kk + 4 * g + tq;
            const int r1 = r0 + 16 < ATB_ROWS ? r0 + 16 : r0;  // tile 9 does not exist: its dS is 0
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = 2 * dt + (tp >> 1);
                const bf16x4_t v0 = lds_tr16(sK + d_off(r0, chunk) + ((tp & 1) << 3));
                const bf16x4_t v1 = lds_tr16(sK + d_off(r1, chunk) + ((tp & 1) << 3));
                const bf16x8_t kf = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, pf, o[dt], 0, 0, 0);
            }
        }
        if (q < L) {
            bf16_t* orow = dqkv + ((size_t)s * L + q) * D3 + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(orow + dt * 16) = make_uint2(pack2bf(o[dt][0], o[dt][1]), pack2bf(o[dt][2], o[dt][3]));
        }
    }

    // ------------------------------ pass B: dK, dV (wave owns key tiles) ------------------------------
    for (int kt = wave; kt < NT; kt += 3) {
        const int key = kt * 16 + fr;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = *(const bf16x8_t*)(sK + d_off(key, (ks << 2) | g));
            vf[ks] = *(const bf16x8_t*)(sV + d_off(key, (ks << 2) | g));
        }
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qq = 0; qq < (ATT_NT + 1) / 2; ++qq) {  // query tiles (2qq, 2qq+1) = one 32-deep k-step
            if (qq * 2 >= NT) break;
            bf16x8_t pf, sf;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * qq + half;
                f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                if (qt < NT) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8_t qf = *(const bf16x8_t*)(sQ + d_off(qt * 16 + fr, (ks << 2) | g));
                        const bf16x8_t gf = *(const bf16x8_t*)(sG + d_off(qt * 16 + fr, (ks << 2) | g));
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], a, 0, 0, 0);  // S  [row q][col key]
                        b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf[ks], b, 0, 0, 0);  // dP [row q][col key]
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qt * 16 + 4 * g + e;
                    const bool ok = (q < L) && (key < L) && (qt < NT);
                    const float pv = ok ? __expf(a[e] * scale - sLse[ok ? q : 0]) : 0.f;
                    const float dsv = pv * (b[e] - sDel[ok ? q : 0]) * scale;
                    pf[half * 4 + e] = (short)f2bf(pv);
                    sf[half * 4 + e] = (short)f2bf(dsv);
                }
            }
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
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT, pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, sf, dk[dt], 0, 0, 0);
            }
        }
        if (key < L) {
            bf16_t* krow = dqkv + ((size_t)s * L + key) * D3 + Dm + h * 64 + 4 * g;
            bf16_t* vrow = krow + Dm;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(uint2*)(krow + dt * 16) = make_uint2(pack2bf(dk[dt][0], dk[dt][1]), pack2bf(dk[dt][2], dk[dt][3]));
                *(uint2*)(vrow + dt * 16) = make_uint2(pack2bf(dv[dt][0], dv[dt][1]), pack2bf(dv[dt][2], dv[dt][3]));
            }
        }
    }
}

int sig_launch_attn_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv,
                        int S, int L, int H, hipStream_t st) {
    SIG_CHECK_ARG(qkv && out && dout && lse && dqkv, "attn_bwd: null pointer");
    SIG_CHECK_ARG(S > 0 && H > 0 && L > 0 && L <= ATT_KROWS, "attn_bwd: L=%d must be in 1..%d", L, ATT_KROWS);
    const int lds = 4 * ATB_ROWS * 128 + 2 * ATB_ROWS * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(S * H), dim3(192), lds, st, qkv, out, dout, lse, dqkv, S, L, H);
    SIG_CHECK_LAUNCH("attn_bwd");
    return 0;
}
