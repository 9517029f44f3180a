// bf16 MFMA GEMMs for the ViT / head contractions (gfx950).
//
//   gemm_nt : C[M,N] = A[M,K] * Bt[N,K]^T      forward (x W^T) and dgrad (dy (W^T)^T, W^T pre-packed)
//   gemm_tn : C[I,J] += P[Mr,I]^T * Q[Mr,J]    wgrad (dy^T x), split over the long row dimension
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves in 2x2, 64x64 per wave), K-step 64,
// operands staged global->LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane) into two buffers so
// the load of step k+1 flies under the MFMAs of step k; LDS images are XOR-swizzled through the
// per-lane SOURCE address (the DMA destination is lane-linear) so fragment reads are conflict-free.
// fp32 accumulation; epilogues fuse bias / residual / QuickGELU / QuickGELU' so the [M, 768..3072]
// activations make one HBM round trip per contraction.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "sig_common.h"
#include "sig_kernels.h"

// ------------------------------------------------------------------------------------------------
// NT
// ------------------------------------------------------------------------------------------------
// LDS image of one operand tile: 128 rows x 64 k (bf16) = 128 B per row, 8 chunks of 16 B.
// physical chunk = chunk ^ ((row >> 1) & 7): a ds_read_b128 lane group (rows r..r+15 at chunk c and
// c+1, MI355X_MICROARCH LDS table) then covers all 16 slots of the 256-B bank row exactly once.
// sigmoid(1.702 u) = 1 / (1 + 2^(-1.702 log2(e) u)): one multiply, v_exp_f32, one add, v_rcp_f32
__device__ __forceinline__ float quick_sigmoid_f(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -2.4554669595930156f));
}
__device__ __forceinline__ float quick_gelu_f(float u) { return u * quick_sigmoid_f(u); }
__device__ __forceinline__ float quick_gelu_grad_f(float u) {
    const float s = quick_sigmoid_f(u);
    return s * (1.0f + 1.702f * u * (1.0f - s));
}

__device__ __forceinline__ float gelu_erf_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}

// ------------------------------------------------------------------------------------------------
// Register-only epilogue (no LDS).  With the swapped MFMA operands lane (fr = lane & 15, g = lane >> 4) holds, for
// 16-row group i and 16-column group j, the 4 CONSECUTIVE columns j*16 + g*4 .. +3 of row i*16 + fr.
//  * everything elementwise (bias, residual, GELU, GELU') happens in that layout;
//  * f32 outputs are stored from it directly: 16 B per lane, 64 B runs per row and instruction;
//  * bf16 outputs are packed (2 dwords per (i, j)) and exchanged between lanes g and g^1 with v_permlane16_swap_b32
//    (gfx950; tools/micro/permlane16_swap.hip): for a column-group pair (2jp, 2jp+1) the even-g lane ends up with columns
//    (2jp)*16 + (g>>1)*8 .. +7 and the odd-g lane with (2jp+1)*16 + (g>>1)*8 .. +7 -> one 16-B store per lane, the
//    same store count as a transposed tile but without ~2 x tile bytes through ds_write_b128 (79 B/clk/CU);
//  * the saved pre-activation u of the GELU' epilogues is loaded 16 B wide in the swapped layout and un-swapped (the
//    exchange is an involution).
// Rows >= M are computed (the swaps need every lane) but neither loaded, stored nor summed.
// ------------------------------------------------------------------------------------------------
// Output stores of the epilogues go through a buffer descriptor so that they can carry a cache policy (aux bits of
// buffer_store: 16 = sc1, 2 = nt, 0 = default).  A tile's output (128-256 KB per 256x256 tile, never re-read by this
// kernel) otherwise write-allocates in the XCD's 4 MB L2 next to the operand panels the resident blocks keep re-reading;
// sc1 stores are write-through and DROP the line (MI355X_MICROARCH.md, stores of each flavour).
// Measured per epilogue at M = 24768 (tools/ab_kernels.sh, same box, plain / sc1 / nt in us): qkv (bias, 16-bit out)
// 96.5 / 94.0 / 100; c_fc 148 / 141-152 / 151; GELU' dgrad 152 / 160 / 139; f32 + residual outputs 50.5 / 66 / 51 and
// 118 / 133 / 120 (sc1 doubles WRITE_SIZE there: the residual lines are read and rewritten).  In the model (tools/
// ab_profile.sh, train step) qkv keeps its gain (13.5 -> 13.3 ms per 144 launches) but the GELU' dgrad LOSES with nt (25.5
// -> 26.4: its saved pre-activations come from HBM there, not from the Infinity Cache as in the micro-benchmark loop).
// So: sc1 for the plain 16-bit projection outputs, default for everything else.  L2-miss traffic of c_fc moved only
// 187 -> 156 MB: the operand re-fetches come from sibling blocks drifting apart in time, not from output pollution, and
// are served by the 256 MB Infinity Cache (A + W = 43 MB).
#ifdef SIG_STORE_AUX
template <int EPI> constexpr int store_aux() { return SIG_STORE_AUX; }      // A/B builds: one policy everywhere
#else
template <int EPI> constexpr int store_aux() { return EPI == SIG_EPI_BIAS_BF16 ? 16 : 0; }
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);   // raw buffer, byte offsets, no swizzle
}
template <int AUX>
__device__ __forceinline__ void store16_policy(__amdgpu_buffer_rsrc_t r, void* base, size_t elem_off_bytes, uint4 v) {
    if constexpr (AUX == 0) {
        *(uint4*)((char*)base + elem_off_bytes) = v;
    } else {
        const u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)elem_off_bytes, 0, AUX);
    }
}

__device__ __forceinline__ void lane_swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// ULDS (256x256 GELU' dgrad only): the saved pre-activations of row groups i < 6 were staged in LDS under the main loop
// (gemm_nt256_kernel, "u prefetch"); only the last two row groups are loaded from global memory here.
//   u_r0: rows i = 0, 1 of both wave rows   [64 region rows x 512 B], region row = (wm ? 32 : 0) + 16 i + fr
//   u_r1: rows i = 2..5                     [128 region rows x 512 B], region row = (wm ? 64 : 0) + 16 (i - 2) + fr
// 16-B chunk c of a row lives at chunk c ^ (row & 15) (the DMA applied the same XOR to its source address).
template <int EPI, int TM, int TN, int DT, bool ULDS = false>
__device__ __forceinline__ void epilogue_regs(const SigGemmNT& p, f32x4_t (&acc)[TM][TN], int m_base, int n_base, int lane,
                                              const char* u_r0 = nullptr, const char* u_r1 = nullptr, int wm = 0, int wn = 0) {
    static_assert(TN % 2 == 0, "column groups are exchanged in pairs");
    constexpr bool HAS_BIAS = EPI == SIG_EPI_BIAS_BF16 || EPI == SIG_EPI_BIAS_F32 || EPI == SIG_EPI_BIAS_RES_F32 ||
                              EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_BIAS_GELUERF_BF16;
    constexpr bool HAS_RES = EPI == SIG_EPI_BIAS_RES_F32 || EPI == SIG_EPI_RES_F32;
    constexpr bool OUT_F32 = EPI == SIG_EPI_F32 || EPI == SIG_EPI_BIAS_F32 || HAS_RES;
    constexpr bool GELU_FWD = EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_BIAS_GELUERF_BF16;
    constexpr bool GELU_BWD = EPI == SIG_EPI_DGELU_BF16 || EPI == SIG_EPI_DGELUERF_BF16;
    constexpr bool QUICK = EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_DGELU_BF16;
    const int fr = lane & 15, g = lane >> 4;
    const int nc = n_base + g * 4;                              // + j*16: this lane's 4 columns in the MFMA layout
    const int ns = n_base + (g & 1) * 16 + (g >> 1) * 8;        // + jp*32: its 8 columns after the exchange
    f32x4_t bias4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias4[j] = HAS_BIAS ? *(const f32x4_t*)(p.bias + nc + j * 16) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t csum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool save_u = GELU_FWD && p.aux != nullptr;
    // column sums (bias gradients) are a by-product of backward GEMMs: the 16-bit forward epilogues carry no code for them
    constexpr bool CAN_SUM = EPI != SIG_EPI_BIAS_BF16 && !GELU_FWD;
    const bool do_sum = CAN_SUM && p.colsum != nullptr;
    const __amdgpu_buffer_rsrc_t r_out = out_rsrc(p.out), r_aux = out_rsrc(p.aux);
    // GELU': the whole tile's saved pre-activations are requested before anything is consumed (TM*TN/2 16-B loads per
    // lane, in the registers the operand fragments no longer need) -- inside the store loop their latency was exposed
    // once per 16-row group
    uint4 uq[GELU_BWD ? TM : 1][GELU_BWD ? TN / 2 : 1];
    if (GELU_BWD) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (ULDS && i < 6) continue;
            const int m = m_base + i * 16 + fr;
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp)
                uq[i][jp] = *(const uint4*)((const bf16_t*)p.aux + (size_t)(m < p.M ? m : 0) * p.ldaux + ns + jp * 32);
        }
    }
    // FULL: every row of this wave's tile is < M (all tiles but the last row of tiles): no exec masking around the stores,
    // no per-row selects in the column sums
    auto rows = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_base + i * 16 + fr;
        const bool live = FULL || m < p.M;
        uint32_t pk[TN][2], pu[TN][2];
        if (GELU_BWD) {
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp) {
                uint4 q;
                if (ULDS && i < 6) {
                    const char* reg = i < 2 ? u_r0 + ((wm ? 32 : 0) + i * 16 + fr) * 512 : u_r1 + ((wm ? 64 : 0) + (i - 2) * 16 + fr) * 512;
                    const int c = (wn >> 3) + (g & 1) * 2 + (g >> 1) + jp * 4;
                    q = *(const uint4*)(reg + ((c ^ fr) << 4));
                } else {
                    q = uq[GELU_BWD ? i : 0][GELU_BWD ? jp : 0];
                }
                lane_swap16(q.x, q.z);
                lane_swap16(q.y, q.w);
                pu[2 * jp][0] = q.x; pu[2 * jp][1] = q.y;
                pu[2 * jp + 1][0] = q.z; pu[2 * jp + 1][1] = q.w;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4_t x = acc[i][j] + bias4[j];
            if (HAS_RES && live) x += *(const f32x4_t*)(p.res + (size_t)m * p.ldr + nc + j * 16);
            if (GELU_FWD) {
                if constexpr (QUICK) {
                    // what backward needs of the pre-activation is only QuickGELU'(u): it is saved INSTEAD of u (16-bit
                    // either way), so the GELU' dgrad epilogue is one multiply per element instead of exp + rcp + 5
                    f32x4_t sg, t = x * -2.4554669595930156f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
                    t += 1.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) sg[e] = __builtin_amdgcn_rcpf(t[e]);
                    if (save_u) {
                        const f32x4_t a = x * 1.702f;
                        const f32x4_t d = sg + sg * (a - a * sg);          // s (1 + 1.702 u (1 - s))
                        pu[j][0] = pack2_t<DT>(d[0], d[1]);
                        pu[j][1] = pack2_t<DT>(d[2], d[3]);
                    }
                    x *= sg;
                } else {
                    pu[j][0] = pack2_t<DT>(x[0], x[1]);
                    pu[j][1] = pack2_t<DT>(x[2], x[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = gelu_erf_f(x[e]);
                }
            }
            if (GELU_BWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t w = pu[j][e >> 1];
                    const float u = cvt16f_t<DT>((bf16_t)((e & 1) ? (w >> 16) : (w & 0xffff)));
                    x[e] *= QUICK ? u : gelu_erf_grad_f(u);                 // QUICK: aux already holds QuickGELU'(u)
                }
            }
            if (OUT_F32) {
                if (live) store16_policy<store_aux<EPI>()>(r_out, p.out, ((size_t)m * p.ldo + nc + j * 16) * 4, __builtin_bit_cast(uint4, x));
            } else {
                pk[j][0] = pack2_t<DT>(x[0], x[1]);
                pk[j][1] = pack2_t<DT>(x[2], x[3]);
            }
            if (do_sum && live) csum[j] += x;   // (uniform flag: forward GEMMs request no column sums)
        }
        if (!OUT_F32) {
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp) {
                lane_swap16(pk[2 * jp][0], pk[2 * jp + 1][0]);
                lane_swap16(pk[2 * jp][1], pk[2 * jp + 1][1]);
                if (live)
                    store16_policy<store_aux<EPI>()>(r_out, p.out, ((size_t)m * p.ldo + ns + jp * 32) * 2,
                                   make_uint4(pk[2 * jp][0], pk[2 * jp][1], pk[2 * jp + 1][0], pk[2 * jp + 1][1]));
            }
        }
        if (GELU_FWD && save_u) {
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp) {
                lane_swap16(pu[2 * jp][0], pu[2 * jp + 1][0]);
                lane_swap16(pu[2 * jp][1], pu[2 * jp + 1][1]);
                if (live)
                    store16_policy<store_aux<EPI>()>(r_aux, p.aux, ((size_t)m * p.ldaux + ns + jp * 32) * 2,
                                   make_uint4(pu[2 * jp][0], pu[2 * jp][1], pu[2 * jp + 1][0], pu[2 * jp + 1][1]));
            }
        }
    }
    };
    // (two copies of the row loop cost registers: the residual / GELU-forward epilogues spill with both, so only the
    // GELU' dgrad -- whose remaining work is the column-sum selects -- gets the split)
    if (EPI == SIG_EPI_DGELU_BF16 && m_base + TM * 16 <= p.M) rows(std::true_type{});
    else rows(std::false_type{});
    // optional bias-gradient by-product: column sums of what was just written
    if (do_sum) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = csum[j][e];
                t += __shfl_xor(t, 1, 64);
                t += __shfl_xor(t, 2, 64);
                t += __shfl_xor(t, 4, 64);
                t += __shfl_xor(t, 8, 64);
                if (fr == 0) atomicAdd(p.colsum + nc + j * 16 + e, t);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Register epilogue in the NATURAL MFMA orientation (D = A-fragment x B-fragment): lane (fr, g) holds, for 16-row group i
// and MFMA column group j, rows i*16 + 4g + e (e = 0..3) of B-slot fr.  The kernel stages the B operand so that slot fr
// of group j is output column 4 fr + j of the wave's 64-column strip (a row permutation of the LDS image, free: it is the
// per-lane source address of the LDS-DMA): the 4 accumulators {acc[i][0..3][e]} of a lane are then 4 CONSECUTIVE columns
// of one row, and the 16 lanes fr = 0..15 of a 16-lane row cover 64 consecutive columns.  One store instruction writes
// 4 rows x (256 B of f32 | 128 B of 16-bit) with consecutive lanes on consecutive addresses: whole cache lines per
// quarter-wave, no lane exchange, no LDS.  Why it matters (tools/micro/store_tail.hip, one 128-KB tile alone on the chip):
// a CU drains lane-consecutive whole lines at 50 B/clk (27 B/clk with 8-B lanes) but only 13.7 B/clk when the lanes of a
// line are scattered over the wave, as in the swapped-operand layout above (16 rows x 64 B per instruction).
// ------------------------------------------------------------------------------------------------
template <int EPI, int TM, int DT>
__device__ __forceinline__ void epilogue_nat(const SigGemmNT& p, f32x4_t (&acc)[TM][4], int m_base, int n_base, int lane) {
    constexpr bool HAS_BIAS = EPI == SIG_EPI_BIAS_BF16 || EPI == SIG_EPI_BIAS_F32 || EPI == SIG_EPI_BIAS_RES_F32 ||
                              EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_BIAS_GELUERF_BF16;
    constexpr bool HAS_RES = EPI == SIG_EPI_BIAS_RES_F32 || EPI == SIG_EPI_RES_F32;
    constexpr bool OUT_F32 = EPI == SIG_EPI_F32 || EPI == SIG_EPI_BIAS_F32 || HAS_RES;
    constexpr bool GELU_FWD = EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_BIAS_GELUERF_BF16;
    constexpr bool GELU_BWD = EPI == SIG_EPI_DGELU_BF16 || EPI == SIG_EPI_DGELUERF_BF16;
    constexpr bool QUICK = EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_DGELU_BF16;
    constexpr bool CAN_SUM = EPI != SIG_EPI_BIAS_BF16 && !GELU_FWD;
    const int fr = lane & 15, g = lane >> 4;
    const int n = n_base + fr * 4;                               // this lane's 4 columns
    const f32x4_t bias4 = HAS_BIAS ? *(const f32x4_t*)(p.bias + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool save_u = GELU_FWD && p.aux != nullptr;
    const bool do_sum = CAN_SUM && p.colsum != nullptr;
    f32x4_t csum = {0.f, 0.f, 0.f, 0.f};
    // Residual rows: `res` may alias `out` (the in-place residual stream), so the compiler keeps every residual load behind the
    // previous row's store -- 40 dependent HBM round trips per wave.  Each lane reads exactly the addresses it writes later, so the
    // loads of RES_AHEAD row groups are requested by hand before the first of their stores (the fragment registers are dead here).
#ifndef SIG_RES_AHEAD
#define SIG_RES_AHEAD 2
#endif
    constexpr int RES_AHEAD = HAS_RES ? SIG_RES_AHEAD : 1;
    f32x4_t rv[RES_AHEAD][4];
    auto res_fetch = [&](int i) {
        if constexpr (HAS_RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m_base + i * 16 + g * 4 + e;
                rv[i % RES_AHEAD][e] = *(const f32x4_t*)(p.res + (size_t)(m < p.M ? m : 0) * p.ldr + n);
            }
        }
    };
    if constexpr (HAS_RES) {
#pragma unroll
        for (int i = 0; i < RES_AHEAD && i < TM; ++i) res_fetch(i);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        uint2 uq[4];
        f32x4_t rcur[4];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rcur[e] = rv[i % RES_AHEAD][e];
            // (the refill for row group i + RES_AHEAD is requested after this group's stores, below)
        }
        if (GELU_BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m_base + i * 16 + g * 4 + e;
                uq[e] = *(const uint2*)((const bf16_t*)p.aux + (size_t)(m < p.M ? m : 0) * p.ldaux + n);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m_base + i * 16 + g * 4 + e;
            const bool live = m < p.M;
            f32x4_t x = (f32x4_t){acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]} + bias4;
            if constexpr (HAS_RES) { if (live) x += rcur[e]; }
            if (GELU_FWD) {
                f32x4_t keep = x;
                if constexpr (QUICK) {
                    f32x4_t sg, t = x * -2.4554669595930156f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = __builtin_amdgcn_exp2f(t[c]);
                    t += 1.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) sg[c] = __builtin_amdgcn_rcpf(t[c]);
                    const f32x4_t a = x * 1.702f;
                    keep = sg + sg * (a - a * sg);                // QuickGELU'(pre): what backward needs (see epilogue_regs)
                    x *= sg;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = gelu_erf_f(x[c]);
                }
                if (save_u && live)
                    *(uint2*)((bf16_t*)p.aux + (size_t)m * p.ldaux + n) = make_uint2(pack2_t<DT>(keep[0], keep[1]), pack2_t<DT>(keep[2], keep[3]));
            }
            if (GELU_BWD) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t w = c < 2 ? uq[e].x : uq[e].y;
                    const float u = cvt16f_t<DT>((bf16_t)((c & 1) ? (w >> 16) : (w & 0xffff)));
                    x[c] *= QUICK ? u : gelu_erf_grad_f(u);
                }
            }
            if (live) {
                if (OUT_F32) *(f32x4_t*)((float*)p.out + (size_t)m * p.ldo + n) = x;
                else *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = make_uint2(pack2_t<DT>(x[0], x[1]), pack2_t<DT>(x[2], x[3]));   // plain: sc1 on these 8-B stores measured qkv 92 -> 105 us
            }
            if (do_sum && live) csum += x;
        }
        if constexpr (HAS_RES) {
            if (i + RES_AHEAD < TM) res_fetch(i + RES_AHEAD);
        }
    }
    if (do_sum) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t = csum[c];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (g == 0) atomicAdd(p.colsum + n + c, t);
        }
    }
}

#ifdef SIG_GEMM_STAMPS   // diagnostic build only (tools/gemm_stamps.py): where does a tile's time go?
__device__ unsigned long long g_stamps[5 * 8192];
extern "C" int sig_debug_read_stamps(unsigned long long* out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (nblocks > 0 ? 4 * nblocks : 5 * 8192)) == hipSuccess ? 0 : 2;
}
#endif

// BM x 128 x 64 tile, BM = 128 or 160.  BM = 160 exists for tile-count quantisation: the 768-column GEMMs at M = 24768
// are 1164 tiles of 128x128 = 3 rounds of the chip's 512 slots, but 930 tiles of 160x128 = 2 rounds of 1.25x the work
// (the same step at B = 128 per GPU, twice the tiles, runs 9 % more triplets per second).  Wave tile (BM/2) x 64 =
// 5 x 4 MFMA tiles with 9 fragment reads per 20 MFMAs (8 per 16 at BM = 128); LDS 2 x 36 KB, still two blocks per CU.
template <int EPI, int BM, int DT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(SigGemmNT p) {
    static_assert(BM == 128 || BM == 160, "row tile");
    constexpr int TM = BM / 32;              // 16-row MFMA tiles per wave (wave tile = BM/2 rows x 64 columns)
    constexpr int PA = BM / 32;              // A pieces (8 rows x 128 B) per wave and stage; B: 4
    constexpr int AB = BM * 128;             // bytes of the A image
    constexpr int STAGE = AB + 16384;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, in an SGPR (LDS-DMA base goes to M0)
    // tile order: blocks of one XCD get a contiguous range of a BAND-major order (band = wb column tiles whose
    // weight rows, wb*128*K*2 B <= ~2.4 MB, stay resident in that XCD's 4 MB L2 while it sweeps the rows)
    const int tn = p.N >> 7, tm = gridDim.x / tn, wb = p.band;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int per = tm * wb, bnd = id / per, rr = id - bnd * per;
    const int tile_m = rr / wb, tile_n = bnd * wb + (rr - tile_m * wb);
    const int m0 = tile_m * BM, n0 = tile_n << 7;

    // ---- staging: wave w moves A pieces w*PA.., B pieces w*4.. (8 rows x 128 B each) ----
    const bf16_t* ag[PA];
    const bf16_t* bg[4];
    // rows of a last 160-row tile past the 128-row padding of the operand buffer read the last padded row instead (their
    // accumulators are never stored: the epilogue predicates every access on row < M)
    const int rlim = ((((p.M + 127) >> 7) << 7)) - 1;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int r = (wave * PA + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int row = BM == 128 ? m0 + r : min(m0 + r, rlim);
        ag[j] = p.A + (size_t)row * p.lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        bg[j] = p.Bt + (size_t)(n0 + r) * p.ldb + c * 8;
    }
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < PA; ++j) glds16_untracked(ag[j] + kt * 64, sa + (wave * PA + j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_untracked(bg[j] + kt * 64, sa + AB + (wave * 4 + j) * 1024);
    };

    // ---- fragment read offsets (bytes inside one stage) ----
    const int fr = lane & 15, g = lane >> 4, sw = fr >> 1;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = (((ks << 2) | g) ^ sw) << 4;
        aoff[ks] = (wm + fr) * 128 + ch;      // (wm + i*16 + fr) >> 1 & 7 == fr >> 1 & 7: wm and i*16 are multiples of 16
        boff[ks] = AB + (wn + fr) * 128 + ch;
    }

    f32x4_t acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#ifdef SIG_GEMM_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
#define SIG_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
    SIG_STAMP(ts0);
#else
#define SIG_STAMP(v)
#endif
    const int nk = p.K >> 6;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // step kt landed for every wave; every wave is done reading the other buffer
#ifdef SIG_GEMM_STAMPS
        if (kt == 0) SIG_STAMP(ts1);
#endif
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* s = smem + (kt & 1) * STAGE;
        // all fragments of the K-step are requested up front (the DMA is issued untracked, so the compiler's waits
        // are counted: the first MFMAs start when the ks = 0 fragments are in, the ks = 1 reads land under them)
        bf16x8_t af[2][TM], bf[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *(const bf16x8_t*)(s + aoff[ks] + i * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[ks][j] = *(const bf16x8_t*)(s + boff[ks] + j * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        // operands swapped on purpose: D[row = n][col = m], so a lane owns 4 CONSECUTIVE n of one output row
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef SIG_GEMM_STAMPS
    SIG_STAMP(ts2);
#endif
    // ---- epilogue ----
    // bf16 outputs: straight from the accumulator registers (epilogue_regs above: no LDS, no barrier; qkv epilogue
    // 5960 -> 3476 cycles per tile).  f32 outputs go through an LDS transpose: a lane of the MFMA layout only holds 16 B
    // of an f32 row, so register stores make 64-B runs and measured slower (c_proj 11.1k -> 18.1k cycles).
    constexpr bool OUT_F32 = EPI == SIG_EPI_F32 || EPI == SIG_EPI_BIAS_F32 || EPI == SIG_EPI_BIAS_RES_F32 || EPI == SIG_EPI_RES_F32;
    if constexpr (!OUT_F32) {
        epilogue_regs<EPI, TM, 4, DT>(p, acc, m0 + wm, n0 + wn, lane);
    } else {
        // The wave transposes its (BM/2) x 64 f32 sub-tile through the now idle LDS, 16 rows per pass (68-float padded
        // rows: conflict-free b128 writes and reads, wave-private area) so a lane owns 8 CONSECUTIVE columns of one row:
        // full 256-B row segments per 8 lanes.  The residual rows are requested before the barrier and the staging.
        constexpr bool HAS_RES = EPI == SIG_EPI_BIAS_RES_F32 || EPI == SIG_EPI_RES_F32;
        constexpr bool HAS_BIAS = EPI == SIG_EPI_BIAS_F32 || EPI == SIG_EPI_BIAS_RES_F32;
        const int t8 = lane & 7, tr = lane >> 3;           // 8 lanes per row, 8 rows per instruction
        const int n = n0 + wn + t8 * 8;
        f32x4_t rres[HAS_RES ? TM : 1][2][2];
        if (HAS_RES) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int m = m0 + wm + i * 16 + q * 8 + tr;
                    const float* r = p.res + (size_t)(m < p.M ? m : 0) * p.ldr + n;   // rows past M: any valid row, never used
                    rres[HAS_RES ? i : 0][q][0] = *(const f32x4_t*)r;
                    rres[HAS_RES ? i : 0][q][1] = *(const f32x4_t*)(r + 4);
                }
        }
        __syncthreads();                                   // every wave is done reading operand fragments
        float* stg = (float*)(smem + wave * (16 * 68 * 4));  // this wave's private staging area
        f32x4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (HAS_BIAS) { b0 = *(const f32x4_t*)(p.bias + n); b1 = *(const f32x4_t*)(p.bias + n + 4); }
        f32x4_t cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
        const __amdgpu_buffer_rsrc_t r_out = out_rsrc(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4_t*)(stg + fr * 68 + j * 16 + g * 4) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: in-order LDS, no barrier needed
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = q * 8 + tr;
                const int m = m0 + wm + i * 16 + row;
                f32x4_t v0 = *(const f32x4_t*)(stg + row * 68 + t8 * 8) + b0, v1 = *(const f32x4_t*)(stg + row * 68 + t8 * 8 + 4) + b1;
                if (m >= p.M) continue;
                if (HAS_RES) { v0 += rres[HAS_RES ? i : 0][q][0]; v1 += rres[HAS_RES ? i : 0][q][1]; }
                const size_t ob = ((size_t)m * p.ldo + n) * 4;
                store16_policy<store_aux<EPI>()>(r_out, p.out, ob, __builtin_bit_cast(uint4, v0));
                store16_policy<store_aux<EPI>()>(r_out, p.out, ob + 16, __builtin_bit_cast(uint4, v1));
                cs0 += v0;
                cs1 += v1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the area
        }
        // optional bias-gradient by-product: column sums of what was just written (this wave's rows x 64 columns)
        if (p.colsum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = e < 4 ? cs0[e & 3] : cs1[e & 3];
                t += __shfl_xor(t, 8, 64);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (tr == 0) atomicAdd(p.colsum + n + e, t);
            }
        }
    }
#ifdef SIG_GEMM_STAMPS
    unsigned long long ts2b = 0;
    SIG_STAMP(ts2b);                                    // every store ISSUED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SIG_STAMP(ts3);                                     // every store acknowledged
    if (tid == 0 && blockIdx.x < 8192) {
        g_stamps[blockIdx.x * 4 + 0] = ts0; g_stamps[blockIdx.x * 4 + 1] = ts1;
        g_stamps[blockIdx.x * 4 + 2] = ts2; g_stamps[blockIdx.x * 4 + 3] = ts3;
        g_stamps[4 * 8192 + blockIdx.x] = ts2b;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4), 128x64 per wave, phase-pipelined.
// Why: the 128x128 kernel's main loop sits on the CU's LDS port -- 64 KB of DMA writes + 128 KB of fragment reads per
// K-step pair ~ 1530 cycles (measured 1507, tools/gemm_stamps.py) for 1024 cycles of MFMA.  A 128x64 wave tile needs 12
// fragment reads per 32 MFMAs instead of 16, and a 256x256 block half the DMA bytes per FLOP: 1792 LDS cycles per
// 2048 MFMA cycles.  That only pays if the 8 waves do NOT burst DMA issue, reads and MFMAs in lockstep, so every K-step
// is cut into 4 phases of 16 MFMAs; in each phase a wave also reads the fragments of the NEXT phase into the other
// register set and issues 2 of its 8 DMA pieces for the next K-stage:
//     P0: MFMA(k0, rows lo)   read A(k0,hi)              DMA B0..B3 of the next stage
//     P1: MFMA(k0, rows hi)   read A(k1,lo), B(k1)
//     P2: MFMA(k1, rows lo)   read A(k1,hi)
//     -- s_waitcnt vmcnt(0) lgkmcnt(0); barrier: next stage landed, this stage fully read --
//     P3: MFMA(k1, rows hi)   read A'(k0,lo), B'(k0) from the NEXT stage      DMA A0..A3 of the stage after
// (all DMA is issued in P3/P0: a piece needs ~900+ cycles to land, a phase lasts ~600)
// One barrier per K-step; LDS image and swizzle as in the 128x128 kernel; same epilogues (LDS-transposed stores).
// ------------------------------------------------------------------------------------------------
template <int EPI, int DT>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(SigGemmNT p) {
    constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = p.N >> 8, tm = gridDim.x / tn, wb = p.band;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int per = tm * wb, bnd = id / per, rr = id - bnd * per;
    const int tile_m = rr / wb, tile_n = bnd * wb + (rr - tile_m * wb);
    const int m0 = tile_m << 8, n0 = tile_n << 8;

    // DMA pieces of this wave: A rows (wave*4+j)*8.., B rows likewise (256 rows = 32 pieces each, 8 waves x 4).
    // Source address = uniform tile base (+ kt*128 B, scalar) + a per-lane byte offset that never changes.
    unsigned ao[4], bo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        ao[j] = (unsigned)(r * p.lda + c * 8) * 2u;
        bo[j] = (unsigned)(r * p.ldb + c * 8) * 2u;
    }
    const bf16_t* abase = p.A + (size_t)m0 * p.lda;
    const bf16_t* bbase = p.Bt + (size_t)n0 * p.ldb;
    auto dma_a = [&](int j, int kt, int stage) { glds16_untracked_s(abase + kt * 64, ao[j], smem + stage * STAGE + (wave * 4 + j) * 1024); };
    auto dma_b = [&](int j, int kt, int stage) { glds16_untracked_s(bbase + kt * 64, bo[j], smem + stage * STAGE + BM * 128 + (wave * 4 + j) * 1024); };

    const int fr = lane & 15, g = lane >> 4, sw = fr >> 1;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = (((ks << 2) | g) ^ sw) << 4;
        aoff[ks] = (wm + fr) * 128 + ch;
        boff[ks] = BM * 128 + (wn + fr) * 128 + ch;
    }
    // Fragment reads are inline asm and their waits are issued by hand (counted, and tied to the fragment registers so
    // no MFMA can be scheduled above its wait).  Left to the compiler, the register allocator recycles the destination
    // of an in-flight ds_read as an MFMA result register and the WAW hazard turns every wait into lgkmcnt(0).
    bf16x8_t aX[4], aY[4], bX[4], bY[4];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#define SIG_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto rd_a = [&](int stage, int ks, auto half_c, bf16x8_t (&a)[4]) {
        constexpr int H = decltype(half_c)::value;
        const unsigned ad = lds0 + stage * STAGE + aoff[ks];
        SIG_RD128(a[0], ad, (H * 4 + 0) * 2048);
        SIG_RD128(a[1], ad, (H * 4 + 1) * 2048);
        SIG_RD128(a[2], ad, (H * 4 + 2) * 2048);
        SIG_RD128(a[3], ad, (H * 4 + 3) * 2048);
    };
    auto rd_b = [&](int stage, int ks, bf16x8_t (&b)[4]) {
        const unsigned ad = lds0 + stage * STAGE + boff[ks];
        SIG_RD128(b[0], ad, 0);
        SIG_RD128(b[1], ad, 2048);
        SIG_RD128(b[2], ad, 4096);
        SIG_RD128(b[3], ad, 6144);
    };
#define SIG_WAIT4(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define SIG_WAIT8(n, a, b)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), \
                 "+v"(b[2]), "+v"(b[3]))
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int half, const bf16x8_t (&a)[4], const bf16x8_t (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[half * 4 + i][j] = mfma16<DT>(b[j], a[i], acc[half * 4 + i][j]);
    };

    const int nk = p.K >> 6;
    // ---- u prefetch (GELU' dgrad): the epilogue multiplies by QuickGELU'(u) and u is a 128-KB tile that used to be loaded
    // in the exposed epilogue (17.7 k of a 52.9 k-cycle tile, tools/gemm_stamps.py; one block per CU, so nothing overlaps
    // it).  Three quarters of it now arrive by LDS-DMA under the main loop: rows i = 0,1 of every wave into the 32 KB of LDS
    // beyond the two operand stages (issued in the prologue), rows i = 2..5 into the operand stage that is free from the
    // barrier of the second-to-last K-step on; rows i = 6,7 stay register loads, requested first and consumed last.
    constexpr bool UPF = EPI == SIG_EPI_DGELU_BF16;
    const bf16_t* ubase = UPF ? (const bf16_t*)p.aux + (size_t)m0 * p.ldaux + n0 : nullptr;
    auto dma_u = [&](int tile_row_of_region_row_lo, int region_row, char* region) {
        // one 1-KB piece = region rows (region_row, region_row + 1); lane -> (row, physical chunk)
        const int rr = region_row + (lane >> 5);
        const int row = rr + tile_row_of_region_row_lo;                       // tile row (same low 4 bits as rr)
        const int c = (lane & 31) ^ (rr & 15);
        glds16_untracked_s(ubase, (unsigned)(row * p.ldaux + c * 8) * 2u, region + region_row * 512);
    };
    char* const u_r0 = smem + 2 * STAGE;
#ifdef SIG_GEMM_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    SIG_STAMP(ts0);
#endif
    // prologue: stage 0 complete, first fragments, first two pieces of stage 1
#pragma unroll
    for (int j = 0; j < 4; ++j) { dma_a(j, 0, 0); dma_b(j, 0, 0); }
    if (UPF) {
        // issued AFTER the operand pieces so that the first barrier only waits for those (vmcnt retires in order: the four u
        // pieces, which come from HBM at ~7 k cycles, stay in flight under the first K-step)
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // region rows 0..31 -> tile rows 0..31, 32..63 -> 128..159
            const int rlo = (wave * 4 + j) * 2;
            dma_u(rlo < 32 ? 0 : 96, rlo, u_r0);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#ifdef SIG_GEMM_STAMPS
    SIG_STAMP(ts1);
#endif
#ifdef SIG_NT256_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // static priority for the second-dispatched half (MI355X_MICROARCH, two waves per SIMD, item 4)
#endif
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    rd_a(0, 0, H0{}, aX);
    rd_b(0, 0, bX);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(j, 1, 1);
    // one K-step; MORE / MORE2 are compile-time so the steady-state body is a single basic block; the last two steps are
    // peeled copies.  Needs nk >= 2.  LDS returns in order, so "lgkmcnt(n)" = everything but the newest n reads landed.
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        constexpr bool U_ISSUE = UPF && MORE && !MORE2;      // second-to-last K-step: its stage is free after the barrier
        constexpr bool U_FLYING = UPF && !MORE;              // last K-step: nothing but the u pieces is in flight
        const int st = kt & 1;
        // P0: outstanding aX,bX (8) + aY (4)
        rd_a(st, 0, H1{}, aY);
        if (MORE) { dma_b(0, kt + 1, st ^ 1); dma_b(1, kt + 1, st ^ 1); dma_b(2, kt + 1, st ^ 1); dma_b(3, kt + 1, st ^ 1); }
        SIG_WAIT8(4, aX, bX);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, aX, bX);
        __builtin_amdgcn_sched_barrier(0);
        // P1: outstanding aY (4) + aX,bY (8)
        rd_a(st, 1, H0{}, aX);
        rd_b(st, 1, bY);
        SIG_WAIT4(8, aY);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, aY, bX);
        __builtin_amdgcn_sched_barrier(0);
        // P2: outstanding aX,bY (8) + aY (4)
        rd_a(st, 1, H1{}, aY);
        SIG_WAIT8(4, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        // stage boundary: this wave's DMA pieces of the next stage landed, all its reads of this stage returned
        if (U_FLYING) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aY[0]), "+v"(aY[1]), "+v"(aY[2]), "+v"(aY[3])::"memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(aY[0]), "+v"(aY[1]), "+v"(aY[2]), "+v"(aY[3])::"memory");
        __builtin_amdgcn_s_barrier();
        // P3
        if (MORE) {
            rd_a(st ^ 1, 0, H0{}, aX);
            rd_b(st ^ 1, 0, bX);
        }
        if (MORE2) { dma_a(0, kt + 2, st); dma_a(1, kt + 2, st); dma_a(2, kt + 2, st); dma_a(3, kt + 2, st); }
        if (U_ISSUE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {      // region rows 0..63 -> tile rows 32..95, 64..127 -> 160..223
                const int rlo = (wave * 8 + j) * 2;
                dma_u(rlo < 64 ? 32 : 96, rlo, smem + st * STAGE);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(1, aY, bY);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    for (int kt = 0; kt < nk - 2; ++kt) step(kt, T_{}, T_{});
    step(nk - 2, T_{}, F_{});
    step(nk - 1, F_{}, F_{});

#ifdef SIG_GEMM_STAMPS
    SIG_STAMP(ts2);
#endif
    // ---- epilogue: straight from the accumulator registers, no LDS and no barrier ----
    if constexpr (UPF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's u pieces (issued >= one K-step ago) landed
        __builtin_amdgcn_s_barrier();                         // ... and everybody else's
        epilogue_regs<EPI, 8, 4, DT, true>(p, acc, m0 + wm, n0 + wn, lane, u_r0, smem + ((nk - 2) & 1) * STAGE, wm, wn);
    } else {
        epilogue_regs<EPI, 8, 4, DT>(p, acc, m0 + wm, n0 + wn, lane);
    }
#ifdef SIG_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SIG_STAMP(ts3);
    if (tid == 0 && blockIdx.x < 8192) {
        g_stamps[blockIdx.x * 4 + 0] = ts0; g_stamps[blockIdx.x * 4 + 1] = ts1;
        g_stamps[blockIdx.x * 4 + 2] = ts2; g_stamps[blockIdx.x * 4 + 3] = ts3;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// NT, 320 x 256 tile: the phase pipeline of gemm_nt256_kernel with 10 x 4 MFMA tiles per wave (8 waves x 160 x 64), for
// the 768-column GEMMs (out_proj, c_proj and every N = 768 dgrad): at M = 24768 they are 78 x 3 = 234 tiles = ONE round
// of the 256 CUs, where 160x128 tiles need two rounds of a simpler, less efficient loop (68 % of the MFMA issue rate
// against 83 %).  LDS 2 x 72 KB, one block per CU; a K-step is 4 phases of 20 MFMAs, 28 fragment reads per 80 MFMAs.
//     P0: MFMA(k0, rows lo)   read A(k0,hi)              DMA B0..B3 of the next stage
//     P1: MFMA(k0, rows hi)   read A(k1,lo), B(k1)
//     P2: MFMA(k1, rows lo)   read A(k1,hi)
//     -- s_waitcnt vmcnt(0) lgkmcnt(0); barrier --
//     P3: MFMA(k1, rows hi)   read A'(k0,lo), B'(k0) from the NEXT stage      DMA A0..A4 of the stage after
// The last row tile reaches past the 128-row padding of the operand buffers: its DMA rows are clamped to the last
// padded row (never stored: the epilogue is row-predicated).
// ------------------------------------------------------------------------------------------------
template <int EPI, int DT>
__global__ __launch_bounds__(512, 2) void gemm_nt320_kernel(SigGemmNT p, int mp) {
    constexpr int BM = 320, BN = 256, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = p.N >> 8, tm = gridDim.x / tn, wb = p.band;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int per = tm * wb, bnd = id / per, rr = id - bnd * per;
    const int tile_m = rr / wb, tile_n = bnd * wb + (rr - tile_m * wb);
    const int m0 = tile_m * BM, n0 = tile_n << 8;

    // DMA pieces (8 rows x 128 B): A 40 = 5 per wave, B 32 = 4 per wave
    unsigned ao[5], bo[4];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = (wave * 5 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int rc = m0 + r < mp ? r : mp - 1 - m0;                 // clamp (last row tile only)
        ao[j] = (unsigned)(rc * p.lda + c * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);             // row of the LDS image
        // ... which holds weight row (= output column) 4 * slot + group of its 64-row strip: epilogue_nat's layout
        const int sr = (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);
        bo[j] = (unsigned)(sr * p.ldb + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u;
    }
    const bf16_t* abase = p.A + (size_t)m0 * p.lda;
    const bf16_t* bbase = p.Bt + (size_t)n0 * p.ldb;
    auto dma_a = [&](int j, int kt, int stage) { glds16_untracked_s(abase + kt * 64, ao[j], smem + stage * STAGE + (wave * 5 + j) * 1024); };
    auto dma_b = [&](int j, int kt, int stage) { glds16_untracked_s(bbase + kt * 64, bo[j], smem + stage * STAGE + BM * 128 + (wave * 4 + j) * 1024); };

    const int fr = lane & 15, g = lane >> 4, sw = fr >> 1;
    const int wm = (wave >> 2) * 160, wn = (wave & 3) * 64;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = (((ks << 2) | g) ^ sw) << 4;
        aoff[ks] = (wm + fr) * 128 + ch;
        boff[ks] = BM * 128 + (wn + fr) * 128 + ch;
    }
    bf16x8_t aX[5], aY[5], bX[4], bY[4];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto rd_a = [&](int stage, int ks, auto half_c, bf16x8_t (&a)[5]) {
        constexpr int H = decltype(half_c)::value;
        const unsigned ad = lds0 + stage * STAGE + aoff[ks];
        SIG_RD128(a[0], ad, (H * 5 + 0) * 2048);
        SIG_RD128(a[1], ad, (H * 5 + 1) * 2048);
        SIG_RD128(a[2], ad, (H * 5 + 2) * 2048);
        SIG_RD128(a[3], ad, (H * 5 + 3) * 2048);
        SIG_RD128(a[4], ad, (H * 5 + 4) * 2048);
    };
    auto rd_b = [&](int stage, int ks, bf16x8_t (&b)[4]) {
        const unsigned ad = lds0 + stage * STAGE + boff[ks];
        SIG_RD128(b[0], ad, 0);
        SIG_RD128(b[1], ad, 2048);
        SIG_RD128(b[2], ad, 4096);
        SIG_RD128(b[3], ad, 6144);
    };
#define SIG_WAIT5(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]))
#define SIG_WAIT9(n, a, b)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), \
                 "+v"(b[1]), "+v"(b[2]), "+v"(b[3]))
    f32x4_t acc[10][4];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int half, const bf16x8_t (&a)[5], const bf16x8_t (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[half * 5 + i][j] = mfma16<DT>(a[i], b[j], acc[half * 5 + i][j]);      // natural orientation: rows = A slots
    };

    const int nk = p.K >> 6;
#ifdef SIG_GEMM_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    SIG_STAMP(ts0);
#endif
#pragma unroll
    for (int j = 0; j < 5; ++j) dma_a(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_b(j, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef SIG_GEMM_STAMPS
    SIG_STAMP(ts1);
#endif
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    rd_a(0, 0, H0{}, aX);
    rd_b(0, 0, bX);
#pragma unroll
    for (int j = 0; j < 5; ++j) dma_a(j, 1, 1);
    // lgkmcnt(n): everything but the newest n LDS reads has landed (in-order returns).  Needs nk >= 2.
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        const int st = kt & 1;
        // P0: outstanding aX,bX (9) + aY (5)
        rd_a(st, 0, H1{}, aY);
        if (MORE) { dma_b(0, kt + 1, st ^ 1); dma_b(1, kt + 1, st ^ 1); dma_b(2, kt + 1, st ^ 1); dma_b(3, kt + 1, st ^ 1); }
        SIG_WAIT9(5, aX, bX);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, aX, bX);
        __builtin_amdgcn_sched_barrier(0);
        // P1: outstanding aY (5) + aX,bY (9)
        rd_a(st, 1, H0{}, aX);
        rd_b(st, 1, bY);
        SIG_WAIT5(9, aY);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, aY, bX);
        __builtin_amdgcn_sched_barrier(0);
        // P2: outstanding aX,bY (9) + aY (5)
        rd_a(st, 1, H1{}, aY);
        SIG_WAIT9(5, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        // stage boundary: this wave's DMA pieces of the next stage landed, all its reads of this stage returned
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(aY[0]), "+v"(aY[1]), "+v"(aY[2]), "+v"(aY[3]), "+v"(aY[4])::"memory");
        __builtin_amdgcn_s_barrier();
        // P3
        if (MORE) {
            rd_a(st ^ 1, 0, H0{}, aX);
            rd_b(st ^ 1, 0, bX);
        }
        if (MORE2) { dma_a(0, kt + 2, st); dma_a(1, kt + 2, st); dma_a(2, kt + 2, st); dma_a(3, kt + 2, st); dma_a(4, kt + 2, st); }
        __builtin_amdgcn_sched_barrier(0);
        mma(1, aY, bY);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    for (int kt = 0; kt < nk - 2; ++kt) step(kt, T_{}, T_{});
    step(nk - 2, T_{}, F_{});
    step(nk - 1, F_{}, F_{});
#ifdef SIG_GEMM_STAMPS
    SIG_STAMP(ts2);
#endif
    epilogue_nat<EPI, 10, DT>(p, acc, m0 + wm, n0 + wn, lane);
#ifdef SIG_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SIG_STAMP(ts3);
    if (tid == 0 && blockIdx.x < 8192) {
        g_stamps[blockIdx.x * 4 + 0] = ts0; g_stamps[blockIdx.x * 4 + 1] = ts1;
        g_stamps[blockIdx.x * 4 + 2] = ts2; g_stamps[blockIdx.x * 4 + 3] = ts3;
    }
#endif
}

// ---- live timing of one GEMM shape (bench.py's roofline leg): HIP events on the launch stream ----
#include <atomic>
#include <vector>
static struct {
    bool on = false;
    int epi = -1, N = 0, K = 0;
    size_t used = 0;
    double flops = 0;
    std::vector<hipEvent_t> ev;  // start/stop pairs
} g_prof;

int sig_prof_begin_impl(int epi, int N, int K, int max_launches) {
    SIG_CHECK_ARG(max_launches > 0 && max_launches <= 65536, "prof_begin: max_launches out of range");
    while (g_prof.ev.size() < (size_t)2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            sig_set_error("prof_begin: hipEventCreate failed");
            return 2;
        }
        g_prof.ev.push_back(e);
    }
    g_prof.on = true; g_prof.epi = epi; g_prof.N = N; g_prof.K = K; g_prof.used = 0; g_prof.flops = 0;
    return 0;
}
int sig_prof_end_impl(double* total_ms, int* launches, double* flops) {
    g_prof.on = false;
    double ms = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) {
            sig_set_error("prof_end: hipEventSynchronize failed");
            return 2;
        }
        float t = 0;
        (void)hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int)(g_prof.used / 2);
    if (flops) *flops = g_prof.flops;
    return 0;
}


bool sig_prof_tn_start(hipStream_t st, int cls, int I, int J) {
    const bool timed = g_prof.on && g_prof.epi == cls && (g_prof.N == 0 || (g_prof.N == I && g_prof.K == J)) &&
                       g_prof.used + 2 <= g_prof.ev.size();
    if (timed) (void)hipEventRecord(g_prof.ev[g_prof.used], st);
    return timed;
}
void sig_prof_tn_stop(hipStream_t st, double flops) {
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
    g_prof.used += 2;
    g_prof.flops += flops;
}

static int choose_band(int tn, int K, int BN) {
    static int force = -1;       // SIG_GEMM_BAND=<column tiles per band> (A/B knob; must divide the tile count)
    if (force < 0) { const char* e = getenv("SIG_GEMM_BAND"); force = e ? atoi(e) : 0; }
    if (force > 0 && tn % force == 0) return force;
    int wmax = (int)(2400000LL / (2LL * BN * K));
    if (wmax < 1) wmax = 1;
    for (int nb = 1; nb <= tn; ++nb)
        if (tn % nb == 0 && tn / nb <= wmax) return tn / nb;
    return 1;
}

// CUs the GEMM launchers may count on.  The big-tile kernels run ONE block per CU and are sized in rounds of the chip: a
// 234-tile launch is one round on 256 free CUs and two as soon as 23 are taken.  While RCCL all-reduces gradient buckets
// under the backward pass its channel workgroups hold CUs for milliseconds (a GEMM block's 8 waves x 256 registers cannot
// share a CU with them), so the trainer reserves CUs for that phase (sig_tune_reserved_cus; SIG_RESERVED_CUS presets it)
// and the tile choice and the wgrad's row split are made for the rest.
// (atomic: set from the training thread around the backward, read by launch_nt / launch_tn on autograd's device thread)
static std::atomic<int> g_reserved_cus{-1};
static int free_cus() {
    if (g_reserved_cus < 0) { const char* e = getenv("SIG_RESERVED_CUS"); g_reserved_cus = e ? atoi(e) : 0; }
    const int f = 256 - g_reserved_cus;
    return f < 64 ? 64 : f;
}
int sig_free_cus() { return free_cus(); }
int sig_tune_reserved_cus_impl(int n) {
    (void)free_cus();                            // resolves the SIG_RESERVED_CUS preset first: restoring `prev` keeps it
    const int cur = g_reserved_cus, prev = cur < 0 ? 0 : cur;
    g_reserved_cus = n < 0 ? 0 : (n > 192 ? 192 : n);
    return prev;
}

// SIG_GEMM_TILE=<128|256|320> / sig_tune_gemm_tile(): pin the NT tile wherever that kernel is legal (tests, A/B runs)
static std::atomic<int> g_force_tile{-1};
int sig_tune_gemm_tile_impl(int tile) {
    if (g_force_tile < 0) { const char* e = getenv("SIG_GEMM_TILE"); g_force_tile = e ? atoi(e) : 0; }   // preset first: restoring `prev` keeps it
    const int cur = g_force_tile, prev = cur < 0 ? 0 : cur;
    g_force_tile = tile;
    return prev;
}

template <int EPI, int DT>
static int launch_nt(const SigGemmNT& p_in, hipStream_t st) {
    SigGemmNT p = p_in;
    if (g_force_tile < 0) { const char* e = getenv("SIG_GEMM_TILE"); g_force_tile = e ? atoi(e) : 0; }
    const int force = g_force_tile;
    // 256x256 phase-pipelined kernel when the problem fills the chip with it and the 128-row padding of the token
    // buffers happens to be a 256 multiple (M = 24768 -> 24832 = 97 * 256); otherwise the 128x128 kernel
    const int mp = ((p.M + 127) >> 7) << 7;
    const bool can256 = (p.N & 255) == 0 && (mp & 255) == 0 && p.K >= 128;
    const int cus = free_cus();
    bool big = can256 && (mp >> 8) * (p.N >> 8) >= 2 * cus;
    if (force == 128) big = false;
    if (force == 256) big = can256;
    if (EPI == SIG_EPI_DGELU_BF16 || EPI == SIG_EPI_BIAS_GELU_BF16) {      // A/B knob: SIG_GEMM_TILE_GELU=128 moves only the GELU epilogues
        static int force_g = -1;
        if (force_g < 0) { const char* e = getenv("SIG_GEMM_TILE_GELU"); force_g = e ? atoi(e) : 0; }
        if (force_g == 128) big = false;
    }
    // 320x256 tiles (gemm_nt320_kernel): SIG_GEMM_TILE=320 wherever legal; by default where one CU-time estimate says so
    const bool can320 = (p.N & 255) == 0 && p.K >= 128;
    const int t320 = ((p.M + 319) / 320) * (p.N >> 8);
    // One CU-time estimate per candidate: rounds of the chip x tile area / the MFMA issue share its main loop reaches
    // (tools/gemm_stamps.py: 0.83 for the phase-pipelined 256- and 320-row kernels, 0.68 for the 128 / 160-row one, two of
    // whose blocks share a CU).  N = 768 at M = 24768: 1 round of 320x256 against 2 rounds of 160x128 (measured 70 vs 80 us
    // for the qkv dgrad, 90 vs 104 for c_fc's); qkv: 3 rounds against 4 of 256x256 (91 vs 94 us).  The GELU' dgrad keeps the
    // 256x256 kernel (its saved derivative arrives by LDS-DMA under the main loop there).
    bool tall320 = false;
    if (can320 && EPI != SIG_EPI_DGELU_BF16 && EPI != SIG_EPI_DGELUERF_BF16) {
        const int tm160 = (p.M + 159) / 160, t160 = tm160 * (p.N >> 7), t128 = (mp >> 7) * (p.N >> 7);
        const float c320 = (float)((t320 + cus - 1) / cus) * 81920.f / 0.83f;
        const float c256 = big ? (float)(((mp >> 8) * (p.N >> 8) + cus - 1) / cus) * 65536.f / 0.83f : 1e30f;
        const float c160 = (float)((t160 + 2 * cus - 1) / (2 * cus)) * 40960.f / 0.68f;
        const float c128 = (float)((t128 + 2 * cus - 1) / (2 * cus)) * 32768.f / 0.68f;
        // a tie with 256x256 (c_fc: 4 rounds x 1.25 = 5 rounds) goes to the kernel whose epilogue suits the outputs: with the
        // saved derivative as a second output the natural-orientation stores win (144 vs 153 us in the train step), with
        // one output the 256x256 kernel does (128 vs 132 us at inference)
        const bool two_out = (EPI == SIG_EPI_BIAS_GELU_BF16 || EPI == SIG_EPI_BIAS_GELUERF_BF16) && p.aux != nullptr;
        tall320 = (c320 < c256 || (c320 == c256 && two_out)) && c320 < c160 && c320 < c128;
    }
    if (force == 320) tall320 = can320;
    if (force != 0 && force != 320) tall320 = false;      // (SIG_GEMM_TILE=1: the choice without this kernel, for A/B runs)
    const bool timed = g_prof.on && g_prof.epi == EPI && g_prof.N == p.N && g_prof.K == p.K && g_prof.used + 2 <= g_prof.ev.size();
    if (timed) (void)hipEventRecord(g_prof.ev[g_prof.used], st);
    if (force == 0 && sig_nt192p_eligible(p, EPI, cus)) {
        p.band = choose_band(p.N >> 8, p.K, 256);
        const int rc = sig_launch_nt192p(p, EPI, cus, st);
        if (rc) return rc;
    } else if (tall320) {
        static std::once_flag attr320;
        std::call_once(attr320, [] {
            (void)hipFuncSetAttribute((const void*)&gemm_nt320_kernel<EPI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            });
        // one round of the chip: every tile runs at once and the CUs walk k in step, so what an XCD's L2 shares is what its
        // tiles have in common at the same k -- all column tiles of a row tile on one XCD read the A panel once (band =
        // all columns: FETCH of the qkv dgrad 342 -> 114 MB of A; 70.5 -> 68 us, c_fc dgrad 91 -> 88.5); with several
        // rounds the weight band that stays L2-resident while the rows sweep by matters instead (choose_band)
        p.band = t320 <= cus ? (p.N >> 8) : choose_band(p.N >> 8, p.K, 256);
        hipLaunchKernelGGL((gemm_nt320_kernel<EPI, DT>), dim3(t320), dim3(512), 147456, st, p, mp);
    } else if (big) {
        static std::once_flag attr256;
        std::call_once(attr256, [] {
            (void)hipFuncSetAttribute((const void*)&gemm_nt256_kernel<EPI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            });
        p.band = choose_band(p.N >> 8, p.K, 256);
        // (the GELU' dgrad stages a quarter of the saved pre-activation tile in the 32 KB beyond the two operand stages)
        const int lds256 = EPI == SIG_EPI_DGELU_BF16 ? 163840 : 131072;
        hipLaunchKernelGGL((gemm_nt256_kernel<EPI, DT>), dim3((mp >> 8) * (p.N >> 8)), dim3(512), lds256, st, p);
    } else {
        static std::once_flag attr_done;
        std::call_once(attr_done, [] {
            (void)hipFuncSetAttribute((const void*)&gemm_nt_kernel<EPI, 128, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            (void)hipFuncSetAttribute((const void*)&gemm_nt_kernel<EPI, 160, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
            });
        p.band = choose_band(p.N >> 7, p.K, 128);
        // 160-row tiles when they need fewer rounds of the chip's 512 slots per unit of work (1.25x a 128-row tile); a last
        // tile that passes the 128-row padding of the operand buffers re-reads the last padded row (B = 32: 12384 rows ->
        // 78 x 6 = 468 tiles in one round for the N = 768 layers, against 582 of 128 rows in two)
        const int tn128 = p.N >> 7, t128 = (mp >> 7) * tn128, tm160 = (p.M + 159) / 160, t160 = tm160 * tn128;
        const int slots = 2 * cus;
        const float c128 = (float)((t128 + slots - 1) / slots), c160 = 1.25f * (float)((t160 + slots - 1) / slots);
        bool tall = c160 < c128;
        static int force_bm = -1;
        if (force_bm < 0) { const char* e = getenv("SIG_GEMM_BM"); force_bm = e ? atoi(e) : 0; }
        if (force_bm == 128) tall = false;
        if (force_bm == 160) tall = true;
        if (tall) hipLaunchKernelGGL((gemm_nt_kernel<EPI, 160, DT>), dim3(t160), dim3(256), 73728, st, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<EPI, 128, DT>), dim3(t128), dim3(256), 65536, st, p);
    }
    if (timed) {
        (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
        g_prof.used += 2;
        g_prof.flops += 2.0 * p.M * p.N * p.K;
    }
    SIG_CHECK_LAUNCH("gemm_nt");
    return 0;
}

template <int DT>
static int dispatch_nt(const SigGemmNT& p, int epi, hipStream_t st) {
    SIG_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    SIG_CHECK_ARG((p.N & 127) == 0 && (p.K & 63) == 0, "gemm_nt: N=%d must be a multiple of 128 and K=%d of 64", p.N, p.K);
    SIG_CHECK_ARG((p.lda & 7) == 0 && (p.ldb & 7) == 0 && (p.ldo & 7) == 0, "gemm_nt: leading dims must keep 16-B alignment");
    SIG_CHECK_ARG(p.lda >= p.K && p.ldb >= p.K && p.ldo >= p.N, "gemm_nt: leading dimension smaller than the row");
    SIG_CHECK_ARG(p.A && p.Bt && p.out, "gemm_nt: null operand");
    SIG_CHECK_ARG(!p.colsum || !(epi == SIG_EPI_BIAS_BF16 || epi == SIG_EPI_BIAS_GELU_BF16 || epi == SIG_EPI_BIAS_GELUERF_BF16),
                  "gemm_nt: the 16-bit forward epilogues (%d) carry no column sums", epi);
    switch (epi) {
        case SIG_EPI_F32: return launch_nt<SIG_EPI_F32, DT>(p, st);
        case SIG_EPI_BF16: return launch_nt<SIG_EPI_BF16, DT>(p, st);
        case SIG_EPI_BIAS_F32: SIG_CHECK_ARG(p.bias, "gemm_nt: bias missing"); return launch_nt<SIG_EPI_BIAS_F32, DT>(p, st);
        case SIG_EPI_BIAS_BF16: SIG_CHECK_ARG(p.bias, "gemm_nt: bias missing"); return launch_nt<SIG_EPI_BIAS_BF16, DT>(p, st);
        case SIG_EPI_BIAS_RES_F32:
            SIG_CHECK_ARG(p.bias && p.res && (p.ldr & 3) == 0, "gemm_nt: bias/residual missing");
            return launch_nt<SIG_EPI_BIAS_RES_F32, DT>(p, st);
        case SIG_EPI_BIAS_GELU_BF16: SIG_CHECK_ARG(p.bias, "gemm_nt: bias missing"); return launch_nt<SIG_EPI_BIAS_GELU_BF16, DT>(p, st);
        case SIG_EPI_DGELU_BF16:
            SIG_CHECK_ARG(p.aux && (p.ldaux & 3) == 0, "gemm_nt: pre-activation missing");
            return launch_nt<SIG_EPI_DGELU_BF16, DT>(p, st);
        case SIG_EPI_BIAS_GELUERF_BF16: SIG_CHECK_ARG(p.bias, "gemm_nt: bias missing"); return launch_nt<SIG_EPI_BIAS_GELUERF_BF16, DT>(p, st);
        case SIG_EPI_DGELUERF_BF16:
            SIG_CHECK_ARG(p.aux && (p.ldaux & 3) == 0, "gemm_nt: pre-activation missing");
            return launch_nt<SIG_EPI_DGELUERF_BF16, DT>(p, st);
        case SIG_EPI_RES_F32:
            SIG_CHECK_ARG(p.res && (p.ldr & 3) == 0, "gemm_nt: residual missing");
            return launch_nt<SIG_EPI_RES_F32, DT>(p, st);
    }
    sig_set_error("gemm_nt: unknown epilogue %d", epi);
    return 1;
}

int sig_launch_gemm_nt(const SigGemmNT& p, int epi, hipStream_t st) {
    SIG_CHECK_DT(p.dt, "gemm_nt");
    return p.dt == SIG_DT_F16 ? dispatch_nt<SIG_DT_F16>(p, epi, st) : dispatch_nt<SIG_DT_BF16>(p, epi, st);
}

// ------------------------------------------------------------------------------------------------
// TN (wgrad):  out[I,J] += sum_m P[m,I] * Q[m,J]   over this block's row chunk
// ------------------------------------------------------------------------------------------------
// LDS image of one operand tile: 64 rows (m) x 128 columns (bf16) = 256 B per row, 16 chunks of 16 B,
// physical chunk = chunk ^ ((row & 3) << 2).  Fragments are read with ds_read_b64_tr_b16: a 16-lane
// group fetches a 4(m) x 16(col) block and each lane gets 4 consecutive m of ONE column, which is the
// k-contiguous MFMA operand shape; the XOR spreads the 8 (row, 32-B block) pairs of a 32-lane half over
// the 8 distinct 32-B ranges of the 256-B bank row.
// MFMA 32x32x16 so that one accumulator register is two 128-B row segments: the shape at which global
// f32 atomics run at full rate (MI355X_MICROARCH 'Global float atomics').
template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(SigGemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, in an SGPR (LDS-DMA base goes to M0)
    const int tj = p.J >> 7, ti = p.I >> 7;
    const int tiles = ti * tj;
    const int split = blockIdx.x / tiles, t = blockIdx.x - split * tiles;
    const int tile_i = t / tj, tile_j = t - tile_i * tj;
    const int i0 = tile_i << 7, j0 = tile_j << 7;
    const int mbeg = split * p.m_chunk;
    int mend = mbeg + p.m_chunk;
    if (mend > p.Mr) mend = p.Mr;
    const int nk = (mend - mbeg) >> 6;
    if (nk <= 0) return;

    const bf16_t* pg[4];
    const bf16_t* qg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        pg[j] = p.P + (size_t)(mbeg + r) * p.ldp + i0 + c * 8;
        qg[j] = p.Q + (size_t)(mbeg + r) * p.ldq + j0 + c * 8;
    }
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * 32768 + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_untracked(pg[j] + (size_t)kt * 64 * p.ldp, sa + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_untracked(qg[j] + (size_t)kt * 64 * p.ldq, sa + 16384 + j * 1024);
    };

    // transposed-read addressing: lane = 16*G + 4*q + pp ; group G: column block 16*(G&1), k half h = G>>1
    const int G = lane >> 4, h = G >> 1, tq = (lane >> 2) & 3, pp = lane & 3;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    int poff[2], qoff[2];  // per 32-wide fragment tile, for row block (8h + q), first half (rows +0..3)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int colp = wi + a * 32 + 16 * (G & 1) + 4 * pp;
        const int colq = wj + a * 32 + 16 * (G & 1) + 4 * pp;
        const int row = 8 * h + tq;
        poff[a] = row * 256 + ((((colp >> 3) ^ (tq << 2)) << 4) | ((pp & 1) << 3));
        qoff[a] = 16384 + row * 256 + ((((colq >> 3) ^ (tq << 2)) << 4) | ((pp & 1) << 3));
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* s = smem + (kt & 1) * 32768;
        // two 16-row slabs of fragments are requested ahead of the MFMAs that use them (the DMA is untracked, so the
        // compiler's waits are counted and the second slab lands under the first slab's MFMAs)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8_t pf[2][2], qf[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = kp * 2 + kk;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const bf16x4_t p0 = lds_tr16(s + poff[a] + ks * 4096);
                    const bf16x4_t p1 = lds_tr16(s + poff[a] + ks * 4096 + 1024);
                    const bf16x4_t q0 = lds_tr16(s + qoff[a] + ks * 4096);
                    const bf16x4_t q1 = lds_tr16(s + qoff[a] + ks * 4096 + 1024);
                    pf[kk][a] = (bf16x8_t){p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                    qf[kk][a] = (bf16x8_t){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = mfma32<DT>(pf[kk][a], qf[kk][b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // D[row = I][col = J]: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ii = i0 + wi + a * 32 + (e & 3) + 8 * (e >> 2) + rbase;
                const int jj = j0 + wj + b * 32 + col;
                atomicAdd(p.out + (size_t)ii * p.ldo + jj, acc[a][b][e]);
            }
}

// ------------------------------------------------------------------------------------------------
// TN, 256x256 output tile, 8 waves (2 x 4), 128(I) x 64(J) per wave = 4 x 2 MFMA 32x32x16, phase-pipelined like
// gemm_nt256_kernel: a K-step (64 rows of m) is 4 phases of 8 MFMAs (one 16-row slab each); in every phase the wave
// also reads the NEXT slab's 12 transposed fragments halves into the other register set (inline asm, counted waits tied
// to the registers) and the LDS-DMA of the following stages is issued in P3 / P0 so that every piece has >= 2 phases
// to land.  One barrier per K-step.  LDS image per operand: 64 rows x 512 B, physical chunk = chunk ^ ((row & 3) << 2)
// (the 128-column image's swizzle: a 512-B row is two bank rows, the XOR acts on the chunk's low 4 bits).
// Fewer, larger blocks also fill the chip better: c_fc wgrad = 36 tiles x 7 row chunks = 252 blocks on 256 CUs,
// against 144 x 3 = 432 blocks of the 128x128 kernel on 512 slots.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(SigGemmTN p) {
    constexpr int ROWB = 512, OPB = 64 * ROWB, STAGE = 2 * OPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tj = p.J >> 8, ti = p.I >> 8, tiles = ti * tj;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / tiles, t = id - split * tiles;
    const int tile_i = t / tj, tile_j = t - tile_i * tj;
    const int i0 = tile_i << 8, j0 = tile_j << 8;
    const int mbeg = split * p.m_chunk;
    int mend = mbeg + p.m_chunk;
    if (mend > p.Mr) mend = p.Mr;
    const int nk = (mend - mbeg) >> 6;
    if (nk <= 0) return;

    // DMA: a 1-KB piece = 2 rows x 512 B; 32 pieces per operand and stage, 4 + 4 per wave.  Source address = uniform
    // base (+ kt * 64 rows, scalar) + a constant per-lane byte offset.
    unsigned po[4], qo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 2 + (lane >> 5);
        const int c = (lane & 31) ^ ((r & 3) << 2);
        po[j] = (unsigned)(r * p.ldp + c * 8) * 2u;
        qo[j] = (unsigned)(r * p.ldq + c * 8) * 2u;
    }
    const bf16_t* pbase = p.P + (size_t)mbeg * p.ldp + i0;
    const bf16_t* qbase = p.Q + (size_t)mbeg * p.ldq + j0;
    const size_t pstep = (size_t)64 * p.ldp, qstep = (size_t)64 * p.ldq;
    auto dma_p = [&](int j, int kt, int stage) { glds16_untracked_s(pbase + kt * pstep, po[j], smem + stage * STAGE + (wave * 4 + j) * 1024); };
    auto dma_q = [&](int j, int kt, int stage) { glds16_untracked_s(qbase + kt * qstep, qo[j], smem + stage * STAGE + OPB + (wave * 4 + j) * 1024); };

    // transposed-read addressing (see gemm_tn_kernel): lane = 16*G + 4*q + pp
    const int G = lane >> 4, h = G >> 1, tq = (lane >> 2) & 3, pp = lane & 3;
    const int wi = (wave >> 2) * 128, wj = (wave & 3) * 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned foff[6];   // 0..3 P column tiles, 4..5 Q column tiles
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int col = wi + a * 32 + 16 * (G & 1) + 4 * pp;
        foff[a] = lds0 + (8 * h + tq) * ROWB + ((((col >> 3) ^ (tq << 2)) << 4) | ((pp & 1) << 3));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = wj + b * 32 + 16 * (G & 1) + 4 * pp;
        foff[4 + b] = lds0 + OPB + (8 * h + tq) * ROWB + ((((col >> 3) ^ (tq << 2)) << 4) | ((pp & 1) << 3));
    }
    struct Frags { bf16x4_t lo[6], hi[6]; };   // 0..3: P column tiles, 4..5: Q column tiles; lo = rows +0..3, hi = rows +4..7
    Frags X, Y;
#define SIG_RDTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto rd = [&](int stage, auto ks_c, Frags& f) {
        constexpr int KS = decltype(ks_c)::value;
        const unsigned so = stage * STAGE;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const unsigned ad = foff[a] + so;
            SIG_RDTR(f.lo[a], ad, KS * 16 * ROWB);
            SIG_RDTR(f.hi[a], ad, KS * 16 * ROWB + 4 * ROWB);
        }
    };
#define SIG_WAITF(n, f)                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                                \
                 : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), \
                   "+v"(f.hi[3]), "+v"(f.lo[4]), "+v"(f.hi[4]), "+v"(f.lo[5]), "+v"(f.hi[5]))
    f32x16_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    auto mma = [&](const Frags& f) {
        bf16x8_t pf[4], qf[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) pf[a] = __builtin_shufflevector(f.lo[a], f.hi[a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int b = 0; b < 2; ++b) qf[b] = __builtin_shufflevector(f.lo[4 + b], f.hi[4 + b], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mfma32<DT>(pf[a], qf[b], acc[a][b]);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

#pragma unroll
    for (int j = 0; j < 4; ++j) { dma_p(j, 0, 0); dma_q(j, 0, 0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(0, K0{}, X);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_p(j, 1, 1);
    }
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        const int st = kt & 1;
        // P0: X (12 reads) outstanding, Y issued behind it
        rd(st, K1{}, Y);
        if (MORE) { dma_q(0, kt + 1, st ^ 1); dma_q(1, kt + 1, st ^ 1); dma_q(2, kt + 1, st ^ 1); dma_q(3, kt + 1, st ^ 1); }
        SIG_WAITF(12, X);
        __builtin_amdgcn_sched_barrier(0);
        mma(X);
        __builtin_amdgcn_sched_barrier(0);
        // P1
        rd(st, K2{}, X);
        SIG_WAITF(12, Y);
        __builtin_amdgcn_sched_barrier(0);
        mma(Y);
        __builtin_amdgcn_sched_barrier(0);
        // P2
        rd(st, K3{}, Y);
        SIG_WAITF(12, X);
        __builtin_amdgcn_sched_barrier(0);
        mma(X);
        __builtin_amdgcn_sched_barrier(0);
        // stage boundary: this wave's pieces of the next stage landed, its reads of this stage returned
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(Y.lo[0]), "+v"(Y.hi[0]), "+v"(Y.lo[1]), "+v"(Y.hi[1]), "+v"(Y.lo[2]), "+v"(Y.hi[2]), "+v"(Y.lo[3]),
                       "+v"(Y.hi[3]), "+v"(Y.lo[4]), "+v"(Y.hi[4]), "+v"(Y.lo[5]), "+v"(Y.hi[5])::"memory");
        __builtin_amdgcn_s_barrier();
        // P3
        if (MORE) rd(st ^ 1, K0{}, X);
        if (MORE2) { dma_p(0, kt + 2, st); dma_p(1, kt + 2, st); dma_p(2, kt + 2, st); dma_p(3, kt + 2, st); }
        __builtin_amdgcn_sched_barrier(0);
        mma(Y);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    for (int kt = 0; kt < nk - 2; ++kt) step(kt, T_{}, T_{});
    if (nk >= 2) step(nk - 2, T_{}, F_{});
    step(nk - 1, F_{}, F_{});

    // D[row = I][col = J]: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): one register = two 128-B rows.
    // The block's 256x256 partial goes to its own workspace tile with plain stores (~6 TB/s); tn_reduce_kernel adds the
    // row chunks into out.  Memory-side f32 atomics (1.3 TB/s) cost ~40 us of a ~130 us weight gradient here.
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float* wt = p.ws ? p.ws + (size_t)id * 65536 : nullptr;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int il = wi + a * 32 + (e & 3) + 8 * (e >> 2) + rbase;
                const int jl = wj + b * 32 + col;
#ifdef SIG_TN_NOFLUSH   // diagnostic build (tools/tn_noflush.py): time the kernel without its flush
                if (acc[a][b][e] == 1.2345e30f) p.out[(size_t)(i0 + il) * p.ldo + j0 + jl] = 0.f;
#else
                if (wt) wt[il * 256 + jl] = acc[a][b][e];
                else atomicAdd(p.out + (size_t)(i0 + il) * p.ldo + j0 + jl, acc[a][b][e]);
#endif
            }
}

// ------------------------------------------------------------------------------------------------
// TN 256x256, second form: MFMA 16x16x32 instead of 32x32x16.  Same tile, waves, LDS-DMA schedule and phase structure as
// gemm_tn256_kernel; per wave 8 x 4 accumulator tiles of 16x16 (the same 128 registers).  Why: on LDS-fed bf16 loops the
// chip holds a higher clock on the 16x16x32 shape at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7:
// 1.12-1.14x), and the 32x32 shape was chosen only for its atomics-friendly accumulator layout, which the partial-tile
// flush no longer needs.  A K-step (64 rows of m) = 2 chunks of 32 rows; each chunk is two phases (P column tiles 0-3 /
// 4-7 against the chunk's 4 Q tiles).  Fragments: lane (col = l & 15, k-group G = l >> 4) holds 8 consecutive m of one
// column = two ds_read_b64_tr_b16 (rows +0..3, +4..7).  LDS image as before but with the XOR extended by row bit 3:
//   physical chunk = chunk ^ (((row & 3) << 2) | (((row >> 3) & 1) << 1))
// -- the two k-groups of a half-wave (rows r, r + 8) read the SAME 16 columns here, and the extra bit moves them to the two
// halves of each 64-B quarter of the bank row (tools/lds_bank_sim.py: conflict-free).
// The block's partial tile is stored in FRAGMENT order (one coalesced 1-KB store per MFMA tile and wave); tn_reduce16_kernel
// un-permutes while it adds the row chunks.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(512, 2) void gemm_tn256x16_kernel(SigGemmTN p) {
    constexpr int ROWB = 512, OPB = 64 * ROWB, STAGE = 2 * OPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tj = p.J >> 8, ti = p.I >> 8, tiles = ti * tj;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / tiles, t = id - split * tiles;
    const int tile_i = t / tj, tile_j = t - tile_i * tj;
    const int i0 = tile_i << 8, j0 = tile_j << 8;
    const int mbeg = split * p.m_chunk;
    int mend = mbeg + p.m_chunk;
    if (mend > p.Mr) mend = p.Mr;
    const int nk = (mend - mbeg) >> 6;
    if (nk <= 0) return;

    unsigned po[4], qo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 2 + (lane >> 5);
        const int c = (lane & 31) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
        po[j] = (unsigned)(r * p.ldp + c * 8) * 2u;
        qo[j] = (unsigned)(r * p.ldq + c * 8) * 2u;
    }
    const bf16_t* pbase = p.P + (size_t)mbeg * p.ldp + i0;
    const bf16_t* qbase = p.Q + (size_t)mbeg * p.ldq + j0;
    const size_t pstep = (size_t)64 * p.ldp, qstep = (size_t)64 * p.ldq;
    auto dma_p = [&](int j, int kt, int stage) { glds16_untracked_s(pbase + kt * pstep, po[j], smem + stage * STAGE + (wave * 4 + j) * 1024); };
    auto dma_q = [&](int j, int kt, int stage) { glds16_untracked_s(qbase + kt * qstep, qo[j], smem + stage * STAGE + OPB + (wave * 4 + j) * 1024); };

    // transposed-read addressing: lane = 16*G + 4*tq + pp addresses row (8G + tq), columns 4pp.. of a 16-column tile and
    // receives column (4tq + pp) = lane & 15, rows 8G .. 8G+3 (+4 for the second read)
    const int G = lane >> 4, tq = (lane >> 2) & 3, pp = lane & 3;
    const int wi = (wave >> 2) * 128, wj = (wave & 3) * 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int swz = (tq << 2) | ((G & 1) << 1);
    unsigned fp[8], fq[4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
        fp[a] = lds0 + (8 * G + tq) * ROWB + (((((wi >> 3) + 2 * a + (pp >> 1)) ^ swz)) << 4) + ((pp & 1) << 3);
#pragma unroll
    for (int b = 0; b < 4; ++b)
        fq[b] = lds0 + OPB + (8 * G + tq) * ROWB + (((((wj >> 3) + 2 * b + (pp >> 1)) ^ swz)) << 4) + ((pp & 1) << 3);
    struct Frag4 { bf16x4_t lo[4], hi[4]; };
    Frag4 pX, pY, qX, qY;
    auto rd_p = [&](int stage, auto c_c, auto h_c, Frag4& f) {
        constexpr int C = decltype(c_c)::value, H = decltype(h_c)::value;
        const unsigned so = stage * STAGE;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const unsigned ad = fp[H * 4 + a] + so;
            SIG_RDTR(f.lo[a], ad, C * 32 * ROWB);
            SIG_RDTR(f.hi[a], ad, C * 32 * ROWB + 4 * ROWB);
        }
    };
    auto rd_q = [&](int stage, auto c_c, Frag4& f) {
        constexpr int C = decltype(c_c)::value;
        const unsigned so = stage * STAGE;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned ad = fq[b] + so;
            SIG_RDTR(f.lo[b], ad, C * 32 * ROWB);
            SIG_RDTR(f.hi[b], ad, C * 32 * ROWB + 4 * ROWB);
        }
    };
#define SIG_WAITF4(n, f)                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                                \
                 : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3]))
#define SIG_WAITF8(n, f, g)                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                                \
                 : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3]), \
                   "+v"(g.lo[0]), "+v"(g.hi[0]), "+v"(g.lo[1]), "+v"(g.hi[1]), "+v"(g.lo[2]), "+v"(g.hi[2]), "+v"(g.lo[3]), "+v"(g.hi[3]))
    f32x4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // rows [r0, r1) of the 4 x 4 tile block `half` : D[i][j] += sum_m P[m][i] Q[m][j]   (A operand = P fragment, B = Q)
    auto mma = [&](int half, const Frag4& pf, const Frag4& qf, auto lo_c, auto hi_c) {
        constexpr int A0 = decltype(lo_c)::value, A1 = decltype(hi_c)::value;
#pragma unroll
        for (int a = A0; a < A1; ++a) {
            const bf16x8_t pa = __builtin_shufflevector(pf.lo[a], pf.hi[a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bf16x8_t qb = __builtin_shufflevector(qf.lo[b], qf.hi[b], 0, 1, 2, 3, 4, 5, 6, 7);
                acc[half * 4 + a][b] = mfma16<DT>(pa, qb, acc[half * 4 + a][b]);
            }
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C4 = std::integral_constant<int, 4>;

#pragma unroll
    for (int j = 0; j < 4; ++j) { dma_p(j, 0, 0); dma_q(j, 0, 0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd_p(0, C0{}, C0{}, pX);
    rd_q(0, C0{}, qX);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_p(j, 1, 1);
    }
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        const int st = kt & 1;
        // P0: pX, qX (16 reads) in flight; pY behind them
        rd_p(st, C0{}, C1{}, pY);
        if (MORE) { dma_q(0, kt + 1, st ^ 1); dma_q(1, kt + 1, st ^ 1); dma_q(2, kt + 1, st ^ 1); dma_q(3, kt + 1, st ^ 1); }
        SIG_WAITF8(8, pX, qX);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, pX, qX, C0{}, C4{});
        __builtin_amdgcn_sched_barrier(0);
        // P1: pY landed long ago; next chunk's P half behind it, its Q tiles issued half-way through the MFMAs (lgkmcnt is a
        // 4-bit counter: never more than 16 reads requested at once)
        rd_p(st, C1{}, C0{}, pX);
        SIG_WAITF4(8, pY);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, pY, qX, C0{}, C2{});
        __builtin_amdgcn_sched_barrier(0);
        rd_q(st, C1{}, qY);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, pY, qX, C2{}, C4{});
        __builtin_amdgcn_sched_barrier(0);
        // P2: outstanding pX, qY, then pY
        rd_p(st, C1{}, C1{}, pY);
        SIG_WAITF8(8, pX, qY);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, pX, qY, C0{}, C4{});
        __builtin_amdgcn_sched_barrier(0);
        // stage boundary
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(pY.lo[0]), "+v"(pY.hi[0]), "+v"(pY.lo[1]), "+v"(pY.hi[1]), "+v"(pY.lo[2]), "+v"(pY.hi[2]), "+v"(pY.lo[3]),
                       "+v"(pY.hi[3])::"memory");
        __builtin_amdgcn_s_barrier();
        // P3
        if (MORE) {
            rd_p(st ^ 1, C0{}, C0{}, pX);
            rd_q(st ^ 1, C0{}, qX);
        }
        if (MORE2) { dma_p(0, kt + 2, st); dma_p(1, kt + 2, st); dma_p(2, kt + 2, st); dma_p(3, kt + 2, st); }
        __builtin_amdgcn_sched_barrier(0);
        mma(1, pY, qY, C0{}, C4{});
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    for (int kt = 0; kt < nk - 2; ++kt) step(kt, T_{}, T_{});
    if (nk >= 2) step(nk - 2, T_{}, F_{});
    step(nk - 1, F_{}, F_{});

    // D: col (j) = lane & 15, row (i) = (lane >> 4) * 4 + reg
    if (p.ws) {
        f32x4_t* wt = (f32x4_t*)(p.ws + (size_t)id * 65536) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) wt[(a * 4 + b) * 64] = acc[a][b];
    } else {
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    atomicAdd(p.out + (size_t)(i0 + wi + a * 16 + (lane >> 4) * 4 + e) * p.ldo + j0 + wj + b * 16 + (lane & 15), acc[a][b][e]);
    }
}

// out += sum over row chunks of the FRAGMENT-order partial tiles of gemm_tn256x16_kernel: float4 q of a tile = (wave, a, b, lane)
// holds rows i = wi + 16a + 4(lane >> 4) .. +3 of column j = wj + 16b + (lane & 15)
__global__ __launch_bounds__(256) void tn_reduce16_kernel(const float* __restrict__ ws, float* __restrict__ out, int I, int J, int ldo,
                                                          int tiles, int split) {
    // a block = the 4 column tiles b of one (wave, a): 16 rows x 64 columns of the output
    __shared__ float tile[16][68];
    const int tj = J >> 8;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;      // float4 index over all tiles (grid = tiles * 64 exactly)
    const int t = (int)(q >> 14), r = (int)(q & 16383);
    const int lane = r & 63, f = r >> 6, wave = f >> 5, a = (f >> 2) & 7, b = f & 3;
    const f32x4_t* src = (const f32x4_t*)(ws + (size_t)t * 65536) + r;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < split; s0 += 8) {          // 8 partial tiles requested before any is added (fixed order: deterministic)
        f32x4_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(s0 + k < split ? s0 + k : s0) * tiles * 16384];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < split) acc += v[k];
    }
    // fragment order -> rows: the read-modify-write of the output is then 16 B per lane on 256-B row segments instead of
    // 4 B per lane (a dword store costs a store instruction's ~50 cycles for 256 B)
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(lane >> 4) * 4 + e][b * 16 + (lane & 15)] = acc[e];
    __syncthreads();
    const int row = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
    const int i = (t / tj) * 256 + (wave >> 2) * 128 + a * 16 + row;
    const int j = (t % tj) * 256 + (wave & 3) * 64 + c4;
    float* o = out + (size_t)i * ldo + j;
    const f32x4_t add = *(const f32x4_t*)&tile[row][c4];
    if ((ldo & 3) == 0) {
        *(f32x4_t*)o = *(const f32x4_t*)o + add;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += add[e];
    }
}

// out[i][j] += sum over row chunks of the partial tiles written by gemm_tn256_kernel (unit = split * tiles + tile)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int I, int J, int ldo,
                                                        int tiles, int split) {
    const int tj = J >> 8;
    const int q = blockIdx.x * 256 + threadIdx.x;           // one float4 of the output
    const int per_row = J >> 2;
    if (q >= I * per_row) return;
    const int i = q / per_row, j = (q - i * per_row) << 2;
    const int t = (i >> 8) * tj + (j >> 8);
    const float* src = ws + (size_t)t * 65536 + (i & 255) * 256 + (j & 255);
    f32x4_t acc = *(const f32x4_t*)src;
    for (int s = 1; s < split; ++s) acc += *(const f32x4_t*)(src + (size_t)s * tiles * 65536);
    float* o = out + (size_t)i * ldo + j;
    if ((ldo & 3) == 0) {
        *(f32x4_t*)o = *(const f32x4_t*)o + acc;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += acc[e];
    }
}

// per-(device, stream, slot) scratch (slot 0: the partial tiles of the 256x256 TN kernel, <= 256 blocks x 256 KB; slot 1: the
// per-block column sums of the LayerNorm backward); grown on demand, never freed
float* sig_stream_scratch(hipStream_t st, size_t bytes, int slot) {
    struct Ent { int dev; hipStream_t st; int slot; float* p; size_t bytes; };
    static Ent ents[32];
    static int n = 0;
    static std::mutex mu;   // forward runs on the caller's thread, backward on autograd's
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < n; ++i)
        if (ents[i].st == st && ents[i].dev == dev && ents[i].slot == slot) {
            if (ents[i].bytes < bytes) {
                (void)hipStreamSynchronize(st);
                (void)hipFree(ents[i].p);
                if (hipMalloc((void**)&ents[i].p, bytes) != hipSuccess) { ents[i].p = nullptr; ents[i].bytes = 0; return nullptr; }
                (void)hipMemsetAsync(ents[i].p, 0, bytes, st);      // new scratch starts zero-filled (arrival counters rely on it)
                ents[i].bytes = bytes;
            }
            return ents[i].p;
        }
    if (n == 32) return nullptr;   // more (device, stream) pairs than slots: the caller falls back to atomics
    float* ptr = nullptr;
    if (hipMalloc((void**)&ptr, bytes) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(ptr, 0, bytes, st);
    ents[n++] = {dev, st, slot, ptr, bytes};
    return ptr;
}
static float* tn_workspace(hipStream_t st, size_t bytes) { return sig_stream_scratch(st, bytes, 0); }

// SIG_GEMM_TN_TILE=<128|256> / sig_tune_tn_path(): pin the weight-gradient path (tests, A/B runs): 128 = the 128x128 kernel
// with f32 atomics, one launch per weight; 256 = the 256x256 kernel, one launch per weight; 0 = default (grouped per block)
static std::atomic<int> g_force_tn{-1};
int sig_tune_tn_path_impl(int path) {
    (void)sig_tn_path();                         // resolves the SIG_GEMM_TN_TILE preset first: restoring `prev` keeps it
    const int cur = g_force_tn, prev = cur < 0 ? 0 : cur;
    g_force_tn = path;
    return prev;
}
int sig_tn_path() {
    if (g_force_tn < 0) { const char* e = getenv("SIG_GEMM_TN_TILE"); g_force_tn = e ? atoi(e) : 0; }
    return g_force_tn;
}

template <int DT>
static int launch_tn(const SigGemmTN& p_in, hipStream_t st) {
    SigGemmTN p = p_in;
    SIG_CHECK_ARG(p.Mr > 0 && (p.Mr & 63) == 0, "gemm_tn: row count %d must be a positive multiple of 64 (pad rows zeroed)", p.Mr);
    SIG_CHECK_ARG((p.I & 127) == 0 && (p.J & 127) == 0 && p.I > 0 && p.J > 0, "gemm_tn: I=%d, J=%d must be multiples of 128", p.I, p.J);
    SIG_CHECK_ARG((p.ldp & 7) == 0 && (p.ldq & 7) == 0 && p.ldp >= p.I && p.ldq >= p.J && p.ldo >= p.J, "gemm_tn: bad leading dimension");
    SIG_CHECK_ARG(p.P && p.Q && p.out, "gemm_tn: null operand");
    static std::once_flag attr_done;
    std::call_once(attr_done, [] {
        (void)hipFuncSetAttribute((const void*)&gemm_tn_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        });
    const int ksteps = p.Mr >> 6;
    if (g_force_tn < 0) { const char* e = getenv("SIG_GEMM_TN_TILE"); g_force_tn = e ? atoi(e) : 0; }
    const int force = g_force_tn;
    // 256x256 tiles for the large weight gradients (one block per CU, <= 256 blocks); the rest on 128x128 tiles
    const bool can256 = (p.I & 255) == 0 && (p.J & 255) == 0 && (p.I >> 8) * (p.J >> 8) <= 128;
    // (also the small square ones: 768x768 out_proj wgrad 63.7 -> 48.2 us) as long as a row chunk keeps >= 8 K-steps
    bool big = can256 && ksteps >= 64 && (long long)ksteps * ((p.I >> 8) * (p.J >> 8)) >= 2048;
    if (force == 128) big = false;
    if (force == 256) big = can256;
    if (big) {
        static std::once_flag attr256;
        std::call_once(attr256, [] {
            (void)hipFuncSetAttribute((const void*)&gemm_tn256_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            (void)hipFuncSetAttribute((const void*)&gemm_tn256x16_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            });
        const int tiles = (p.I >> 8) * (p.J >> 8);
        int split = p.split > 0 ? p.split : free_cus() / tiles;      // one block per CU, one round
        if (split < 1) split = 1;
        if (split > ksteps) split = ksteps;
        const int per = sig_ceil_div(ksteps, split);
        split = sig_ceil_div(ksteps, per);
        p.m_chunk = per * 64;
        static int use_ws = -1;
        if (use_ws < 0) { const char* e = getenv("SIG_GEMM_TN_ATOMICS"); use_ws = e && atoi(e) ? 0 : 1; }
        p.ws = use_ws ? tn_workspace(st, (size_t)tiles * split * 65536 * sizeof(float)) : nullptr;
        // bench.py's roofline leg: class SIG_PROF_TN256 times every launch of this kernel (N/K = 0) or one (I, J) shape
        const bool timed = g_prof.on && g_prof.epi == SIG_PROF_TN256 && (g_prof.N == 0 || (g_prof.N == p.I && g_prof.K == p.J)) &&
                           g_prof.used + 2 <= g_prof.ev.size();
        static int shape = -1;      // SIG_TN256_SHAPE=32: the 32x32x16 form (A/B); default: 16x16x32
        if (shape < 0) { const char* e = getenv("SIG_TN256_SHAPE"); shape = e ? atoi(e) : 16; }
        if (timed) (void)hipEventRecord(g_prof.ev[g_prof.used], st);
        if (shape == 32) hipLaunchKernelGGL(gemm_tn256_kernel<DT>, dim3(tiles * split), dim3(512), 131072, st, p);
        else hipLaunchKernelGGL(gemm_tn256x16_kernel<DT>, dim3(tiles * split), dim3(512), 131072, st, p);
        if (timed) {
            (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
            g_prof.used += 2;
            g_prof.flops += 2.0 * p.Mr * p.I * p.J;
        }
        SIG_CHECK_LAUNCH("gemm_tn256");
        if (p.ws && shape != 32) {
            hipLaunchKernelGGL(tn_reduce16_kernel, dim3(tiles * 64), dim3(256), 0, st, p.ws, p.out, p.I, p.J, p.ldo, tiles, split);
            SIG_CHECK_LAUNCH("tn_reduce16");
        } else if (p.ws) {
            hipLaunchKernelGGL(tn_reduce_kernel, dim3(sig_ceil_div(p.I * (p.J >> 2), 256)), dim3(256), 0, st, p.ws, p.out, p.I, p.J, p.ldo,
                               tiles, split);
            SIG_CHECK_LAUNCH("tn_reduce");
        }
        return 0;
    }
    const int tiles = (p.I >> 7) * (p.J >> 7);
    // every split re-adds a whole 128x128 f32 tile with atomics (64 KB per workgroup, ~1.3 TB/s chip-wide), so use the
    // FEWEST row chunks that still fill the chip once: tiles * split <= 512 resident workgroups (2 per CU)
    // (measured, tools/tn_split_sweep.py: with few tiles the atomic flush dominates and ~one block per CU wins --
    // 768x768: 14 chunks 71 us, 7 chunks 62 us; 768x512: 21 chunks 53 us, 8 chunks 43 us)
    const int slots = tiles < 100 ? 256 : 512;
    int split = p.split > 0 ? p.split : (slots / tiles > 0 ? slots / tiles : 1);
    if (split > ksteps) split = ksteps;
    const int per = sig_ceil_div(ksteps, split);
    split = sig_ceil_div(ksteps, per);
    p.m_chunk = per * 64;
    hipLaunchKernelGGL(gemm_tn_kernel<DT>, dim3(tiles * split), dim3(256), 65536, st, p);
    SIG_CHECK_LAUNCH("gemm_tn");
    return 0;
}

int sig_launch_gemm_tn(const SigGemmTN& p, hipStream_t st) {
    SIG_CHECK_DT(p.dt, "gemm_tn");
    return p.dt == SIG_DT_F16 ? launch_tn<SIG_DT_F16>(p, st) : launch_tn<SIG_DT_BF16>(p, st);
}
