// Persistent, software-pipelined NT GEMM for the wide K = 768 contractions (qkv, c_fc, GELU' dgrad).
//
// Why: at K = 768 a 256x256 / 320x256 tile is 12 K-steps of main loop behind a ~2 k-cycle prologue and an 8-16 k-cycle
// store tail, and every CU of a round reaches its tail at the same time: the 114-304 MB of a launch's outputs leave the
// chip in bursts that are HBM-bound (14-16 B/clk per CU) while no MFMA runs.  This kernel keeps ONE
// workgroup per CU that walks its tiles with a K-step sequence that never drains:
//   * the LDS-DMA of the next tile's first two stages is issued from the last two K-steps of the current tile, so the
//     stage ring runs straight through the tile boundary (no prologue after the first tile);
//   * at a tile's last K-step the accumulators are finished in registers (bias / GELU, rounded to 16 bit) and PARKED:
//     half of the wave's 24 store units (4 rows x 128 B each) in 24 VGPRs, half in the 48 KB of LDS beyond the two
//     operand stages (lane-private 8-B slots, no transposition).  The units are stored during the NEXT tile's main
//     loop, two to four store instructions per K-step and wave, issued behind the K-step's DMA pieces so that the
//     stage-boundary wait (in-order vmcnt) never sits on them;
//   * the tile is 192 x 256 (8 waves x 96 x 64, 6 x 4 MFMA 16x16x32 tiles = 96 accumulator registers) so that the
//     parked units and the fragment double buffer fit next to the accumulators at two waves per SIMD -- a 256x256
//     tile fills the register file with accumulators and fragments alone.  24768 token rows are 129 x 192: no ragged tile.
// Natural MFMA orientation and the B-row permutation of gemm_nt320_kernel (a lane's four accumulators of a row are
// four consecutive columns): a store unit is one 8-B store per lane, consecutive lanes on consecutive addresses.
//
// Waits are counted by hand: LDS returns in order, so reads placed BEFORE a phase's fragment reads never change the
// phase's lgkmcnt arguments; vector-memory operations retire in order, so the stage-boundary wait is vmcnt(number of
// stores issued after this K-step's last DMA piece).  Under-counting that number only over-waits.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "sig_common.h"
#include "sig_kernels.h"

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#ifdef SIG_GEMM_STAMPS   // diagnostic build only (tools/persist_stamps.py): where does a workgroup's time go?
// per workgroup: [0] start, [1] first stage landed, then per tile t < 7: [2 + 2t] last K-step done, [3 + 2t] conversion done; [16] end
__device__ unsigned long long g_pstamps[256 * 17];
extern "C" int sig_debug_read_pstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 256 * 17) == hipSuccess ? 0 : 2;
}
#define PP_STAMP(slot)                                                                                        \
    do {                                                                                                      \
        unsigned long long v_;                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");                            \
        if (tid == 0 && blockIdx.x < 256 && (slot) < 17) g_pstamps[blockIdx.x * 17 + (slot)] = v_;              \
    } while (0)
#else
#define PP_STAMP(slot)
#endif

#define PP_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PP_WAIT3(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]))
#define PP_WAIT7(n, a, b) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]))

// 8-B store, address = uniform base (SGPR pair) + per-lane byte offset; counted by hand (see header)
__device__ __forceinline__ void pp_store8(const void* uniform_base, unsigned lane_off, u32x2_t v) {
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(lane_off), "v"(v), "s"(uniform_base) : "memory");
}

template <int EPI, int DT>
__global__ __launch_bounds__(512, 2) void gemm_nt192p_kernel(SigGemmNT p, int ntm, int tiles) {
    constexpr int BM = 192, STAGE = (BM + 256) * 128;      // 57344 B per stage
    constexpr int HELD = 2 * STAGE;                        // 8 waves x 12 slots x 512 B of parked store units
    constexpr bool HAS_BIAS = EPI == SIG_EPI_BIAS_BF16 || EPI == SIG_EPI_BIAS_GELU_BF16;
    constexpr bool GELU_FWD = EPI == SIG_EPI_BIAS_GELU_BF16;
    constexpr bool UMUL = EPI == SIG_EPI_DGELU_BF16;          // out = acc * aux (the saved QuickGELU'), + optional column sums
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 6;
    PP_STAMP(0);

    // ---- this workgroup's tiles: XCD x owns a contiguous range of the band-major tile order, its workgroups take
    //      consecutive tiles of it round by round (what an XCD's L2 sees per round is 32 neighbouring tiles)
    const int spx = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tq = tiles >> 3, tr = tiles & 7;
    const int cstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int csize = tq + (xcd < tr ? 1 : 0);
    if (slot >= csize) return;                              // (uniform; never with grid <= tiles)
    const int wb = p.band, per = ntm * wb;
    auto tile_base = [&](int l, int& m0, int& n0) {
        const int id = cstart + l;
        const int bnd = id / per, rr = id - bnd * per, tile_m = rr / wb;
        m0 = tile_m * BM;
        n0 = (bnd * wb + rr - tile_m * wb) << 8;
    };

    // ---- DMA pieces (8 rows x 128 B): A 24 = 3 per wave, B 32 = 4 per wave; per-lane offsets never change
    unsigned ao[3], bo[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = (wave * 3 + j) * 8 + (lane >> 3);
        ao[j] = (unsigned)(r * p.lda + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);             // row of the LDS image
        const int sr = (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);   // ... holds weight row 4 * slot + group of its strip
        bo[j] = (unsigned)(sr * p.ldb + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u;
    }
    auto dma_a = [&](int j, const bf16_t* base, int stage) { glds16_untracked_s(base, ao[j], smem + stage * STAGE + (wave * 3 + j) * 1024); };
    auto dma_b = [&](int j, const bf16_t* base, int stage) { glds16_untracked_s(base, bo[j], smem + stage * STAGE + BM * 128 + (wave * 4 + j) * 1024); };

    const int fr = lane & 15, g = lane >> 4, sw = fr >> 1;
    const int wm = (wave >> 2) * 96, wn = (wave & 3) * 64;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = (((ks << 2) | g) ^ sw) << 4;
        aoff[ks] = (wm + fr) * 128 + ch;
        boff[ks] = BM * 128 + (wn + fr) * 128 + ch;
    }
    bf16x8_t aX[3], aY[3], bX[4], bY[4];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto rd_a = [&](int stage, int ks, auto half_c, bf16x8_t (&a)[3]) {
        constexpr int H = decltype(half_c)::value;
        const unsigned ad = lds0 + stage * STAGE + aoff[ks];
        PP_RD128(a[0], ad, (H * 3 + 0) * 2048);
        PP_RD128(a[1], ad, (H * 3 + 1) * 2048);
        PP_RD128(a[2], ad, (H * 3 + 2) * 2048);
    };
    auto rd_b = [&](int stage, int ks, bf16x8_t (&b)[4]) {
        const unsigned ad = lds0 + stage * STAGE + boff[ks];
        PP_RD128(b[0], ad, 0);
        PP_RD128(b[1], ad, 2048);
        PP_RD128(b[2], ad, 4096);
        PP_RD128(b[3], ad, 6144);
    };
    f32x4_t acc[6][4];
    auto mma = [&](auto first_c, int half, const bf16x8_t (&a)[3], const bf16x8_t (&b)[4]) {
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[half * 3 + i][j] = mfma16<DT>(a[i], b[j], FIRST ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[half * 3 + i][j]);
    };

    // ---- parked store units of the previous tile.  Unit u = 4 i + e is rows i*16 + 4 g + e of this wave's strip
    //      (4 rows x 128 B per instruction); units 0..11 wait in registers, 12..23 in lane-private LDS slots.
    //      GELU' dgrad (out = acc * aux): the SAME storage carries the next conversion's multiplier in between (see the plan
    //      in `step`); the conversion multiplies in place.  24 registers + 6 KB of LDS per wave either way.
    u32x2_t hr[12];
    const unsigned hbase = lds0 + HELD + wave * 6144 + lane * 8;
    const unsigned lane_out = (unsigned)((wm + 4 * g) * p.ldo + wn + 4 * fr) * 2u;       // byte offset inside a tile's output
    const unsigned lane_aux = (UMUL || GELU_FWD) ? (unsigned)((wm + 4 * g) * p.ldaux + wn + 4 * fr) * 2u : 0u;
    const bool save_u = GELU_FWD && p.aux != nullptr;      // c_fc in training: QuickGELU'(pre-activation) is a second output
    const unsigned lane_bias = (unsigned)(wn + 4 * fr) * 4u;
    const size_t ldo2 = (size_t)p.ldo * 2, ldx2 = (size_t)p.ldaux * 2;
    const char* pout = nullptr;                                                            // previous tile's output base (uniform)
    const char* uin = nullptr;                                                             // this tile's aux base (uniform; GELU' only)
    auto unit_base = [&](const char* tile_out, int u) { return tile_out + (size_t)((u >> 2) * 16 + (u & 3)) * ldo2; };
    auto aux_base = [&](int u) { return uin + (size_t)((u >> 2) * 16 + (u & 3)) * ldx2; };
    auto load8 = [&](u32x2_t& dst, const void* uniform_base, unsigned lane_off) {
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(uniform_base) : "memory");
    };

    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    // One K-step.  KT = its index inside the tile when < 12 (compile-time: everything that happens beside the MFMAs is a
    // fixed plan per K-step), -1 for the K-steps beyond the twelfth (K > 768: nothing to drain any more).
    //   bsrc / bval: B panel of the NEXT K-step (same tile or the next one); asrc / aval: A panel of the K-step after it.
    auto step = [&](auto kt_c, int st, const bf16_t* bsrc, bool bval, const bf16_t* asrc, bool aval) {
        constexpr int KT = decltype(kt_c)::value;
        constexpr bool FIRST = KT == 0;
        // the plan: K-steps 0..2 the twelve register units leave (four each), 3..5 the twelve LDS units (four each).  GELU'
        // dgrad: from K-step 6 on the 24 unit registers are free; K-step 6 requests the aux values of the LDS units INTO them,
        // K-step 7 moves them to the (emptied) slots, K-step 8 requests the register units' own aux values.  Two batches of
        // twelve loads, each right behind its K-step's DMA pieces: vector memory retires in order, so the next K-step's stage
        // boundary waits for them -- HBM latency against the DMA's L2 latency -- and two such boundaries cost less than six.
        constexpr int RU = (KT >= 0 && KT <= 2) ? 4 * KT : -1;
        constexpr int LU = (KT >= 3 && KT <= 5) ? 12 + 4 * (KT - 3) : -1;
        constexpr bool ULS = UMUL && KT == 6, UWS = UMUL && KT == 7, ULH = UMUL && KT == 8;
        constexpr int NLOAD = (ULS || ULH) ? 12 : 0, NSTORE = (RU >= 0 ? 4 : 0) + (LU >= 0 ? 4 : 0);
        u32x2_t l[4];
        // P0: outstanding aX,bX (7) + aY (3).  The LDS units of this K-step are read whether or not there is a previous tile
        // (an asm output written on one side of a branch makes the compiler merge registers with copies, and a copy of a
        // register whose ds_read is still in flight copies stale bits); they sit BEFORE the fragment reads: the counted waits
        // below only name the newest reads.
        if constexpr (LU >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l[q]) : "v"(hbase), "n"((LU - 12) * 512 + q * 512));
        }
        rd_a(st, 0, H1{}, aY);
        if (bval) { dma_b(0, bsrc, st ^ 1); dma_b(1, bsrc, st ^ 1); dma_b(2, bsrc, st ^ 1); dma_b(3, bsrc, st ^ 1); }
        if constexpr (ULS || ULH) {      // (behind the K-step's last DMA piece: the boundary wait below leaves exactly these in flight)
#pragma unroll
            for (int q = 0; q < 12; ++q) load8(hr[q], aux_base((ULS ? 12 : 0) + q), lane_aux);
        }
        PP_WAIT7(3, aX, bX);
        if constexpr (LU >= 0) asm volatile("" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));      // (landed: older than aX, bX)
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<bool, FIRST>{}, 0, aX, bX);
        __builtin_amdgcn_sched_barrier(0);
        // P1: outstanding aY (3) + aX,bY (7)
        rd_a(st, 1, H0{}, aX);
        rd_b(st, 1, bY);
        PP_WAIT3(7, aY);
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<bool, FIRST>{}, 1, aY, bX);
        __builtin_amdgcn_sched_barrier(0);
        // the K-step's stores and aux requests: behind its last DMA piece, so that the boundary wait below can leave exactly
        // them in flight (vector-memory operations retire in order)
        if constexpr (RU >= 0) {
            if (pout) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pp_store8(unit_base(pout, RU + q), lane_out, hr[RU + q]);
            }
        }
        // P2: outstanding aX,bY (7) + aY (3)
        rd_a(st, 1, H1{}, aY);
        PP_WAIT7(3, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        mma(F_{}, 0, aX, bY);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LU >= 0) {
            if (pout) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pp_store8(unit_base(pout, LU + q), lane_out, l[q]);
            }
        }
        // stage boundary: this wave's DMA pieces of the next stage landed, all its LDS reads of this stage returned
        // (the vmcnt statements carry no register operands: nothing to merge across the branches)
        if (NSTORE && pout) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD + NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aY[0]), "+v"(aY[1]), "+v"(aY[2])::"memory");
        __builtin_amdgcn_s_barrier();
        // P3
        rd_a(st ^ 1, 0, H0{}, aX);      // (unconditional: after the workgroup's last K-step these fragments are never used)
        rd_b(st ^ 1, 0, bX);
        if (aval) { dma_a(0, asrc, st); dma_a(1, asrc, st); dma_a(2, asrc, st); }
        if constexpr (UWS) {            // requested one K-step ago, behind that K-step's DMA pieces: landed with THIS boundary wait
            asm volatile("" : "+v"(hr[0]), "+v"(hr[1]), "+v"(hr[2]), "+v"(hr[3]), "+v"(hr[4]), "+v"(hr[5]), "+v"(hr[6]), "+v"(hr[7]),
                         "+v"(hr[8]), "+v"(hr[9]), "+v"(hr[10]), "+v"(hr[11]));
#pragma unroll
            for (int q = 0; q < 12; ++q) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(hbase), "v"(hr[q]), "n"(q * 512) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(F_{}, 1, aY, bY);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue of the first tile: stage 0 complete, first fragments, A of stage 1
    const int ntl = (csize - slot + spx - 1) / spx;        // tiles of this workgroup
    int m0, n0;
    tile_base(slot, m0, n0);
    const bf16_t* acur = p.A + (size_t)m0 * p.lda;
    const bf16_t* bcur = p.Bt + (size_t)n0 * p.ldb;
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_a(j, acur, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_b(j, bcur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);
    rd_a(0, 0, H0{}, aX);
    rd_b(0, 0, bX);
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_a(j, acur + 64, 1);

    for (int t = 0; t < ntl; ++t) {
        const bool has_next = t + 1 < ntl;
        int m1 = 0, n1 = 0;
        if (has_next) tile_base(slot + (t + 1) * spx, m1, n1);
        const bf16_t* anext = p.A + (size_t)m1 * p.lda;
        const bf16_t* bnext = p.Bt + (size_t)n1 * p.ldb;
        if constexpr (UMUL) uin = (const char*)p.aux + ((size_t)m0 * p.ldaux + n0) * 2;
        // bias of this tile's columns: requested by hand ahead of K-step 0's DMA pieces (older than them, so K-step 0's
        // boundary wait covers it); a compiler-tracked load would be waited for with vmcnt(0) at its first use -- the
        // tile's end, where the next tile's first stages are in flight
        f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bias4) : "v"(lane_bias), "s"(p.bias + n0) : "memory");
        // the twelve planned K-steps (needs nk >= 12); the last two K-steps of the tile request the next tile's first stages
        auto run = [&](auto kt_c) {
            constexpr int KT = decltype(kt_c)::value;
            const bool b_in = KT + 1 < nk, a_in = KT + 2 < nk;
            const bf16_t* bsrc = b_in ? bcur + (KT + 1) * 64 : bnext;
            const bf16_t* asrc = a_in ? acur + (KT + 2) * 64 : anext + (KT + 2 - nk) * 64;
            step(kt_c, KT & 1, bsrc, b_in || has_next, asrc, a_in || has_next);
        };
        run(std::integral_constant<int, 0>{});
        if constexpr (HAS_BIAS) asm volatile("" : "+v"(bias4));
        run(std::integral_constant<int, 1>{});
        run(std::integral_constant<int, 2>{});
        run(std::integral_constant<int, 3>{});
        run(std::integral_constant<int, 4>{});
        run(std::integral_constant<int, 5>{});
        run(std::integral_constant<int, 6>{});
        run(std::integral_constant<int, 7>{});
        run(std::integral_constant<int, 8>{});
        run(std::integral_constant<int, 9>{});
        if constexpr (UMUL) {           // (the register units' aux values, requested in K-step 8: landed with K-step 9's boundary wait)
            asm volatile("" : "+v"(hr[0]), "+v"(hr[1]), "+v"(hr[2]), "+v"(hr[3]), "+v"(hr[4]), "+v"(hr[5]), "+v"(hr[6]), "+v"(hr[7]),
                         "+v"(hr[8]), "+v"(hr[9]), "+v"(hr[10]), "+v"(hr[11]));
        }
        run(std::integral_constant<int, 10>{});
        run(std::integral_constant<int, 11>{});
        for (int kt = 12; kt < nk; ++kt) {
            const bool b_in = kt + 1 < nk, a_in = kt + 2 < nk;
            const bf16_t* bsrc = b_in ? bcur + (kt + 1) * 64 : bnext;
            const bf16_t* asrc = a_in ? acur + (kt + 2) * 64 : anext + (kt + 2 - nk) * 64;
            step(std::integral_constant<int, -1>{}, kt & 1, bsrc, b_in || has_next, asrc, a_in || has_next);
        }
        PP_STAMP(2 + 2 * t);
        // ---- the tile is complete: finish it in registers and park it (every unit of the previous tile left in K-steps 0..8)
        const char* tout = (const char*)p.out + ((size_t)m0 * p.ldo + n0) * 2;
        // the second output of the training c_fc has no parking space left (24 units of y fill the registers and the LDS slots): its
        // units are stored as they are produced, behind this tile's last DMA piece and ahead of the next tile's first -- K-step 0's
        // boundary wait, which leaves only what was issued after ITS last piece in flight, covers them
        const char* uout = save_u ? (const char*)p.aux + ((size_t)m0 * p.ldaux + n0) * 2 : nullptr;
        f32x4_t csum = {0.f, 0.f, 0.f, 0.f};
        // one store unit: rows i*16 + 4g + e, this lane's four columns; `w` = its aux values (GELU' dgrad only)
        auto finish_unit = [&](int i, int e, u32x2_t w) -> u32x2_t {
            f32x4_t x = (f32x4_t){acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]} + bias4;
            if constexpr (GELU_FWD) {
                f32x4_t sg, tt = x * -2.4554669595930156f;
#pragma unroll
                for (int c = 0; c < 4; ++c) tt[c] = __builtin_amdgcn_exp2f(tt[c]);
                tt += 1.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) sg[c] = __builtin_amdgcn_rcpf(tt[c]);
                if (save_u) {
                    const f32x4_t a = x * 1.702f;
                    const f32x4_t d = sg + sg * (a - a * sg);          // s (1 + 1.702 u (1 - s)), as the per-tile epilogues save it
                    pp_store8(uout + (size_t)(i * 16 + e) * ldx2, lane_aux, (u32x2_t){pack2_t<DT>(d[0], d[1]), pack2_t<DT>(d[2], d[3])});
                }
                x *= sg;
            }
            if constexpr (UMUL) {
                x[0] *= cvt16f_t<DT>((bf16_t)(w[0] & 0xffff)); x[1] *= cvt16f_t<DT>((bf16_t)(w[0] >> 16));
                x[2] *= cvt16f_t<DT>((bf16_t)(w[1] & 0xffff)); x[3] *= cvt16f_t<DT>((bf16_t)(w[1] >> 16));
                csum += x;
            }
            return (u32x2_t){pack2_t<DT>(x[0], x[1]), pack2_t<DT>(x[2], x[3])};
        };
        // LDS units first in program order for the GELU' dgrad: their aux values are read back two units at a time, the next
        // pair requested before the current one is multiplied (24 live registers for all twelve at once spilled the parked
        // units -- and a spill of a register that an asm load has not filled yet stores stale bits)
        u32x2_t ua[2] = {{0u, 0u}, {0u, 0u}}, ub[2] = {{0u, 0u}, {0u, 0u}};
        if constexpr (UMUL) {
            asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(ua[0]) : "v"(hbase));
            asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(ua[1]) : "v"(hbase));
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) hr[u] = finish_unit(u >> 2, u & 3, hr[u]);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            u32x2_t(&cur)[2] = (pr & 1) ? ub : ua;
            u32x2_t(&nxt)[2] = (pr & 1) ? ua : ub;
            if constexpr (UMUL) {
                if (pr + 1 < 6) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(nxt[0]) : "v"(hbase), "n"((2 * pr + 2) * 512));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(nxt[1]) : "v"(hbase), "n"((2 * pr + 3) * 512));
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]));      // all but the two just requested
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]));
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = 12 + 2 * pr + k;
                const u32x2_t v = finish_unit(u >> 2, u & 3, cur[k]);
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(hbase), "v"(v), "n"((2 * pr + k) * 512) : "memory");
            }
        }
        if constexpr (UMUL) {
            if (p.colsum) {      // bias-gradient by-product: column sums of what was just produced (f32, before rounding)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tsum = csum[c];
                    tsum += __shfl_xor(tsum, 16, 64);
                    tsum += __shfl_xor(tsum, 32, 64);
                    if (g == 0) atomicAdd(p.colsum + n0 + wn + 4 * fr + c, tsum);
                }
            }
        }
        PP_STAMP(3 + 2 * t);
        pout = tout;
        m0 = m1; n0 = n1;
        acur = anext; bcur = bnext;
    }
    // ---- the last tile's units
#pragma unroll
    for (int u = 0; u < 12; ++u) pp_store8(unit_base(pout, u), lane_out, hr[u]);
#pragma unroll
    for (int u = 12; u < 24; ++u) {
        u32x2_t v;
        asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(hbase), "n"((u - 12) * 512) : "memory");
        pp_store8(unit_base(pout, u), lane_out, v);
    }
#ifdef SIG_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_STAMP(16);
#endif
}

// eligibility and launch -----------------------------------------------------------------------------------------------
#include <atomic>
#ifndef SIG_NT_PERSIST_DEFAULT
#define SIG_NT_PERSIST_DEFAULT 1
#endif
static std::atomic<int> g_persist{-1};      // SIG_NT_PERSIST / sig_tune_nt_persist: 0 = off, 1 = wherever legal
static int persist_setting() {
    int v = g_persist.load();
    if (v < 0) {
        const char* e = getenv("SIG_NT_PERSIST");
        v = e ? atoi(e) : SIG_NT_PERSIST_DEFAULT;
        g_persist = v;
    }
    return v;
}
int sig_tune_nt_persist_impl(int on) {
    const int prev = persist_setting();     // (resolves the environment preset first: restoring `prev` keeps it)
    g_persist = on < 0 ? 0 : on;
    return prev;
}
bool sig_nt192p_eligible(const SigGemmNT& p, int epi, int cus) {
    if (!persist_setting()) return false;
    // The QuickGELU forward has a persistent form too (sig_tune_nt_persist(2) admits it), but its ~2.9 k cycles of exp / rcp per
    // wave and tile sit between two tiles' MFMAs here where the per-tile kernels hide them under their store tail: measured
    // 119 -> 123 us for c_fc at inference (tools/persist_ab.py), so it stays with the 256x256 / 320x256 kernels.
    // Level 3 adds the TRAINING c_fc (QuickGELU' as a second output, stored unit by unit between two tiles): 136-143 us against
    // 134-136 for the 320-row kernel, so off as well (profiles/r04_experiments.md).
    const bool epi_ok = epi == SIG_EPI_BF16 || epi == SIG_EPI_BIAS_BF16 || epi == SIG_EPI_DGELU_BF16 ||
                        (epi == SIG_EPI_BIAS_GELU_BF16 && p.aux == nullptr && persist_setting() >= 2) ||
                        (epi == SIG_EPI_BIAS_GELU_BF16 && p.aux != nullptr && !(p.ldaux & 3) && persist_setting() >= 3);
    if (!epi_ok || (p.colsum && epi != SIG_EPI_DGELU_BF16)) return false;
    const int nk = p.K >> 6;
    if ((p.N & 255) || (p.K & 63) || nk < 12 || (nk & 1)) return false;
    if (epi == SIG_EPI_DGELU_BF16) {
        static int dgelu_on = -1;      // SIG_NT_PERSIST_DGELU=0: the GELU' dgrad stays with the 256x256 kernel (A/B runs)
        if (dgelu_on < 0) { const char* e = getenv("SIG_NT_PERSIST_DGELU"); dgelu_on = e ? atoi(e) : 1; }
        if (!dgelu_on || nk != 12 || !p.aux || (p.ldaux & 3)) return false;                     // its aux plan is twelve K-steps long
    }
    if (p.M % 192) return false;                               // whole row tiles only (24768 = 129 x 192): no row predicates
    const int tiles = (p.M / 192) * (p.N >> 8);
    return tiles >= 2 * cus;                                   // a launch of several tiles per CU
}

template <int EPI, int DT>
static int launch_192p(const SigGemmNT& p, int cus, hipStream_t st) {
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)&gemm_nt192p_kernel<EPI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); });
    const int ntm = p.M / 192, tiles = ntm * (p.N >> 8);
    int grid = cus & ~7;
    if (grid > tiles) grid = tiles & ~7;
    hipLaunchKernelGGL((gemm_nt192p_kernel<EPI, DT>), dim3(grid), dim3(512), 163840, st, p, ntm, tiles);
    return 0;
}

int sig_launch_nt192p(const SigGemmNT& p, int epi, int cus, hipStream_t st) {
    const bool h = p.dt == SIG_DT_F16;
    switch (epi) {
        case SIG_EPI_BF16: return h ? launch_192p<SIG_EPI_BF16, SIG_DT_F16>(p, cus, st) : launch_192p<SIG_EPI_BF16, SIG_DT_BF16>(p, cus, st);
        case SIG_EPI_BIAS_BF16: return h ? launch_192p<SIG_EPI_BIAS_BF16, SIG_DT_F16>(p, cus, st) : launch_192p<SIG_EPI_BIAS_BF16, SIG_DT_BF16>(p, cus, st);
        case SIG_EPI_BIAS_GELU_BF16:
            return h ? launch_192p<SIG_EPI_BIAS_GELU_BF16, SIG_DT_F16>(p, cus, st) : launch_192p<SIG_EPI_BIAS_GELU_BF16, SIG_DT_BF16>(p, cus, st);
        case SIG_EPI_DGELU_BF16:
            return h ? launch_192p<SIG_EPI_DGELU_BF16, SIG_DT_F16>(p, cus, st) : launch_192p<SIG_EPI_DGELU_BF16, SIG_DT_BF16>(p, cus, st);
    }
    sig_set_error("nt192p: epilogue %d has no persistent form", epi);
    return 1;
}
