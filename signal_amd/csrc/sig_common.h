// Shared device/host helpers for the Signal hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/signal_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits; all bf16 tensors cross the C ABI as uint16_t*

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define SIG_WAVE 64

// ---- error plumbing (C ABI returns int, message via sig_last_error) -------------------------
void sig_set_error(const char* fmt, ...);
#define SIG_CHECK_ARG(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            sig_set_error(__VA_ARGS__);           \
            return 1;                             \
        }                                         \
    } while (0)
#define SIG_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            sig_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return 2;                                                               \
        }                                                                           \
    } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast keeps NaN a NaN and lowers to v_cvt_pk_bf16_f32 (MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
// two floats -> one dword of two bf16 (lo in bits 0..15), round-to-nearest-even: ONE v_cvt_pk_bf16_f32.  (Converting the
// halves separately and merging them cost a convert per value plus a shift and an OR per pair.)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2_native_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_native_t __attribute__((ext_vector_type(2)));
    const f32x2_native_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_native_t));
}

// ---- f16 <-> f32, and the operand-type switch ---------------------------------------------------
// Every 16-bit tensor of the path holds ONE operand type per model: bf16 (DT = SIG_DT_BF16) or IEEE f16 (SIG_DT_F16).
// Same MFMA rate on gfx950 (v_mfma_f32_16x16x32_{bf16,f16}); f16 keeps 3 more mantissa bits (operand rounding 2^-11
// instead of 2^-8), which is what the reference's CUDA autocast computes in (engine/processor.py:165).  The
// matrix-core kernels take the type as a template parameter; the HBM-bound row kernels take it as a (uniform) argument.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;   // (SIG_DT_BF16 / SIG_DT_F16 come from include/signal_hip.h)
__device__ __forceinline__ float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2h(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {     // ONE v_cvt_pk_f16_f32 (round to nearest even)
    typedef _Float16 f16x2_native_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_native_t __attribute__((ext_vector_type(2)));
    const f32x2_native_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_native_t));
}
__device__ __forceinline__ float cvt16f(bf16_t v, int dt) { return dt == SIG_DT_F16 ? h2f(v) : bf2f(v); }
__device__ __forceinline__ bf16_t f2cvt16(float f, int dt) { return dt == SIG_DT_F16 ? f2h(f) : f2bf(f); }
__device__ __forceinline__ uint32_t pack2_16(float lo, float hi, int dt) { return dt == SIG_DT_F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
template <int DT> __device__ __forceinline__ float cvt16f_t(bf16_t v) { return DT == SIG_DT_F16 ? h2f(v) : bf2f(v); }
template <int DT> __device__ __forceinline__ uint32_t pack2_t(float lo, float hi) { return DT == SIG_DT_F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
template <int DT> __device__ __forceinline__ bf16_t f2cvt16_t(float f) { return DT == SIG_DT_F16 ? f2h(f) : f2bf(f); }
// MFMA on raw 16-bit fragments (the bit patterns are typed by DT)
template <int DT> __device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    if constexpr (DT == SIG_DT_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    if constexpr (DT == SIG_DT_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#define SIG_CHECK_DT(dt, who) SIG_CHECK_ARG((dt) == SIG_DT_BF16 || (dt) == SIG_DT_F16, "%s: operand dtype %d is neither bf16 (0) nor f16 (1)", who, (int)(dt))

// ---- wave reductions (64 lanes) ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- LDS-DMA: 16 B per lane, LDS destination = wave-uniform base + lane*16 ---------------------
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// Same transfer issued through inline asm.  The compiler's waitcnt pass models an LDS-DMA as an LGKM event, so with a
// builtin DMA in flight every fragment wait becomes `s_waitcnt lgkmcnt(0)`; the hardware counts it in vmcnt only
// (tools/micro/lgkm_dma.hip: lgkmcnt(0) returns 114 cycles after a cold DMA + ds_read, vmcnt(0) 870 cycles later).
// Issued this way the DMA is invisible to that pass: the CALLER must `s_waitcnt vmcnt(0)` before anyone reads the
// destination and before any compiler-tracked global load is consumed (its counted vmcnt would be off).
__device__ __forceinline__ void glds16_untracked(const void* gsrc, void* lds_wave_base) {
    // lds_wave_base must be wave-uniform IN AN SGPR (derive it from __builtin_amdgcn_readfirstlane(tid >> 6)): a
    // readfirstlane here would cost a VALU->SALU hop per piece inside the K loop (+7 % on gemm_tn256_kernel)
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "memory");   // m0 is reserved: not clobberable; kernels using this helper must not mix it with glds16()
}
// saddr form: source = 64-bit wave-uniform base (SGPR pair) + 32-bit per-lane byte offset.  The per-lane offsets of a
// tile's pieces never change, the base moves by a scalar add per K-step: no vector instruction per piece.
__device__ __forceinline__ void glds16_untracked_s(const void* uniform_base, unsigned lane_byte_off, void* lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_byte_off), "s"(uniform_base), "s"(l) : "memory");
}
// transposed LDS read: per 16-lane group a 4x16 block of 16-bit elements, delivered column-major
__device__ __forceinline__ bf16x4_t lds_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)p);
}

// XCD-aware bijective block remap: blocks b and b+8 share an XCD (speed only, never correctness)
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

static inline int sig_ceil_div(int a, int b) { return (a + b - 1) / b; }
