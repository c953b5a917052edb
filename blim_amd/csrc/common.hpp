// Shared device/host helpers for the BLiM scoring engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

#define WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// f32 -> bf16, round to nearest even.  A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (NaN-safe).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(uint32_t, r);
}

// ---- 16-bit compute dtype (runtime choice per engine, compile-time per kernel): 0 = bf16, 1 = fp16.
// fp16 is what the reference itself runs on GPU (main.py:97 .half(), training_utils.py:142 autocast(float16)); both
// MFMA forms have the same rate on gfx950.
#define DT_BF16 0
#define DT_F16 1
// fp8 mode (BASELINE config 5): the five big decoder GEMMs and lm_head take OCP e4m3 operands (per-row f32 scales on both
// sides, MX block scales fixed at 1) on the block-scaled MFMA; everything 16-bit around them is fp16.
#define DT_F8 2
#define FP8_MAX 448.0f
#define F6_TILE_BYTES 25600   // "lo6" operand tiles (gemm.hpp: A6 / W6): 24 KiB of packed e2m3 + 1 KiB of E8M0 scale bytes per (256-row tile, 128-value K-step)
template <int DT> struct out16 { static constexpr int value = DT == DT_F8 ? DT_F16 : DT; };

// two floats -> two e4m3 bytes in the low (hi = false) or high half of `old` (round to nearest even, v_cvt_pk_fp8_f32)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE.FP16_OVFL (hwreg MODE, bit 23) for the rest of this wave: an fp16 VALU result that overflows -- here: the f32 -> f16 conversions of the
// epilogues -- becomes +-65504 instead of +-inf; true infinities and NaNs pass through unchanged (tools/f16_ovfl_probe.hip, measured on gfx950).
// Costs one scalar instruction per wave.  The fp16 scoring path sets it wherever it stores 16-bit activations: a q / k / v, a SwiGLU output
// or a norm output beyond fp16's range saturates (a defined, finite value) instead of turning the row's scores into NaN; the f32
// residual stream keeps its range.  Gradient stores of the trainer do NOT set it: the loss scaler's overflow detection needs the inf.
// CAUTION (measured, tools/nan_probe.py): while the bit is set the fp16 MFMA reads a NaN operand as 0 and an infinite one as +-65504 -- NaN
// would no longer propagate through a GEMM.  So the bit is set only AROUND the conversions (after a tile's last MFMA) and cleared before the
// next MFMA: NaN / inf operands still poison the accumulator, and a NaN accumulator is stored as NaN.
__device__ __forceinline__ void f16_saturate_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
__device__ __forceinline__ void f16_saturate_off() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 0); }

template <int DT> __device__ __forceinline__ float from16(uint16_t b) {
    if constexpr (DT == DT_BF16) return __uint_as_float(((uint32_t)b) << 16);
    else return (float)__builtin_bit_cast(_Float16, b);
}
template <int DT> __device__ __forceinline__ uint16_t to16(float f) {
    if constexpr (DT == DT_BF16) return f32_to_bf16(f);
    else { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
}
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (DT == DT_BF16) return pack_bf16x2(lo, hi);
    else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {lo, hi};
        h2 r = __builtin_convertvector(v, h2);
        return __builtin_bit_cast(uint32_t, r);
    }
}
template <int DT> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// max / sum of a value over the two 32-lane halves of a wave (lanes l and l ^ 32), in every lane: one v_permlane32_swap (VALU) instead of the
// ds_bpermute trip through the LDS crossbar that __shfl_xor(v, 32) compiles to
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Host-side error plumbing: every C-ABI entry returns 0 or a negative code and records a message.
#define BLIM_OK 0
#define BLIM_ERR_ARG (-1)
#define BLIM_ERR_HIP (-2)
#define BLIM_ERR_STATE (-3)
#define BLIM_ERR_NOMEM (-4)

void blim_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            blim_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return BLIM_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define ARG_CHECK(cond)                                                                      \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            blim_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);           \
            return BLIM_ERR_ARG;                                                             \
        }                                                                                    \
    } while (0)
