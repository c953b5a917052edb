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
