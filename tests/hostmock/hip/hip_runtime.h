// TEST INFRASTRUCTURE ONLY (tests/test_host_sanitizers.py): a host stand-in for the few HIP runtime calls the engine's HOST code uses, so that csrc/engine.hip,
// adapters.hip and train.hip compile as plain C++ for x86 and run under AddressSanitizer / UBSan in the build container (GPU sanitizers are not available on the
// pool).  "Device" memory is calloc'ed host memory, copies are memcpy, kernel launches and events do nothing.  Nothing here computes: the driver exercises the
// bookkeeping -- allocation, workspace growth, adapter tables, option parsing, error paths -- not the numerics.  Never part of the product build.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
using std::max;
using std::min;

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct mock_stream* hipStream_t;
typedef struct mock_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; }; struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; }; struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static const dim3 threadIdx, blockIdx, blockDim, gridDim;

// allocation limit of the mock device (bytes; 0 = none): lets the driver walk the engine's out-of-memory paths
extern "C" size_t mock_hip_mem_limit;
extern "C" size_t mock_hip_mem_in_use;
extern "C" int mock_hip_device_count;
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset2D(void* d, size_t pitch, int v, size_t w, size_t h) { for (size_t r = 0; r < h; ++r) memset((char*)d + r * pitch, v, w); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) { for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory (mock)" : "hipError (mock)"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = mock_hip_device_count; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
// a launch evaluates its arguments (so that what the host passes is read under the sanitizers) and runs nothing
template <typename... T> static inline void mock_launch_args(const T&...) {}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) do { (void)(grid); (void)(block); (void)(stream); mock_launch_args(__VA_ARGS__); } while (0)

// device-side intrinsics that appear in code shared with the host compile (common.hpp and the few plain kernels of engine.hip / adapters.hip): never executed
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
template <typename T> static inline T __shfl_xor(T v, int, int = 64) { return v; }
static inline void __syncthreads() {}
struct mock_u2 { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
#define __builtin_amdgcn_permlane32_swap(a, b, c, d) (mock_u2{{(uint32_t)(a), (uint32_t)(b)}})
#define __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, hi) ((int)(old))
#define __builtin_amdgcn_s_setreg(a, b) ((void)0)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) (c)
