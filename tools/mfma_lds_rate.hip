// How much matrix throughput does LDS fragment traffic cost under the power limit?  8 waves per CU, each loop iteration issues 64
// bf16 16x16x32 MFMAs (32 independent accumulators) and R ds_read_b128 of fresh fragments from a conflict-free LDS image
// (R = 24 is the GEMM's ratio with 128x64 wave tiles, 16 what 128x128 wave tiles would need, 0 none).  No global traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_lds_rate tools/mfma_lds_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(512) void rate(const uint4* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 lds[4096];   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    v8s f[24];
    for (int i = 0; i < 24; ++i) f[i] = __builtin_bit_cast(v8s, lds[(w * 64 + l + 64 * i) & 4095]);
    v4f acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    int off = w * 64 + l;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) f[i] = __builtin_bit_cast(v8s, lds[(off + 64 * i) & 4095]);   // lane-linear 16-B reads: conflict-free
        off = (off + 512) & 4095;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, f[ks * 12 + i]), __builtin_bit_cast(v8bf, f[ks * 12 + 8 + j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int R>
static void run(const uint4* src, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<R>, dim3(256), dim3(512), 0, 0, src, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<R>, dim3(256), dim3(512), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 64 * 2.0 * 16 * 16 * 32;
    printf("%2d ds_read_b128 per 64 MFMAs: %.2f ms  %.0f TFLOP/s\n", R, ms, flops / ms / 1e9);
}

int main() {
    uint4* src; float* out; hipMalloc(&src, 65536); hipMalloc(&out, 256 * 512 * 4);
    uint32_t h[16384]; uint32_t x = 12345;
    for (int i = 0; i < 16384; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x & 0x807f807fu) | 0x3f003f00u; }   // bf16 values in [0.5, 1), random sign/mantissa
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { run<0>(src, out); run<8>(src, out); run<16>(src, out); run<24>(src, out); }
    return 0;
}
