// MFMA-only rate (no memory traffic in the loop): bf16 16x16x32 vs MX-fp8 16x16x128 (unit scales), random operands.
// 256 workgroups x 8 waves (two per SIMD), 32 independent accumulators per wave.  Power-limited clocks show up here.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int F8>
__global__ __launch_bounds__(512) void rate(const int* src, float* out, int iters) {
    const int l = threadIdx.x & 63;
    v8i a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(l * 8 + i) * 8 + j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[4096 + (l * 4 + i) * 8 + j];
    v4f acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F8) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else {
                    v8bf x = __builtin_bit_cast(v8bf, (v8s){(short)a[i][0], (short)(a[i][0] >> 16), (short)a[i][1], (short)(a[i][1] >> 16), (short)a[i][2], (short)(a[i][2] >> 16), (short)a[i][3], (short)(a[i][3] >> 16)});
                    v8bf y = __builtin_bit_cast(v8bf, (v8s){(short)b[j][0], (short)(b[j][0] >> 16), (short)b[j][1], (short)(b[j][1] >> 16), (short)b[j][2], (short)(b[j][2] >> 16), (short)b[j][3], (short)(b[j][3] >> 16)});
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i][j], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    int* src; float* out; hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
    int h[8192]; uint32_t x = 12345;
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 8192; ++i) {   // random bytes with exponent fields kept moderate (valid, finite fp8 / bf16)
            x = x * 1664525u + 1013904223u; uint32_t v = x;
            v = (v & 0x87878787u) | 0x38383838u;          // e4m3: sign + 3-bit mantissa random, exponent 7 (|v| in [1, 2))
            h[i] = zero ? 0 : (int)v;
        }
        hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
        for (int f8 = 0; f8 < 2; ++f8) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (f8) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, src, out, 100); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, src, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (f8) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, src, out, iters); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * 8 * iters * 32 * 2.0 * 16 * 16 * (f8 ? 128 : 32);
            printf("%s operands, %s: %.2f ms  %.0f TFLOP/s\n", zero ? "zero" : "random", f8 ? "mx-fp8 16x16x128" : "bf16 16x16x32", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
