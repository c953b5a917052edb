// Prototype of a 4-wave GEMM main loop (one wave per SIMD, 128x128 per-wave tiles, accumulators in AGPRs): does cutting the LDS
// fragment traffic by a third (16 instead of 24 ds_read_b128 per 64 MFMAs) and halving the barriers beat the 8-wave ping-pong
// loop of blim_amd/csrc/gemm.hip?  Main loop only: accumulators are reduced to one float per lane, nothing else is stored.
// Same 256x256 tile, 128-byte K-steps, [k-quarter][row][32 B] LDS image and LDS-DMA pattern as the product kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm4_proto tools/gemm4_proto.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define TILE 32768

__global__ __launch_bounds__(256) void gemm4(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // double buffer: {A, W} x 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntm = M / 256, ntn = N / 256, nwg = ntm * ntn;
    const int nk = K / 64;
    const int sq = lane >> 4, sr = (lane >> 1) & 7, sc = 2 * sq + (lane & 1);
    const int fr = lane & 15, fc = lane >> 4;
    const int frag_off = (fr >> 3) * 1024 + (fr & 7) * 32 + (fc >> 1) * 256 + (fc & 1) * 16;
    const int a_off = (16 * wm) * 1024 + frag_off;      // + mi*2048 + ks*512
    const int b_off = TILE + (16 * wn) * 1024 + frag_off;
    float total = 0.f;
    for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
        int pid;
        { const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7; pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3); }
        const int width = 8 * ntn, first_m = (pid / width) * 8, gsz = min(ntm - first_m, 8);
        const int tm = first_m + (pid % width) % gsz, tn = (pid % width) / gsz;
        const int row0 = tm * 256, col0 = tn * 256;
        uint32_t offA[8], offW[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = wave + 4 * i;
            offA[i] = (uint32_t)(((int64_t)(row0 + 8 * b + sr) * K + 8 * sc) * 2);
            offW[i] = (uint32_t)(((int64_t)(col0 + 8 * b + sr) * K + 8 * sc) * 2);
        }
        const char* baseA = (const char*)A; const char* baseW = (const char*)W;
        auto dma = [&](int buf, int kt, int i, bool isW) __attribute__((always_inline)) {   // one 1-KB block
            uint32_t o = isW ? offW[i] : offA[i];
            asm volatile("" : "+v"(o));
            const char* g = (isW ? baseW : baseA) + (int64_t)kt * 128 + o;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + buf * 2 * TILE + (isW ? TILE : 0) + (wave + 4 * i) * 1024), 16, 0, 0);
        };
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[2][8], fb[2][8];
        auto rd = [&](int buf, int ks, int i, bool isB) __attribute__((always_inline)) {
            if (isB) fb[ks][i] = *(const bf16x8*)(smem + buf * 2 * TILE + b_off + i * 2048 + ks * 512);
            else fa[ks][i] = *(const bf16x8*)(smem + buf * 2 * TILE + a_off + i * 2048 + ks * 512);
        };
        // prologue: tiles 0 and 1 in flight, fragments ks=0 of tile 0 loaded
#pragma unroll
        for (int i = 0; i < 8; ++i) { dma(0, 0, i, false); dma(0, 0, i, true); }
        if (nk > 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { dma(1, 1, i, false); dma(1, 1, i, true); }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) { rd(0, 0, i, false); rd(0, 0, i, true); }
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            // ---- sub-step 0: compute ks=0 of tile kt; meanwhile read ks=1 of tile kt (and, from kt >= 1, stage half of tile kt+1:
            //      the other buffer was freed by the barrier below in the previous iteration)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                rd(buf, 1, g, false); rd(buf, 1, g, true);
                if (kt >= 1 && kt + 1 < nk) dma(buf ^ 1, kt + 1, g, true);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, fa[0][g]), __builtin_bit_cast(v8bf, fb[0][j]), acc[g][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // tile kt fully read by this wave; tile kt+1 must have landed (own DMA) before anyone reads it
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- sub-step 1: compute ks=1 of tile kt; read ks=0 of tile kt+1; stage A half of tile kt+2 into the freed buffer
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (kt + 1 < nk) { rd(buf ^ 1, 0, g, false); rd(buf ^ 1, 0, g, true); }
                if (kt + 2 < nk) dma(buf, kt + 2, g, false);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, fa[1][g]), __builtin_bit_cast(v8bf, fb[1][j]), acc[g][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        total += s;
    }
    sink[(size_t)blockIdx.x * 256 + tid] = total;
}

int main() {
    const int shapes[3][3] = {{32768, 37888, 3584}, {32768, 3584, 18944}, {8192, 8192, 8192}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        uint16_t *A, *W; float* sink;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&sink, 256 * 256 * 4);
        // pseudo-random bf16 in (-2, 2): fill on host once for a 64-MB pattern, replicate
        const size_t pat = 32u << 20; uint16_t* h = (uint16_t*)malloc(pat * 2); uint32_t x = 777;
        for (size_t i = 0; i < pat; ++i) { x = x * 1664525u + 1013904223u; h[i] = (uint16_t)(((x >> 16) & 0x807f) | 0x3f00 | ((x >> 8) & 0x0080)); }
        for (size_t o = 0; o < (size_t)M * K; o += pat) hipMemcpy(A + o, h, std::min(pat, (size_t)M * K - o) * 2, hipMemcpyHostToDevice);
        for (size_t o = 0; o < (size_t)N * K; o += pat) hipMemcpy(W + o, h + 12345, std::min(pat - 12345, (size_t)N * K - o) * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(gemm4, dim3(256), dim3(256), 0, 0, A, W, M, N, K, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm4, dim3(256), dim3(256), 0, 0, A, W, M, N, K, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("gemm4 main loop %dx%dx%d: %.3f ms  %.1f TFLOP/s (%s)\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9, hipGetErrorString(hipGetLastError()));
        hipFree(A); hipFree(W); hipFree(sink); free(h);
    }
    return 0;
}
