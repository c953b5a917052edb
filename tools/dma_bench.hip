// Load-path microbenchmark for the GEMM's LDS-DMA pattern (no MFMA): every workgroup streams the A and W K-panels of
// its 256x256 tile exactly as gemm_kernel does (same tile order, same 1-KiB blocks), with a configurable number of
// 64-KB K-steps in flight (DEPTH), a barrier per step or not, and the DMA burst split over SPLIT issue points.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_bench tools/dma_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define BM 256
#define BN 256
#define BK 64
#define GROUP_M 8
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int DEPTH, int BARRIER, int SPLIT, int STAGES>
__global__ __launch_bounds__(512) void dma_kernel(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[STAGES * 65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN, nwg = ntm * ntn;
    int pid;
    { const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7; pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int width = GROUP_M * ntn;
    const int first_m = (pid / width) * GROUP_M;
    const int gsz = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (pid % width) % gsz, tn = (pid % width) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;
    const int sq = lane >> 4, sr = (lane >> 1) & 7, sc = 2 * sq + (lane & 1);
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = wave + 8 * i;
        offA[i] = (uint32_t)(((int64_t)min(row0 + 8 * b + sr, M - 1) * K + 8 * sc) * 2);
        offB[i] = (uint32_t)(((int64_t)min(col0 + 8 * b + sr, N - 1) * K + 8 * sc) * 2);
    }
    const char* baseA = (const char*)A; const char* baseW = (const char*)W;
    const int nk = K / BK;
    auto issue = [&](int kt, int part) __attribute__((always_inline)) {   // part in [0, SPLIT): 8/SPLIT LDS-DMA per wave
        char* base = smem + (kt % STAGES) * 65536;
        const char* ga = baseA + (int64_t)kt * 128; const char* gw = baseW + (int64_t)kt * 128;
#pragma unroll
        for (int j = 0; j < 8 / SPLIT; ++j) {
            const int idx = part * (8 / SPLIT) + j;   // 0..7: even -> A block, odd -> W block
            const int i = idx >> 1;
            const int b = wave + 8 * i;
            if (idx & 1) __builtin_amdgcn_global_load_lds((gptr_t)(gw + offB[i]), (lptr_t)(base + 32768 + b * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)(ga + offA[i]), (lptr_t)(base + b * 1024), 16, 0, 0);
        }
    };
    // prologue: DEPTH steps in flight
    for (int d = 0; d < DEPTH && d < nk; ++d)
        for (int part = 0; part < SPLIT; ++part) issue(d, part);
    for (int kt = 0; kt < nk; ++kt) {
        // wait until step kt has landed: allow (DEPTH-1) newer steps outstanding
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        // "consume" (nothing), then refill DEPTH steps ahead, optionally in SPLIT pieces separated by short sleeps
#pragma unroll
        for (int part = 0; part < SPLIT; ++part) {
            if (kt + DEPTH < nk) issue(kt + DEPTH, part);
            else {  // keep the vmcnt arithmetic uniform at the tail: issue dummy re-loads of the last step
                issue(nk - 1, part);
            }
            if (SPLIT > 1) __builtin_amdgcn_s_sleep(4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = (float)smem[0];
}

template <int DEPTH, int BARRIER>
__global__ __launch_bounds__(512) void dma32_kernel(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN, nwg = ntm * ntn;
    int pid;
    { const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7; pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int width = GROUP_M * ntn;
    const int first_m = (pid / width) * GROUP_M;
    const int gsz = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (pid % width) % gsz, tn = (pid % width) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;
    const int c = lane >> 4, r = (lane >> 1) & 7, h = lane & 1;   // LDS slot lane -> row 8h + r, 16-B chunk c
    uint32_t offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = wave + 8 * i;   // 16-row block index (16 blocks per operand)
        offA[i] = (uint32_t)(((int64_t)min(row0 + 16 * b + 8 * h + r, M - 1) * K + 8 * c) * 2);
        offB[i] = (uint32_t)(((int64_t)min(col0 + 16 * b + 8 * h + r, N - 1) * K + 8 * c) * 2);
    }
    const char* baseA = (const char*)A; const char* baseW = (const char*)W;
    const int nk = K / 32;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* base = smem + (kt % 5) * 32768;
        const char* ga = baseA + (int64_t)kt * 64; const char* gw = baseW + (int64_t)kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((gptr_t)(ga + offA[i]), (lptr_t)(base + b * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gw + offB[i]), (lptr_t)(base + 16384 + b * 1024), 16, 0, 0);
        }
    };
    for (int d = 0; d < DEPTH && d < nk; ++d) issue(d);
    for (int kt = 0; kt < nk; ++kt) {
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        issue(kt + DEPTH < nk ? kt + DEPTH : nk - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = (float)smem[0];
}

// half-step granularity: one operand tile (32 KB = 4 LDS-DMA per wave) per issue, D operand tiles in flight
template <int D>
__global__ __launch_bounds__(512) void dmah_kernel(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN, nwg = ntm * ntn;
    int pid;
    { const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7; pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int width = GROUP_M * ntn;
    const int first_m = (pid / width) * GROUP_M;
    const int gsz = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (pid % width) % gsz, tn = (pid % width) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;
    const int sq = lane >> 4, sr = (lane >> 1) & 7, sc = 2 * sq + (lane & 1);
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = wave + 8 * i;
        offA[i] = (uint32_t)(((int64_t)min(row0 + 8 * b + sr, M - 1) * K + 8 * sc) * 2);
        offB[i] = (uint32_t)(((int64_t)min(col0 + 8 * b + sr, N - 1) * K + 8 * sc) * 2);
    }
    const char* baseA = (const char*)A; const char* baseW = (const char*)W;
    const int nh = 2 * (K / BK);
    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* base = smem + (hs % 5) * 32768;
        const int kt = hs >> 1;
        const char* g = ((hs & 1) ? baseW : baseA) + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + ((hs & 1) ? offB[i] : offA[i])), (lptr_t)(base + b * 1024), 16, 0, 0);
        }
    };
    for (int d = 0; d < D && d < nh; ++d) issue(d);
    for (int hs = 0; hs < nh; ++hs) {
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(hs + D < nh ? hs + D : nh - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = (float)smem[0];
}
template <int D>
static void runh(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dmah_kernel<D>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dmah_kernel<D>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)ntm * ntn * (K / 64) * 65536.0;
    const double flops = 2.0 * M * (double)N * K;
    printf("%-34s M=%d N=%d K=%d: %.3f ms  %.1f GB/s per CU  (%.2f TB/s chip)  == %.0f TFLOP/s if compute were free\n", name, M, N, K, ms,
           bytes / ms / 1e6 / 256, bytes / ms / 1e9, flops / ms / 1e9);
}

template <int DEPTH, int BARRIER>
static void run32(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dma32_kernel<DEPTH, BARRIER>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dma32_kernel<DEPTH, BARRIER>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)ntm * ntn * (K / 64) * 65536.0;
    const double flops = 2.0 * M * (double)N * K;
    printf("%-34s M=%d N=%d K=%d: %.3f ms  %.1f GB/s per CU  (%.2f TB/s chip)  == %.0f TFLOP/s if compute were free\n", name, M, N, K, ms,
           bytes / ms / 1e6 / 256, bytes / ms / 1e9, flops / ms / 1e9);
}

template <int DEPTH, int BARRIER, int SPLIT, int STAGES>
static void run(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dma_kernel<DEPTH, BARRIER, SPLIT, STAGES>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dma_kernel<DEPTH, BARRIER, SPLIT, STAGES>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)ntm * ntn * (K / 64) * 65536.0;
    const double flops = 2.0 * M * (double)N * K;
    printf("%-34s M=%d N=%d K=%d: %.3f ms  %.1f GB/s per CU  (%.2f TB/s chip)  == %.0f TFLOP/s if compute were free\n", name, M, N, K, ms,
           bytes / ms / 1e6 / 256, bytes / ms / 1e9, flops / ms / 1e9);
}

int main() {
    const int M = 32768, N = 37888, K = 3584;
    uint16_t *A, *W; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&sink, 1 << 20);
    hipMemset(A, 0x11, (size_t)M * K * 2); hipMemset(W, 0x22, (size_t)N * K * 2);
    run<1, 1, 1, 2>("depth1 barrier", A, W, M, N, K, sink);
    run<2, 1, 1, 2>("depth2 barrier (LDS hazard ignored)", A, W, M, N, K, sink);
    run<2, 0, 1, 2>("depth2 nobarrier", A, W, M, N, K, sink);
    run<3, 1, 1, 2>("depth3 barrier", A, W, M, N, K, sink);
    run<3, 0, 1, 2>("depth3 nobarrier", A, W, M, N, K, sink);
    run<4, 0, 1, 2>("depth4 nobarrier", A, W, M, N, K, sink);
    run<2, 1, 2, 2>("depth2 barrier split2", A, W, M, N, K, sink);
    run<2, 1, 4, 2>("depth2 barrier split4", A, W, M, N, K, sink);
    run<3, 1, 4, 2>("depth3 barrier split4", A, W, M, N, K, sink);
    run<1, 1, 4, 2>("depth1 barrier split4", A, W, M, N, K, sink);
    runh<1>("half-step D1 (32KB in flight)", A, W, M, N, K, sink);
    runh<2>("half-step D2 (64KB in flight)", A, W, M, N, K, sink);
    runh<3>("half-step D3 (96KB in flight)", A, W, M, N, K, sink);
    runh<4>("half-step D4 (128KB in flight)", A, W, M, N, K, sink);
    run32<1, 1>("BK32 depth1 (32KB) barrier", A, W, M, N, K, sink);
    run32<2, 1>("BK32 depth2 (64KB) barrier", A, W, M, N, K, sink);
    run32<3, 1>("BK32 depth3 (96KB) barrier", A, W, M, N, K, sink);
    run32<4, 1>("BK32 depth4 (128KB) barrier", A, W, M, N, K, sink);
    run32<4, 0>("BK32 depth4 (128KB) nobarrier", A, W, M, N, K, sink);
    // second shape: long K (down_proj)
    {
        const int M2 = 32768, N2 = 3584, K2 = 18944;
        uint16_t *A2, *W2; hipMalloc(&A2, (size_t)M2 * K2 * 2); hipMalloc(&W2, (size_t)N2 * K2 * 2);
        hipMemset(A2, 0x11, (size_t)M2 * K2 * 2); hipMemset(W2, 0x22, (size_t)N2 * K2 * 2);
        run<1, 1, 1, 2>("depth1 barrier", A2, W2, M2, N2, K2, sink);
        run<2, 1, 1, 2>("depth2 barrier", A2, W2, M2, N2, K2, sink);
        run<3, 0, 1, 2>("depth3 nobarrier", A2, W2, M2, N2, K2, sink);
    }
    return 0;
}
