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
__global__ __launch_bounds__(512) void dmah_kernel(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink, int same = 0, int nat = 0) {
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
    const int row0 = same ? 0 : tm * BM, col0 = same ? 0 : tn * BN;   // same: every workgroup streams one L2-resident panel pair
    // nat = 1: 8 consecutive lanes read one whole 128-B line (row = lane / 8, XOR-swizzled chunk order inside the line);
    // nat = 0: the product kernel's [k-quarter][row][32 B] image, where a lane PAIR reads 32 B of a row
    const int sq = lane >> 4, sr = nat ? (lane >> 3) : ((lane >> 1) & 7), sc = nat ? ((lane & 7) ^ ((lane >> 4) & 7)) : (2 * sq + (lane & 1));
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


// register path: global_load_dwordx4 -> VGPR -> ds_write_b128 (the classic pipeline), DEPTH 64-KB steps in flight
template <int DEPTH>
__global__ __launch_bounds__(512) void reg_kernel(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
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
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    u4 regs[DEPTH][8];
    auto issue = [&](int kt, int d) __attribute__((always_inline)) {
        const char* ga = baseA + (int64_t)kt * 128; const char* gw = baseW + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            regs[d][2 * i] = *(const u4*)(ga + offA[i]);
            regs[d][2 * i + 1] = *(const u4*)(gw + offB[i]);
        }
    };
    auto drain = [&](int kt, int d) __attribute__((always_inline)) {
        char* base = smem + (kt & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = wave + 8 * i;
            *(u4*)(base + b * 1024 + lane * 16) = regs[d][2 * i];
            *(u4*)(base + 32768 + b * 1024 + lane * 16) = regs[d][2 * i + 1];
        }
    };
    static_assert(DEPTH == 1 || DEPTH == 2, "");
    issue(0, 0);
    if (DEPTH == 2) issue(1 < nk ? 1 : 0, 1);
    for (int kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            drain(kt + d, d);                          // compiler inserts the vmcnt wait for regs[d]
            const int nx = kt + d + DEPTH;
            issue(nx < nk ? nx : nk - 1, d);
            __builtin_amdgcn_s_barrier();
        }
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += (float)regs[d][i].x;
    if (sink && (tid == 0 || acc == 12345.f)) sink[blockIdx.x] = (float)smem[0] + acc;
}
template <int DEPTH>
static void runreg(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink) {
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((reg_kernel<DEPTH>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((reg_kernel<DEPTH>), dim3(ntm * ntn), dim3(512), 0, 0, A, W, M, N, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)ntm * ntn * (K / 64) * 65536.0;
    printf("%-34s M=%d N=%d K=%d: %.3f ms  %.1f GB/s per CU  (%.2f TB/s chip)\n", name, M, N, K, ms, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}
// the half-step LDS-DMA kernel on only `nwg` workgroups (one per CU on nwg CUs): per-CU limit or chip-wide limit?
template <int D>
static void runh_few(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink, int nwg, int same = 0, int nat = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dmah_kernel<D>), dim3(nwg), dim3(512), 0, 0, A, W, M, N, K, sink, same, nat);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((dmah_kernel<D>), dim3(nwg), dim3(512), 0, 0, A, W, M, N, K, sink, same, nat);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double bytes = (double)nwg * (K / 64) * 65536.0;
    printf("%-34s nwg=%d: %.4f ms  %.1f GB/s per active CU (%.2f TB/s chip)\n", name, nwg, ms, bytes / ms / 1e6 / nwg, bytes / ms / 1e9);
}


// contiguous pattern: every LDS-DMA instruction reads 1 KB of consecutive bytes (a tile-blocked operand layout), the
// workgroup streams `nk` 32-KB blocks; SEG = contiguous bytes per row segment (1024 = fully linear, 256 / 128 = strided rows)
template <int D, int SEG>
__global__ __launch_bounds__(512) void lin_kernel(const char* A, size_t bytes_per_wg, int nk, float* sink, size_t wrap, int share = 1) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // share = G: G workgroups of one XCD stream the same bytes at the same time (the GEMM's panel sharing)
    const int grp = (blockIdx.x & 7) * 64 + (int)(blockIdx.x >> 3) / share;
    const char* base = A + ((size_t)grp * bytes_per_wg) % wrap;
    // lane -> byte offset inside the wave's 1-KB share: SEG-byte segments at a 7168-B row pitch
    const int per = SEG / 16;                     // lanes per segment
    const uint32_t loff = (uint32_t)((lane / per) * (SEG == 1024 ? 1024 : 7168) + (lane % per) * 16);
    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* l = smem + (hs % 5) * 32768;
        const char* g = base + (SEG == 1024 ? (size_t)(hs % 56) * 32768 : (size_t)(hs % (7168 / SEG)) * SEG);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = wave + 8 * i;
            const size_t goff = SEG == 1024 ? (size_t)b * 1024 : (size_t)b * (1024 / SEG) * 7168;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + goff + loff), (lptr_t)(l + b * 1024), 16, 0, 0);
        }
    };
    for (int d = 0; d < D && d < nk; ++d) issue(d);
    for (int hs = 0; hs < nk; ++hs) {
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(hs + D < nk ? hs + D : nk - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = (float)smem[0];
}
template <int D, int SEG>
static void runlin(const char* name, const char* A, int nwg, int nk, float* sink, size_t wrap, int share = 1) {
    const size_t per_wg = (size_t)4 << 20;        // 4 MB apart
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lin_kernel<D, SEG>), dim3(nwg), dim3(512), 0, 0, A, per_wg, nk, sink, wrap, share);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((lin_kernel<D, SEG>), dim3(nwg), dim3(512), 0, 0, A, per_wg, nk, sink, wrap, share);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double bytes = (double)nwg * nk * 32768.0;
    printf("%-44s SEG=%4d nwg=%d: %.4f ms  %.1f GB/s per CU (%.2f TB/s chip)\n", name, SEG, nwg, ms, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}


// the GEMM's sharing structure on synchronised workgroups: per XCD 32 workgroups = 8 (tm) x 4 (tn); a workgroup alternates
// between its A panel (shared with the 3 others of the same tm) and its W panel (shared with the 7 others of the same tn)
template <int D>
__global__ __launch_bounds__(512) void lin2_kernel(const char* A, int nk, float* sink, size_t pitch, int gm) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int tm = w % gm, tn = w / gm;
    const char* pa = A + (size_t)(xcd * 40 + tm) * pitch;
    const char* pw = A + (size_t)(xcd * 40 + 32 + tn) * pitch;
    const uint32_t loff = (uint32_t)((lane / 8) * 7168 + (lane % 8) * 16);
    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* l = smem + (hs % 5) * 32768;
        const char* g = ((hs & 1) ? pw : pa) + (size_t)((hs >> 1) % 56) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + (size_t)b * 8 * 7168 + loff), (lptr_t)(l + b * 1024), 16, 0, 0);
        }
    };
    for (int d = 0; d < D && d < nk; ++d) issue(d);
    for (int hs = 0; hs < nk; ++hs) {
        if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(hs + D < nk ? hs + D : nk - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = (float)smem[0];
}
template <int D>
static void runlin2(const char* name, const char* A, int nk, float* sink, size_t pitch, int gm) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lin2_kernel<D>), dim3(256), dim3(512), 0, 0, A, nk, sink, pitch, gm);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((lin2_kernel<D>), dim3(256), dim3(512), 0, 0, A, nk, sink, pitch, gm);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double bytes = 256.0 * nk * 32768.0;
    printf("%-44s pitch=%zu gm=%d: %.4f ms  %.1f GB/s per CU (%.2f TB/s chip)\n", name, pitch, gm, ms, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
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
    if (getenv("DMA_ALL"))
    {
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
    }
    runreg<1>("register path depth1", A, W, M, N, K, sink);
    runreg<2>("register path depth2", A, W, M, N, K, sink);
    for (int nat : {0, 1, 0, 1}) {
        runh_few<4>(nat ? "GEMM map, whole-line lanes, 18944 WGs" : "GEMM map, 32-B lane pairs, 18944 WGs", A, W, M, N, K, sink, 18944, 0, nat);
        runh_few<4>(nat ? "GEMM map, whole-line lanes, 256 WGs" : "GEMM map, 32-B lane pairs, 256 WGs", A, W, M, N, K, sink, 256, 0, nat);
        runh_few<2>(nat ? "GEMM map, whole-line lanes, 256 WGs D2" : "GEMM map, 32-B lane pairs, 256 WGs D2", A, W, M, N, K, sink, 256, 0, nat);
    }
    for (int nwg : {8, 64, 256}) runh_few<4>("half-step D4, few CUs", A, W, M, N, K, sink, nwg);
    for (int nwg : {8, 64, 256, 2048}) runh_few<4>("half-step D4, one L2-resident panel pair", A, W, M, N, K, sink, nwg, 1);
    for (int nwg : {8, 256, 2048}) runh_few<2>("half-step D2, one L2-resident panel pair", A, W, M, N, K, sink, nwg, 1);
    for (int nwg : {8, 256, 2048}) runh_few<4>("half-step D4, L2-resident, K=512", A, W, M, N, 512, sink, nwg, 1);
    {
        const size_t all = (size_t)M * K * 2;      // 235 MB: wrap inside A
        runlin<4, 1024>("linear 1-KB instr, distinct (HBM/MALL)", (const char*)A, 256, 224, sink, all - ((size_t)8 << 20));
        runlin<4, 256>("256-B row segments, distinct", (const char*)A, 256, 224, sink, all - ((size_t)8 << 20));
        runlin<4, 128>("128-B row segments, distinct", (const char*)A, 256, 224, sink, all - ((size_t)8 << 20));
        runlin<4, 1024>("linear 1-KB instr, L2-resident", (const char*)A, 256, 224, sink, (size_t)4 << 20);
        runlin<4, 256>("256-B row segments, L2-resident", (const char*)A, 256, 224, sink, (size_t)4 << 20);
        runlin<4, 128>("128-B row segments, L2-resident", (const char*)A, 256, 224, sink, (size_t)4 << 20);
        for (int g : {1, 2, 4, 8, 16, 32}) { char nm[64]; snprintf(nm, 64, "128-B rows, %d WGs share a stream, D4", g); runlin<4, 128>(nm, (const char*)A, 256, 224, sink, all - ((size_t)8 << 20), g); }
        for (int g : {4, 8}) { char nm[64]; snprintf(nm, 64, "128-B rows, %d WGs share a stream, D2", g); runlin<2, 128>(nm, (const char*)A, 256, 224, sink, all - ((size_t)8 << 20), g); }
        for (int g : {4, 8}) { char nm[64]; snprintf(nm, 64, "linear, %d WGs share a stream, D4", g); runlin<4, 1024>(nm, (const char*)A, 256, 224, sink, all - ((size_t)8 << 20), g); }
        char* big; hipMalloc(&big, (size_t)640 << 20); hipMemset(big, 0x33, (size_t)640 << 20);
        runlin2<4>("GEMM sharing 8x4, D4", (const char*)big, 448, sink, (size_t)1835008, 8);
        runlin2<2>("GEMM sharing 8x4, D2", (const char*)big, 448, sink, (size_t)1835008, 8);
        runlin2<4>("GEMM sharing 8x4, D4, odd pitch", (const char*)big, 448, sink, (size_t)1835008 + 4096 + 256, 8);
        runlin2<4>("GEMM sharing 4x8, D4", (const char*)big, 448, sink, (size_t)1835008, 4);
        runlin2<4>("GEMM sharing 16x2, D4", (const char*)big, 448, sink, (size_t)1835008, 16);
        runlin2<4>("GEMM sharing 32x1, D4", (const char*)big, 448, sink, (size_t)1835008, 32);
        runlin<2, 1024>("linear 1-KB instr, L2-resident D2", (const char*)A, 256, 224, sink, (size_t)4 << 20);
    }
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
