// Kernels of the adapters-apart scoring path (adapters.hpp).  gfx950 only.
#include "adapters.hpp"

#include <algorithm>

#define DISPATCH_DT(dtype, CALL)            \
    do {                                    \
        if ((dtype) == DT_F16) { constexpr int DT = DT_F16; CALL; } \
        else { constexpr int DT = DT_BF16; CALL; }                  \
    } while (0)
#define LAUNCH_CHECK()                                                                                  \
    do {                                                                                                \
        hipError_t _e = hipGetLastError();                                                              \
        if (_e != hipSuccess) { blim_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); return BLIM_ERR_HIP; } \
    } while (0)

#define AD_MAX_R 16

__device__ __forceinline__ int ad_stored_row_of_nat(int n) {      // gemm.hpp: qkv_perm_row inverted, head offset kept
    const int h = n >> 7, d = n & 127;
    return (h << 7) + 32 * ((d & 63) >> 4) + 16 * (d >> 6) + (d & 15);
}

template <int DT>
__global__ void adapter_a16_kernel(uint16_t* A16, const float* A, int K, int r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 32 * K) return;
    const int row = i / K, k = i - row * K;
    const int j = row & 15;
    uint16_t v = 0;
    if (j < r) {
        const float a = A[(int64_t)j * K + k];
        const uint16_t hi = to16<DT>(a);
        v = row < 16 ? hi : to16<DT>(a - from16<DT>(hi));
    }
    A16[i] = v;
}
int launch_adapter_a16(uint16_t* A16, const float* A, int K, int r, int dtype, hipStream_t s) {
    ARG_CHECK(A16 && A && r > 0 && r <= AD_MAX_R && K > 0);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(adapter_a16_kernel<DT>, dim3((32 * K + 255) / 256), dim3(256), 0, s, A16, A, K, r));
    LAUNCH_CHECK();
    return BLIM_OK;
}

template <int DT>
__global__ void adapter_b_aug_kernel(uint16_t* w_aug, int64_t ld, int64_t row0, int col_hi, int col_lo, const float* B, int N, int r, int row_mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * r) return;
    const int n = i / r, j = i - n * r;
    const int64_t srow = row0 + (row_mode == 1 ? ad_stored_row_of_nat(n) : n);
    const float b = B[i];
    const uint16_t hi = to16<DT>(b);
    w_aug[srow * ld + col_hi + j] = hi;
    w_aug[srow * ld + col_lo + j] = to16<DT>(b - from16<DT>(hi));
}
int launch_adapter_b_aug(uint16_t* w_aug, int64_t ld, int64_t row0, int col_hi, int col_lo, const float* B, int N, int r, int row_mode, int dtype, hipStream_t s) {
    ARG_CHECK(w_aug && B && N > 0 && r > 0 && r <= AD_MAX_R);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(adapter_b_aug_kernel<DT>, dim3((N * r + 255) / 256), dim3(256), 0, s, w_aug, ld, row0, col_hi, col_lo, B, N, r, row_mode));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// u = scale * A x on the matrix cores: one workgroup = 32 tokens, its 4 waves split K; one 32x32x16 MFMA per 16 columns (and per activation half)
// with the 32-row A16 (hi rows 0-15, lo rows 16-31) as the second operand, both operands K-contiguous straight from memory; the waves' partial sums
// and the hi / lo row pairs meet in LDS.
template <int DT, bool LO>
__global__ __launch_bounds__(256) void adapter_down_kernel(uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, AdapterDownArgs a, int r, float scale, int aug) {
    __shared__ float red[4][3][32][33];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int row = lane & 31, kg = lane >> 5;
    const int64_t t = min(t0 + row, T - 1);
    const int steps = K / 16, per = (steps + 3) / 4;
    const int s0 = w * per, s1 = min(steps, s0 + per);
    f32x16 acc[3];
#pragma unroll
    for (int sg = 0; sg < 3; ++sg)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[sg][i] = 0.f;
    const uint16_t* xp = x16 + t * ldx + 8 * kg;
    for (int st = s0; st < s1; ++st) {
        const bf16x8 xa = *(const bf16x8*)(xp + 16 * st);
        bf16x8 xl = xa;
        if constexpr (LO) xl = *(const bf16x8*)(xp + lo_off + 16 * st);
#pragma unroll
        for (int sg = 0; sg < 3; ++sg) {
            if (sg < a.n && a.A16[sg]) {
                const bf16x8 bb = *(const bf16x8*)(a.A16[sg] + (int64_t)row * K + 16 * st + 8 * kg);
                acc[sg] = mfma32<DT>(xa, bb, acc[sg]);
                if constexpr (LO) acc[sg] = mfma32<DT>(xl, bb, acc[sg]);
            }
        }
    }
    // acc[sg][4 g + jj] <-> token t0 + 8 g + 4 kg + jj, column (A16 row) = lane & 31
#pragma unroll
    for (int sg = 0; sg < 3; ++sg)
        if (sg < a.n)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) red[w][sg][8 * g + 4 * kg + jj][row] = acc[sg][4 * g + jj];
    __syncthreads();
    const int nr = a.n * r;
    for (int i = threadIdx.x; i < 32 * aug; i += 256) {
        const int m = i / aug, c = i - m * aug;
        if (t0 + m >= T) continue;
        float u = 0.f;
        if (c < 2 * nr) {
            const int cc = c < nr ? c : c - nr;
            const int sg = cc / r, j = cc - sg * r;
            u = scale * ((red[0][sg][m][j] + red[1][sg][m][j] + red[2][sg][m][j] + red[3][sg][m][j]) +
                         (red[0][sg][m][16 + j] + red[1][sg][m][16 + j] + red[2][sg][m][16 + j] + red[3][sg][m][16 + j]));
        }
        const uint16_t hi = to16<DT>(u);
        uint16_t* dst = x16 + (t0 + m) * ldx + K + c;
        *dst = hi;
        if constexpr (LO) dst[lo_off] = to16<DT>(u - from16<DT>(hi));
    }
}
int launch_adapter_down(uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, const AdapterDownArgs& a, int r, float scale, int aug, int dtype, hipStream_t s) {
    ARG_CHECK(x16 && r > 0 && r <= AD_MAX_R && K % 16 == 0 && ldx % 8 == 0 && lo_off % 8 == 0 && (a.n == 1 || a.n == 3) && T > 0);
    ARG_CHECK(aug >= 2 * a.n * r && aug % 8 == 0 && ldx >= K + aug && (lo_off == 0 || (lo_off >= K + aug && ldx >= lo_off + K + aug)));
    const dim3 grid((unsigned)((T + 31) / 32));
    if (lo_off > 0) DISPATCH_DT(dtype, hipLaunchKernelGGL((adapter_down_kernel<DT, true>), grid, dim3(256), 0, s, x16, ldx, lo_off, T, K, a, r, scale, aug));
    else DISPATCH_DT(dtype, hipLaunchKernelGGL((adapter_down_kernel<DT, false>), grid, dim3(256), 0, s, x16, ldx, lo_off, T, K, a, r, scale, aug));
    LAUNCH_CHECK();
    return BLIM_OK;
}

__global__ void make_aug_kernel(uint16_t* dst, const uint16_t* src, int64_t N, int K, int aug) {
    const int Ka = K + aug, c8 = Ka / 8;
    const int64_t total = N * c8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / c8;
        const int c = (int)(i - n * c8) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < K) v = *(const uint4*)(src + n * K + c);
        *(uint4*)(dst + n * Ka + c) = v;
    }
}
int launch_make_aug(uint16_t* dst, const uint16_t* src, int64_t N, int K, int aug, hipStream_t s) {
    ARG_CHECK(dst && src && N > 0 && K % 8 == 0 && aug % 8 == 0);
    const int64_t total = N * ((K + aug) / 8);
    hipLaunchKernelGGL(make_aug_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, dst, src, N, K, aug);
    LAUNCH_CHECK();
    return BLIM_OK;
}

__global__ void copy_rows16_kernel(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n, int K) {
    const int c8 = K / 8;
    const int64_t total = n * c8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c8;
        const int c = (int)(i - row * c8) * 8;
        *(uint4*)(dst + row * ldd + c) = *(const uint4*)(src + row * lds + c);
    }
}
int launch_copy_rows16(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n, int K, hipStream_t s) {
    ARG_CHECK(dst && src && n > 0 && K % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0);
    const int64_t total = n * (K / 8);
    hipLaunchKernelGGL(copy_rows16_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, dst, ldd, src, lds, n, K);
    LAUNCH_CHECK();
    return BLIM_OK;
}
