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

// u = scale * A x on the matrix cores.  One workgroup = 128 tokens, its four waves own 32 tokens each over the whole K (no cross-wave reduction);
// K is walked in chunks of 64 columns: the x tile [128 rows x 128 B] (+ the lo tile) and the adapters' A16 tiles [32 rows x 128 B] are fetched with
// whole-line global loads (eight consecutive lanes = one 128-B row segment), parked in registers while the previous chunk is computed, and handed to
// the MFMA lanes through LDS (row stride 144 B: the 16 lanes of a ds_read_b128 lane group cover all 64 banks once).  One 32x32x16 MFMA per 16
// columns, adapter and activation half, with the 32-row A16 (hi rows 0-15, lo rows 16-31) as the second operand; the hi / lo column pairs are added
// across lanes at the end and the u columns leave as whole 128-B row segments.  (The first version let every lane fetch its own 16 bytes of a row --
// 32-B pieces of 32 different lines per instruction -- and read all of A16 per 32 tokens: 225 us per layer for q, k, v at 32,560 tokens, against
// the 50 us the 233 MB of x cost at HBM speed.)
#define AD_TOK 128
#define AD_KC 64
#define AD_RS 144                      // LDS row stride in bytes (128 + 16)
template <int DT, bool LO, int NSEG, int DEPTH>
__global__ __launch_bounds__(256) void adapter_down_kernel(uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, AdapterDownArgs a, int r, float scale, int aug) {
    constexpr int LDS_LOOP = (LO ? 2 : 1) * AD_TOK * AD_RS + 3 * 32 * AD_RS, LDS_OUT = (LO ? 2 : 1) * AD_TOK * 128 * 2;     // K loop's tiles / the u tile at aug = 128
    __shared__ __attribute__((aligned(16))) char lds[LDS_LOOP > LDS_OUT ? LDS_LOOP : LDS_OUT];
    char* xs = lds;                                    // x tile (hi), then (LO) the lo tile
    char* as = lds + (LO ? 2 : 1) * AD_TOK * AD_RS;    // A16 tiles of the (up to three) adapters
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * AD_TOK;
    const int lrow = tid >> 3, lch = tid & 7;          // loader role: row (of a 32-row group) and 16-B chunk
    const int row = lane & 31, kg = lane >> 5;         // MFMA role
    f32x16 acc[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[sg][i] = 0.f;
    // Register prefetch, DEPTH chunks deep: the kernel has almost no arithmetic, so its speed is the number of bytes it keeps in flight -- with one
    // chunk ahead (28 KB per CU) an iteration cost a full memory latency, 4.4 us, and the whole call 246 us.  Everything in the K loop is
    // unconditional (NSEG and DEPTH are compile-time, the chunk index of the tail's surplus prefetches is clamped): a conditional load makes its
    // destination a phi, and the copies the compiler then inserts wait for the load on the spot.
    // (the register sets are plain scalars spelled out by macros: as arrays / structs of uint4 handed to lambdas they were left in scratch memory)
    const uint16_t* xg0 = x16 + min(t0 + lrow, T - 1) * ldx + 8 * lch;
    const uint16_t* xg1 = x16 + min(t0 + 32 + lrow, T - 1) * ldx + 8 * lch;
    const uint16_t* xg2 = x16 + min(t0 + 64 + lrow, T - 1) * ldx + 8 * lch;
    const uint16_t* xg3 = x16 + min(t0 + 96 + lrow, T - 1) * ldx + 8 * lch;
    const uint16_t* ag0 = a.A16[0] + (int64_t)lrow * K + 8 * lch;
    const uint16_t* ag1 = a.A16[NSEG > 1 ? 1 : 0] + (int64_t)lrow * K + 8 * lch;
    const uint16_t* ag2 = a.A16[NSEG > 2 ? 2 : 0] + (int64_t)lrow * K + 8 * lch;
    const int nkc = K / AD_KC;                         // a multiple of DEPTH (launch_adapter_down)
    char* xw = xs + lrow * AD_RS + 16 * lch;           // this thread's LDS slot of row group 0 (groups: + 32 AD_RS each)
    char* aw = as + lrow * AD_RS + 16 * lch;
    const char* xr = xs + (32 * w + row) * AD_RS + 16 * kg;
    const char* ar = as + row * AD_RS + 16 * kg;
#define AD_DECL(S) uint4 S##x0, S##x1, S##x2, S##x3, S##l0, S##l1, S##l2, S##l3, S##a0, S##a1, S##a2
#define AD_LOAD(S, KC)                                                                                   \
    do {                                                                                                 \
        const int64_t off_ = (int64_t)min((KC), nkc - 1) * AD_KC;                                        \
        S##x0 = *(const uint4*)(xg0 + off_); S##x1 = *(const uint4*)(xg1 + off_);                        \
        S##x2 = *(const uint4*)(xg2 + off_); S##x3 = *(const uint4*)(xg3 + off_);                        \
        if constexpr (LO) {                                                                              \
            S##l0 = *(const uint4*)(xg0 + lo_off + off_); S##l1 = *(const uint4*)(xg1 + lo_off + off_);  \
            S##l2 = *(const uint4*)(xg2 + lo_off + off_); S##l3 = *(const uint4*)(xg3 + lo_off + off_);  \
        }                                                                                                \
        S##a0 = *(const uint4*)(ag0 + off_);                                                             \
        if constexpr (NSEG > 1) { S##a1 = *(const uint4*)(ag1 + off_); S##a2 = *(const uint4*)(ag2 + off_); } \
    } while (0)
#define AD_STEP(S, KC)                                                                                   \
    do {                                                                                                 \
        __syncthreads();                               /* the previous chunk's fragment reads are complete */ \
        *(uint4*)(xw) = S##x0; *(uint4*)(xw + 32 * AD_RS) = S##x1; *(uint4*)(xw + 64 * AD_RS) = S##x2; *(uint4*)(xw + 96 * AD_RS) = S##x3; \
        if constexpr (LO) {                                                                              \
            *(uint4*)(xw + AD_TOK * AD_RS) = S##l0; *(uint4*)(xw + (AD_TOK + 32) * AD_RS) = S##l1;       \
            *(uint4*)(xw + (AD_TOK + 64) * AD_RS) = S##l2; *(uint4*)(xw + (AD_TOK + 96) * AD_RS) = S##l3; \
        }                                                                                                \
        *(uint4*)(aw) = S##a0;                                                                           \
        if constexpr (NSEG > 1) { *(uint4*)(aw + 32 * AD_RS) = S##a1; *(uint4*)(aw + 64 * AD_RS) = S##a2; } \
        __syncthreads();                                                                                 \
        AD_LOAD(S, (KC) + DEPTH);                      /* this register set is free again */             \
        _Pragma("unroll") for (int st = 0; st < AD_KC / 16; ++st) {                                      \
            const bf16x8 xa = *(const bf16x8*)(xr + 32 * st);                                            \
            bf16x8 xl = xa;                                                                              \
            if constexpr (LO) xl = *(const bf16x8*)(xr + AD_TOK * AD_RS + 32 * st);                      \
            _Pragma("unroll") for (int sg = 0; sg < NSEG; ++sg) {                                        \
                const bf16x8 bb = *(const bf16x8*)(ar + sg * 32 * AD_RS + 32 * st);                      \
                acc[sg] = mfma32<DT>(xa, bb, acc[sg]);                                                   \
                if constexpr (LO) acc[sg] = mfma32<DT>(xl, bb, acc[sg]);                                 \
            }                                                                                            \
        }                                                                                                \
    } while (0)
    AD_DECL(p0_); AD_DECL(p1_); AD_DECL(p2_); AD_DECL(p3_);
    AD_LOAD(p0_, 0);
    if constexpr (DEPTH > 1) AD_LOAD(p1_, 1);
    if constexpr (DEPTH > 2) { AD_LOAD(p2_, 2); AD_LOAD(p3_, 3); }
    for (int kc = 0; kc < nkc; kc += DEPTH) {
        AD_STEP(p0_, kc);
        if constexpr (DEPTH > 1) AD_STEP(p1_, kc + 1);
        if constexpr (DEPTH > 2) { AD_STEP(p2_, kc + 2); AD_STEP(p3_, kc + 3); }
    }
#undef AD_DECL
#undef AD_LOAD
#undef AD_STEP
    // acc[sg][4 g + jj] <-> token 32 w + 8 g + 4 kg + jj, column (A16 row) = lane & 31: columns j and 16 + j (A_hi and A_lo rows of adapter row j) are added
    // across lanes; the u tile [128 tokens][aug] is assembled in LDS (the x tile's space) and leaves as whole row segments
    __syncthreads();
    uint16_t* uh = (uint16_t*)lds;                     // [AD_TOK][aug] hi, then (LO) lo
    uint16_t* ul = uh + AD_TOK * aug;
    for (int i = tid; i < AD_TOK * aug / 8; i += 256) {
        ((uint4*)uh)[i] = make_uint4(0, 0, 0, 0);
        if constexpr (LO) ((uint4*)ul)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int nr = NSEG * r, j = lane & 15;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = acc[sg][i] + __shfl_xor(acc[sg][i], 16);
            if ((lane & 16) == 0 && j < r) {
                const int tok = 32 * w + 8 * (i >> 2) + 4 * kg + (i & 3);
                const float u = scale * v;
                const uint16_t hi = to16<DT>(u);
                uh[tok * aug + sg * r + j] = hi; uh[tok * aug + nr + sg * r + j] = hi;
                if constexpr (LO) { const uint16_t lo = to16<DT>(u - from16<DT>(hi)); ul[tok * aug + sg * r + j] = lo; ul[tok * aug + nr + sg * r + j] = lo; }
            }
        }
    }
    __syncthreads();
    const int cpr = aug / 8;                           // 16-B chunks per row
    for (int i = tid; i < AD_TOK * cpr; i += 256) {
        const int tok = i / cpr, c = i - tok * cpr;
        if (t0 + tok >= T) continue;
        uint16_t* dst = x16 + (t0 + tok) * ldx + K + 8 * c;
        *(uint4*)dst = ((const uint4*)uh)[i];
        if constexpr (LO) *(uint4*)(dst + lo_off) = ((const uint4*)ul)[i];
    }
}
template <int DT, bool LO, int NSEG>
static void adapter_down_dispatch(dim3 grid, hipStream_t s, int depth, uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, const AdapterDownArgs& a, int r, float scale, int aug) {
    if (depth == 4) hipLaunchKernelGGL((adapter_down_kernel<DT, LO, NSEG, 4>), grid, dim3(256), 0, s, x16, ldx, lo_off, T, K, a, r, scale, aug);
    else if (depth == 2) hipLaunchKernelGGL((adapter_down_kernel<DT, LO, NSEG, 2>), grid, dim3(256), 0, s, x16, ldx, lo_off, T, K, a, r, scale, aug);
    else hipLaunchKernelGGL((adapter_down_kernel<DT, LO, NSEG, 1>), grid, dim3(256), 0, s, x16, ldx, lo_off, T, K, a, r, scale, aug);
}
int launch_adapter_down(uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, const AdapterDownArgs& a, int r, float scale, int aug, int dtype, hipStream_t s) {
    ARG_CHECK(x16 && r > 0 && r <= AD_MAX_R && K % AD_KC == 0 && ldx % 8 == 0 && lo_off % 8 == 0 && (a.n == 1 || a.n == 3) && T > 0);
    ARG_CHECK(aug >= 2 * a.n * r && aug % 8 == 0 && aug <= 128 && ldx >= K + aug && (lo_off == 0 || (lo_off >= K + aug && ldx >= lo_off + K + aug)));
    for (int sg = 0; sg < a.n; ++sg) ARG_CHECK(a.A16[sg] != nullptr);      // an absent adapter of a q / k / v triple: a zeroed table (engine.hip: build_aug)
    const dim3 grid((unsigned)((T + AD_TOK - 1) / AD_TOK));
    const int nkc = K / AD_KC, depth = nkc % 4 == 0 ? 4 : nkc % 2 == 0 ? 2 : 1;
#define AD_GO(LO_, NSEG_) DISPATCH_DT(dtype, (adapter_down_dispatch<DT, LO_, NSEG_>(grid, s, depth, x16, ldx, lo_off, T, K, a, r, scale, aug)))
    if (lo_off > 0) { if (a.n == 3) AD_GO(true, 3); else AD_GO(true, 1); }
    else { if (a.n == 3) AD_GO(false, 3); else AD_GO(false, 1); }
#undef AD_GO
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

template <int DT>
__global__ void split3_f32_kernel(uint16_t* dst, uint16_t* dst1, const float* src, int64_t n, int K, int w_side) {
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const float x = src[i];
        const uint16_t hi = to16<DT>(x), lo = to16<DT>(x - from16<DT>(hi));
        uint16_t* d = dst + r * 3 * K + k;
        d[0] = hi; d[K] = w_side ? lo : hi; d[2 * K] = w_side ? hi : lo;
        if (dst1) dst1[i] = hi;
    }
}
int launch_split3_f32(uint16_t* dst, uint16_t* dst1, const float* src, int64_t n, int K, int w_side, int dtype, hipStream_t s) {
    ARG_CHECK(dst && src && n > 0 && K > 0);
    const int64_t total = n * K;
    DISPATCH_DT(dtype, hipLaunchKernelGGL(split3_f32_kernel<DT>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, dst, dst1, src, n, K, w_side));
    LAUNCH_CHECK();
    return BLIM_OK;
}
__global__ void split3_hilo_kernel(uint16_t* dst, const uint16_t* src, int64_t lds, int64_t n, int K) {
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const uint16_t hi = src[r * lds + k], lo = src[r * lds + K + k];
        uint16_t* d = dst + r * 3 * K + k;
        d[0] = hi; d[K] = hi; d[2 * K] = lo;
    }
}
int launch_split3_hilo(uint16_t* dst, const uint16_t* src, int64_t lds, int64_t n, int K, hipStream_t s) {
    ARG_CHECK(dst && src && n > 0 && K > 0 && lds >= 2 * K);
    const int64_t total = n * K;
    hipLaunchKernelGGL(split3_hilo_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, dst, src, lds, n, K);
    LAUNCH_CHECK();
    return BLIM_OK;
}
