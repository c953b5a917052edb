// Kernels of the fine-tuning step (train.hpp).  gfx950 only.  Everything here works around the big GEMMs (which run on gemm.hip's
// kernel, forward and backward): the rank-r adapter products and the attention backward's batched products run on 32x32x16 MFMA
// tiles (operands whose contraction index is their row index come through the transposing LDS load, tr_frag); the rest are
// one-pass HBM-bound kernels (RMSNorm / RoPE / GELU / cross-entropy backward, casts, AdamW).
#include "train.hpp"

#include <string.h>

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

// ---------------------------------------------------------------------------- dropout (LoRA input, peft: lora_dropout)
// keep(t, k) of adapter `site` in step `seed`: counter-based, so the backward regenerates the forward's mask.  One 64-bit hash serves
// the four elements idx & ~3 .. + 3 (16 bits each: the drop probability is floor(65536 p) / 65536) -- hashing every element cost 3 % of
// the step.  oracle/train_oracle.py:drop_mult restates this rule for the tests.
__device__ __forceinline__ uint64_t drop_hash(uint64_t seed, uint32_t site, uint64_t grp) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1) + grp * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// multipliers of the four elements idx .. idx + 3 (idx % 4 == 0)
__device__ __forceinline__ void drop_mult4(uint64_t seed, uint32_t site, uint64_t idx, float p, float* m) {
    const uint64_t z = drop_hash(seed, site, idx >> 2);
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    const float keep = 1.0f / (1.0f - p);
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = (((uint32_t)(z >> (16 * q))) & 0xFFFFu) >= thr ? keep : 0.0f;
}

__device__ __forceinline__ int nat_row_of_stored(int r) {   // gemm.hpp qkv_perm_row, head offset kept
    const int h = r >> 7, c = r & 127;
    return (h << 7) + 16 * (c >> 5) + (c & 15) + 64 * ((c >> 4) & 1);
}
__device__ __forceinline__ int stored_row_of_nat(int n) {
    const int h = n >> 7, d = n & 127;
    return (h << 7) + 32 * ((d & 63) >> 4) + 16 * (d >> 6) + (d & 15);
}

// ---------------------------------------------------------------------------- layout
__global__ void transpose16_kernel(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n_rows, int n_cols, int mode, int rope_rows) {
    __shared__ uint16_t tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.y * 32;
    const int c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int64_t r = r0 + i; const int c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < n_rows && c < n_cols) ? src[r * lds + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i; const int64_t r = r0 + threadIdx.x;
        if (c < n_cols && r < n_rows) {
            const int64_t rn = (mode == 1 && r < rope_rows) ? nat_row_of_stored((int)r) : r;
            dst[(int64_t)c * ldd + rn] = tile[threadIdx.x][i];
        }
    }
}
int launch_transpose16(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n_rows, int n_cols, int mode, int rope_rows, hipStream_t s) {
    dim3 grid((n_cols + 31) / 32, (unsigned)((n_rows + 31) / 32));
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(32, 8), 0, s, dst, ldd, src, lds, n_rows, n_cols, mode, rope_rows);
    LAUNCH_CHECK();
    return BLIM_OK;
}

template <int DT>
__global__ void lora_b_to_aug_kernel(uint16_t* w_aug, int64_t ld, int64_t row0, int col0, const float* B, int N, int r, int row_mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * r) return;
    const int n = i / r, j = i - n * r;
    const int64_t srow = row0 + (row_mode == 1 ? stored_row_of_nat(n) : n);
    w_aug[srow * ld + col0 + j] = to16<DT>(B[i]);
}
int launch_lora_b_to_aug(uint16_t* w_aug, int64_t ld, int64_t row0, int col0, const float* B, int N, int r, int row_mode, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_b_to_aug_kernel<DT>, dim3((N * r + 255) / 256), dim3(256), 0, s, w_aug, ld, row0, col0, B, N, r, row_mode));
    LAUNCH_CHECK();
    return BLIM_OK;
}

template <int DT>
__global__ void lora_merge_kernel(uint16_t* dst, const uint16_t* base_aug, int64_t ld_aug, int64_t row0, const float* B, const float* A, int N, int K, int r, float scale, int row_mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int sr = (int)(i / K), k = (int)(i - (int64_t)sr * K);          // sr: stored row inside this projection's block
    const int n = row_mode == 1 ? nat_row_of_stored(sr) : sr;
    float acc = 0.f;
    for (int j = 0; j < r; ++j) acc += B[n * r + j] * A[(int64_t)j * K + k];
    dst[(row0 + sr) * K + k] = to16<DT>(from16<DT>(base_aug[(row0 + sr) * ld_aug + k]) + scale * acc);
}
int launch_lora_merge(uint16_t* dst, const uint16_t* base_aug, int64_t ld_aug, int64_t row0, const float* B, const float* A, int N, int K, int r, float scale,
                      int row_mode, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)N * K;
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_merge_kernel<DT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dst, base_aug, ld_aug, row0, B, A, N, K, r, scale, row_mode));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// MFMA fragment of a row-major LDS tile [k][STRIDE columns] whose contraction index is the ROW: lane (col32 + lane & 31) gets its 8
// k values (k = 16 s + 4 hf + {0..3} and + 8) through the transposing LDS load ds_read_b64_tr_b16 -- the scheme of the forward attention
// kernel's P.V product.  16-byte chunks are XOR-swizzled by the row (STRIDE 128: 16 chunks, STRIDE 32: 4 chunks).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <int STRIDE>
__device__ __forceinline__ int tr_off(int row, int col) {
    const int ch = col >> 3, sw = STRIDE == 128 ? ((row & 3) << 2) : (row & 3);
    return row * STRIDE + 8 * (ch ^ sw) + (col & 7);
}
template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* tile, int col32, int s, int lane) {
    const int i16 = lane & 15, g16 = (lane >> 4) & 1, hf = lane >> 5;
    const int col = col32 + 16 * g16 + 4 * (i16 & 3);
    const int kr0 = 16 * s + 4 * hf + (i16 >> 2), kr1 = kr0 + 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + tr_off<STRIDE>(kr0, col)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + tr_off<STRIDE>(kr1, col)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ---------------------------------------------------------------------------- LoRA
#define LORA_MAX_R 16
__device__ __forceinline__ float block_sum_256(float v, float* red) {   // red: [4] floats of LDS per call site
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// A16[j, k] = 16-bit(A[j, k]) for j < r (rows r..15 stay zero): the B operand of the MFMA below
template <int DT>
__global__ void lora_a16_kernel(uint16_t* A16, const float* A, int K, int r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r * K) A16[i] = to16<DT>(A[i]);
}
int launch_lora_a16(uint16_t* A16, const float* A, int K, int r, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_a16_kernel<DT>, dim3((r * K + 255) / 256), dim3(256), 0, s, A16, A, K, r));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// u~ = scale * drop(x) A^T on the matrix cores: one workgroup = 32 tokens, its 4 waves split K; a 32x32x16 MFMA per 16 columns with the
// 16-row A16 as the second operand (j < r real), both operands K-contiguous straight from memory; the waves' partial sums meet in LDS.
// The adapters of one call (q, k, v) share the x fragment when there is no dropout, otherwise each masks its own copy.
template <int DT>
__global__ __launch_bounds__(256) void lora_down_kernel(uint16_t* x16, int64_t ldx, int64_t T, int K, LoraDownArgs a, int r, float scale, float drop_p, uint64_t seed, uint32_t site) {
    __shared__ float red[4][3][32][LORA_MAX_R];
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
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t* xp = x16 + t * ldx + 8 * kg;
    for (int st = s0; st < s1; ++st) {
        const bf16x8 xa = *(const bf16x8*)(xp + 16 * st);
#pragma unroll
        for (int sg = 0; sg < 3; ++sg) {
            if (sg < a.n) {
                bf16x8 xs = xa;
                if (drop_p > 0.f) {
                    float m[8];
                    const uint64_t b0 = (uint64_t)t * K + 16 * st + 8 * kg;
                    drop_mult4(seed, site + sg, b0, drop_p, m); drop_mult4(seed, site + sg, b0 + 4, drop_p, m + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xs[e] = (short)to16<DT>(from16<DT>((uint16_t)xa[e]) * m[e]);
                }
                const bf16x8 bb = row < 16 ? *(const bf16x8*)(a.A16[sg] + (int64_t)row * K + 16 * st + 8 * kg) : zero;
                acc[sg] = mfma32<DT>(xs, bb, acc[sg]);
            }
        }
    }
    // acc[sg][4g + jj] <-> token t0 + 8 g + 4 kg + jj, column j = lane & 31
    if (row < r) {
#pragma unroll
        for (int sg = 0; sg < 3; ++sg)
            if (sg < a.n)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) red[w][sg][8 * g + 4 * kg + jj][row] = acc[sg][4 * g + jj];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n * 32 * r; i += 256) {
        const int sg = i / (32 * r), rem = i - sg * 32 * r;
        const int m = rem / r, j = rem - m * r;
        if (t0 + m < T) x16[(t0 + m) * ldx + K + sg * r + j] = to16<DT>(scale * (red[0][sg][m][j] + red[1][sg][m][j] + red[2][sg][m][j] + red[3][sg][m][j]));
    }
}
int launch_lora_down(uint16_t* x16, int64_t ldx, int64_t T, int K, const LoraDownArgs& a, int r, float scale, float drop_p, uint64_t seed, uint32_t site, int dtype, hipStream_t s) {
    ARG_CHECK(r > 0 && r <= LORA_MAX_R && K % 16 == 0 && ldx % 8 == 0 && a.n >= 1 && a.n <= 3 && T > 0);
    for (int sg = 0; sg < a.n; ++sg) ARG_CHECK(a.A16[sg] != nullptr);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_down_kernel<DT>, dim3((unsigned)((T + 31) / 32)), dim3(256), 0, s, x16, ldx, T, K, a, r, scale, drop_p, seed, site));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// out[c, j] += sum_t X[t, c] * U[t, j]  (the two rank-r weight gradients: dB with X = dy, U = u~;  dA with X = drop(x), U = du, stored
// transposed) on the matrix cores.  The contraction index t is the ROW index of both operands, so the tiles are staged row-major
// (coalesced 16-byte copies of X) and the fragments come from the transposing LDS load (tr_frag).  One workgroup = 128 columns x 1024
// rows, 64 rows per stage, wave w owns columns 32 w .. 32 w + 31 and one 32x32 accumulator (n = j < r real, the rest zero padding).
// An f32 U (du carries the loss scale: up to ~1e4 and down to ~1e-3 in one tensor) enters as hi + lo with hi = 16-bit(u / 256),
// lo = 16-bit(u - 256 hi): two MFMAs, out = 256 acc_hi + acc_lo, exact to ~3e-5 relative over that whole range in fp16.
// Time splits meet through PARTIALS, not float atomics: split y stores its [C, r] contribution to part + y * C * r (every element of it is
// written by exactly one lane) and ordered_sum_kernel adds the splits to `out` in split order -- the gradients are reproducible bit for bit
// from run to run, like the reference's autograd (training_utils.py:81-91).
// out[i] = (accumulate ? out[i] : 0) + sum_{s < ns, in order} part[s * stride + i]
__global__ void ordered_sum_kernel(float* out, const float* part, int64_t n, int ns, int64_t stride, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = accumulate ? out[i] : 0.f;
    for (int sidx = 0; sidx < ns; ++sidx) a += part[(int64_t)sidx * stride + i];
    out[i] = a;
}
static int launch_ordered_sum(float* out, const float* part, int64_t n, int ns, int64_t stride, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, part, n, ns, stride, accumulate);
    LAUNCH_CHECK();
    return BLIM_OK;
}
#define LORA_TSPLIT 1024
size_t lora_wgrad_scratch_bytes(int64_t T, int C, int r) { return (size_t)((T + LORA_TSPLIT - 1) / LORA_TSPLIT) * C * r * 4; }
template <int DT, bool U_F32, bool OUT_T>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(float* part, const uint16_t* X, int64_t ldx, const void* Uv, int64_t ldu, int64_t T, int C, int r,
                                                         float drop_p, uint64_t seed, uint32_t site, int drop_k) {
    __shared__ __attribute__((aligned(16))) uint16_t x_lds[64 * 128];
    __shared__ __attribute__((aligned(16))) uint16_t u_hi[64 * 32];
    __shared__ __attribute__((aligned(16))) uint16_t u_lo[64 * 32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5;
    const int c0 = blockIdx.x * 128;
    const int64_t tb = (int64_t)blockIdx.y * LORA_TSPLIT, te = min(T, tb + LORA_TSPLIT);
    for (int i = tid; i < 64 * 32; i += 256) { u_hi[i] = 0; u_lo[i] = 0; }          // columns >= r stay zero
    f32x16 acc_hi, acc_lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_hi[i] = 0.f; acc_lo[i] = 0.f; }
    for (int64_t t0 = tb; t0 < te; t0 += 64) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u;
            const int row = idx >> 4, ch = idx & 15;
            const int64_t t = t0 + row;
            const int col = c0 + 8 * ch;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t < te && col < C) {
                v = *(const uint4*)(X + t * ldx + col);
                if (drop_p > 0.f) {
                    uint16_t* e = (uint16_t*)&v;
                    float m[8];
                    const uint64_t b0 = (uint64_t)t * drop_k + col;
                    drop_mult4(seed, site, b0, drop_p, m); drop_mult4(seed, site, b0 + 4, drop_p, m + 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) e[q] = to16<DT>(from16<DT>(e[q]) * m[q]);
                }
            }
            *(uint4*)(x_lds + tr_off<128>(row, 8 * ch)) = v;
        }
        for (int i = tid; i < 64 * r; i += 256) {
            const int row = i / r, j = i - row * r;
            const int64_t t = t0 + row;
            uint16_t h = 0, l = 0;
            if (t < te) {
                if (U_F32) {
                    const float uv = ((const float*)Uv)[t * ldu + j];
                    h = to16<DT>(uv * (1.0f / 256.0f));
                    l = to16<DT>(uv - 256.0f * from16<DT>(h));
                } else h = ((const uint16_t*)Uv)[t * ldu + j];
            }
            u_hi[tr_off<32>(row, j)] = h;
            if (U_F32) u_lo[tr_off<32>(row, j)] = l;
        }
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const bf16x8 af = tr_frag<128>(x_lds, 32 * w, s4, lane);
            acc_hi = mfma32<DT>(af, tr_frag<32>(u_hi, 0, s4, lane), acc_hi);
            if (U_F32) acc_lo = mfma32<DT>(af, tr_frag<32>(u_lo, 0, s4, lane), acc_lo);
        }
    }
    const int n = lane & 31;
    if (n < r) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = c0 + 32 * w + 8 * g + 4 * hf + jj;
                if (m >= C) continue;
                const float v = U_F32 ? 256.0f * acc_hi[4 * g + jj] + acc_lo[4 * g + jj] : acc_hi[4 * g + jj];
                part[(int64_t)blockIdx.y * C * r + (OUT_T ? (int64_t)n * C + m : (int64_t)m * r + n)] = v;
            }
    }
}
int launch_lora_dB(float* dB, const uint16_t* dy16, int64_t ldy, const uint16_t* u16, int64_t ldu, int64_t T, int N, int r, int dtype, float* scratch, hipStream_t s) {
    ARG_CHECK(r <= LORA_MAX_R && ldy % 8 == 0 && ldy >= (N + 7) / 8 * 8 && scratch);      // the last 16-byte chunk of a row may reach into the row's padding columns
    dim3 grid((N + 127) / 128, (unsigned)((T + LORA_TSPLIT - 1) / LORA_TSPLIT));
    DISPATCH_DT(dtype, hipLaunchKernelGGL((lora_wgrad_kernel<DT, false, false>), grid, dim3(256), 0, s, scratch, dy16, ldy, (const void*)u16, ldu, T, N, r, 0.f, 0ull, 0u, 0));
    LAUNCH_CHECK();
    return launch_ordered_sum(dB, scratch, (int64_t)N * r, (int)grid.y, (int64_t)N * r, 1, s);
}
int launch_lora_dA(float* dA, const float* du, const uint16_t* x16, int64_t ldx, int64_t T, int K, int r, float drop_p, uint64_t seed, uint32_t site, int dtype, float* scratch,
                   hipStream_t s) {
    ARG_CHECK(r <= LORA_MAX_R && K % 8 == 0 && ldx % 8 == 0 && scratch);
    dim3 grid((K + 127) / 128, (unsigned)((T + LORA_TSPLIT - 1) / LORA_TSPLIT));
    DISPATCH_DT(dtype, hipLaunchKernelGGL((lora_wgrad_kernel<DT, true, true>), grid, dim3(256), 0, s, scratch, x16, ldx, (const void*)du, (int64_t)r, T, K, r, drop_p, seed, site, K));
    LAUNCH_CHECK();
    return launch_ordered_sum(dA, scratch, (int64_t)K * r, (int)grid.y, (int64_t)K * r, 1, s);
}

// Bt16[j, n] = 16-bit(B[n, j]) (j < r; rows r..15 and columns N.. stay zero): the B operand of the MFMA below
template <int DT>
__global__ void lora_bt_kernel(uint16_t* Bt16, int64_t ldb, const float* B, int N, int r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * r) return;
    const int n = i / r, j = i - n * r;
    Bt16[(int64_t)j * ldb + n] = to16<DT>(B[i]);
}
int launch_lora_bt(uint16_t* Bt16, int64_t ldb, const float* B, int N, int r, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_bt_kernel<DT>, dim3((N * r + 255) / 256), dim3(256), 0, s, Bt16, ldb, B, N, r));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// du[t, j] += scale * sum_n dy[t, n] * B[n, j] on the matrix cores: one wave = 32 rows x one slice of N, a 32x32x16 MFMA per 16 columns
// with B^T (16 rows: j < r real, the rest zero) as the second operand -- 3/4 of the tile is padding, but the kernel is bound by
// reading dy once.  Slices of N meet through partials summed in slice order (ordered_sum_kernel above), not atomics.
template <int DT>
__global__ __launch_bounds__(256) void lora_du_kernel(float* part, const uint16_t* dy16, int64_t ldy, const uint16_t* Bt16, int64_t ldb, int64_t T, int steps_total, int steps_per_split,
                                                      int r, float scale) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + w) * 32;
    if (t0 >= T) return;
    const int row = lane & 31, kg = lane >> 5;
    const int s0 = blockIdx.y * steps_per_split, s1 = min(steps_total, s0 + steps_per_split);
    const bool rv = t0 + row < T, bv = row < 16;
    const uint16_t* ap = dy16 + (t0 + row) * ldy + 8 * kg;
    const uint16_t* bp = Bt16 + (int64_t)row * ldb + 8 * kg;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    int st = s0;
    for (; st + 4 <= s1; st += 4) {          // four loads of each operand in flight per MFMA group
        bf16x8 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = rv ? *(const bf16x8*)(ap + 16 * (st + u)) : zero;
            b[u] = bv ? *(const bf16x8*)(bp + 16 * (st + u)) : zero;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma32<DT>(a[u], b[u], acc);
    }
    for (; st < s1; ++st) {
        const bf16x8 a = rv ? *(const bf16x8*)(ap + 16 * st) : zero;
        const bf16x8 b = bv ? *(const bf16x8*)(bp + 16 * st) : zero;
        acc = mfma32<DT>(a, b, acc);
    }
    if (row < r) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int64_t t = t0 + 8 * g + 4 * kg + jj;
                if (t < T) part[(int64_t)blockIdx.y * T * r + t * r + row] = scale * acc[4 * g + jj];
            }
    }
}
static void lora_du_splits(int64_t T, int N, int* per, int* n_slices) {
    const int row_blocks = (int)((T + 127) / 128), steps_total = N / 16;
    int n_split = (1024 + row_blocks - 1) / row_blocks;
    n_split = std::max(1, std::min(n_split, (steps_total + 15) / 16));
    *per = (steps_total + n_split - 1) / n_split;
    *n_slices = (steps_total + *per - 1) / *per;
}
size_t lora_du_scratch_bytes(int64_t T, int N, int r) {
    int per, ns; lora_du_splits(T, N, &per, &ns);
    return (size_t)ns * T * r * 4;
}
int launch_lora_du(float* du, const uint16_t* dy16, int64_t ldy, const uint16_t* Bt16, int64_t ldb, int64_t T, int N, int r, float scale, int dtype, float* scratch, hipStream_t s) {
    ARG_CHECK(r <= LORA_MAX_R && T > 0 && ldy % 8 == 0 && ldb % 8 == 0 && N % 16 == 0 && scratch);
    int per, ns; lora_du_splits(T, N, &per, &ns);
    dim3 grid((unsigned)((T + 127) / 128), ns);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_du_kernel<DT>, grid, dim3(256), 0, s, scratch, dy16, ldy, Bt16, ldb, T, N / 16, per, r, scale));
    LAUNCH_CHECK();
    return launch_ordered_sum(du, scratch, T * r, ns, T * r, 0, s);     // every (t < T, j < r) element of every slice was written: no clearing needed
}

// sum over the adapters reading x of keep_seg(t, k..k+3) / (1 - p) * sum_j du_seg[t, j] * A_seg[j, k..k+3]
__device__ __forceinline__ float4 lora_dx4(const LoraDxArgs& a, int64_t t, int k, int K, int r, float drop_p, uint64_t seed, uint32_t site) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sg = 0; sg < a.n; ++sg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < r; ++j) {
            const float d = a.du[sg][t * r + j];
            const float4 av = *(const float4*)(a.A[sg] + (int64_t)j * K + k);
            acc.x += d * av.x; acc.y += d * av.y; acc.z += d * av.z; acc.w += d * av.w;
        }
        if (drop_p > 0.f) {
            const uint64_t b = (uint64_t)t * K + k;
            float m[4];
            drop_mult4(seed, site + sg, b, drop_p, m);
            acc.x *= m[0]; acc.y *= m[1]; acc.z *= m[2]; acc.w *= m[3];
        }
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    }
    return o;
}

// dx[t, k] += the rank-r term (one read-modify-write pass); out16 (optional): 16-bit copy of the result
template <int DT>
__global__ void lora_dx_kernel(float* dx, int64_t ldd, LoraDxArgs a, int64_t T, int K, int r, float drop_p, uint64_t seed, uint32_t site, uint16_t* out16, int64_t ldo) {
    const int k4 = K / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * k4) return;
    const int64_t t = i / k4; const int k = (int)(i - t * k4) * 4;
    float4 o = *(float4*)(dx + t * ldd + k);
    const float4 l = lora_dx4(a, t, k, K, r, drop_p, seed, site);
    o.x += l.x; o.y += l.y; o.z += l.z; o.w += l.w;
    if (out16) *(uint2*)(out16 + t * ldo + k) = make_uint2(pack2<DT>(o.x, o.y), pack2<DT>(o.z, o.w));
    else *(float4*)(dx + t * ldd + k) = o;
}
int launch_lora_dx(float* dx, int64_t ldd, const LoraDxArgs& a, int64_t T, int K, int r, float drop_p, uint64_t seed, uint32_t site, hipStream_t s, uint16_t* out16, int64_t ldo, int dtype) {
    ARG_CHECK(K % 4 == 0 && ldd % 4 == 0 && a.n >= 1 && a.n <= 3);
    const int64_t total = T * (K / 4);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(lora_dx_kernel<DT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dx, ldd, a, T, K, r, drop_p, seed, site, out16, ldo));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- RMSNorm backward
// dy_eff = dy + (optional) the adapters' rank-r input gradient, formed on the fly and kept in registers between the two passes.  The
// adapter variant handles RB rows per workgroup: every A chunk a thread loads (A is [r, K] f32 per adapter, 344 KB for q + k + v at 7B,
// L2-resident) serves RB rows instead of one.
#define RB_MAXV 8      // float4 per thread and row: H <= 8192 (plain variant)
#define RB_ROWS 4      // adapter variant: rows per workgroup, H <= 4096
template <int DT, bool LORA>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(float* dx, const float* dy, const float* x, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps, int accumulate,
                                                          uint16_t* out16, LoraDxArgs la, int r, float drop_p, uint64_t seed, uint32_t site) {
    __shared__ float red[4];
    constexpr int NR = LORA ? RB_ROWS : 1;
    constexpr int NV = LORA ? 4 : RB_MAXV;
    const int64_t i0 = (int64_t)blockIdx.x * NR;
    float4 dv[NR][NV];
    float ss[NR], dot[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) { ss[q] = 0.f; dot[q] = 0.f; }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int k = (threadIdx.x + 256 * u) * 4;
        if (k < H) {
            const float4 wv = *(const float4*)(w + k);
            float4 l[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) l[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (LORA) {
                for (int sg = 0; sg < la.n; ++sg) {
                    float4 acc[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int j = 0; j < r; ++j) {
                        const float4 av = *(const float4*)(la.A[sg] + (int64_t)j * H + k);
#pragma unroll
                        for (int q = 0; q < NR; ++q) {
                            const float d = la.du[sg][min(i0 + q, n_rows - 1) * r + j];
                            acc[q].x += d * av.x; acc[q].y += d * av.y; acc[q].z += d * av.z; acc[q].w += d * av.w;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        if (drop_p > 0.f) {
                            const uint64_t b = (uint64_t)min(i0 + q, n_rows - 1) * H + k;
                            float m[4];
                            drop_mult4(seed, site + sg, b, drop_p, m);
                            acc[q].x *= m[0]; acc[q].y *= m[1]; acc[q].z *= m[2]; acc[q].w *= m[3];
                        }
                        l[q].x += acc[q].x; l[q].y += acc[q].y; l[q].z += acc[q].z; l[q].w += acc[q].w;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int64_t i = min(i0 + q, n_rows - 1);
                const int64_t row = rows ? rows[i] : i;
                const float4 xv = *(const float4*)(x + row * H + k);
                float4 d = *(const float4*)(dy + i * H + k);
                d.x += l[q].x; d.y += l[q].y; d.z += l[q].z; d.w += l[q].w;
                dv[q][u] = d;
                ss[q] += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
                dot[q] += wv.x * d.x * xv.x + wv.y * d.y * xv.y + wv.z * d.z * xv.z + wv.w * d.w * xv.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const float s2 = block_sum_256(ss[q], red);
        const float dt2 = block_sum_256(dot[q], red);
        const int64_t i = i0 + q;
        if (i >= n_rows) continue;                       // uniform per workgroup
        const int64_t row = rows ? rows[i] : i;
        const float rs = rsqrtf(s2 / (float)H + eps);
        const float c = rs * rs * rs * dt2 / (float)H;
        float* o = dx + row * H;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int k = (threadIdx.x + 256 * u) * 4;
            if (k < H) {
                const float4 xv = *(const float4*)(x + row * H + k), wv = *(const float4*)(w + k), d = dv[q][u];
                float4 g = make_float4(rs * wv.x * d.x - xv.x * c, rs * wv.y * d.y - xv.y * c, rs * wv.z * d.z - xv.z * c, rs * wv.w * d.w - xv.w * c);
                if (accumulate) { const float4 pp = *(const float4*)(o + k); g.x += pp.x; g.y += pp.y; g.z += pp.z; g.w += pp.w; }
                *(float4*)(o + k) = g;
                if (out16) *(uint2*)(out16 + row * H + k) = make_uint2(pack2<DT>(g.x, g.y), pack2<DT>(g.z, g.w));
            }
        }
    }
}
int launch_rmsnorm_bwd(float* dx, const float* dy, const float* x, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps, int accumulate, uint16_t* out16, int dtype,
                       hipStream_t s, const LoraDxArgs* la, int r, float drop_p, uint64_t seed, uint32_t site) {
    ARG_CHECK(n_rows > 0 && H % 4 == 0 && H <= 1024 * RB_MAXV && (!out16 || !rows) && (!la || (!rows && H <= 4096)));
    LoraDxArgs z; memset(&z, 0, sizeof(z));
    if (la) DISPATCH_DT(dtype, hipLaunchKernelGGL((rmsnorm_bwd_kernel<DT, true>), dim3((unsigned)((n_rows + RB_ROWS - 1) / RB_ROWS)), dim3(256), 0, s, dx, dy, x, rows, n_rows, H, w, eps, accumulate,
                                                  out16, *la, r, drop_p, seed, site));
    else DISPATCH_DT(dtype, hipLaunchKernelGGL((rmsnorm_bwd_kernel<DT, false>), dim3((unsigned)n_rows), dim3(256), 0, s, dx, dy, x, rows, n_rows, H, w, eps, accumulate, out16, z, 0, 0.f,
                                               0ull, 0u));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- casts / GELU
template <int DT>
__global__ void f32_to_16_kernel(uint16_t* out, int64_t ldo, const float* in, int64_t ldi, int64_t rows, int cols, float scale) {
    const int c4 = cols / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c4) return;
    const int64_t r = i / c4; const int c = (int)(i - r * c4) * 4;
    const float4 v = *(const float4*)(in + r * ldi + c);
    *(uint2*)(out + r * ldo + c) = make_uint2(pack2<DT>(v.x * scale, v.y * scale), pack2<DT>(v.z * scale, v.w * scale));
}
int launch_f32_to_16(uint16_t* out, int64_t ldo, const float* in, int64_t ldi, int64_t rows, int cols, float scale, int dtype, hipStream_t s) {
    ARG_CHECK(cols % 4 == 0 && ldo % 4 == 0 && ldi % 4 == 0 && rows > 0);
    const int64_t total = rows * (cols / 4);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(f32_to_16_kernel<DT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, ldo, in, ldi, rows, cols, scale));
    LAUNCH_CHECK();
    return BLIM_OK;
}

template <int DT, bool BWD>
__global__ void gelu_kernel(uint16_t* out, int64_t ldo, const uint16_t* pre16, const float* dh, int64_t rows, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const int64_t r = i / H; const int k = (int)(i - r * H);
    const float x = from16<DT>(pre16[i]);
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    if (!BWD) out[r * ldo + k] = to16<DT>(x * cdf);
    else out[r * ldo + k] = to16<DT>(dh[i] * (cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x)));
}
int launch_gelu_fwd(uint16_t* h16, int64_t ldo, const uint16_t* pre16, int64_t rows, int H, int dtype, hipStream_t s) {
    const int64_t total = rows * H;
    DISPATCH_DT(dtype, hipLaunchKernelGGL((gelu_kernel<DT, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h16, ldo, pre16, nullptr, rows, H));
    LAUNCH_CHECK();
    return BLIM_OK;
}
int launch_gelu_bwd(uint16_t* dpre16, const float* dh, const uint16_t* pre16, int64_t rows, int H, int dtype, hipStream_t s) {
    const int64_t total = rows * H;
    DISPATCH_DT(dtype, hipLaunchKernelGGL((gelu_kernel<DT, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpre16, (int64_t)H, pre16, dh, rows, H));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- cross-entropy forward + backward
template <int DT>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* logits, int64_t ldl, int V, const int32_t* labels, int label_div, float coef, uint16_t* dl16, float* dl32,
                                                         int64_t ldd, float* row_loss) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const float* lg = logits + r * ldl;
    const int lab = labels[r / label_div];
    float m = -3.0e38f;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, lg[v]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) sum += __expf(lg[v] - m);
    sum = block_sum_256(sum, red);
    const float inv = 1.0f / sum;
    if (threadIdx.x == 0) row_loss[r] = (lab >= 0 && lab < V) ? -(lg[lab] - m - __logf(sum)) : 0.f;
    const float c = (lab >= 0 && lab < V) ? coef : 0.f;
    for (int v = threadIdx.x; v < (int)ldd; v += 256) {
        const float d = v < V ? c * (__expf(lg[v] - m) * inv - (v == lab ? 1.0f : 0.0f)) : 0.f;
        if (dl16) dl16[r * ldd + v] = to16<DT>(d);
        else dl32[r * ldd + v] = d;
    }
}
// *loss += sum of x[0..n) in a fixed order (one workgroup: strided partial sums per thread, then the block tree) -- the loss value is
// reproducible bit for bit, which a float atomic per row is not
__global__ __launch_bounds__(256) void fixed_order_sum_kernel(float* loss, const float* x, int64_t n) {
    __shared__ float red[4];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += x[i];
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) *loss += a;
}
int launch_ce_fwd_bwd(const float* logits, int64_t ldl, int V, const int32_t* labels, int label_div, int64_t n_rows, float coef, uint16_t* dl16, float* dl32, int64_t ldd,
                      float* loss, int dtype, float* scratch, hipStream_t s) {
    ARG_CHECK(n_rows > 0 && ldd >= V && label_div >= 1 && (dl16 || dl32) && scratch);          // scratch: n_rows floats (per-row losses)
    DISPATCH_DT(dtype, hipLaunchKernelGGL(ce_fwd_bwd_kernel<DT>, dim3((unsigned)n_rows), dim3(256), 0, s, logits, ldl, V, labels, label_div, coef, dl16, dl32, ldd, scratch));
    LAUNCH_CHECK();
    hipLaunchKernelGGL(fixed_order_sum_kernel, dim3(1), dim3(256), 0, s, loss, scratch, n_rows);
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- TVG head
template <int DT>
__global__ void tvg_dvh_kernel(float* dvh, const float* dl, const uint16_t* vocab16, int C, int N, int M, float scale) {
    const int bc = blockIdx.x;
    const int m = blockIdx.y * 256 + threadIdx.x;
    if (m >= M) return;
    const uint16_t* voc = vocab16 + (int64_t)(bc % C) * N * M;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += dl[(int64_t)bc * N + n] * from16<DT>(voc[(int64_t)n * M + m]);
    dvh[(int64_t)bc * M + m] = scale * acc;
}
int launch_tvg_dvh(float* dvh, const float* dl, const uint16_t* vocab16, int n_rows, int C, int N, int M, float scale, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(tvg_dvh_kernel<DT>, dim3(n_rows, (M + 255) / 256), dim3(256), 0, s, dvh, dl, vocab16, C, N, M, scale));
    LAUNCH_CHECK();
    return BLIM_OK;
}
template <int DT>
__global__ void outer_acc_kernel(float* dW, const float* dvh, const uint16_t* h16, int64_t ldh, int n_rows, int M, int H) {
    const int m = blockIdx.x;
    const int h = blockIdx.y * 256 + threadIdx.x;
    if (h >= H) return;
    float acc = 0.f;
    for (int b = 0; b < n_rows; ++b) acc += dvh[(int64_t)b * M + m] * from16<DT>(h16[(int64_t)b * ldh + h]);
    dW[(int64_t)m * H + h] += acc;
}
int launch_outer_acc(float* dW, const float* dvh, const uint16_t* h16, int64_t ldh, int n_rows, int M, int H, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(outer_acc_kernel<DT>, dim3(M, (H + 255) / 256), dim3(256), 0, s, dW, dvh, h16, ldh, n_rows, M, H));
    LAUNCH_CHECK();
    return BLIM_OK;
}
__global__ void rows_matmul_kernel(float* out, const float* dvh, const float* W, int M, int H) {
    const int b = blockIdx.x;
    const int h = blockIdx.y * 256 + threadIdx.x;
    if (h >= H) return;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += dvh[(int64_t)b * M + m] * W[(int64_t)m * H + h];
    out[(int64_t)b * H + h] = acc;
}
int launch_rows_matmul(float* out, const float* dvh, const float* W, int n_rows, int M, int H, hipStream_t s) {
    hipLaunchKernelGGL(rows_matmul_kernel, dim3(n_rows, (H + 255) / 256), dim3(256), 0, s, out, dvh, W, M, H);
    LAUNCH_CHECK();
    return BLIM_OK;
}

// d embeds -> d projector outputs.  Token t with src_index[t] = -(f + 1): f < F is row f of the `mlp` output (a VTG video token):
// dout_a[f] = dres[t]; f >= F is clip mean f - F of the `tvg_mlp` output: dout_b[(f - F) * group + g] = dres[t] / group, g < group.
template <int DT>
__global__ void feat_grad_kernel(uint16_t* dout_a, uint16_t* dout_b, const float* dres, const int32_t* src_index, int H, int64_t F, int group) {
    const int64_t t = blockIdx.x;
    const int src = src_index[t];
    if (src >= 0) return;
    const int64_t f = -(int64_t)src - 1;
    if (f < F) {
        for (int k = threadIdx.x; k < H; k += blockDim.x) dout_a[f * H + k] = to16<DT>(dres[t * H + k]);
    } else {
        const float inv = 1.0f / (float)group;
        for (int k = threadIdx.x; k < H; k += blockDim.x) {
            const uint16_t v = to16<DT>(dres[t * H + k] * inv);
            for (int g = 0; g < group; ++g) dout_b[((f - F) * group + g) * H + k] = v;
        }
    }
}
int launch_feat_grad(uint16_t* dout_a, uint16_t* dout_b, const float* dres, const int32_t* src_index, int64_t T, int H, int64_t F, int group, int dtype, hipStream_t s) {
    DISPATCH_DT(dtype, hipLaunchKernelGGL(feat_grad_kernel<DT>, dim3((unsigned)T), dim3(256), 0, s, dout_a, dout_b, dres, src_index, H, F, group));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- attention backward
// P = exp(scale * Q K^T - lse) is re-materialised per (sequence, head) as a 16-bit [Lm, Lm] matrix (Lm = round_up(longest row, 64); lse =
// the forward kernel's log-sum-exp per row, so no softmax pass), then  dV = P^T dO,  dS = scale * P o (dO V^T - D) with
// D = rowsum(dO o O) (formed in the product's epilogue),  dQ = dS K,  dK = dS^T Q: five batched products on 64x64 tiles (4 waves x one
// 32x32x16 MFMA accumulator each).  Training rows are a few hundred tokens: P and dS are a few hundred MB per layer and the products
// ~0.4 % of the step's flops.
int64_t attn_bwd_lm(int max_len) { return (max_len + 63) / 64 * 64; }

enum { AB_S = 0, AB_DP = 1, AB_DQ = 2, AB_DV = 3, AB_DK = 4 };
#define AB_LD 40    // LDS row stride (16-bit elements) of a [64][32] operand tile: 80 B keeps 16-B alignment and spreads banks

template <int DT, int MODE>
__global__ __launch_bounds__(256) void attn_bgemm_kernel(AttnBwdParams p, int Lm) {
    __shared__ __attribute__((aligned(16))) uint16_t As[64 * AB_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[64 * AB_LD];
    const int nh = p.num_heads, nkv = p.num_kv_heads, grp = nh / nkv;
    const int HB = (MODE == AB_DV || MODE == AB_DK) ? nkv : nh;
    const int s = blockIdx.z / HB, hb = blockIdx.z % HB;
    const int L = p.seq_len[s];
    const int64_t t0 = p.seq_start[s];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    if (m0 >= L) return;
    if ((MODE == AB_S || MODE == AB_DP) && n0 > m0 + 63) {
        // above the diagonal: never read, except the tile that shares a 128-column block with a diagonal tile (attn_tn_kernel reads
        // 128-wide column blocks of P / dS): that one is written as zeros
        if (n0 < (m0 / 128 + 1) * 128) {
            uint16_t* dst = (MODE == AB_S ? p.P16 : p.dS16) + ((int64_t)s * p.num_heads + hb) * (int64_t)Lm * Lm;
            for (int i = threadIdx.x; i < 64 * 8; i += 256) {
                const int r = m0 + (i >> 3), c = n0 + 8 * (i & 7);
                if (r < L) *(uint4*)(dst + (int64_t)r * Lm + c) = make_uint4(0, 0, 0, 0);
            }
        }
        return;
    }
    if ((MODE == AB_S || MODE == AB_DP) && n0 >= L) {      // columns beyond the row's length: zeros (same readers)
        uint16_t* dst = (MODE == AB_S ? p.P16 : p.dS16) + ((int64_t)s * p.num_heads + hb) * (int64_t)Lm * Lm;
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int r = m0 + (i >> 3), c = n0 + 8 * (i & 7);
            if (r < L) *(uint4*)(dst + (int64_t)r * Lm + c) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const int kvh = HB == nh ? hb / grp : hb;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int64_t mat = ((int64_t)s * nh) * (int64_t)Lm * Lm;      // + h * Lm * Lm
    const int n_hh = (MODE == AB_DV || MODE == AB_DK) ? grp : 1;
    for (int hh = 0; hh < n_hh; ++hh) {
        const int h = (MODE == AB_DV || MODE == AB_DK) ? kvh * grp + hh : hb;
        const uint16_t* PS = (MODE == AB_DV ? p.P16 : p.dS16) + mat + (int64_t)h * Lm * Lm;
        int k_begin = 0, k_end = 128;
        if (MODE == AB_DQ) { k_begin = 0; k_end = min((L + 31) / 32 * 32, m0 + 64); }
        if (MODE == AB_DV || MODE == AB_DK) { k_begin = m0; k_end = (L + 31) / 32 * 32; }
        for (int k0 = k_begin; k0 < k_end; k0 += 32) {
            __syncthreads();
            // ---- A tile: As[m][k]
            if (MODE == AB_S || MODE == AB_DP || MODE == AB_DQ) {     // k-contiguous source
                const int row = tid >> 2, kc = (tid & 3) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                const int i = m0 + row;
                if (i < L) {
                    if (MODE == AB_S) v = *(const uint4*)(p.qkv + (t0 + i) * p.ldq + hb * 128 + k0 + kc);
                    else if (MODE == AB_DP) v = *(const uint4*)(p.dout + (t0 + i) * p.ldo + hb * 128 + k0 + kc);
                    else v = *(const uint4*)(PS + (int64_t)i * Lm + k0 + kc);
                }
                *(uint4*)(As + row * AB_LD + kc) = v;
            } else {                                                  // m-contiguous source: A(m = j, k = i) = P/dS[i][j]
                const int k = tid >> 3, mc = (tid & 7) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                const int i = k0 + k;
                if (i < L) v = *(const uint4*)(PS + (int64_t)i * Lm + m0 + mc);
                const uint16_t* e = (const uint16_t*)&v;
#pragma unroll
                for (int q = 0; q < 8; ++q) As[(mc + q) * AB_LD + k] = e[q];
            }
            // ---- B tile: Bs[n][k]
            if (MODE == AB_S || MODE == AB_DP) {                      // B(n = j, k = d): k-contiguous rows of K / V
                const int row = tid >> 2, kc = (tid & 3) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                const int j = n0 + row;
                if (j < L) v = *(const uint4*)(p.qkv + (t0 + j) * p.ldq + (MODE == AB_S ? nh + kvh : nh + nkv + kvh) * 128 + k0 + kc);
                *(uint4*)(Bs + row * AB_LD + kc) = v;
            } else {                                                  // B(n = d, k = token): n-contiguous rows
                const int k = tid >> 3, nc = (tid & 7) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                const int tok = k0 + k;
                if (tok < L) {
                    if (MODE == AB_DQ) v = *(const uint4*)(p.qkv + (t0 + tok) * p.ldq + (nh + kvh) * 128 + n0 + nc);
                    else if (MODE == AB_DV) v = *(const uint4*)(p.dout + (t0 + tok) * p.ldo + h * 128 + n0 + nc);
                    else v = *(const uint4*)(p.qkv + (t0 + tok) * p.ldq + h * 128 + n0 + nc);
                }
                const uint16_t* e = (const uint16_t*)&v;
#pragma unroll
                for (int q = 0; q < 8; ++q) Bs[(nc + q) * AB_LD + k] = e[q];
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 a = *(const bf16x8*)(As + (32 * wm + (lane & 31)) * AB_LD + 16 * kk + 8 * (lane >> 5));
                const bf16x8 b = *(const bf16x8*)(Bs + (32 * wn + (lane & 31)) * AB_LD + 16 * kk + 8 * (lane >> 5));
                acc = mfma32<DT>(a, b, acc);
            }
        }
    }
    // ---- store: acc[4g + j] <-> m = 32 wm + 8 g + 4 (lane >> 5) + j, n = 32 wn + (lane & 31)
    const int n = n0 + 32 * wn + (lane & 31);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + 32 * wm + 8 * g + 4 * (lane >> 5) + j;
            if (m >= L) continue;
            const float v = acc[4 * g + j];
            if (MODE == AB_S) {            // P = exp(scale * q.k - lse) on the visible keys j <= i (the forward's softmax, re-materialised)
                const bool vis = n <= m && n < L && p.key_visible[t0 + n];
                const float pv = vis ? __expf(v * p.scale - p.lse[(t0 + m) * nh + hb]) : 0.f;
                p.P16[mat + (int64_t)hb * Lm * Lm + (int64_t)m * Lm + n] = to16<DT>(pv);
            } else if (MODE == AB_DP) {    // dS = scale * P o (dP - D),  D = rowsum(dO o O) = rowsum(P o dP)
                const int64_t at = mat + (int64_t)hb * Lm * Lm + (int64_t)m * Lm + n;
                const float pv = from16<DT>(p.P16[at]);
                p.dS16[at] = to16<DT>(p.scale * pv * (v - p.D[(t0 + m) * nh + hb]));
            } else if (MODE == AB_DQ) p.dqkv[(t0 + m) * p.ldq + hb * 128 + n] = v;
            else if (MODE == AB_DK) p.dqkv[(t0 + m) * p.ldq + (nh + kvh) * 128 + n] = v;
            else p.dqkv[(t0 + m) * p.ldq + (nh + nkv + kvh) * 128 + n] = v;
        }
}

// dV = P^T dO and dK = dS^T Q: both operands have the contraction index (the query row i) as their ROW index, so the tiles are staged
// row-major (coalesced 16-byte copies, XOR-swizzled chunks) and the MFMA fragments are read with the transposing LDS load
// (ds_read_b64_tr_b16), the scheme of the forward kernel's P.V product.  One workgroup = 128 key rows j x 128 head dims of one
// (sequence, kv head); wave w owns key rows 32 w .. 32 w + 31; the contraction runs over the group's q heads and the query rows i >= j.
template <int DT, int MODE>
__global__ __launch_bounds__(256) void attn_tn_kernel(AttnBwdParams p, int Lm) {
    __shared__ __attribute__((aligned(16))) uint16_t a_lds[32 * 128];
    __shared__ __attribute__((aligned(16))) uint16_t b_lds[32 * 128];
    const int nh = p.num_heads, nkv = p.num_kv_heads, grp = nh / nkv;
    const int s = blockIdx.z / nkv, kvh = blockIdx.z % nkv;
    const int L = p.seq_len[s];
    const int64_t t0 = p.seq_start[s];
    const int m0 = blockIdx.y * 128;
    if (m0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5;
    f32x16 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    const int64_t mat = ((int64_t)s * nh) * (int64_t)Lm * Lm;
    const int k_end = (L + 31) / 32 * 32;
    for (int hh = 0; hh < grp; ++hh) {
        const int h = kvh * grp + hh;
        const uint16_t* PS = (MODE == AB_DV ? p.P16 : p.dS16) + mat + (int64_t)h * Lm * Lm;
        for (int i0 = m0; i0 < k_end; i0 += 32) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + 256 * u;
                const int row = idx >> 4, ch = idx & 15;
                const int i = i0 + row;
                uint4 av = make_uint4(0, 0, 0, 0), bv = make_uint4(0, 0, 0, 0);
                if (i < L) {
                    if (m0 + 8 * ch + 8 <= Lm) av = *(const uint4*)(PS + (int64_t)i * Lm + m0 + 8 * ch);
                    bv = MODE == AB_DV ? *(const uint4*)(p.dout + (t0 + i) * p.ldo + h * 128 + 8 * ch) : *(const uint4*)(p.qkv + (t0 + i) * p.ldq + h * 128 + 8 * ch);
                }
                const int off = row * 128 + 8 * (ch ^ ((row & 3) << 2));
                *(uint4*)(a_lds + off) = av;
                *(uint4*)(b_lds + off) = bv;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 af = tr_frag<128>(a_lds, 32 * w, s2, lane);
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] = mfma32<DT>(af, tr_frag<128>(b_lds, 32 * db, s2, lane), o[db]);
            }
        }
    }
    const int col0 = (MODE == AB_DK ? nh + kvh : nh + nkv + kvh) * 128;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = m0 + 32 * w + 8 * g + 4 * hf + jj;
                if (m < L) p.dqkv[(t0 + m) * p.ldq + col0 + 32 * db + (lane & 31)] = o[db][4 * g + jj];
            }
}

// D[t, h] = sum_d dO[t, h, d] * O[t, h, d]  (= rowsum(P o dP)): one wave per (token, head), 4 per workgroup
template <int DT>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(float* D, const uint16_t* dout, int64_t ldo, const uint16_t* o16, int64_t ldo16, int64_t n_pairs, int nh) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_pairs) return;
    const int64_t t = i / nh; const int h = (int)(i - t * nh);
    const int lane = threadIdx.x & 63;
    const uint32_t a = *(const uint32_t*)(dout + t * ldo + h * 128 + 2 * lane), b = *(const uint32_t*)(o16 + t * ldo16 + h * 128 + 2 * lane);
    float v = from16<DT>((uint16_t)(a & 0xFFFF)) * from16<DT>((uint16_t)(b & 0xFFFF)) + from16<DT>((uint16_t)(a >> 16)) * from16<DT>((uint16_t)(b >> 16));
    v = wave_sum(v);
    if (lane == 0) D[i] = v;
}

template <int DT>
static int attention_bwd_t(const AttnBwdParams& p, int64_t n_tokens, hipStream_t s) {
    const int Lm = (int)attn_bwd_lm(p.max_len);
    const int nt = Lm / 64;
    const dim3 blk(256);
    const int64_t n_pairs = n_tokens * p.num_heads;
    hipLaunchKernelGGL(attn_rowdot_kernel<DT>, dim3((unsigned)((n_pairs + 3) / 4)), blk, 0, s, p.D, p.dout, p.ldo, p.o16, p.ldo16, n_pairs, p.num_heads);
    hipLaunchKernelGGL((attn_bgemm_kernel<DT, AB_S>), dim3(nt, nt, p.n_seqs * p.num_heads), blk, 0, s, p, Lm);
    hipLaunchKernelGGL((attn_bgemm_kernel<DT, AB_DP>), dim3(nt, nt, p.n_seqs * p.num_heads), blk, 0, s, p, Lm);
    hipLaunchKernelGGL((attn_bgemm_kernel<DT, AB_DQ>), dim3(2, nt, p.n_seqs * p.num_heads), blk, 0, s, p, Lm);
    hipLaunchKernelGGL((attn_tn_kernel<DT, AB_DV>), dim3(1, (Lm + 127) / 128, p.n_seqs * p.num_kv_heads), blk, 0, s, p, Lm);
    hipLaunchKernelGGL((attn_tn_kernel<DT, AB_DK>), dim3(1, (Lm + 127) / 128, p.n_seqs * p.num_kv_heads), blk, 0, s, p, Lm);
    LAUNCH_CHECK();
    return BLIM_OK;
}
int launch_attention_bwd(const AttnBwdParams& p, int64_t n_tokens, hipStream_t s) {
    ARG_CHECK(p.n_seqs > 0 && p.max_len > 0 && p.num_heads % p.num_kv_heads == 0 && p.ldq % 8 == 0 && p.ldo % 8 == 0 && p.ldo16 % 2 == 0);
    ARG_CHECK(p.lse && p.D && p.o16 && p.P16 && p.dS16 && n_tokens > 0);
    if ((int64_t)p.n_seqs * p.num_heads > 65535) { blim_set_error("attention backward: %d sequences x %d heads exceed the grid's z range; split the batch", p.n_seqs, p.num_heads); return BLIM_ERR_ARG; }
    if (p.dtype == DT_F16) return attention_bwd_t<DT_F16>(p, n_tokens, s);
    return attention_bwd_t<DT_BF16>(p, n_tokens, s);
}

// ---------------------------------------------------------------------------- RoPE backward
template <int DT>
__global__ void rope_bwd_kernel(uint16_t* out16, const float* dqkv, int64_t T, int qkv_n, int rope_cols, const int32_t* pos, const float* cosb, const float* sinb, int n_pos) {
    const int half = qkv_n / 2;       // one thread per (token, head, d < 64) pair: qkv_n / 2 pairs per token
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * half) return;
    const int64_t t = i / half;
    const int q = (int)(i - t * half);
    const int head = q >> 6, d = q & 63;
    const int c1 = head * 128 + d, c2 = c1 + 64;
    const float y1 = dqkv[t * qkv_n + c1], y2 = dqkv[t * qkv_n + c2];
    float x1 = y1, x2 = y2;
    if (c1 < rope_cols) {
        const int pp = min(max(pos[t], 0), n_pos - 1);
        const float c = cosb[(int64_t)pp * 64 + d], sn = sinb[(int64_t)pp * 64 + d];
        x1 = y1 * c + y2 * sn;
        x2 = y2 * c - y1 * sn;
    }
    out16[t * qkv_n + c1] = to16<DT>(x1);
    out16[t * qkv_n + c2] = to16<DT>(x2);
}
int launch_rope_bwd(uint16_t* out16, const float* dqkv, int64_t T, int qkv_n, int rope_cols, const int32_t* pos, const float* cosb, const float* sinb, int n_pos, int dtype, hipStream_t s) {
    const int64_t total = T * (qkv_n / 2);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(rope_bwd_kernel<DT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out16, dqkv, T, qkv_n, rope_cols, pos, cosb, sinb, n_pos));
    LAUNCH_CHECK();
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- optimizer
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, float inv_scale, float c1, float c2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * inv_scale;
    float pv = p[i] * (1.0f - lr * wd);
    const float mv = b1 * m[i] + (1.0f - b1) * gr;
    const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mv; v[i] = vv;
    pv -= (lr / c1) * mv / (sqrtf(vv) / sqrtf(c2) + eps);
    p[i] = pv;
}
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, float inv_scale, float c1, float c2, hipStream_t s) {
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, inv_scale, c1, c2);
    LAUNCH_CHECK();
    return BLIM_OK;
}
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* g, int64_t n, float inv_scale, float* stats, float* part) {
    __shared__ float red[4];
    float ss = 0.f; int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i] * inv_scale;
        if (!(fabsf(v) <= 3.0e38f)) bad = 1;
        ss += v * v;
    }
    ss = block_sum_256(ss, red);
    if (threadIdx.x == 0) part[blockIdx.x] = ss;
    if (bad) stats[1] = 1.0f;
}
int launch_grad_stats(const float* g, int64_t n, float inv_scale, float* stats, float* scratch, hipStream_t s) {
    ARG_CHECK(scratch);                                           // 1024 floats: one partial per workgroup, summed in a fixed order
    const int grid = (int)min((int64_t)1024, (n + 255) / 256);
    hipLaunchKernelGGL(grad_stats_kernel, dim3(grid), dim3(256), 0, s, g, n, inv_scale, stats, scratch);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(fixed_order_sum_kernel, dim3(1), dim3(256), 0, s, stats, scratch, (int64_t)grid);
    LAUNCH_CHECK();
    return BLIM_OK;
}
