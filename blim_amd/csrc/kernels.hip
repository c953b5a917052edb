// HBM-bound helper kernels (see kernels.hpp).  All loads/stores are 8-16 B per lane.
#include "kernels.hpp"
#include <stdlib.h>

#define LAUNCH_CHECK(name)                                                           \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            blim_set_error("%s launch failed: %s", name, hipGetErrorString(_e));     \
            return BLIM_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

static inline int grid_for(int64_t n, int per_block, int cap = 8192) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------- synthetic fill
// Rule stated in blim_amd/synth.py (bit-exact with the numpy statement).
__device__ __forceinline__ float bell_value(uint64_t seed, uint64_t tid, uint64_t idx, float scale, float mean) {
    const uint64_t K0 = 0x9E3779B97F4A7C15ull, K1 = 0xBF58476D1CE4E5B9ull, K2 = 0x94D049BB133111EBull;
    uint64_t z = seed * K0 + tid * K1 + idx + K0;
    z = (z ^ (z >> 30)) * K1;
    z = (z ^ (z >> 27)) * K2;
    z = z ^ (z >> 31);
    const int s = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)(z >> 48);
    return __fadd_rn(__fmul_rn((float)(s - 131070), scale), mean);  // no fma contraction: matches numpy
}

__global__ void fill_bell_bf16_kernel(bf16_t* out, int64_t n, uint64_t seed, uint64_t tid, float scale, float mean) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            uint2 pk;
            pk.x = pack_bf16x2(bell_value(seed, tid, i, scale, mean), bell_value(seed, tid, i + 1, scale, mean));
            pk.y = pack_bf16x2(bell_value(seed, tid, i + 2, scale, mean), bell_value(seed, tid, i + 3, scale, mean));
            *(uint2*)(out + i) = pk;
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = f32_to_bf16(bell_value(seed, tid, j, scale, mean));
        }
    }
}
__global__ void fill_bell_f32_kernel(float* out, int64_t n, uint64_t seed, uint64_t tid, float scale, float mean, int round_bf16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = bell_value(seed, tid, i, scale, mean);
        if (round_bf16) v = bf16_to_f32(f32_to_bf16(v));
        out[i] = v;
    }
}
int launch_fill_bell_bf16(bf16_t* out, int64_t n, uint64_t seed, uint64_t tensor_id, float scale, float mean, hipStream_t s) {
    ARG_CHECK(out && n > 0 && ((uintptr_t)out & 7) == 0);
    hipLaunchKernelGGL(fill_bell_bf16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, out, n, seed, tensor_id, scale, mean);
    LAUNCH_CHECK("fill_bell_bf16");
    return BLIM_OK;
}
int launch_fill_bell_f32(float* out, int64_t n, uint64_t seed, uint64_t tensor_id, float scale, float mean, int round_bf16, hipStream_t s) {
    ARG_CHECK(out && n > 0);
    hipLaunchKernelGGL(fill_bell_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, out, n, seed, tensor_id, scale, mean, round_bf16);
    LAUNCH_CHECK("fill_bell_f32");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- assemble (K2)
__global__ void assemble_kernel(bf16_t* out, const int32_t* src, int64_t n_tokens, int H, const bf16_t* table, const bf16_t* feats) {
    const int chunks = H / 8;  // 16-B chunks per row
    const int64_t total = n_tokens * chunks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks);
        const int32_t sidx = src[t];
        const bf16_t* row = sidx >= 0 ? table + (int64_t)sidx * H : feats + (int64_t)(-(sidx + 1)) * H;
        *(uint4*)(out + t * H + 8 * c) = *(const uint4*)(row + 8 * c);
    }
}
// compensated mode: rows are [hi | lo] of width 2H -- embedding-table rows are exact in 16 bits (lo = 0), feature rows carry both halves
__global__ void assemble_split_kernel(bf16_t* out, const int32_t* src, int64_t n_tokens, int H, const bf16_t* table, const bf16_t* feats) {
    const int chunks = H / 4;  // 16-B chunks per [hi | lo] row of 2H
    const int64_t total = n_tokens * chunks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks);
        const int32_t sidx = src[t];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (sidx >= 0) { if (8 * c < H) v = *(const uint4*)(table + (int64_t)sidx * H + 8 * c); }
        else v = *(const uint4*)(feats + (int64_t)(-(sidx + 1)) * 2 * H + 8 * c);
        *(uint4*)(out + t * 2 * H + 8 * c) = v;
    }
}
int launch_assemble(bf16_t* out, const int32_t* src_index, int64_t n_tokens, int H, const bf16_t* table, const bf16_t* feats, hipStream_t s, bool split) {
    ARG_CHECK(out && src_index && n_tokens > 0 && H % 8 == 0 && table);
    if (split) {
        hipLaunchKernelGGL(assemble_split_kernel, dim3(grid_for(n_tokens * (H / 4), 256)), dim3(256), 0, s, out, src_index, n_tokens, H, table, feats);
        LAUNCH_CHECK("assemble");
        return BLIM_OK;
    }
    hipLaunchKernelGGL(assemble_kernel, dim3(grid_for(n_tokens * (H / 8), 256)), dim3(256), 0, s, out, src_index, n_tokens, H, table, feats);
    LAUNCH_CHECK("assemble");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- dtype converts
template <int DT>
__global__ void h16_to_f32_kernel(float* out, const bf16_t* in, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const uint2 v = *(const uint2*)(in + i);
            *(float4*)(out + i) = make_float4(from16<DT>((uint16_t)(v.x & 0xFFFF)), from16<DT>((uint16_t)(v.x >> 16)),
                                              from16<DT>((uint16_t)(v.y & 0xFFFF)), from16<DT>((uint16_t)(v.y >> 16)));
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = from16<DT>(in[j]);
        }
    }
}
__global__ void f32_to_bf16_kernel(bf16_t* out, const float* in, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(in + i);
            *(uint2*)(out + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = f32_to_bf16(in[j]);
        }
    }
}
// resid[t, c] = f32(hi[t, c]) + f32(lo[t, c]) for [hi | lo] rows of width 2H (compensated mode)
template <int DT>
__global__ void hilo_to_f32_kernel(float* out, const bf16_t* in, int64_t n_rows, int H) {
    const int64_t total = n_rows * H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / H;
        const int c = (int)(i - t * H);
        out[i] = from16<DT>(in[t * 2 * H + c]) + from16<DT>(in[t * 2 * H + H + c]);
    }
}
int launch_hilo_to_f32(float* out, const bf16_t* in, int64_t n_rows, int H, int dtype, hipStream_t s) {
    ARG_CHECK(out && in && n_rows > 0 && H > 0);
    if (dtype == DT_F16) hipLaunchKernelGGL(hilo_to_f32_kernel<DT_F16>, dim3(grid_for(n_rows * H, 256)), dim3(256), 0, s, out, in, n_rows, H);
    else hipLaunchKernelGGL(hilo_to_f32_kernel<DT_BF16>, dim3(grid_for(n_rows * H, 256)), dim3(256), 0, s, out, in, n_rows, H);
    LAUNCH_CHECK("hilo_to_f32");
    return BLIM_OK;
}
int launch_h16_to_f32(float* out, const bf16_t* in, int64_t n, int dtype, hipStream_t s) {
    ARG_CHECK(out && in && n > 0);
    if (dtype == DT_F16) hipLaunchKernelGGL(h16_to_f32_kernel<DT_F16>, dim3(grid_for(n, 1024)), dim3(256), 0, s, out, in, n);
    else hipLaunchKernelGGL(h16_to_f32_kernel<DT_BF16>, dim3(grid_for(n, 1024)), dim3(256), 0, s, out, in, n);
    LAUNCH_CHECK("h16_to_f32");
    return BLIM_OK;
}
int launch_f32_to_bf16(bf16_t* out, const float* in, int64_t n, hipStream_t s) {
    ARG_CHECK(out && in && n > 0);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, out, in, n);
    LAUNCH_CHECK("f32_to_bf16");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- row gather (last-layer pruning, engine.hip run_layers)
// dst[r] = src[rows[r]] in 16-byte chunks; an index outside [0, n_src) gives a row of `fill` words (NaN bits for the f32 residual stream:
// the row's score is poisoned instead of wild memory being read, as rmsnorm_kernel does for its gathered rows)
__global__ void gather_rows_kernel(uint4* dst, const uint4* src, const int32_t* rows, int64_t n_rows, int chunks, int64_t n_src, uint32_t fill) {
    const int64_t total = n_rows * chunks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        const int64_t sr = rows[r];
        dst[i] = (sr >= 0 && sr < n_src) ? src[sr * chunks + c] : make_uint4(fill, fill, fill, fill);
    }
}
int launch_gather_rows(void* dst, const void* src, const int32_t* rows, int64_t n_rows, int64_t row_bytes, int64_t n_src, uint32_t fill, hipStream_t s) {
    ARG_CHECK(dst && src && rows && n_rows > 0 && row_bytes % 16 == 0);
    const int chunks = (int)(row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_rows * chunks, 256)), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, rows, n_rows, chunks, n_src, fill);
    LAUNCH_CHECK("gather_rows");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- RMSNorm (K3/K9)
// One wave per row; the row (H f32) is read once in float4 pieces and kept in registers when H <= 64*4*16.
template <int MAXV, int DT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, int64_t ldx, const int32_t* rows, int64_t n_rows, int H,
                                                      const float* w, float eps, bf16_t* out_bf16, float* out_f32, int64_t n_src,
                                                      int64_t ldo, bf16_t* out_lo, int saturate) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    if constexpr (DT == DT_F16) { if (saturate) f16_saturate_on(); }
    const int64_t src = rows ? rows[r] : r;
    const int nv = H / 4;  // float4 per row
    if (src < 0 || src >= n_src) {   // a gather index outside the packed batch: poison the row (NaN score) instead of reading wild memory
        const float qnan = __builtin_nanf("");
        for (int c = lane; c < nv; c += 64) {
            if (out_bf16) *(uint2*)(out_bf16 + r * ldo + 4 * c) = make_uint2(pack2<DT>(qnan, qnan), pack2<DT>(qnan, qnan));
            if (out_lo) *(uint2*)(out_lo + r * ldo + 4 * c) = make_uint2(0u, 0u);
            if (out_f32) *(float4*)(out_f32 + r * H + 4 * c) = make_float4(qnan, qnan, qnan, qnan);
        }
        return;
    }
    const float* xr = x + src * ldx;
    float4 v[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = *(const float4*)(xr + 4 * c);
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = *(const float4*)(w + 4 * c);
            const float o0 = g.x * (v[i].x * inv), o1 = g.y * (v[i].y * inv), o2 = g.z * (v[i].z * inv), o3 = g.w * (v[i].w * inv);
            if (out_bf16) *(uint2*)(out_bf16 + r * ldo + 4 * c) = make_uint2(pack2<DT>(o0, o1), pack2<DT>(o2, o3));
            if (out_lo)      // compensated mode: lo = 16-bit(x - f32(hi)), stored beside hi ([hi | lo] along K of the consuming GEMM)
                *(uint2*)(out_lo + r * ldo + 4 * c) = make_uint2(pack2<DT>(o0 - from16<DT>(to16<DT>(o0)), o1 - from16<DT>(to16<DT>(o1))),
                                                                 pack2<DT>(o2 - from16<DT>(to16<DT>(o2)), o3 - from16<DT>(to16<DT>(o3))));
            if (out_f32) *(float4*)(out_f32 + r * H + 4 * c) = make_float4(o0, o1, o2, o3);
        }
    }
}
// Wide form (H % 8 == 0, H <= 4096): a lane owns chunks of EIGHT consecutive elements, so the 16-bit output leaves as 16-byte stores (the form above stores
// 8 bytes per lane: 512 B per wave-instruction) and the row is read as two adjacent float4 per chunk.  Same arithmetic per element; only the order in which the
// lanes' partial sums of squares are formed differs.
template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_wide_kernel(const float* x, int64_t ldx, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps,
                                                           bf16_t* out_bf16, float* out_f32, int64_t n_src, int64_t ldo, bf16_t* out_lo, int saturate, uint8_t* out6) {
    // out6: the four rows' lo parts are staged in LDS as 16-bit values and leave as e2m3 operand tiles (below); the grid then covers the rows up to the next
    // multiple of 256 (zero rows of the last tile)
    extern __shared__ __attribute__((aligned(16))) char lo_s[];          // [4][H + 8] 16-bit values when out6
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nc = H / 8;
    const int64_t lrs = (int64_t)(H + 8) * 2;                            // staged row stride (bytes)
    char* lrow = lo_s + (threadIdx.x >> 6) * lrs;
    auto tiles = [&]() __attribute__((always_inline)) {
        __syncthreads();
        const int nblk = H / 32, nk6 = H / 128;
        for (int item = threadIdx.x; item < 4 * nblk; item += 256) {
            const int rw = item & 3, blk = item >> 2;
            const int64_t row = (int64_t)blockIdx.x * 4 + rw;
            float f[32];
            const char* src = lo_s + rw * lrs + blk * 64;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const uint4 v = *(const uint4*)(src + 16 * c4);
                const uint16_t* e16 = (const uint16_t*)&v;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[8 * c4 + j] = from16<DT>(e16[j]);
            }
            const F6Block q = e2m3_block(f);
            const int rl = (int)(row & 255), fb = rl >> 4, rr = rl & 15, g = blk & 3;
            uint8_t* t = out6 + ((row >> 8) * nk6 + (blk >> 2)) * F6_TILE_BYTES;
            *(uint4*)(t + fb * 1536 + g * 256 + rr * 16) = make_uint4(q.d[0], q.d[1], q.d[2], q.d[3]);
            *(uint2*)(t + fb * 1536 + 1024 + g * 128 + rr * 8) = make_uint2(q.d[4], q.d[5]);
            t[24576 + ((rl >> 7) * 4 + g) * 128 + (rl & 15) * 8 + ((rl >> 4) & 7)] = (uint8_t)q.e8;
        }
    };
    if (r >= n_rows) {
        if (out6) { for (int c = lane; c < nc; c += 64) *(uint4*)(lrow + 16 * c) = make_uint4(0u, 0u, 0u, 0u); tiles(); }
        return;
    }
    if constexpr (DT == DT_F16) { if (saturate) f16_saturate_on(); }
    const int64_t src = rows ? rows[r] : r;
    if (src < 0 || src >= n_src) {
        const float qnan = __builtin_nanf("");
        const uint32_t q2 = pack2<DT>(qnan, qnan);
        for (int c = lane; c < nc; c += 64) {
            if (out_bf16) *(uint4*)(out_bf16 + r * ldo + 8 * c) = make_uint4(q2, q2, q2, q2);
            if (out_lo) *(uint4*)(out_lo + r * ldo + 8 * c) = make_uint4(0u, 0u, 0u, 0u);
            if (out_f32) { *(float4*)(out_f32 + r * H + 8 * c) = make_float4(qnan, qnan, qnan, qnan); *(float4*)(out_f32 + r * H + 8 * c + 4) = make_float4(qnan, qnan, qnan, qnan); }
            if (out6) *(uint4*)(lrow + 16 * c) = make_uint4(0u, 0u, 0u, 0u);
        }
        if (out6) tiles();
        return;
    }
    const float* xr = x + src * ldx;
    float4 v[8][2];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            v[i][0] = *(const float4*)(xr + 8 * c); v[i][1] = *(const float4*)(xr + 8 * c + 4);
            ss += v[i][0].x * v[i][0].x + v[i][0].y * v[i][0].y + v[i][0].z * v[i][0].z + v[i][0].w * v[i][0].w;
            ss += v[i][1].x * v[i][1].x + v[i][1].y * v[i][1].y + v[i][1].z * v[i][1].z + v[i][1].w * v[i][1].w;
        }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            const float4 g0 = *(const float4*)(w + 8 * c), g1 = *(const float4*)(w + 8 * c + 4);
            const float o[8] = {g0.x * (v[i][0].x * inv), g0.y * (v[i][0].y * inv), g0.z * (v[i][0].z * inv), g0.w * (v[i][0].w * inv),
                                g1.x * (v[i][1].x * inv), g1.y * (v[i][1].y * inv), g1.z * (v[i][1].z * inv), g1.w * (v[i][1].w * inv)};
            if (out_bf16) *(uint4*)(out_bf16 + r * ldo + 8 * c) = make_uint4(pack2<DT>(o[0], o[1]), pack2<DT>(o[2], o[3]), pack2<DT>(o[4], o[5]), pack2<DT>(o[6], o[7]));
            if (out_lo || out6) {
                float d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = o[j] - from16<DT>(to16<DT>(o[j]));
                const uint4 dv = make_uint4(pack2<DT>(d[0], d[1]), pack2<DT>(d[2], d[3]), pack2<DT>(d[4], d[5]), pack2<DT>(d[6], d[7]));
                if (out_lo) *(uint4*)(out_lo + r * ldo + 8 * c) = dv;
                if (out6) *(uint4*)(lrow + 16 * c) = dv;
            }
            if (out_f32) { *(float4*)(out_f32 + r * H + 8 * c) = make_float4(o[0], o[1], o[2], o[3]); *(float4*)(out_f32 + r * H + 8 * c + 4) = make_float4(o[4], o[5], o[6], o[7]); }
        }
    }
    if (out6) tiles();
}
static int g_rmsnorm_wide = getenv("BLIM_RMSNORM_WIDE") ? atoi(getenv("BLIM_RMSNORM_WIDE")) : 1;
bool rmsnorm_can_write_tiles(int H, int64_t ldx, int64_t ldo) {
    if (ldo == 0) ldo = H;
    return g_rmsnorm_wide && H % 128 == 0 && H <= 4096 && H > 256 && ldo % 8 == 0 && ldx % 4 == 0;
}
int launch_rmsnorm(const float* x, int64_t ldx, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps,
                   bf16_t* out_h16, int dtype, float* out_f32, hipStream_t s, int64_t n_src, int64_t ldo, bf16_t* out_lo, bool saturate, uint8_t* out6) {
    const int sat = saturate ? 1 : 0;
    if (!rows) n_src = n_rows;
    if (ldo == 0) ldo = H;
    ARG_CHECK(ldo % 4 == 0 && (!out_lo || out_h16));
    ARG_CHECK(x && w && n_rows > 0 && H % 4 == 0 && ldx % 4 == 0 && (out_h16 || out_f32));
    ARG_CHECK(!out6 || (out_h16 && rmsnorm_can_write_tiles(H, ldx, ldo)));
    const int nv = H / 4;
    const dim3 grid((unsigned)((n_rows + 3) / 4));
    if (g_rmsnorm_wide && H % 8 == 0 && H <= 4096 && H > 256 && ldo % 8 == 0 && ldx % 4 == 0) {
        const dim3 grid_w(out6 ? (unsigned)(((n_rows + 255) / 256) * 64) : grid.x);
        const size_t lds = out6 ? (size_t)4 * (H + 8) * 2 : 0;
        if (dtype == DT_F16) hipLaunchKernelGGL((rmsnorm_wide_kernel<DT_F16>), grid_w, dim3(256), lds, s, x, ldx, rows, n_rows, H, w, eps, out_h16, out_f32, n_src, ldo, out_lo, sat, out6);
        else hipLaunchKernelGGL((rmsnorm_wide_kernel<DT_BF16>), grid_w, dim3(256), lds, s, x, ldx, rows, n_rows, H, w, eps, out_h16, out_f32, n_src, ldo, out_lo, sat, out6);
        LAUNCH_CHECK("rmsnorm");
        return BLIM_OK;
    }
#define RMS_LAUNCH(MV)                                                                                                              \
    do {                                                                                                                            \
        if (dtype == DT_F16) hipLaunchKernelGGL((rmsnorm_kernel<MV, DT_F16>), grid, dim3(256), 0, s, x, ldx, rows, n_rows, H, w, eps, out_h16, out_f32, n_src, ldo, out_lo, sat); \
        else hipLaunchKernelGGL((rmsnorm_kernel<MV, DT_BF16>), grid, dim3(256), 0, s, x, ldx, rows, n_rows, H, w, eps, out_h16, out_f32, n_src, ldo, out_lo, sat);                \
    } while (0)
    if (nv <= 64 * 4) RMS_LAUNCH(4);
    else if (nv <= 64 * 16) RMS_LAUNCH(16);
    else {
        blim_set_error("rmsnorm: hidden size %d > 4096 not supported", H);
        return BLIM_ERR_ARG;
    }
#undef RMS_LAUNCH
    LAUNCH_CHECK("rmsnorm");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- group mean (TVG clip tokens)
template <int DT>
__global__ void group_mean_kernel(bf16_t* out, const bf16_t* in, int64_t n_out, int group, int H) {
    const int chunks = H / 4;
    const int64_t total = n_out * chunks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t o = i / chunks;
        const int c = (int)(i - o * chunks);
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int j = 0; j < group; ++j) {
            const uint2 v = *(const uint2*)(in + (o * group + j) * H + 4 * c);
            a0 += from16<DT>((uint16_t)(v.x & 0xFFFF)); a1 += from16<DT>((uint16_t)(v.x >> 16));
            a2 += from16<DT>((uint16_t)(v.y & 0xFFFF)); a3 += from16<DT>((uint16_t)(v.y >> 16));
        }
        const float inv = 1.0f / (float)group;
        *(uint2*)(out + o * H + 4 * c) = make_uint2(pack2<DT>(a0 * inv, a1 * inv), pack2<DT>(a2 * inv, a3 * inv));
    }
}
// compensated mode: in [n_out * group, 2H] rows of [hi | lo] -> out [n_out, 2H]: the f32 mean of hi + lo, split again
template <int DT>
__global__ void group_mean_split_kernel(bf16_t* out, const bf16_t* in, int64_t n_out, int group, int H) {
    const int64_t total = n_out * H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t o = i / H;
        const int c = (int)(i - o * H);
        float a = 0.f;
        for (int j = 0; j < group; ++j) { const bf16_t* r = in + (o * group + j) * 2 * H; a += from16<DT>(r[c]) + from16<DT>(r[H + c]); }
        a *= 1.0f / (float)group;
        const uint16_t hi = to16<DT>(a);
        out[o * 2 * H + c] = hi;
        out[o * 2 * H + H + c] = to16<DT>(a - from16<DT>(hi));
    }
}
int launch_group_mean(bf16_t* out, const bf16_t* in, int64_t n_out, int group, int H, int dtype, hipStream_t s, bool split) {
    ARG_CHECK(out && in && n_out > 0 && group > 0 && H % 4 == 0);
    if (split) {
        if (dtype == DT_F16) hipLaunchKernelGGL(group_mean_split_kernel<DT_F16>, dim3(grid_for(n_out * H, 256)), dim3(256), 0, s, out, in, n_out, group, H);
        else hipLaunchKernelGGL(group_mean_split_kernel<DT_BF16>, dim3(grid_for(n_out * H, 256)), dim3(256), 0, s, out, in, n_out, group, H);
        LAUNCH_CHECK("group_mean");
        return BLIM_OK;
    }
    if (dtype == DT_F16) hipLaunchKernelGGL(group_mean_kernel<DT_F16>, dim3(grid_for(n_out * (H / 4), 256)), dim3(256), 0, s, out, in, n_out, group, H);
    else hipLaunchKernelGGL(group_mean_kernel<DT_BF16>, dim3(grid_for(n_out * (H / 4), 256)), dim3(256), 0, s, out, in, n_out, group, H);
    LAUNCH_CHECK("group_mean");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- LSE combine (K11)
// One wave per row: combine the per-tile (max, sumexp) partials.
__global__ __launch_bounds__(256) void lse_combine_kernel(const float2* part, int n_tiles, const float* label_logit, const int32_t* labels, int64_t n_rows, float* logprob) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float2* pr = part + r * n_tiles;
    float mx = -INFINITY;
    for (int i = lane; i < n_tiles; i += 64) mx = fmaxf(mx, pr[i].x);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int i = lane; i < n_tiles; i += 64) {
        const float2 v = pr[i];
        if (v.x > -INFINITY) sm += v.y * expf(v.x - mx);
    }
    sm = wave_sum(sm);
    if (lane == 0) logprob[r] = labels[r] < 0 ? 0.f : label_logit[r] - (mx + logf(sm));
}
int launch_lse_combine(const float2* part, int n_tiles, const float* label_logit, const int32_t* labels, int64_t n_rows, float* logprob, hipStream_t s) {
    ARG_CHECK(part && label_logit && labels && logprob && n_rows > 0 && n_tiles > 0);
    hipLaunchKernelGGL(lse_combine_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, part, n_tiles, label_logit, labels, n_rows, logprob);
    LAUNCH_CHECK("lse_combine");
    return BLIM_OK;
}

// score = sum / count_nonzero  (retrieval_utils.py:32, sign already folded: logprob = -loss)
__global__ void segment_mean_kernel(const float* logprob, const int32_t* row_start, int n_pairs, int mode, float* score) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    float sm = 0.f;
    int nz = 0;
    for (int r = row_start[p]; r < row_start[p + 1]; ++r) {
        const float v = logprob[r];
        sm += v;
        nz += (v != 0.f);
    }
    score[p] = sm / (float)(mode == 0 ? nz : row_start[p + 1] - row_start[p]);
}
int launch_segment_mean(const float* logprob, const int32_t* row_start, int n_pairs, int mode, float* score, hipStream_t s) {
    ARG_CHECK(logprob && row_start && score && n_pairs > 0);
    hipLaunchKernelGGL(segment_mean_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, s, logprob, row_start, n_pairs, mode, score);
    LAUNCH_CHECK("segment_mean");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- CE on materialised logits
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* logits, int64_t ld, int V, const int32_t* labels, int64_t n_rows, float* logprob) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const int lab = labels[r];
    if (lab < 0) {
        if (threadIdx.x == 0) logprob[r] = 0.f;
        return;
    }
    const float* row = logits + r * ld;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sm = 0.f;
    for (int i = tid; i < V; i += 256) sm += expf(row[i] - mx);
    sm = wave_sum(sm);
    if (lane == 0) red[4 + w] = sm;
    __syncthreads();
    if (tid == 0) logprob[r] = row[lab] - (mx + logf(red[4] + red[5] + red[6] + red[7]));
}
int launch_ce_rows(const float* logits, int64_t ld, int V, const int32_t* labels, int64_t n_rows, float* logprob, hipStream_t s) {
    ARG_CHECK(logits && labels && logprob && n_rows > 0 && V > 0);
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, s, logits, ld, V, labels, n_rows, logprob);
    LAUNCH_CHECK("ce_rows");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- TVG criterion (K15)
// One workgroup per pair; each of its 4 waves takes clips round-robin.
__global__ __launch_bounds__(256) void tvg_score_kernel(const float* logits, int64_t ld, int n_vocab, const int32_t* labels, int n_pairs, int clips, float* score) {
    __shared__ float part[4];
    const int p = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lab = labels[p];
    float acc = 0.f;
    for (int c = w; c < clips; c += 4) {
        const float* row = logits + ((int64_t)p * clips + c) * ld;
        float mx = -INFINITY;
        for (int i = lane; i < n_vocab; i += 64) mx = fmaxf(mx, row[i]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int i = lane; i < n_vocab; i += 64) sm += expf(row[i] - mx);
        sm = wave_sum(sm);
        acc += row[lab] - (mx + logf(sm));
    }
    if (lane == 0) part[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) score[p] = (part[0] + part[1] + part[2] + part[3]) / (float)clips;
}
int launch_tvg_score(const float* logits, int64_t ld, int n_vocab, const int32_t* labels, int n_pairs, int clips, float* score, hipStream_t s) {
    ARG_CHECK(logits && labels && score && n_pairs > 0 && clips > 0 && n_vocab > 0);
    hipLaunchKernelGGL(tvg_score_kernel, dim3(n_pairs), dim3(256), 0, s, logits, ld, n_vocab, labels, n_pairs, clips, score);
    LAUNCH_CHECK("tvg_score");
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- fp8 quantisers (DT_F8 mode)
template <int DT>
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* in, int64_t ld, int64_t n_rows, int K, uint8_t* out8, float* scale) {
    constexpr int MAXC = 10;                      // 8-element chunks per thread: K <= 256 * 8 * 10
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const bf16_t* row = in + r * ld;
    const int nchunk = K / 8;
    uint4 v[MAXC];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nchunk) {
            v[i] = *(const uint4*)(row + 8 * c);
            const uint16_t* e = (const uint16_t*)&v[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(from16<DT>(e[j])));
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = mx > 0.f ? mx / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    if (threadIdx.x == 0) scale[r] = sc;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nchunk) {
            const uint16_t* e = (const uint16_t*)&v[i];
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = from16<DT>(e[j]) * inv;
            *(uint2*)(out8 + r * K + 8 * c) = make_uint2(pack_fp8x4(f[0], f[1], f[2], f[3]), pack_fp8x4(f[4], f[5], f[6], f[7]));
        }
    }
}
int launch_quant_rows(const bf16_t* in, int64_t ld, int64_t n_rows, int K, int dtype, uint8_t* out8, float* scale, hipStream_t s) {
    ARG_CHECK(in && out8 && scale && n_rows > 0 && K > 0 && K % 8 == 0 && ld % 8 == 0 && K <= 256 * 8 * 10);
    if (dtype == DT_BF16) hipLaunchKernelGGL((quant_rows_kernel<DT_BF16>), dim3((unsigned)n_rows), dim3(256), 0, s, in, ld, n_rows, K, out8, scale);
    else hipLaunchKernelGGL((quant_rows_kernel<DT_F16>), dim3((unsigned)n_rows), dim3(256), 0, s, in, ld, n_rows, K, out8, scale);
    LAUNCH_CHECK("quant_rows");
    return BLIM_OK;
}

template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_f8_kernel(const float* x, int64_t ldx, int64_t n_rows, int H, const float* w, float eps, uint8_t* out8, float* scale) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float* xr = x + r * ldx;
    const int nv = H / 4;
    float4 v[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = *(const float4*)(xr + 4 * c);
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)H + eps);
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = *(const float4*)(w + 4 * c);
            v[i] = make_float4(g.x * (v[i].x * inv), g.y * (v[i].y * inv), g.z * (v[i].z * inv), g.w * (v[i].w * inv));
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        }
    }
    mx = wave_max(mx);
    const float sc = mx > 0.f ? mx / FP8_MAX : 1.0f;
    const float qi = 1.0f / sc;
    if (lane == 0) scale[r] = sc;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) *(uint32_t*)(out8 + r * H + 4 * c) = pack_fp8x4(v[i].x * qi, v[i].y * qi, v[i].z * qi, v[i].w * qi);
    }
}
int launch_rmsnorm_f8(const float* x, int64_t ldx, int64_t n_rows, int H, const float* w, float eps, uint8_t* out8, float* scale, hipStream_t s) {
    ARG_CHECK(x && w && out8 && scale && n_rows > 0 && H % 4 == 0 && ldx % 4 == 0);
    const int nv = H / 4;
    const dim3 grid((unsigned)((n_rows + 3) / 4));
    if (nv <= 64 * 4) hipLaunchKernelGGL((rmsnorm_f8_kernel<4>), grid, dim3(256), 0, s, x, ldx, n_rows, H, w, eps, out8, scale);
    else if (nv <= 64 * 16) hipLaunchKernelGGL((rmsnorm_f8_kernel<16>), grid, dim3(256), 0, s, x, ldx, n_rows, H, w, eps, out8, scale);
    else { blim_set_error("rmsnorm_f8: hidden size %d > 4096 not supported", H); return BLIM_ERR_ARG; }
    LAUNCH_CHECK("rmsnorm_f8");
    return BLIM_OK;
}

// rows of a 16-bit matrix whose flag byte is zero are cleared (engine option "masked_query_zero": the attention output of masked query positions)
__global__ __launch_bounds__(256) void zero_rows_kernel(uint16_t* x, int64_t ld, const uint8_t* keep, int64_t n_rows, int width) {
    const int64_t r = blockIdx.x;
    if (keep[r]) return;
    uint4* row = (uint4*)(x + r * ld);
    for (int c = threadIdx.x; c < width / 8; c += 256) row[c] = make_uint4(0u, 0u, 0u, 0u);
}
int launch_zero_rows(bf16_t* x, int64_t ld, const uint8_t* keep, int64_t n_rows, int width, hipStream_t s) {
    ARG_CHECK(x && keep && n_rows > 0 && width > 0 && width % 8 == 0 && ld % 8 == 0 && ld >= width);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, s, x, ld, keep, n_rows, width);
    LAUNCH_CHECK("zero_rows");
    return BLIM_OK;
}
// ---------------------------------------------------------------------------- lo6 quantiser (kernels.hpp; image: gemm.hpp A6 / W6)
template <int DT>
__device__ __forceinline__ void load32(const bf16_t* src, float (&f)[32]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = *(const uint4*)(src + 8 * q);
        const uint16_t* e = (const uint16_t*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[8 * q + j] = from16<DT>(e[j]);
    }
}
// One workgroup = one 16-row fragment group; thread = (row r of the group, one of 16 consecutive 32-value blocks), looping over K: the 16 lanes of a block write 256
// (resp. 128) consecutive bytes of the tile image.
template <int DT, bool W_SIDE>
__global__ __launch_bounds__(256) void f6_tiles_kernel(const bf16_t* in, int64_t ld, int64_t n_rows, int K, uint8_t* out) {
    const int r = threadIdx.x & 15, bl = threadIdx.x >> 4;
    const int64_t fbg = blockIdx.x, row = fbg * 16 + r, tile = fbg >> 4;
    const int fb = (int)(fbg & 15), rl = fb * 16 + r, nk6 = K / 128, nblk = K / 32;
    for (int b0 = 0; b0 < nblk; b0 += 16) {
        const int blk = b0 + bl;
        if (blk >= nblk) break;
        F6Block q;
        if (row < n_rows) { float f[32]; load32<DT>(in + row * ld + 32 * blk, f); q = e2m3_block(f); }
        else { q.d[0] = q.d[1] = q.d[2] = q.d[3] = q.d[4] = q.d[5] = 0u; q.e8 = 0u; }
        const int st = blk >> 2, g = blk & 3;
        uint8_t* t = out + ((int64_t)tile * nk6 + st) * F6_TILE_BYTES;
        *(uint4*)(t + fb * 1536 + g * 256 + r * 16) = make_uint4(q.d[0], q.d[1], q.d[2], q.d[3]);
        *(uint2*)(t + fb * 1536 + 1024 + g * 128 + r * 8) = make_uint2(q.d[4], q.d[5]);
        const int sidx = W_SIDE ? ((rl >> 6) * 4 + g) * 64 + (rl & 15) * 4 + ((rl >> 4) & 3) : ((rl >> 7) * 4 + g) * 128 + (rl & 15) * 8 + ((rl >> 4) & 7);
        t[24576 + sidx] = (uint8_t)q.e8;
    }
}
int launch_f6_tiles(const bf16_t* in, int64_t ld, int64_t n_rows, int K, int dtype, bool w_side, uint8_t* out, hipStream_t s) {
    ARG_CHECK(in && out && n_rows > 0 && K > 0 && K % 128 == 0 && ld % 8 == 0 && ld >= K);
    const dim3 grid((unsigned)((n_rows + 255) / 256 * 16));
    if (dtype == DT_BF16) { if (w_side) hipLaunchKernelGGL((f6_tiles_kernel<DT_BF16, true>), grid, dim3(256), 0, s, in, ld, n_rows, K, out); else hipLaunchKernelGGL((f6_tiles_kernel<DT_BF16, false>), grid, dim3(256), 0, s, in, ld, n_rows, K, out); }
    else { if (w_side) hipLaunchKernelGGL((f6_tiles_kernel<DT_F16, true>), grid, dim3(256), 0, s, in, ld, n_rows, K, out); else hipLaunchKernelGGL((f6_tiles_kernel<DT_F16, false>), grid, dim3(256), 0, s, in, ld, n_rows, K, out); }
    LAUNCH_CHECK("f6_tiles");
    return BLIM_OK;
}
