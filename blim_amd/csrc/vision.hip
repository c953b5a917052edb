// Offline feature extraction on MI355X: UMT-L vision encoder + ToMe token merging behind the C ABI
// blim_vision_* (include/blim.h).  Replaces what extract.py:96-110 runs per video:
//   frames [16, 3, S, S] -> 4 clips x 4 frames -> ViT-L/16 (23 blocks; vision_tower_builder.py:272-433)
//                        -> bipartite soft matching 4*(S/16)^2 -> 64 tokens per clip (mm_projector_builder.py:6-130) -> [4, 64, 1024]
// The linear layers run on the engine's MFMA GEMM (gemm.hip: bias / GELU / residual epilogues); this file adds the patch gather,
// LayerNorm, a non-causal head_dim-64 flash attention, and ToMe in f32 (metric, best-match search, a bitonic argsort and a
// deterministic merge: HBM-bound index work, no MFMA).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/blim.h"
#include "common.hpp"
#include "gemm.hpp"
#include "kernels.hpp"

#define TRY(expr)                       \
    do {                                \
        int _rc = (expr);               \
        if (_rc != BLIM_OK) return _rc; \
    } while (0)
#define KCHECK(name)                                                                  \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess) {                                                       \
            blim_set_error("%s launch failed: %s", name, hipGetErrorString(_e));      \
            return BLIM_ERR_HIP;                                                      \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------------------- small kernels
// f32 / bf16 source -> 16-bit of the compute dtype (weights)
template <bool SRC_F32, int DT>
__global__ void to_h16_kernel(bf16_t* dst, const void* src, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = SRC_F32 ? ((const float*)src)[i] : bf16_to_f32(((const bf16_t*)src)[i]);
        dst[i] = to16<DT>(v);
    }
}
template <bool SRC_F32>
__global__ void to_f32_kernel(float* dst, const void* src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = SRC_F32 ? ((const float*)src)[i] : bf16_to_f32(((const bf16_t*)src)[i]);
}

// patches[row, col]: row = ((clip * T + t) * G + gy) * G + gx, col = (c * P + ky) * P + kx  <-  frames[clip, t, c, gy*P+ky, gx*P+kx]
// (Conv3d(3, D, kernel (1,P,P), stride the same) + flatten(2).transpose(1,2): vision_tower_builder.py:170-184).  8 kx per thread.
__global__ void patchify_kernel(bf16_t* out, const bf16_t* frames, int64_t n_rows, int T, int G, int P, int S) {
    const int cols = 3 * P * P, chunks = cols / 8;
    const int64_t total = n_rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunks;
        const int col = (int)(i - row * chunks) * 8;
        const int gx = (int)(row % G), gy = (int)((row / G) % G);
        const int64_t ft = row / ((int64_t)G * G);                     // clip * T + t
        const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
        const bf16_t* src = frames + ((ft * 3 + c) * S + (gy * P + ky)) * (int64_t)S + gx * P + kx;
        *(uint4*)(out + row * cols + col) = *(const uint4*)src;
    }
}
// resid[tok, :] = pos[tok % L, :] + bias   (the patch GEMM then accumulates into it: x = conv(frames) + bias + pos_embed, :353)
__global__ void init_resid_kernel(float* resid, const float* pos, const float* bias, int64_t n_tok, int L, int D) {
    const int chunks = D / 4;
    const int64_t total = n_tok * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks) * 4;
        const float4 p = *(const float4*)(pos + (t % L) * D + c), b = *(const float4*)(bias + c);
        *(float4*)(resid + t * D + c) = make_float4(p.x + b.x, p.y + b.y, p.z + b.z, p.w + b.w);
    }
}
// LayerNorm (nn.LayerNorm, biased variance), one wave per row of D <= 4096 f32; 16-bit and / or f32 output
template <int DT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int64_t n_rows, int D, const float* w, const float* b, float eps,
                                                        bf16_t* out16, float* out32) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int nv = D / 4;
    float4 v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) { v[i] = *(const float4*)(x + r * D + 4 * c); s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) { const float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu; q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3; }
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = *(const float4*)(w + 4 * c), h = *(const float4*)(b + 4 * c);
            const float o0 = (v[i].x - mu) * inv * g.x + h.x, o1 = (v[i].y - mu) * inv * g.y + h.y, o2 = (v[i].z - mu) * inv * g.z + h.z, o3 = (v[i].w - mu) * inv * g.w + h.w;
            if (out16) *(uint2*)(out16 + r * D + 4 * c) = make_uint2(pack2<DT>(o0, o1), pack2<DT>(o2, o3));
            if (out32) *(float4*)(out32 + r * D + 4 * c) = make_float4(o0, o1, o2, o3);
        }
    }
}

// Wide form for the encoder's inner norms (16-bit output only, D % 8 == 0, D <= 4096): a lane owns chunks of EIGHT consecutive elements, so the output leaves as
// 16-byte stores (kernels.hip: rmsnorm_wide_kernel, same reason).  Same arithmetic per element; only the order of the lanes' partial sums differs.
template <int DT>
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const float* x, int64_t n_rows, int D, const float* w, const float* b, float eps, bf16_t* out16) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int nc = D / 8;
    const float* xr = x + r * D;
    float4 v[8][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            v[i][0] = *(const float4*)(xr + 8 * c); v[i][1] = *(const float4*)(xr + 8 * c + 4);
            s += (v[i][0].x + v[i][0].y + v[i][0].z + v[i][0].w) + (v[i][1].x + v[i][1].y + v[i][1].z + v[i][1].w);
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a0 = v[i][h].x - mu, a1 = v[i][h].y - mu, a2 = v[i][h].z - mu, a3 = v[i][h].w - mu;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
        }
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 g = *(const float4*)(w + 8 * c + 4 * h), t = *(const float4*)(b + 8 * c + 4 * h);
                o[2 * h] = pack2<DT>((v[i][h].x - mu) * inv * g.x + t.x, (v[i][h].y - mu) * inv * g.y + t.y);
                o[2 * h + 1] = pack2<DT>((v[i][h].z - mu) * inv * g.z + t.z, (v[i][h].w - mu) * inv * g.w + t.w);
            }
            *(uint4*)(out16 + r * D + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---------------------------------------------------------------------------- ViT attention: non-causal, head_dim 64
// One workgroup = NW waves = NW consecutive 32-query blocks of ONE (clip, head); the K / V tiles are staged in LDS once for all of them.
// Same swapped-product scheme as attention.hip: S^T = K.Q^T (keys on MFMA rows, a lane owns one query), the exponentiated accumulator packed
// to 16 bits is the B operand of O^T = V^T.P^T, V^T through ds_read_b64_tr_b16.
// At head_dim 64 the softmax's VALU work (one fma + v_exp_f32 + add per score, 660 issue cycles per 64 keys) outweighs the 16 MFMAs (512 pipe
// cycles, 128 of issue), so the loop is built to issue little else: 64 keys per iteration out of a double-buffered LDS image (ONE barrier per
// 64 keys; the tile after the next already in flight in registers), every LDS address a per-lane register plus an immediate (the loop is
// unrolled by two so the buffer index is a constant), the cross-half maximum by v_permlane32_swap instead of a trip through the LDS crossbar,
// row sums kept per lane half until the end (both halves rescale by the same factor), raw v_exp_f32 (libm's exp2f adds a range check and
// a ldexp per call), and MFMA results straight into VGPRs (Makefile: FLAGS_vision).  Round-3 history at 8 videos per call (32 clips x 3,136
// tokens x 16 heads, tools/vit_attn_bench.py): 3.18 ms per layer -> 2.58 (raw v_exp) -> 2.04 (64-key tiles, double buffer, permlane) -> 1.79
// (VGPR-form MFMA) -> 1.73 (immediate offsets, 8 waves per workgroup) = 746 TFLOP/s.
#define VHD 64
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define VKT2 64
template <int DT, int NW>                              // NW waves = 32 * NW queries share the staged K / V tiles
__global__ __launch_bounds__(64 * NW) void vit_attn_kernel(const bf16_t* qkv, int64_t ldq, int L, int D, bf16_t* out, int64_t ldo, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t kv_lds[2][2][VKT2 * VHD];      // [buffer][K | V][64 keys x 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y, clip = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qi = lane & 31, hf = lane >> 5;
    const int64_t tok0 = (int64_t)clip * L;
    const int64_t qtok = tok0 + min(q0 + qi, L - 1);
    bf16x8 qf[4];
    {
        const bf16_t* qrow = qkv + qtok * ldq + head * VHD + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    const float NEG = -1.0e30f;
    float m_run = NEG, l_run = 0.f;                    // l_run: this lane half's share of the row sum
    const float c_log2 = scale * 1.4426950408889634f;
    const int n_tiles = (L + VKT2 - 1) / VKT2;
    // staging: 1024 16-B chunks per tile (512 K + 512 V): a thread moves the 16-B column tid & 7 of row tid >> 3 of K and of V, and with four waves also
    // of row + 32.  Named registers, not an array: as an array captured by two lambdas the staging values were kept in scratch memory
    const int srow = tid >> 3, sch = tid & 7;
    const bf16_t* gk = qkv + D + head * VHD + 8 * sch;
    const bf16_t* gv = qkv + 2 * D + head * VHD + 8 * sch;
    const int lk0 = srow * VHD + 8 * (sch ^ (srow & 7)), lk1 = (srow + 32) * VHD + 8 * (sch ^ (srow & 7));
    const int lv0 = srow * VHD + 8 * (sch ^ ((srow & 3) << 1)), lv1 = (srow + 32) * VHD + 8 * (sch ^ ((srow & 3) << 1));
    uint4 st0, st1, st2, st3;
#define VIT_LOAD_TILE(t)                                                                   \
    {                                                                                      \
        const int64_t r0 = (tok0 + min((t) * VKT2 + srow, L - 1)) * ldq;                   \
        st0 = *(const uint4*)(gk + r0); st2 = *(const uint4*)(gv + r0);                    \
        if constexpr (NW == 4) {                                                           \
            const int64_t r1 = (tok0 + min((t) * VKT2 + srow + 32, L - 1)) * ldq;          \
            st1 = *(const uint4*)(gk + r1); st3 = *(const uint4*)(gv + r1);                \
        }                                                                                  \
    }
#define VIT_STORE_TILE(b)                                                                  \
    {                                                                                      \
        *(uint4*)(&kv_lds[b][0][lk0]) = st0; *(uint4*)(&kv_lds[b][1][lv0]) = st2;          \
        if constexpr (NW == 4) { *(uint4*)(&kv_lds[b][0][lk1]) = st1; *(uint4*)(&kv_lds[b][1][lv1]) = st3; } \
    }
    VIT_LOAD_TILE(0);
    VIT_STORE_TILE(0);
    if (n_tiles > 1) VIT_LOAD_TILE(1);
    // one tile; the buffer index is a compile-time constant (the loop below is unrolled by two) so that every LDS address is a per-lane register
    // plus an immediate offset
    auto tile = [&](auto BUF, const int t) __attribute__((always_inline)) {
        constexpr int B = decltype(BUF)::value;
        __syncthreads();                               // tile t is visible in buffer B; nobody reads buffer B ^ 1 any more
        if (t + 1 < n_tiles) {
            VIT_STORE_TILE(B ^ 1);
            if (t + 2 < n_tiles) VIT_LOAD_TILE(t + 2);
        }
        const bf16_t* k_lds = kv_lds[B][0];
        const bf16_t* v_lds = kv_lds[B][1];
        const int k0 = t * VKT2;
        f32x16 sacc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[h][r] = 0.f;
        {
            const int key = lane & 31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x8 kf = *(const bf16x8*)(k_lds + (32 * h + key) * VHD + 8 * ((2 * ks + hf) ^ (key & 7)));
                    sacc[h] = mfma32<DT>(kf, qf[ks], sacc[h]);
                }
            }
        }
        if (k0 + VKT2 > L) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = 4 * g + j; if (k0 + 32 * h + 8 * g + 4 * hf + j >= L) sacc[h][r] = NEG; }
        }
        float tmax = NEG;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[h][r]);
        tmax = xhalf_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float mc = m_new * c_log2;
        float pv[2][16];
        float rsum = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[h][r], c_log2, -mc));      // native v_exp_f32: arguments <= 0, masked scores give exp2(-huge) = 0
                pv[h][r] = e;
                rsum += e;
            }
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_log2);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        l_run += rsum;
        m_run = m_new;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {               // 16 keys per step: s4 >> 1 = the 32-key half, s4 & 1 = its 16-key half
            const int h = s4 >> 1, s2 = s4 & 1;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack2<DT>(pv[h][8 * s2 + 2 * j], pv[h][8 * s2 + 2 * j + 1]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int i16 = lane & 15, g16 = (lane >> 4) & 1;
                const int dcol = 32 * db + 16 * g16 + 4 * (i16 & 3);
                const int ch = dcol >> 3, within = dcol & 7;
                const int kr0 = 32 * h + 16 * s2 + 4 * hf + (i16 >> 2);
                const int kr1 = kr0 + 8;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(v_lds + kr0 * VHD + 8 * (ch ^ ((kr0 & 3) << 1)) + within));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(v_lds + kr1 * VHD + 8 * (ch ^ ((kr1 & 3) << 1)) + within));
                const bf16x8 vf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[db] = mfma32<DT>(vf, pf, o[db]);
            }
        }
    };
    for (int t = 0; t < n_tiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < n_tiles) tile(std::integral_constant<int, 1>{}, t + 1);
    }
    const float l_tot = xhalf_sum(l_run);
    if (q0 + qi < L) {
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        bf16_t* orow = out + (tok0 + q0 + qi) * ldo + head * VHD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * db + 8 * g + 4 * hf;
                *(uint2*)(orow + d) = make_uint2(pack2<DT>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv), pack2<DT>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
            }
    }
}

// ---------------------------------------------------------------------------- ToMe (f32; mm_projector_builder.py:6-130)
// metric[b, t, :] = mean over heads of x[b, t, h, :], L2-normalised (:119, :23).  One wave per token; dim == 64.
__global__ __launch_bounds__(256) void tome_metric_kernel(const float* x, int64_t n_tok, int heads, float* metric) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tok) return;
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += x[t * heads * 64 + h * 64 + lane];
    s /= (float)heads;
    const float n = sqrtf(wave_sum(s * s));
    metric[t * 64 + lane] = s / n;
}
// for every even ("a") token i of batch b: best odd ("b") token by dot product; ties -> lowest index (torch.max on CPU)
__global__ __launch_bounds__(256) void tome_match_kernel(const float* metric, int p, float* node_max, int32_t* node_idx) {
    const int lane = threadIdx.x & 63;
    const int t1 = p / 2;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= t1) return;
    const float* m = metric + (int64_t)b * p * 64;
    float a[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) { const float4 v = *(const float4*)(m + (int64_t)(2 * i) * 64 + d); a[d] = v.x; a[d + 1] = v.y; a[d + 2] = v.z; a[d + 3] = v.w; }
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int j = lane; j < t1; j += 64) {
        const float* br = m + (int64_t)(2 * j + 1) * 64;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; d += 4) { const float4 v = *(const float4*)(br + d); s = fmaf(a[d], v.x, s); s = fmaf(a[d + 1], v.y, s); s = fmaf(a[d + 2], v.z, s); s = fmaf(a[d + 3], v.w, s); }
        if (s > best) { best = s; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { node_max[(int64_t)b * t1 + i] = best; node_idx[(int64_t)b * t1 + i] = bi; }
}
// edge[b, :] = argsort(node_max[b, :], descending), equal keys in ascending index order; bitonic sort of (key, index) in LDS, t1 <= 2048
__global__ __launch_bounds__(1024) void tome_sort_kernel(const float* node_max, int t1, int32_t* edge) {
    __shared__ float key[2048];
    __shared__ int32_t idx[2048];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 1024) { key[i] = i < t1 ? node_max[(int64_t)b * t1 + i] : -INFINITY; idx[i] = i < t1 ? i : 0x7fffffff; }
    __syncthreads();
    // "a before b" <=> key[a] > key[b] or (key equal and idx[a] < idx[b])
    for (int k = 2; k <= 2048; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < 2048; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const float ki = key[i], kl = key[l]; const int ii = idx[i], il = idx[l];
                    const bool l_first = kl > ki || (kl == ki && il < ii);       // element l should precede element i
                    if (l_first == up) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < t1; i += 1024) edge[(int64_t)b * t1 + i] = idx[i];
}
// per batch entry: the first r edges grouped by destination (stable: sources of one destination stay in edge order, the order in which
// scatter_add visits them on CPU).  off[b, 0..t1] = start of every odd token's source list, list[b, 0..r) = even TOKEN indices.
__global__ __launch_bounds__(1024) void tome_group_kernel(const int32_t* edge, const int32_t* node_idx, int t1, int r, int32_t* off, int32_t* list) {
    __shared__ int32_t cnt[2049];
    __shared__ int32_t src[2048], dst[2048];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t* eb = edge + (int64_t)b * t1;
    const int32_t* nb = node_idx + (int64_t)b * t1;
    for (int i = tid; i <= t1; i += 1024) cnt[i] = 0;
    for (int i = tid; i < r; i += 1024) { const int a = eb[i]; src[i] = a; dst[i] = nb[a]; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < r; ++i) cnt[dst[i] + 1] += 1;
        for (int j = 0; j < t1; ++j) cnt[j + 1] += cnt[j];                 // cnt[j] = start of destination j
    }
    __syncthreads();
    for (int i = tid; i <= t1; i += 1024) off[(int64_t)b * (t1 + 1) + i] = cnt[i];
    __syncthreads();
    if (tid == 0)
        for (int i = 0; i < r; ++i) list[(int64_t)b * t1 + cnt[dst[i]]++] = 2 * src[i];
}
// one workgroup per OUTPUT token (merge_wavg, :58-74 with the `merge` closure :35-42): rows [0, t1-r) = unmerged even tokens in
// edge order, rows [t1-r, p-r) = odd tokens, each with its merged sources added in edge order; out = sum(x * size) / sum(size),
// size_out = sum(size).  size == nullptr means all ones.  No fma contraction: the same roundings as the reference's mul, add, div.
__global__ __launch_bounds__(256) void tome_merge_kernel(const float* x, const float* size, int p, int c, int r, const int32_t* edge, const int32_t* off,
                                                         const int32_t* list, float* xo, float* so) {
    const int t1 = p / 2, b = blockIdx.y, o = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (int64_t)b * p * c;
    const float* sb = size ? size + (int64_t)b * p : nullptr;
    float* xrow = xo + ((int64_t)b * (p - r) + o) * c;
    if (o < t1 - r) {                                           // unmerged even token
        const int tok = 2 * edge[(int64_t)b * t1 + r + o];
        const float s = sb ? sb[tok] : 1.0f;
        for (int ch = tid; ch < c; ch += 256) xrow[ch] = __fdiv_rn(__fmul_rn(xb[(int64_t)tok * c + ch], s), s);
        if (tid == 0) so[(int64_t)b * (p - r) + o] = s;
        return;
    }
    const int j = o - (t1 - r);                                 // odd token 2j+1
    const int32_t* lb = list + (int64_t)b * t1;
    const int s_lo = off[(int64_t)b * (t1 + 1) + j], s_hi = off[(int64_t)b * (t1 + 1) + j + 1];
    const int tok = 2 * j + 1;
    const float s0 = sb ? sb[tok] : 1.0f;
    float stot = s0;
    for (int i = s_lo; i < s_hi; ++i) stot = __fadd_rn(stot, sb ? sb[lb[i]] : 1.0f);
    for (int ch = tid; ch < c; ch += 256) {
        float acc = __fmul_rn(xb[(int64_t)tok * c + ch], s0);
        for (int i = s_lo; i < s_hi; ++i) { const int st = lb[i]; acc = __fadd_rn(acc, __fmul_rn(xb[(int64_t)st * c + ch], sb ? sb[st] : 1.0f)); }
        xrow[ch] = __fdiv_rn(acc, stot);
    }
    if (tid == 0) so[(int64_t)b * (p - r) + o] = stot;
}

// ---------------------------------------------------------------------------- engine
struct VBuf { void* p = nullptr; size_t bytes = 0; };
static int vensure(VBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return BLIM_OK;
    if (b.p) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    HIP_TRY(hipMalloc(&b.p, bytes + 4096));
    b.bytes = bytes + 4096;
    return BLIM_OK;
}
struct VBlock {
    float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    bf16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;
};
struct blim_vision {
    blim_vision_config c;
    int G = 0, L = 0;
    bf16_t* patch_w = nullptr; float* patch_b = nullptr; float* norm_w = nullptr; float* norm_b = nullptr; float* pos = nullptr;
    std::vector<VBlock> B;
    std::vector<void*> owned;
    std::map<std::string, bool> loaded;
    bool pos_set = false;
    VBuf patches, resid, xn, qkv, attn, act, feat, stage, tome_x[2], tome_s[2], metric, nmax, nidx, edge, goff, glist;
};

static int valloc(blim_vision* v, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    HIP_TRY(hipMemset(*p, 0, bytes));
    v->owned.push_back(*p);
    return BLIM_OK;
}
struct VSlot { int kind; void* dst; int64_t n; };   // kind 0: 16-bit matrix, 1: f32 vector
static bool vfind(blim_vision* v, const std::string& name, VSlot& s) {
    const int D = v->c.hidden_size, Hm = v->c.mlp_hidden, P = v->c.patch_size;
    if (name == "vit.patch.w") { s = {0, v->patch_w, (int64_t)D * 3 * P * P}; return true; }
    if (name == "vit.patch.b") { s = {1, v->patch_b, D}; return true; }
    if (name == "vit.norm.w") { s = {1, v->norm_w, D}; return true; }
    if (name == "vit.norm.b") { s = {1, v->norm_b, D}; return true; }
    int i = -1; char rest[32] = "";
    if (sscanf(name.c_str(), "vit.blocks.%d.%31s", &i, rest) == 2 && i >= 0 && i < v->c.depth) {
        VBlock& b = v->B[i];
        const std::string r(rest);
        if (r == "norm1.w") { s = {1, b.n1w, D}; return true; }
        if (r == "norm1.b") { s = {1, b.n1b, D}; return true; }
        if (r == "norm2.w") { s = {1, b.n2w, D}; return true; }
        if (r == "norm2.b") { s = {1, b.n2b, D}; return true; }
        if (r == "q_bias") { s = {1, b.qkv_b, D}; return true; }                 // qkv bias = [q_bias, 0, v_bias] (:106-108)
        if (r == "v_bias") { s = {1, b.qkv_b + 2 * D, D}; return true; }
        if (r == "qkv.w") { s = {0, b.qkv_w, (int64_t)3 * D * D}; return true; }
        if (r == "proj.w") { s = {0, b.proj_w, (int64_t)D * D}; return true; }
        if (r == "proj.b") { s = {1, b.proj_b, D}; return true; }
        if (r == "fc1.w") { s = {0, b.fc1_w, (int64_t)Hm * D}; return true; }
        if (r == "fc1.b") { s = {1, b.fc1_b, Hm}; return true; }
        if (r == "fc2.w") { s = {0, b.fc2_w, (int64_t)D * Hm}; return true; }
        if (r == "fc2.b") { s = {1, b.fc2_b, D}; return true; }
    }
    return false;
}
static std::vector<std::string> vnames(const blim_vision* v) {
    std::vector<std::string> n = {"vit.patch.w", "vit.patch.b", "vit.norm.w", "vit.norm.b"};
    for (int i = 0; i < v->c.depth; ++i)
        for (const char* t : {"norm1.w", "norm1.b", "q_bias", "v_bias", "qkv.w", "proj.w", "proj.b", "norm2.w", "norm2.b", "fc1.w", "fc1.b", "fc2.w", "fc2.b"})
            n.push_back("vit.blocks." + std::to_string(i) + "." + t);
    return n;
}

extern "C" int blim_vision_create(const blim_vision_config* cfg, blim_vision** out) {
    ARG_CHECK(cfg && out);
    ARG_CHECK(cfg->image_size > 0 && cfg->patch_size > 0 && cfg->image_size % cfg->patch_size == 0 && cfg->patch_size % 8 == 0);
    ARG_CHECK(cfg->num_frames > 0 && cfg->depth > 0 && cfg->num_heads > 0 && cfg->tome_tokens > 0);
    if (cfg->hidden_size != cfg->num_heads * 64) { blim_set_error("vision tower: head_dim %d unsupported (the attention / ToMe kernels are built for 64)", cfg->hidden_size / cfg->num_heads); return BLIM_ERR_ARG; }
    ARG_CHECK(cfg->hidden_size % 64 == 0 && cfg->hidden_size <= 4096 && cfg->mlp_hidden % 64 == 0 && (3 * cfg->patch_size * cfg->patch_size) % 64 == 0);
    ARG_CHECK(cfg->compute_dtype == BLIM_COMPUTE_BF16 || cfg->compute_dtype == BLIM_COMPUTE_F16);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { blim_set_error("no HIP device visible: the vision encoder has no CPU fallback"); return BLIM_ERR_HIP; }
    blim_vision* v = new blim_vision();
    v->c = *cfg;
    v->G = cfg->image_size / cfg->patch_size;
    v->L = cfg->num_frames * v->G * v->G;
    if (v->L / 2 > 2048 || v->L <= cfg->tome_tokens) { blim_set_error("vision tower: %d tokens per clip unsupported (ToMe sorts at most 2048 candidates; must exceed tome_tokens)", v->L); delete v; return BLIM_ERR_ARG; }
    const int D = cfg->hidden_size, Hm = cfg->mlp_hidden, PK = 3 * cfg->patch_size * cfg->patch_size;
    v->B.resize(cfg->depth);
    int rc = BLIM_OK;
#define A(ptr, count, type) do { if (rc == BLIM_OK) rc = valloc(v, (void**)&(ptr), (size_t)(count) * sizeof(type)); } while (0)
    A(v->patch_w, (int64_t)D * PK, bf16_t); A(v->patch_b, D, float); A(v->norm_w, D, float); A(v->norm_b, D, float); A(v->pos, (int64_t)v->L * D, float);
    for (auto& b : v->B) {
        A(b.n1w, D, float); A(b.n1b, D, float); A(b.n2w, D, float); A(b.n2b, D, float); A(b.qkv_b, 3 * D, float); A(b.proj_b, D, float);
        A(b.fc1_b, Hm, float); A(b.fc2_b, D, float);
        A(b.qkv_w, (int64_t)3 * D * D, bf16_t); A(b.proj_w, (int64_t)D * D, bf16_t); A(b.fc1_w, (int64_t)Hm * D, bf16_t); A(b.fc2_w, (int64_t)D * Hm, bf16_t);
    }
#undef A
    if (rc != BLIM_OK) { blim_vision_destroy(v); return rc; }
    *out = v;
    return BLIM_OK;
}
extern "C" void blim_vision_destroy(blim_vision* v) {
    if (!v) return;
    hipDeviceSynchronize();
    for (void* p : v->owned) hipFree(p);
    VBuf* bufs[] = {&v->patches, &v->resid, &v->xn, &v->qkv, &v->attn, &v->act, &v->feat, &v->stage, &v->tome_x[0], &v->tome_x[1], &v->tome_s[0], &v->tome_s[1],
                    &v->metric, &v->nmax, &v->nidx, &v->edge, &v->goff, &v->glist};
    for (VBuf* b : bufs) if (b->p) hipFree(b->p);
    delete v;
}
static int vplace(blim_vision* v, const std::string& name, const void* dev_src, int dtype) {
    VSlot s;
    if (!vfind(v, name, s)) { blim_set_error("unknown vision weight name '%s'", name.c_str()); return BLIM_ERR_ARG; }
    if (s.kind == 0) {
        const int grid = (int)std::min<int64_t>((s.n + 255) / 256, 16384);
        const bool f16 = v->c.compute_dtype == BLIM_COMPUTE_F16;
        if (dtype == BLIM_DTYPE_F32) { if (f16) hipLaunchKernelGGL((to_h16_kernel<true, DT_F16>), dim3(grid), dim3(256), 0, 0, (bf16_t*)s.dst, dev_src, s.n); else hipLaunchKernelGGL((to_h16_kernel<true, DT_BF16>), dim3(grid), dim3(256), 0, 0, (bf16_t*)s.dst, dev_src, s.n); }
        else { if (f16) hipLaunchKernelGGL((to_h16_kernel<false, DT_F16>), dim3(grid), dim3(256), 0, 0, (bf16_t*)s.dst, dev_src, s.n); else hipLaunchKernelGGL((to_h16_kernel<false, DT_BF16>), dim3(grid), dim3(256), 0, 0, (bf16_t*)s.dst, dev_src, s.n); }
    } else {
        const int grid = (int)((s.n + 255) / 256);
        if (dtype == BLIM_DTYPE_F32) hipLaunchKernelGGL(to_f32_kernel<true>, dim3(grid), dim3(256), 0, 0, (float*)s.dst, dev_src, s.n);
        else hipLaunchKernelGGL(to_f32_kernel<false>, dim3(grid), dim3(256), 0, 0, (float*)s.dst, dev_src, s.n);
    }
    KCHECK("vision weight placement");
    v->loaded[name] = true;
    return BLIM_OK;
}
extern "C" int blim_vision_load_weight(blim_vision* v, const char* name, const void* data, int32_t dtype, int32_t on_device) {
    ARG_CHECK(v && name && data && (dtype == BLIM_DTYPE_F32 || dtype == BLIM_DTYPE_BF16));
    VSlot s;
    if (!vfind(v, name, s)) { blim_set_error("unknown vision weight name '%s'", name); return BLIM_ERR_ARG; }
    const void* src = data;
    if (!on_device) {
        const size_t bytes = (size_t)s.n * (dtype == BLIM_DTYPE_F32 ? 4 : 2);
        TRY(vensure(v->stage, bytes));
        HIP_TRY(hipMemcpy(v->stage.p, data, bytes, hipMemcpyHostToDevice));
        src = v->stage.p;
    }
    TRY(vplace(v, name, src, dtype));
    HIP_TRY(hipDeviceSynchronize());
    return BLIM_OK;
}
static uint64_t vfnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ull; }
    return h;
}
extern "C" int blim_vision_init_synthetic_weights(blim_vision* v, uint64_t seed) {
    ARG_CHECK(v);
    const double kSigma4 = 37837.22723328507;
    for (const std::string& name : vnames(v)) {
        VSlot s;
        if (!vfind(v, name, s)) return BLIM_ERR_STATE;
        const bool is_gain = name.size() >= 7 && (name.compare(name.size() - 7, 7, "norm1.w") == 0 || name.compare(name.size() - 7, 7, "norm2.w") == 0 || name == "vit.norm.w");
        const float std_ = is_gain ? 0.1f : 0.02f, mean = is_gain ? 1.0f : 0.0f;
        TRY(vensure(v->stage, (size_t)s.n * 4));
        TRY(launch_fill_bell_f32((float*)v->stage.p, s.n, seed, vfnv1a64(name.c_str()), (float)((double)std_ / kSigma4), mean, 1, 0));
        TRY(vplace(v, name, v->stage.p, BLIM_DTYPE_F32));
    }
    HIP_TRY(hipDeviceSynchronize());
    return BLIM_OK;
}
extern "C" int blim_vision_set_pos_embed(blim_vision* v, const float* table_host) {
    ARG_CHECK(v && table_host);
    HIP_TRY(hipMemcpy(v->pos, table_host, (size_t)v->L * v->c.hidden_size * 4, hipMemcpyHostToDevice));
    v->pos_set = true;
    return BLIM_OK;
}
extern "C" int blim_vision_ready(const blim_vision* v) {
    ARG_CHECK(v);
    for (const std::string& n : vnames(v)) if (!v->loaded.count(n)) { blim_set_error("vision weight '%s' not loaded", n.c_str()); return BLIM_ERR_STATE; }
    if (!v->pos_set) { blim_set_error("position table not set (blim_vision_set_pos_embed)"); return BLIM_ERR_STATE; }
    return BLIM_OK;
}

static GemmParams vgp(int dt, const void* A, int64_t lda, const void* W, int64_t M, int N, int K, void* C, int64_t ldc, const float* bias) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt; p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.M = (int)M; p.N = N; p.K = K; p.C = C; p.ldc = ldc; p.scale = 1.0f; p.bias = bias;
    p.f16_saturate = 1;
    return p;
}

// ToMe on f32 tokens x [b, p, c] (c = heads * 64) down to `target` tokens per batch entry; result in out [b, target, c]
static int tome_merge_tokens(blim_vision* v, const float* x, int b, int p, int c, int heads, int target, float* out, hipStream_t s) {
    ARG_CHECK(p > target && c == heads * 64 && p / 2 <= 2048);
    TRY(vensure(v->tome_x[0], (size_t)b * p * c * 4)); TRY(vensure(v->tome_x[1], (size_t)b * p * c * 4));
    TRY(vensure(v->tome_s[0], (size_t)b * p * 4)); TRY(vensure(v->tome_s[1], (size_t)b * p * 4));
    TRY(vensure(v->metric, (size_t)b * p * 64 * 4)); TRY(vensure(v->nmax, (size_t)b * p * 4)); TRY(vensure(v->nidx, (size_t)b * p * 4)); TRY(vensure(v->edge, (size_t)b * p * 4));
    TRY(vensure(v->goff, (size_t)b * (p + 2) * 4)); TRY(vensure(v->glist, (size_t)b * p * 4));
    const float* cur = x; const float* cur_s = nullptr;
    int which = 0, tmp = p;
    while (tmp != target) {                                               // merge_tokens' schedule (:108-115)
        const int r = (tmp - target <= tmp / 2) ? tmp - target : tmp / 2;
        if (tmp % 2 != 0) { blim_set_error("ToMe: %d tokens in a round -- odd counts (an unpaired even token) are not implemented; 4 x (S/16)^2 -> 64 never produces one", tmp); return BLIM_ERR_ARG; }
        const int t1 = tmp / 2;
        const int64_t n_tok = (int64_t)b * tmp;
        hipLaunchKernelGGL(tome_metric_kernel, dim3((unsigned)((n_tok + 3) / 4)), dim3(256), 0, s, cur, n_tok, heads, (float*)v->metric.p);
        hipLaunchKernelGGL(tome_match_kernel, dim3((t1 + 3) / 4, b), dim3(256), 0, s, (const float*)v->metric.p, tmp, (float*)v->nmax.p, (int32_t*)v->nidx.p);
        hipLaunchKernelGGL(tome_sort_kernel, dim3(b), dim3(1024), 0, s, (const float*)v->nmax.p, t1, (int32_t*)v->edge.p);
        float* xo = (tmp - r == target) ? out : (float*)v->tome_x[which].p;
        hipLaunchKernelGGL(tome_group_kernel, dim3(b), dim3(1024), 0, s, (const int32_t*)v->edge.p, (const int32_t*)v->nidx.p, t1, r, (int32_t*)v->goff.p, (int32_t*)v->glist.p);
        hipLaunchKernelGGL(tome_merge_kernel, dim3(tmp - r, b), dim3(256), 0, s, cur, cur_s, tmp, c, r, (const int32_t*)v->edge.p, (const int32_t*)v->goff.p,
                           (const int32_t*)v->glist.p, xo, (float*)v->tome_s[which].p);
        KCHECK("tome");
        cur = xo; cur_s = (const float*)v->tome_s[which].p;
        which ^= 1;
        tmp -= r;
    }
    return BLIM_OK;
}

// softmax(q k^T / sqrt(64)) v per (clip, head) over the fused [tokens, 3 D] projection; out [tokens, D]
static int launch_vit_attn(int dt, const bf16_t* qkv, int L, int D, int heads, int n_clips, bf16_t* out, hipStream_t s) {
    // eight waves (256 queries) share a staged tile where the sequence is long enough to fill such workgroups (-1 % at L = 3,136), four otherwise
    if (L >= 1024) {
        const dim3 grid((L + 255) / 256, heads, n_clips);
        if (dt == DT_F16) hipLaunchKernelGGL((vit_attn_kernel<DT_F16, 8>), grid, dim3(512), 0, s, qkv, (int64_t)3 * D, L, D, out, (int64_t)D, 0.125f);
        else hipLaunchKernelGGL((vit_attn_kernel<DT_BF16, 8>), grid, dim3(512), 0, s, qkv, (int64_t)3 * D, L, D, out, (int64_t)D, 0.125f);
    } else {
        const dim3 grid((L + 127) / 128, heads, n_clips);
        if (dt == DT_F16) hipLaunchKernelGGL((vit_attn_kernel<DT_F16, 4>), grid, dim3(256), 0, s, qkv, (int64_t)3 * D, L, D, out, (int64_t)D, 0.125f);
        else hipLaunchKernelGGL((vit_attn_kernel<DT_BF16, 4>), grid, dim3(256), 0, s, qkv, (int64_t)3 * D, L, D, out, (int64_t)D, 0.125f);
    }
    KCHECK("vit attention");
    return BLIM_OK;
}

extern "C" int blim_vit_attention(const void* qkv, int32_t n_clips, int32_t L, int32_t heads, int32_t dtype16, void* out, void* stream) {
    ARG_CHECK(qkv && out && n_clips > 0 && L > 0 && heads > 0 && (dtype16 == BLIM_COMPUTE_F16 || dtype16 == BLIM_COMPUTE_BF16));
    return launch_vit_attn(dtype16 == BLIM_COMPUTE_F16 ? DT_F16 : DT_BF16, (const bf16_t*)qkv, L, heads * VHD, heads, n_clips, (bf16_t*)out, (hipStream_t)stream);
}

extern "C" int blim_tome_merge(blim_vision* v, const float* x, int32_t b, int32_t p, int32_t c, int32_t heads, int32_t target, float* out, void* stream) {
    ARG_CHECK(v && x && out && b > 0);
    return tome_merge_tokens(v, x, b, p, c, heads, target, out, (hipStream_t)stream);
}

extern "C" int blim_vision_encode(blim_vision* v, const void* frames, int32_t n_clips, float* out_feat, float* out_tome, void* stream) {
    ARG_CHECK(v && frames && n_clips > 0 && (out_feat || out_tome));
    TRY(blim_vision_ready(v));
    hipStream_t s = (hipStream_t)stream;
    const blim_vision_config& c = v->c;
    const int D = c.hidden_size, Hm = c.mlp_hidden, P = c.patch_size, PK = 3 * P * P, L = v->L, dt = c.compute_dtype;
    const int64_t M = (int64_t)n_clips * L, Mp = (M + 255) / 256 * 256;
    TRY(vensure(v->patches, (size_t)Mp * PK * 2)); TRY(vensure(v->resid, (size_t)Mp * D * 4)); TRY(vensure(v->xn, (size_t)Mp * D * 2));
    TRY(vensure(v->qkv, (size_t)Mp * 3 * D * 2)); TRY(vensure(v->attn, (size_t)Mp * D * 2)); TRY(vensure(v->act, (size_t)Mp * Hm * 2));
    float* feat = out_feat;
    if (!feat) { TRY(vensure(v->feat, (size_t)Mp * D * 4)); feat = (float*)v->feat.p; }
    float* resid = (float*)v->resid.p;
    bf16_t* xn = (bf16_t*)v->xn.p; bf16_t* qkv = (bf16_t*)v->qkv.p; bf16_t* attn = (bf16_t*)v->attn.p; bf16_t* act = (bf16_t*)v->act.p;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)std::min<int64_t>((M * (PK / 8) + 255) / 256, 65535)), dim3(256), 0, s, (bf16_t*)v->patches.p, (const bf16_t*)frames, M, c.num_frames, v->G, P, c.image_size);
    hipLaunchKernelGGL(init_resid_kernel, dim3((unsigned)std::min<int64_t>((M * (D / 4) + 255) / 256, 65535)), dim3(256), 0, s, resid, v->pos, v->patch_b, M, L, D);
    KCHECK("patchify");
    { GemmParams p = vgp(dt, v->patches.p, PK, v->patch_w, M, D, PK, resid, D, nullptr); TRY(launch_gemm(EPI_RESID, p, s)); }
    const dim3 ln_grid((unsigned)((M + 3) / 4));
    static const int ln_wide = getenv("BLIM_LN_WIDE") ? atoi(getenv("BLIM_LN_WIDE")) : 1;
    auto layernorm = [&](const float* w, const float* b, float eps, bf16_t* o16, float* o32) -> int {
        if (ln_wide && o16 && !o32 && D % 8 == 0 && D <= 4096) {
            if (dt == DT_F16) hipLaunchKernelGGL(layernorm_wide_kernel<DT_F16>, ln_grid, dim3(256), 0, s, (const float*)resid, M, D, w, b, eps, o16);
            else hipLaunchKernelGGL(layernorm_wide_kernel<DT_BF16>, ln_grid, dim3(256), 0, s, (const float*)resid, M, D, w, b, eps, o16);
            KCHECK("layernorm");
            return BLIM_OK;
        }
        if (dt == DT_F16) hipLaunchKernelGGL(layernorm_kernel<DT_F16>, ln_grid, dim3(256), 0, s, (const float*)resid, M, D, w, b, eps, o16, o32);
        else hipLaunchKernelGGL(layernorm_kernel<DT_BF16>, ln_grid, dim3(256), 0, s, (const float*)resid, M, D, w, b, eps, o16, o32);
        KCHECK("layernorm");
        return BLIM_OK;
    };
    for (int i = 0; i < c.depth; ++i) {
        const VBlock& b = v->B[i];
        TRY(layernorm(b.n1w, b.n1b, 1e-6f, xn, nullptr));
        { GemmParams p = vgp(dt, xn, D, b.qkv_w, M, 3 * D, D, qkv, 3 * D, b.qkv_b); TRY(launch_gemm(EPI_BF16, p, s)); }
        TRY(launch_vit_attn(dt, qkv, L, D, c.num_heads, n_clips, attn, s));
        { GemmParams p = vgp(dt, attn, D, b.proj_w, M, D, D, resid, D, b.proj_b); TRY(launch_gemm(EPI_RESID, p, s)); }
        TRY(layernorm(b.n2w, b.n2b, 1e-6f, xn, nullptr));
        { GemmParams p = vgp(dt, xn, D, b.fc1_w, M, Hm, D, act, Hm, b.fc1_b); p.act = 1; TRY(launch_gemm(EPI_BF16, p, s)); }
        { GemmParams p = vgp(dt, act, Hm, b.fc2_w, M, D, Hm, resid, D, b.fc2_b); TRY(launch_gemm(EPI_RESID, p, s)); }
    }
    TRY(layernorm(v->norm_w, v->norm_b, 1e-12f, nullptr, feat));
    if (out_tome) TRY(tome_merge_tokens(v, feat, n_clips, L, D, c.num_heads, c.tome_tokens, out_tome, s));
    return BLIM_OK;
}
