// HBM-bound helper kernels of the scoring path (gfx950): synthetic fill, embedding gather (K2),
// RMSNorm (K3/K9), log-sum-exp combine + masked mean (K11), cross-entropy on materialised logits,
// TVG criterion (K15).
#pragma once
#include "common.hpp"

int launch_fill_bell_bf16(bf16_t* out, int64_t n, uint64_t seed, uint64_t tensor_id, float scale, float mean, hipStream_t s);
int launch_fill_bell_f32(float* out, int64_t n, uint64_t seed, uint64_t tensor_id, float scale, float mean, int round_bf16, hipStream_t s);

// out[t, :] = src_index[t] >= 0 ? table[src_index[t], :] : feats[-(src_index[t]+1), :]      (bf16 rows of width H)
int launch_assemble(bf16_t* out, const int32_t* src_index, int64_t n_tokens, int H, const bf16_t* table, const bf16_t* feats, hipStream_t s,
                    bool split = false);   // split (compensated mode): out / feats rows are [hi | lo] of width 2H

// resid[t, :] = f32(embeds[t, :])   (16-bit input of the engine's compute dtype)
int launch_h16_to_f32(float* out, const bf16_t* in, int64_t n, int dtype, hipStream_t s);
int launch_gather_rows(void* dst, const void* src, const int32_t* rows, int64_t n_rows, int64_t row_bytes, int64_t n_src, uint32_t fill, hipStream_t s);   // dst[r] = src[rows[r]]
int launch_hilo_to_f32(float* out, const bf16_t* in, int64_t n_rows, int H, int dtype, hipStream_t s);   // [hi | lo] 16-bit rows of width 2H -> f32 [n_rows, H]
int launch_f32_to_bf16(bf16_t* out, const float* in, int64_t n, hipStream_t s);
int launch_zero_rows(bf16_t* x, int64_t ld, const uint8_t* keep, int64_t n_rows, int width, hipStream_t s);      // x[r, :width] = 0 where keep[r] == 0

// out[i, :] = bf16( w * x[rows ? rows[i] : i, :] * rsqrt(mean(x^2) + eps) ); optionally also f32 copy.
int launch_rmsnorm(const float* x, int64_t ldx, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps,
                   bf16_t* out_h16, int dtype, float* out_f32, hipStream_t s, int64_t n_src = 0,    // n_src: valid source rows when `rows` gathers
                   int64_t ldo = 0, bf16_t* out_lo = nullptr,    // ldo: row stride of out_h16 / out_lo (0 = H); out_lo: compensated mode, lo = 16-bit(x - f32(hi))
                   bool saturate = true,                         // fp16 stores saturate at +-65504 (scoring path); false: overflow to inf (the trainer: its loss scaler must see it)
                   uint8_t* out6 = nullptr);                     // the lo parts ALSO (or, with out_lo == nullptr, ONLY) as the e2m3 operand tiles of the consuming GEMM (gemm.hpp: A6; bit-identical
                                                                 // to launch_f6_tiles on the lo rows, rows up to the next multiple of 256 zero-filled); needs rmsnorm_can_write_tiles(...)
bool rmsnorm_can_write_tiles(int H, int64_t ldx, int64_t ldo);   // (the wide kernel's shapes: H % 128 == 0, 256 < H <= 4096)

// mean over groups of `group` consecutive rows: out[i,:] = mean_j in[i*group + j, :]   (16-bit in/out, f32 accumulate)
int launch_group_mean(bf16_t* out, const bf16_t* in, int64_t n_out, int group, int H, int dtype, hipStream_t s, bool split = false);

// logprob[r] = labels[r] < 0 ? 0 : label_logit[r] - logsumexp over the n_tiles (max, sumexp) partials of row r
int launch_lse_combine(const float2* part, int n_tiles, const float* label_logit, const int32_t* labels, int64_t n_rows, float* logprob, hipStream_t s);

// score[p] = sum(logprob[rows of p]) / (mode 0: count_nonzero, mode 1: row count)  for rows [row_start[p], row_start[p+1])
int launch_segment_mean(const float* logprob, const int32_t* row_start, int n_pairs, int mode, float* score, hipStream_t s);

// logprob[r] = log_softmax(logits[r, :V])[label[r]] (label < 0 -> 0), one workgroup per row
int launch_ce_rows(const float* logits, int64_t ld, int V, const int32_t* labels, int64_t n_rows, float* logprob, hipStream_t s);

// TVG criterion: logits [n_pairs*clips, n_vocab] f32 (row p*clips+c = pair p, clip c), label[p] -> score[p] = mean_c log_softmax[label]
int launch_tvg_score(const float* logits, int64_t ld, int n_vocab, const int32_t* labels, int n_pairs, int clips, float* score, hipStream_t s);

// ---- fp8 mode (DT_F8): per-row symmetric quantisation to OCP e4m3, scale = absmax / 448 (1 when the row is all zero)
// in: 16-bit [n_rows, K] (row stride ld, dtype DT_BF16/DT_F16) -> out8 [n_rows, K] + scale [n_rows].  K % 8 == 0, K <= 20480.
int launch_quant_rows(const bf16_t* in, int64_t ld, int64_t n_rows, int K, int dtype, uint8_t* out8, float* scale, hipStream_t s);
// RMSNorm whose output is quantised per row (same arithmetic as launch_rmsnorm, then the rule above)
int launch_rmsnorm_f8(const float* x, int64_t ldx, int64_t n_rows, int H, const float* w, float eps, uint8_t* out8, float* scale, hipStream_t s);

// ---- "lo6" (gemm.hpp: A6 / W6 / K6): the operands of the compensated GEMMs' second pass, e2m3 with one E8M0 (power-of-two) scale per 32 values that the block-scaled
// MFMA applies itself, written as the LDS image of the pass's operand tiles (layout: gemm.hpp).
// in: 16-bit [n_rows, K] (row stride ld) -> out: f6_tiles_bytes(n_rows, K) bytes; rows n_rows .. the next multiple of 256 are written as zeros.  K % 128 == 0.
// w_side: the scale table of a W operand (4 fragments per lane) instead of an A operand's (8).
int launch_f6_tiles(const bf16_t* in, int64_t ld, int64_t n_rows, int K, int dtype, bool w_side, uint8_t* out, hipStream_t s);
static inline size_t f6_tiles_bytes(int64_t n_rows, int K) { return (size_t)((n_rows + 255) / 256) * (K / 128) * 25600; }

// ---- the e2m3 block quantiser of the "lo6" operand tiles (device code: kernels.hip's tile writer and the SwiGLU epilogue of gemm.hip, which writes the tiles of its
// own output's lo part)
__device__ __forceinline__ int pow2_exp_ge(float x) {            // smallest e with 2^e >= x (x > 0, finite)
    int ex; const float m = frexpf(x, &ex);                      // x = m 2^ex, m in [0.5, 1)
    return m == 0.5f ? ex - 1 : ex;
}
// 32 values -> their e2m3 block: six dwords of packed 6-bit codes (value j at bits [6 j, 6 j + 6): sign | 2-bit exponent | 3-bit mantissa, OCP MX: magnitudes
// m / 8 (exponent 0) and (1 + m / 8) 2^(e - 1), largest 7.5; round to nearest even on that grid) and the E8M0 byte of the block's power-of-two scale -- the smallest
// 2^s with max|x| 2^-s <= 7.5.  A block that is all zero or holds an inf is stored as zeros (codes 0, scale byte 0), and so is every NaN element (the hi part carries
// non-finite values through the first pass).
struct F6Block { uint32_t d[6]; uint32_t e8; };
// the block's scale from its largest magnitude: E8M0 byte (return value) and 2^-s (inv; 0 for an all-zero or non-finite block)
__device__ __forceinline__ uint32_t e2m3_scale(float m, float& inv) {
    const bool live = m > 0.f && m < 3.0e38f;
    int ex = live ? pow2_exp_ge(m * (1.0f / 7.5f)) : -127;
    ex = max(-127, min(127, ex));
    inv = live ? ldexpf(1.0f, -ex) : 0.f;
    return (uint32_t)(ex + 127);
}
// one value's 6-bit code under that scale (0 in a block stored as zeros: inv == 0; a NaN beside live values -- the block maximum's fmaxf skips it -- is stored as 0 as well)
__device__ __forceinline__ uint32_t e2m3_code(float f, float inv) {
    if (!(f == f) || inv == 0.f) return 0u;
    const float a = fminf(fabsf(f) * inv, 7.5f);
    const int bin = a < 2.f ? 0 : a < 4.f ? 1 : 2;                       // step 1/8 below 2 (subnormals and the first binade share it), 1/4 below 4, 1/2 above
    const int q = (int)rintf(a * (bin == 0 ? 8.f : bin == 1 ? 4.f : 2.f));
    const int code = min(q + 8 * bin, 31);                                 // [0, 16] | 8 + [8, 16] | 16 + [8, 15]: continuous across the binades
    return (uint32_t)code | (f < 0.f ? 32u : 0u);
}
__device__ __forceinline__ F6Block e2m3_block(const float (&f)[32]) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(f[j]));
    F6Block b;
    float inv;
    b.e8 = e2m3_scale(m, inv);
    uint32_t c[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) c[j] = e2m3_code(f[j], inv);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                          // 16 codes = 96 bits = three dwords
        const uint32_t* k = c + 16 * h;
        b.d[3 * h + 0] = k[0] | k[1] << 6 | k[2] << 12 | k[3] << 18 | k[4] << 24 | k[5] << 30;
        b.d[3 * h + 1] = k[5] >> 2 | k[6] << 4 | k[7] << 10 | k[8] << 16 | k[9] << 22 | k[10] << 28;
        b.d[3 * h + 2] = k[10] >> 4 | k[11] << 2 | k[12] << 8 | k[13] << 14 | k[14] << 20 | k[15] << 26;
    }
    return b;
}
