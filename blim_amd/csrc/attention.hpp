// GQA attention over packed tokens with a shared prefix segment and a key-visibility mask (K6).
// Semantics = the reference's eager/SDPA path (modeling_qwen2_flash.py:288-310, mask :1025-1040):
//   query token i of sequence s attends to keys { all prefix tokens of s } U { own tokens 0..i },
//   restricted to key_visible[token] != 0; query rows at invisible positions are still computed.
//   A query with no visible key gets a zero output (the reference averages V uniformly there; such
//   rows never reach a score -- DESIGN.md "fully masked rows").
#pragma once
#include "common.hpp"

struct AttnParams {
    int dtype;  // DT_BF16 / DT_F16
    const bf16_t* qkv;  // [T, ldq]: q heads | k heads | v heads, head_dim 128, RoPE already applied
    int64_t ldq;
    int num_heads, num_kv_heads;
    const uint8_t* key_visible;  // [T]
    const int32_t* seq_start;    // [S] first own token
    const int32_t* seq_len;      // [S]
    const int32_t* pfx_start;    // [S] first token of the shared prefix (ignored when pfx_len == 0)
    const int32_t* pfx_len;      // [S]
    const int32_t* blk_seq;      // [n_blocks] sequence of each 32-query block
    const int32_t* blk_q0;       // [n_blocks] first query (offset inside the sequence) of the block
    const int32_t* own_start;    // [T] or nullptr: first own-segment index (inside the sequence) a token attends to (blim.h: blim_batch.own_start)
    int n_blocks;
    bf16_t* out;  // [T, num_heads*128]
    int64_t ldo;
    float scale;  // 1/sqrt(head_dim)
    // compensated ("precise") mode, fp16: Q, K, V = hi + lo with the lo parts v_lo_off columns after the hi parts in `qkv`; the
    // products S = K.Q^T and O = V^T.P^T (P = P_hi + P_lo as well) are formed as hi.hi + hi.lo + lo.hi (three MFMA passes), and
    // the output is written as hi at `out`, lo = f16(o - f32(hi)) at out + out_lo_off.  0 / 0 = plain.
    int64_t v_lo_off, out_lo_off;
    // fp8 mode, fused quantisation: when out8 != nullptr the output is written as e4m3 bytes [T, ldo8] instead of 16-bit, one E8M0 scale per
    // (token, head) = per 128-deep K-step of the o_proj GEMM, into out_mx (layout: gemm.hpp `a_mx`; mx_stride = bytes per K-step)
    uint8_t* out8;
    int64_t ldo8;
    uint8_t* out_mx;
    int64_t mx_stride;
    // training: when non-null, the log-sum-exp of every (token, head) row, f32 [T, num_heads] in natural-log units of the SCALED scores
    // (what the backward needs to re-materialise P = exp(scale * q.k - lse)); 1e30 for a row without a visible key
    float* lse_out;
};

int launch_attention(const AttnParams& p, int use_tr_read, hipStream_t stream);
