// LoRA adapters KEPT APART in the scoring path (the reference's `--eval --resume` flow: main.py:96-105 wraps q/k/v/o_proj, lm_head and the two
// projector MLPs in peft and evaluates y = W x + b + (alpha / r) B (A x) with the adapters as separate matrices, main.py:125-128).
//
// Every adapted Linear runs on a K-AUGMENTED copy of its base weight, [W | B_hi | B_lo | 0] with `aug` (64 or 128) extra K columns, against
// activations [x | u | u | 0], u = (alpha / r) A x: the rank-r update rides in the base product's MFMA accumulation (before bias / RoPE /
// the epilogue's rounding) at +1.8 % of the K loop, W stays exactly the checkpoint's 16-bit value and B, A and u keep ~16 - 21 significant
// bits (hi + lo 16-bit pairs).  Merging instead (W + s B A rounded to the engine's 16-bit format) costs nothing per call but rounds the sum:
// harmless in fp16 for a bf16 checkpoint, 8 % of the update in bf16, most of it in e4m3.
#pragma once
#include "common.hpp"

// A16 [32, K] 16-bit: rows j < r = hi(A[j, :]), rows 16 + j = lo = 16-bit(A - f32(hi)); all other rows zero.  (r <= 16)
int launch_adapter_a16(uint16_t* A16, const float* A, int K, int r, int dtype, hipStream_t s);
// w_aug[stored_row(n), col_hi + j] = hi(B[n, j]), w_aug[.., col_lo + j] = lo(B[n, j])  for n < N, j < r   (row_mode 1: natural row n of a
// q/k head -> its RoPE-pair-interleaved stored row, gemm.hpp: qkv_perm_row)
int launch_adapter_b_aug(uint16_t* w_aug, int64_t ld, int64_t row0, int col_hi, int col_lo, const float* B, int N, int r, int row_mode, int dtype, hipStream_t s);

struct AdapterDownArgs {
    const uint16_t* A16[3];   // up to three adapters reading the same x (q, k, v); nullptr = absent (its u columns are written as zeros)
    int n;                    // segments (1 or 3)
};
// u[t, sg * r + j] = scale * sum_k x[t, k] * A_sg[j, k] with x = x_hi (+ x_lo at x16 + lo_off when lo_off > 0), written into the augmented columns
// of the SAME rows: columns K + [0, n r) = u_hi, K + [n r, 2 n r) = u_hi again (the B_lo columns' partner), the rest of the `aug` columns zero;
// with lo_off > 0 the lo half of the row (x16 + lo_off + K ..) gets u_lo = 16-bit(u - f32(u_hi)) in the same pattern.
int launch_adapter_down(uint16_t* x16, int64_t ldx, int64_t lo_off, int64_t T, int K, const AdapterDownArgs& a, int r, float scale, int aug, int dtype, hipStream_t s);
// dst [N, K + aug] <- [src [N, K] | 0]
int launch_make_aug(uint16_t* dst, const uint16_t* src, int64_t N, int K, int aug, hipStream_t s);
// dst [n, ldd] (first K columns) <- src [n, lds]; 16-bit rows
int launch_copy_rows16(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n, int K, hipStream_t s);

// ---- three-term compensated product of two f32-valued operands on 16-bit MFMAs: (a_hi + a_lo) . (w_hi + w_lo) ~ a_hi w_hi + a_hi w_lo + a_lo w_hi as ONE
// GEMM of depth 3 K: A rows [hi | hi | lo] against W rows [hi | lo | hi].  Used for the TVG logits (retrieval_utils.py:106: visual-head output . video vocabulary),
// whose W side -- the vocabulary of clip-mean features -- is data, not a 16-bit checkpoint tensor.
// dst [n, 3 K] 16-bit <- src f32 [n, K] (w_side = 0: [hi | hi | lo], 1: [hi | lo | hi]); dst1 (optional) [n, K] = hi alone
int launch_split3_f32(uint16_t* dst, uint16_t* dst1, const float* src, int64_t n, int K, int w_side, int dtype, hipStream_t s);
// dst [n, 3 K] = [hi | hi | lo] <- src 16-bit rows [hi | lo] of width 2 K (row stride lds)
int launch_split3_hilo(uint16_t* dst, const uint16_t* src, int64_t lds, int64_t n, int K, hipStream_t s);
