// Kernels of the fine-tuning step (training_utils.py:39-104, main.py:96-150) that the scoring path does not
// have: LoRA down-projection / gradient kernels, backward passes of RMSNorm / SwiGLU / RoPE / GELU / cross-entropy, the attention
// backward (batched MFMA products over materialised P / dS) and AdamW.  All tensors are packed tokens as in the scoring path.
#pragma once
#include "common.hpp"

// dst[c, map(r)] = src[r, c]  (16-bit).  mode 1: rows < rope_rows are q/k rows stored pair-interleaved inside each 128-row head
// (gemm.hpp: qkv_perm_row) and land at their natural column; other rows / mode 0: identity.
int launch_transpose16(uint16_t* dst, int64_t ldd, const uint16_t* src, int64_t lds, int64_t n_rows, int n_cols, int mode, int rope_rows, hipStream_t s);

// W_aug[stored_row(n), col0 + j] = 16-bit(B[n, j]) for n < N, j < r  (row_mode 1: natural row n of a q/k head -> its interleaved stored row)
int launch_lora_b_to_aug(uint16_t* w_aug, int64_t ld, int64_t row0, int col0, const float* B, int N, int r, int row_mode, int dtype, hipStream_t s);

// dst[n, k] = 16-bit( f32(base_aug[n, k]) + scale * sum_j Baug[n, col0 + j] ... ) -- merge for the scoring engine:
// dst [N, K] (row stride K) = base (first K columns of the augmented copy, row stride ld_aug) + scale * B[N, r] . A[r, K]   (stored row order of dst/base)
int launch_lora_merge(uint16_t* dst, const uint16_t* base_aug, int64_t ld_aug, int64_t row0, const float* B, const float* A, int N, int K, int r, float scale,
                      int row_mode, int dtype, hipStream_t s);

struct LoraDownArgs {
    const uint16_t* A16[3];   // up to three adapters reading the same x (q, k, v): 16-bit copy of A_j, [16, K] with rows >= r zero (launch_lora_a16)
    int n;                    // adapters
};
int launch_lora_a16(uint16_t* A16, const float* A, int K, int r, int dtype, hipStream_t s);
// u~[t, seg*r + j] = 16-bit( scale * sum_k drop_seg(x)[t, k] * A_seg[j, k] ), written into x16[t, K + seg*r + j]  (columns K.. of the augmented row)
int launch_lora_down(uint16_t* x16, int64_t ldx, int64_t T, int K, const LoraDownArgs& a, int r, float scale, float drop_p, uint64_t seed, uint32_t site, int dtype, hipStream_t s);
// dB[n, j] += sum_t dy[t, n] * u~[t, j]        (dy16 [T, ldy] columns n0.., u16 = x16 + K + seg*r)
// (scratch: lora_wgrad_scratch_bytes(T, N, r) -- the time splits' partials, summed in split order: no float atomics, reproducible bit for bit)
int launch_lora_dB(float* dB, const uint16_t* dy16, int64_t ldy, const uint16_t* u16, int64_t ldu, int64_t T, int N, int r, int dtype, float* scratch, hipStream_t s);
size_t lora_wgrad_scratch_bytes(int64_t T, int C, int r);
size_t lora_du_scratch_bytes(int64_t T, int N, int r);
// Bt16 [16, ldb] 16-bit <- transposed B [N, r] f32 (rows r.. and columns N.. must already be zero)
int launch_lora_bt(uint16_t* Bt16, int64_t ldb, const float* B, int N, int r, int dtype, hipStream_t s);
// du[t, j] = scale * sum_n dy[t, n] * B[n, j]   (N % 16 == 0: pad dy / Bt16 with zero columns)
int launch_lora_du(float* du, const uint16_t* dy16, int64_t ldy, const uint16_t* Bt16, int64_t ldb, int64_t T, int N, int r, float scale, int dtype, float* scratch, hipStream_t s);
// dA[j, k] += sum_t du[t, j] * drop(x)[t, k]
int launch_lora_dA(float* dA, const float* du, const uint16_t* x16, int64_t ldx, int64_t T, int K, int r, float drop_p, uint64_t seed, uint32_t site, int dtype, float* scratch,
                   hipStream_t s);
struct LoraDxArgs {
    const float* du[3];  // du_seg [T, r]
    const float* A[3];   // A_seg [r, K]
    int n;
};
// dx[t, k] += sum_seg keep_seg(t, k) / (1 - p) * sum_j du_seg[t, j] * A_seg[j, k]      (dx f32 [T, ldd]; adapter seg uses dropout site `site + seg`)
// out16 != nullptr: dx is only read and the sum is written as 16-bit rows (stride ldo) instead -- the cast that would follow, folded in
int launch_lora_dx(float* dx, int64_t ldd, const LoraDxArgs& a, int64_t T, int K, int r, float drop_p, uint64_t seed, uint32_t site, hipStream_t s, uint16_t* out16 = nullptr,
                   int64_t ldo = 0, int dtype = DT_F16);

// out[rows ? rows[i] : i, :] (+)= d/dx of  y = w * x * rsqrt(mean(x^2) + eps)  applied to dy[i, :]   (x row = rows ? rows[i] : i)
// out16 (optional, rows == nullptr only): 16-bit copy of the updated dx rows (the next GEMM's A operand)
// la (optional, rows == nullptr only): dy is taken as dy + the adapters' rank-r term of launch_lora_dx, formed on the fly (saves that pass)
int launch_rmsnorm_bwd(float* dx, const float* dy, const float* x, const int32_t* rows, int64_t n_rows, int H, const float* w, float eps, int accumulate, uint16_t* out16, int dtype,
                       hipStream_t s, const LoraDxArgs* la = nullptr, int r = 0, float drop_p = 0.f, uint64_t seed = 0, uint32_t site = 0);

int launch_f32_to_16(uint16_t* out, int64_t ldo, const float* in, int64_t ldi, int64_t rows, int cols, float scale, int dtype, hipStream_t s);
// h16[t, :H] = gelu(pre16[t, :H]) (exact erf)
int launch_gelu_fwd(uint16_t* h16, int64_t ldo, const uint16_t* pre16, int64_t rows, int H, int dtype, hipStream_t s);
// dpre16 = dh (f32) * gelu'(pre16)
int launch_gelu_bwd(uint16_t* dpre16, const float* dh, const uint16_t* pre16, int64_t rows, int H, int dtype, hipStream_t s);

// Cross-entropy over rows of f32 logits: loss[0] += sum_r -log_softmax(logits[r])[label[r]];  d = coef * (softmax - onehot) written as 16-bit
// (dl16, row stride ldd, columns V..ldd zeroed) or f32 (dl32, row stride ldd).  label per row = labels[r / label_div].
int launch_ce_fwd_bwd(const float* logits, int64_t ldl, int V, const int32_t* labels, int label_div, int64_t n_rows, float coef, uint16_t* dl16, float* dl32, int64_t ldd,
                      float* loss, int dtype, float* scratch, hipStream_t s);     // scratch: n_rows floats (per-row losses, summed in a fixed order)

// TVG head pieces (training_utils.py:76-79): vocab16 is clip-major [C][N][M]
int launch_tvg_dvh(float* dvh, const float* dl, const uint16_t* vocab16, int n_rows, int C, int N, int M, float scale, int dtype, hipStream_t s);   // dvh[bc, m] = scale * sum_n dl[bc, n] * vocab[c][n][m]
int launch_outer_acc(float* dW, const float* dvh, const uint16_t* h16, int64_t ldh, int n_rows, int M, int H, int dtype, hipStream_t s);             // dW[m, h] += sum_bc dvh[bc, m] * h16[bc, h]
int launch_rows_matmul(float* out, const float* dvh, const float* W, int n_rows, int M, int H, hipStream_t s);                                       // out[bc, h] = sum_m dvh[bc, m] * W[m, h]

// d embeds -> d projector outputs: token t with src_index[t] = -(f + 1): f < F -> dout_a[f, :] = 16-bit(dres[t, :]) (a VTG video token);
// f >= F -> dout_b[(f - F) * group + g, :] = 16-bit(dres[t, :] / group), g < group (a TVG clip token = the mean of `group` projector rows)
int launch_feat_grad(uint16_t* dout_a, uint16_t* dout_b, const float* dres, const int32_t* src_index, int64_t T, int H, int64_t F, int group, int dtype, hipStream_t s);

// ---- attention backward over materialised scores.  Sequences have no shared prefix; token of (s, i) = seq_start[s] + i.
struct AttnBwdParams {
    int dtype;
    const uint16_t* qkv;   // [T, ldq] saved q | k | v (RoPE applied)
    int64_t ldq;
    const uint16_t* dout;  // [T, ldo] d(attention output), q-head major
    int64_t ldo;
    const uint16_t* o16;   // [T, ldo16] the forward's attention output
    int64_t ldo16;
    const float* lse;      // [T, num_heads] the forward's log-sum-exp per row (attention.hpp: lse_out)
    int num_heads, num_kv_heads;
    const uint8_t* key_visible;
    const int32_t* seq_start; const int32_t* seq_len;
    int n_seqs, max_len;   // max_len = longest sequence of the batch
    float scale;
    float* D;              // workspace [T, num_heads]
    uint16_t* P16; uint16_t* dS16;   // workspaces [n_seqs][num_heads][Lm][Lm], Lm = round_up(max_len, 64)
    float* dqkv;           // out: f32 [T, ldq] gradient w.r.t. the post-RoPE q | k | v
};
int64_t attn_bwd_lm(int max_len);
int launch_attention_bwd(const AttnBwdParams& p, int64_t n_tokens, hipStream_t s);

// dqkv16[t, :] = 16-bit of the gradient w.r.t. the PRE-RoPE q | k | v: inverse rotation of the q/k columns (positions -> cos/sin tables [n_pos, 64])
int launch_rope_bwd(uint16_t* out16, const float* dqkv, int64_t T, int qkv_n, int rope_cols, const int32_t* pos, const float* cosb, const float* sinb, int n_pos, int dtype, hipStream_t s);

// AdamW (torch.optim.AdamW): g = grad * inv_scale; p *= 1 - lr*wd; m, v updates; p -= lr/c1 * m / (sqrt(v)/sqrt(c2) + eps)
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, float inv_scale, float c1, float c2, hipStream_t s);
// stats[0] += sum (g * inv_scale)^2 ; stats[1] = 1 if any g is inf / nan
int launch_grad_stats(const float* g, int64_t n, float inv_scale, float* stats, float* scratch, hipStream_t s);     // scratch: 1024 floats
