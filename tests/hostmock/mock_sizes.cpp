// TEST INFRASTRUCTURE: the three host-side sizing helpers of csrc/train_kernels.hip (a file of kernels, not compiled for the host), restated for the sanitizer build:
// train_kernels.hip:233 (lora_wgrad_scratch_bytes), :385 (lora_du_scratch_bytes: an upper bound here -- every slice), :699 (attn_bwd_lm).
#include "train.hpp"
size_t lora_wgrad_scratch_bytes(int64_t T, int C, int r) { return (size_t)((T + 1023) / 1024) * C * r * 4; }
size_t lora_du_scratch_bytes(int64_t T, int N, int r) { return (size_t)((N + 15) / 16) * T * r * 4; }
int64_t attn_bwd_lm(int max_len) { return (max_len + 63) / 64 * 64; }
