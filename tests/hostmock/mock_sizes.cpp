// TEST INFRASTRUCTURE: the host-side sizing / capability helpers of csrc/train_kernels.hip and csrc/kernels.hip (files of kernels, not compiled for the host), restated for the sanitizer build:
// train_kernels.hip:233 (lora_wgrad_scratch_bytes), :385 (lora_du_scratch_bytes: an upper bound here -- every slice), :699 (attn_bwd_lm).
#include "train.hpp"
size_t lora_wgrad_scratch_bytes(int64_t T, int C, int r) { return (size_t)((T + 1023) / 1024) * C * r * 4; }
size_t lora_du_scratch_bytes(int64_t T, int N, int r) { return (size_t)((N + 15) / 16) * T * r * 4; }
int64_t attn_bwd_lm(int max_len) { return (max_len + 63) / 64 * 64; }
#include "kernels.hpp"
bool rmsnorm_can_write_tiles(int H, int64_t ldx, int64_t ldo) { if (ldo == 0) ldo = H; return H % 128 == 0 && H <= 4096 && H > 256 && ldo % 8 == 0 && ldx % 4 == 0; }   // kernels.hip
