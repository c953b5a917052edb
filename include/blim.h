/* blim.h -- C ABI of the MI355X-native BLiM likelihood-scoring engine (libblim_hip.so).
 *
 * The reference (mlvlab/BLiM) is pure Python/PyTorch and has no FFI layer; its boundary for the hot path
 * is the duck-typed Python surface that retrieval_utils.py calls.  This header is
 * the C-ABI the host-side mirror of that surface (blim_amd/modeling.py, blim_amd/retrieval_utils.py)
 * binds with ctypes; every entry point names the reference code it replaces.  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative BLIM_ERR_* code and never throws;
 * blim_last_error() returns a message for the calling thread.  All data pointers are DEVICE pointers
 * (HIP) unless the parameter is documented as host; the caller owns every buffer passed in.  Calls are
 * asynchronous on the given hipStream_t (passed as void*; NULL = default stream); the caller synchronises.
 * An engine handle is not thread-safe; distinct handles are independent.  Buffers documented as "bf16" below hold raw
 * 16-bit values of the engine's compute_dtype (bf16 or f16).
 */
#ifndef BLIM_H_
#define BLIM_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BLIM_ABI_VERSION 9
#define BLIM_ERR_ARG (-1)
#define BLIM_ERR_HIP (-2)
#define BLIM_ERR_STATE (-3)
#define BLIM_ERR_NOMEM (-4)

#define BLIM_DTYPE_F32 0
#define BLIM_DTYPE_BF16 1

/* 16-bit compute format of an engine: activations handed across the ABI ("h16" below), the engine's weight copy and the
 * MFMA operand type.  F16 is what the reference itself runs on GPU (main.py:97 .half(), training_utils.py:142 autocast
 * float16) and the default of the Python host; BF16 has the same MFMA rate and an 8-bit mantissa (it holds the 1e-3 bar in the compensated
 * mode, option "precise").  Range: an fp16 engine's 16-bit activation STORES saturate at +-65504 instead of overflowing to inf (the f32
 * residual stream keeps f32's range; NaN / inf operands still propagate) -- where the reference's `.half()` forward turns a SwiGLU product
 * beyond 65504 into NaN scores, the engine returns finite ones; a bf16 engine has f32's range and reproduces the fp32 result
 * (tests/test_gpu_parity.py::test_activations_beyond_fp16_range, tests/golden/saturation.npz recorded from the reference). */
#define BLIM_COMPUTE_BF16 0
#define BLIM_COMPUTE_F16 1
/* fp8 mode (BASELINE.json config 5, "fp8 weights on CDNA4 fp8 MFMA"): q/k/v, o, gate|up, down and
 * lm_head weights are quantised at load to OCP e4m3 with one f32 scale per output row, their input activations per token
 * (scale = absmax / 448; the attention and SwiGLU outputs: one power-of-two E8M0 scale per token and 128 values, written by their
 * producers and fed to the MFMA as its A-side block scale), and multiplied on the block-scaled fp8 MFMA (the f32 scales applied to
 * the f32 accumulator).  Everything else -- attention, RoPE, residual stream, norms, projector, visual head, criteria -- and
 * every 16-bit buffer crossing the ABI is fp16 as with BLIM_COMPUTE_F16.  Scores differ from the fp32 reference at the 1e-2
 * level (reported, not asserted to 1e-3). */
#define BLIM_COMPUTE_F8 2

typedef struct blim_engine blim_engine;

/* Model dimensions: Qwen2Config + mm_* fields of the checkpoint the reference loads (main.py:96-97). */
typedef struct blim_config {
    int32_t vocab_size, hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads;
    int32_t mm_hidden_size; /* projector input width (1024) */
    int32_t num_clips;      /* --num_clips, main.py:59 (4) */
    int32_t max_positions;  /* RoPE table length */
    int32_t compute_dtype;  /* BLIM_COMPUTE_BF16 / BLIM_COMPUTE_F16 / BLIM_COMPUTE_F8 */
    float rms_eps, rope_theta;
} blim_config;

/* A batch of packed token sequences.  Sequence s owns tokens [seq_start[s], seq_start[s]+seq_len[s]) of the
 * packed arrays and may name a shared prefix [pfx_start[s], +pfx_len[s]) (another sequence's tokens, e.g. the
 * video+prompt prefix shared by all text candidates of one query).  Token i of s attends to every prefix token
 * and to own tokens 0..i, restricted to key_visible != 0 (causal AND key-padding AND CPN mask of
 * modeling_qwen2_flash.py:1025-1040 / modeling_videochat_flash.py:409-433).  blk_* list the 32-query blocks:
 * for every sequence, q0 = 0, 32, 64, ... < seq_len. */
typedef struct blim_batch {
    int64_t n_tokens;
    int32_t n_seqs;
    int32_t n_blocks;
    const int32_t* positions;   /* [n_tokens] RoPE position (modeling_qwen2_flash.py:998-1003); every value < blim_config.max_positions
                                 * (device data: not checked by the library, the Python host checks it when it packs the batch) */
    const uint8_t* key_visible; /* [n_tokens] */
    const int32_t* seq_start;   /* [n_seqs] */
    const int32_t* seq_len;     /* [n_seqs] */
    const int32_t* pfx_start;   /* [n_seqs] */
    const int32_t* pfx_len;     /* [n_seqs] (0 = no prefix) */
    const int32_t* blk_seq;     /* [n_blocks] */
    const int32_t* blk_q0;      /* [n_blocks] */
    const int32_t* own_start;   /* [n_tokens] or NULL (ABI v7).  Token i of a sequence attends to its own tokens own_start[i] .. i instead of 0 .. i
                                 * (index inside the sequence): several short continuations of ONE prefix -- the three clip tokens of every candidate
                                 * video of a text, retrieval_utils.py:99-107 -- are packed into one sequence whose segments do not see each other, so the
                                 * 32-query attention blocks are dense instead of holding 3 queries each.  NULL = 0 everywhere (plain causal). */
} blim_batch;

int blim_abi_version(void);
const char* blim_last_error(void);

/* ---- lifetime.  Replaces VideoChatFlashQwenForCausalLM.from_pretrained(...).half().to(device), main.py:96-97. */
int blim_create(const blim_config* cfg, blim_engine** out);
void blim_destroy(blim_engine* e);

/* ---- weights.  `name` is a canonical tensor name (blim_amd/synth.py:weight_shapes); `data` holds the tensor in
 * the checkpoint's natural [out, in] layout as f32 or bf16, on host (on_device = 0) or device.  The engine keeps
 * its own bf16 copy in kernel layout (fused q|k|v with RoPE pair interleave, interleaved gate|up).
 * Replaces load_state_dict / util/misc.py:303-311. */
int blim_load_weight(blim_engine* e, const char* name, const void* data, int32_t dtype, int32_t on_device);
/* Seeded synthetic weights generated on device by the rule of blim_amd/synth.py (no checkpoint available offline). */
int blim_init_synthetic_weights(blim_engine* e, uint64_t seed);
/* 0 when every tensor has been loaded, BLIM_ERR_STATE (message lists a missing tensor) otherwise. */
int blim_weights_ready(const blim_engine* e);

/* ---- LoRA adapters of a fine-tuned checkpoint, KEPT APART as the reference keeps them.  Replaces main.py:96-105 (peft get_peft_model on the
 * projector `mlp` / `tvg_mlp` Linear "0" / "2", on every q/k/v/o_proj and on lm_head) + main.py:125-128 (load_state_dict of the resume file): the
 * reference evaluates y = W x + b + (alpha / r) B (A x) with A, B as separate matrices; so does the engine for every `weight_name` an adapter was
 * loaded for -- the rank-r term rides in the base product's accumulation (K-augmented operands [x | u] . [W | B]^T, u = (alpha / r) A x, 64 extra K
 * columns), W stays exactly the checkpoint's value and A, B, u travel as hi + lo 16-bit pairs.  `A` [lora_r, in] and `B` [out, lora_r] are HOST
 * float32 arrays in peft's layout (lora_A.weight / lora_B.weight); weight_name is the canonical name of the adapted weight
 * ("layers.<i>.q_proj.w", "lm_head", "mlp.0.w", "tvg_mlp.2.w", ...).  All adapters of an engine share lora_r (<= 16) and lora_alpha (one LoraConfig,
 * main.py:96-101).  Base weights and adapters may be loaded in any order; the augmented weight copies are rebuilt on the next call after either
 * changes.  On an fp8 engine the adapted projections (q/k/v/o, lm_head) run in fp16 once adapters are loaded (the MLP, not adapted, stays e4m3).
 * The alternative -- folding W + (alpha / r) B A into the engine's 16-bit weight on the host before blim_load_weight (blim_amd/checkpoint.py,
 * lora_mode = "merge") -- needs no entry point; it rounds the sum to the engine's format.  An engine whose base weights
 * received a merged update from blim_train_merge refuses adapters (BLIM_ERR_STATE: the update would apply twice) until base weights are loaded again. */
int blim_load_adapter(blim_engine* e, const char* weight_name, const float* A, const float* B, int32_t lora_r, float lora_alpha);
/* Drops every loaded adapter (the engine scores with the base weights again). */
int blim_clear_adapters(blim_engine* e);
/* Number of adapters currently loaded (negative on a NULL engine). */
int blim_num_adapters(blim_engine* e);

/* Pre-size workspaces (optional; they grow on demand, which synchronises the device).  On an engine in the compensated mode (option "precise" = 1 at the time of the
 * call: blim_amd/engine.py's reserve(compensated=True) brackets the call with it) with option "precise_lo6" and its weights in place, this call also builds the
 * weights' e2m3 images and the tile workspaces: running out of device memory is reported here (BLIM_ERR_NOMEM, the matrix named) instead of inside the first
 * compensated scoring call.  Plain calls never build or read the images. */
int blim_reserve(blim_engine* e, int64_t max_tokens, int64_t max_rows);

/* ---- K1: projector.  feats bf16 [n_rows, mm_hidden] -> out bf16 [n_rows, hidden]; which = 0 `mlp`, 1 `tvg_mlp`.
 * Replaces ToMe16_mlp_hd64.forward(video_feature=True), mm_projector_builder.py:156-159. */
int blim_project_video(blim_engine* e, const void* feats, int64_t n_rows, int32_t which, void* out, void* stream);
/* mean over groups of `group` consecutive rows (TVG clip tokens, modeling_videochat_flash.py:243). */
int blim_group_mean(blim_engine* e, const void* in, int64_t n_out, int32_t group, void* out, void* stream);

/* ---- K2: sequence assembly.  out[t] = src_index[t] >= 0 ? embed_tokens[src_index[t]] : feats[-(src_index[t]+1)].
 * Replaces embed_tokens + the splice of modeling_videochat_flash.py:388-440. */
int blim_assemble(blim_engine* e, const int32_t* src_index, int64_t n_tokens, const void* feats, void* out_embeds, void* stream);

/* ---- K3-K9: Qwen2 decoder over a packed batch; writes the final-norm hidden state of rows out_rows[0..n_out)
 * (out_rows = NULL: all tokens) as bf16 and/or f32 (either may be NULL).
 * Replaces Qwen2Model_Flash.forward, modeling_qwen2_flash.py:952-1156. */
int blim_decode(blim_engine* e, const blim_batch* b, const void* embeds, const int32_t* out_rows, int64_t n_out,
                void* out_hidden_bf16, float* out_hidden_f32, void* stream);

/* ---- K10+K11: log P(label | hidden) without materialising logits: logprob[r] = log_softmax(lm_head(hidden[r]))[labels[r]],
 * 0 where labels[r] < 0.  Replaces lm_head + .float() (modeling_qwen2_flash.py:1452-1453) + CrossEntropyLoss of
 * VTGCriterion (retrieval_utils.py:23-31). */
int blim_vtg_logprobs(blim_engine* e, const void* hidden_bf16, const int32_t* labels, int64_t n_rows, float* logprob, void* stream);
/* score[p] = sum(logprob[row_start[p] .. row_start[p+1])) / d, d = count_nonzero(...) for mode 0 (VTGCriterion,
 * retrieval_utils.py:32-33) or the row count for mode 1 (TVGCriterion's mean, :42).  e may be NULL. */
int blim_segment_mean(blim_engine* e, const float* logprob, const int32_t* row_start, int32_t n_pairs, int32_t mode, float* score, void* stream);

/* Literal path: logits f32 [n_rows, vocab] = lm_head(hidden) and the criterion on materialised logits. */
int blim_lm_head(blim_engine* e, const void* hidden_bf16, int64_t n_rows, float* logits, void* stream);
/* (e may be NULL for blim_ce_rows) */
int blim_ce_rows(blim_engine* e, const float* logits, int64_t ld, int32_t n_cols, const int32_t* labels, int64_t n_rows, float* logprob, void* stream);

/* ---- K13: visual_head, bf16 [n_rows, hidden] -> bf16 [n_rows, mm_hidden] (modeling_videochat_flash.py:598-599). */
int blim_visual_head(blim_engine* e, const void* hidden_bf16, int64_t n_rows, void* out_bf16, void* stream);
/* The same on FLOAT32 hidden states with a float32 result (device pointers): the head -- an fp32 tensor of the resume file, main.py:104-107 -- and the
 * hidden rows both enter as hi + lo 16-bit operands (three-term product).  What the literal path's forward_visual calls on 16-bit engines. */
int blim_visual_head_f32(blim_engine* e, const float* hidden_f32, int64_t n_rows, float* out_f32, void* stream);
/* ---- K14+K15: vh bf16 [n_pairs, clips, mm_hidden]; vocab bf16 [clips, n_vocab, mm_hidden] (clip-major copy of
 * video_vocab); score[p] = mean_c log_softmax_n(vh[p,c].vocab[c,n] / sqrt(mm_hidden))[labels[p]].
 * Replaces the bmm + TVGCriterion of retrieval_utils.py:106-107, 40-43. */
int blim_tvg_scores(blim_engine* e, const void* vh_bf16, const void* vocab_bf16, int32_t n_vocab, const int32_t* labels,
                    int32_t n_pairs, float* score, void* stream);
/* the logits alone: f32 [n_pairs, clips, n_vocab] (literal path, retrieval_utils.py:106) */
int blim_tvg_logits(blim_engine* e, const void* vh_bf16, const void* vocab_bf16, int32_t n_vocab, int32_t n_pairs, float* logits, void* stream);
/* Registers the video vocabulary (model.module.set_video_vocab, modeling_videochat_flash.py:589-590; retrieval_utils.py:209): `vocab_f32` = DEVICE float32
 * [num_clips][n_vocab][mm_hidden] (clip-major), the clip-mean features of every test video (base_dataset.py:33-37).  The engine keeps it as hi + lo 16-bit
 * operands; blim_tvg_logits / blim_tvg_scores / blim_score_tvg called with vocab_bf16 == NULL then score against it, and in the compensated mode (option
 * "precise": every TVG call of a 16-bit engine) the logits are the three-term product (vh_hi + vh_lo) . (v_hi + v_lo) -- a 16-bit vocabulary alone bounded the
 * bf16 engine's TVG scores at 2 - 7e-4 of the fp32 reference. */
int blim_set_video_vocab(blim_engine* e, const float* vocab_f32, int32_t n_vocab, void* stream);
/* retrieval_utils.py:104-106 for the literal path: logits [n_pairs, num_clips, n_vocab] of FLOAT32 visual-head outputs vh_f32 [n_pairs * num_clips, mm_hidden]
 * (device) against the registered vocabulary, three-term compensated product, / sqrt(mm_hidden). */
int blim_tvg_logits_f32(blim_engine* e, const float* vh_f32, int32_t n_pairs, float* logits, void* stream);

/* ---- Fused scoring: decode + head + criterion in one call.
 * VTG: rows[r] = packed token whose hidden state predicts labels[r]; pair p owns rows [row_start[p], row_start[p+1]). */
int blim_score_vtg(blim_engine* e, const blim_batch* b, const void* embeds, const int32_t* rows, const int32_t* labels,
                   int64_t n_rows, const int32_t* row_start, int32_t n_pairs, float* score, void* stream);
/* TVG: rows[p*clips + c] = packed token whose hidden state predicts clip c of pair p. */
int blim_score_tvg(blim_engine* e, const blim_batch* b, const void* embeds, const int32_t* rows, const void* vocab_bf16,
                   int32_t n_vocab, const int32_t* labels, int32_t n_pairs, float* score, void* stream);

/* ---- Literal model.forward(inputs_embeds=[B,L,H] bf16, attention_mask=[B,L] u8) -> logits f32 [B,L,V] (may be NULL),
 * hidden f32 [B,L,H] (may be NULL).  Replaces VideoChatFlashQwenForCausalLM.forward, modeling_videochat_flash.py:601-629. */
int blim_forward(blim_engine* e, const void* embeds, const uint8_t* mask, int32_t B, int32_t L, float* logits, float* hidden, void* stream);

/* ---- synthetic data + plain GEMM (bench / tests) */
int blim_fill_bell_bf16(void* out, int64_t n, uint64_t seed, const char* name, float std, float mean, void* stream);
int blim_fill_bell_f32(float* out, int64_t n, uint64_t seed, const char* name, float std, float mean, int32_t round_bf16, void* stream);
/* C [M, ldc] = A [M, lda] . W [N, K]^T, all bf16 (resp. f16) */
int blim_gemm_f16(const void* A, int64_t lda, const void* W, int32_t M, int32_t N, int32_t K, void* C, int64_t ldc, void* stream);
int blim_gemm_bf16(const void* A, int64_t lda, const void* W, int32_t M, int32_t N, int32_t K, void* C, int64_t ldc, void* stream);
/* The compensated GEMM of fp16 engines as a building block (tests; option "precise_lo6"): A_hilo [M, 2 K] fp16 rows [hi | lo], W [N, K] fp16 ->
 * C f32 [M, N] = hi . W^T (fp16 MFMA) + e2m3(lo) . e2m3(W)^T (block-scaled MFMA: e2m3 values with one power-of-two scale per 32), one kernel, one set of
 * accumulators.  a6 / w6 (blim_f6_tiles_bytes(M, K) / (N, K) bytes) receive the e2m3 operand tiles of the lo part and of W in the layout the kernel stages:
 * per (256-row tile, 128-value K-step) 24 KiB of packed 6-bit values in MFMA-lane order + 1 KiB of E8M0 scale bytes (csrc/gemm.hpp).  K % 128 == 0. */
int64_t blim_f6_tiles_bytes(int64_t n_rows, int32_t K);
int blim_gemm_f16_lo6(const void* A_hilo, const void* W, int32_t M, int32_t N, int32_t K, void* a6, void* w6, float* C, void* stream);
/* fp8 building blocks (tests / bench): per-row e4m3 quantisation of a 16-bit matrix (dtype16 = BLIM_COMPUTE_BF16 / _F16;
 * out8 [n_rows, K] bytes, scale [n_rows] = absmax / 448, 1 for an all-zero row), and
 * C f16 [M, ldc] = (A8 [M, lda] . W8 [N, K]^T) * a_scale[m] * w_scale[n] on the block-scaled fp8 MFMA.  K % 128 == 0. */
int blim_quant_rows(const void* in16, int64_t ld, int64_t n_rows, int32_t K, int32_t dtype16, void* out8, float* scale, void* stream);
int blim_gemm_f8(const void* A8, int64_t lda, const float* a_scale, const void* W8, const float* w_scale, int32_t M, int32_t N, int32_t K,
                 void* C, int64_t ldc, void* stream);

/* ---- per-kernel-class timing (hipEvents on the launch stream).  Classes: see blim_timing_class_name. */
int blim_timing_enable(blim_engine* e, int32_t on);
int blim_timing_num_classes(void);
const char* blim_timing_class_name(int32_t cls);
/* Synchronises the recorded events; ms[c] = total ms, calls[c] = launches, flops[c] = FLOPs issued (host arrays of
 * blim_timing_num_classes() entries); then clears the record. */
int blim_timing_report(blim_engine* e, double* ms, int64_t* calls, double* flops);

/* Bring-up aid: copies the first `bytes` bytes of an internal workspace ("resid" f32 [T,H], "xn" bf16 [T,H],
 * "qkv" bf16 [T,(nh+2nkv)*128], "attn" bf16 [T,H], "act" bf16 [T,I]) as the last blim_decode left it. */
int blim_debug_read(blim_engine* e, const char* which, void* dst, int64_t bytes, void* stream);

/* Bring-up aid: when non-NULL, every GEMM workgroup writes s_memrealtime stamps {entry, main loop start, main loop end, exit,
 * C staged in LDS, stores issued} to device_buf[workgroup*8 ..] (u64, 100 MHz); NULL turns it off. */
int blim_debug_gemm_stamps(void* device_buf);

/* "precise" (0/1, fp16 and bf16 engines; fp8 engines refuse it): compensated arithmetic for the following calls (activations as hi + lo,
 *   GEMMs walk K twice): ~21 significant bits per activation on fp16 engines, 16 on bf16 engines (against exact bf16 weights: VTG 2e-6 / TVG
 *   <= 7e-4 from the fp32 reference at 28 layers of the 7B configuration).  The host turns it on for the TVG calls of both dtypes and for the
 *   VTG calls of bf16 engines -- the mode in which BASELINE.json's named dtype holds the 1e-3 bar (blim_amd/modeling.py: vtg_precise);
 * "precise_embeds" (0/1): in precise mode the input embeddings / projector outputs are [hi | lo] rows of width 2 * hidden as well;
 * "precise_mlp" (0/1, default 1): 0 leaves the MLP branch plain in precise mode -- the TVG calls' "attn" mode (1.6x faster than fully compensated; TVG deviation at
 *   7B depth 8e-4 instead of 4e-5 on Gaussian weights, 3e-3 on weights with massive activations: `--tvg_precise auto` measures which one a checkpoint needs);
 * "precise_lo6" (0/1; 16-bit engines with hidden / intermediate sizes that are multiples of 128.  fp16 engines: default 1, env BLIM_PRECISE_LO6=0 turns it off.  bf16 engines
 *   (round 6): default 0 -- their parity mode walks K a second time in bf16, 1 - 3e-6 at 7B depth -- and 1 is an opt-in (env BLIM_PRECISE_LO6=1, `--second_pass e2m3 | auto`):
 *   a bf16 value's lo part is 2^-9 of it, hi + e2m3(lo) carry about what one fp16 rounding keeps: VTG scores 3 - 7e-5 from the fp32 reference at 7B depth on N(0, 0.02^2)
 *   weights at 1.41x the rate of the bf16 second pass.  Set it before loading adapters (their K extension is 128 columns when the option may be used).  fp8 engines refuse 1):
 *   in precise mode the decoder GEMMs' (and lm_head's) second walk over K -- the product of W with the activations' LO parts, 2^-11 of the values -- runs on the
 *   block-scaled MFMA with e2m3 operands (6 bits, one power-of-two scale per 32 values: four times the 16-bit MFMA rate on gfx950), inside the same kernel and into
 *   the same accumulators.  Weights: e2m3 tile images built on the first compensated call (+0.78 byte per decoder / head weight: 5.9 GB at 7B; a failed allocation
 *   is BLIM_ERR_NOMEM with the matrix named); activations: the lo parts a producer wrote are re-written as operand tiles (one HBM-bound pass per GEMM input).  A
 *   fully compensated call costs 1.49x a plain one instead of 2x and stays within 4e-5 of the fp32 reference where the fp16 second pass reads 4e-6 (28 layers of
 *   the 7B configuration); on weights with a trained checkpoint's massive activations the two differ by <= 1.6e-4 over 16,000 scores (rms 9e-6);
 * "masked_query_zero" (0/1, default 0; 16-bit engines; PARITY-UNPINNED): query positions the key mask hides (blim_batch.key_visible == 0) write a ZERO attention output
 *   instead of attending to their visible keys.  The default is the semantics parity is pinned to -- the reference's eager / SDPA attention classes
 *   (modeling_qwen2_flash.py:288-310, 701-709), where a masked query row is computed like any other.  Its flash-attention-2 class drops such positions before the
 *   kernel and pads zeros back (modeling_qwen2_flash.py:526-563); main.py:96 passes no attn_implementation, so which class a real run used depends on the
 *   checkpoint's config and on whether flash-attn was installed (setup.sh:7).  The only scores that differ are the TVG-CPN prior's (its first gathered row is a
 *   masked position: modeling_videochat_flash.py:414-417).  flash-attn cannot be imported in the build environment: this option restates those lines, it is not
 *   checked against a recorded run (tests/test_gpu_parity.py::test_masked_query_zero_option_matches_its_restatement);
 * "prune_last" (0/1, default 1): calls that name the rows they read (blim_decode with out_rows, blim_score_*) run the LAST layer's o_proj / norm / MLP
 *   on those rows only (same values bit for bit; the other rows' K / V are still produced); after such a call the "resid" / "attn" / "act" workspaces of
 *   blim_debug_read hold the last layer's state of the live rows only -- bring-up code reads them after calls without out_rows, or sets 0;
 * "f8_fuse" (0/1, fp8 engines): quantise the attention / SwiGLU outputs inside their producers (default 1);
 * tuning switches: "attn_tr_read" (0/1); "f8_mask" (BLIM_COMPUTE_F8 engines: which GEMMs take fp8 operands, bit 0 qkv, 1 o_proj,
 * 2 gate|up, 3 down, 4 lm_head; default 31 = all; the others run in fp16 from the retained 16-bit weights) */
int blim_set_option(blim_engine* e, const char* key, int32_t value);

/* ------------------------------------------------------------------------------------------------------------------
 * Offline feature extraction: the UMT-L vision encoder + ToMe token merging that produce the
 * ./data/<DS>/features/<vid>.pth files the scoring path reads.  Replaces model.encode_video_image(video, ...,
 * return_video_feature=True) as extract.py:104 calls it (modeling_videochat_flash.py:126-181 -> UMTVisionTower.forward,
 * vision_tower_builder.py:558-571 -> ToMe16_mlp_hd64.forward(compress=True, local_num_frames=4, return_video_feature=True),
 * mm_projector_builder.py:134-154).  A separate handle: extraction needs no language-model weights. */
typedef struct blim_vision blim_vision;
typedef struct blim_vision_config {
    int32_t image_size;    /* 448 ("umt-hd", vision_tower_builder.py:613-614) */
    int32_t patch_size;    /* 16 */
    int32_t num_frames;    /* frames per clip = mm_local_num_frames (4) */
    int32_t hidden_size;   /* 1024 */
    int32_t num_heads;     /* 16 (head_dim must be 64) */
    int32_t mlp_hidden;    /* 4096 */
    int32_t depth;         /* blocks actually run: encoder_depth 24 + mm_vision_select_layer (-2) + 1 = 23 (vision_tower_builder.py:293) */
    int32_t tome_tokens;   /* tokens per clip after merging: 16 * num_frames = 64 (mm_projector_builder.py:147) */
    int32_t compute_dtype; /* BLIM_COMPUTE_F16 / BLIM_COMPUTE_BF16: format of the GEMM / attention operands (residual stream, LayerNorm and ToMe are f32) */
} blim_vision_config;
int blim_vision_create(const blim_vision_config* cfg, blim_vision** out);
void blim_vision_destroy(blim_vision* v);
/* names: vit.patch.{w,b}, vit.blocks.N.{norm1,norm2}.{w,b}, .q_bias, .v_bias, .qkv.w, .proj.{w,b}, .fc1.{w,b}, .fc2.{w,b}, vit.norm.{w,b}
 * (blim_amd/vision.py:vision_weight_shapes); natural [out, in] layout, f32 or bf16, host or device. */
int blim_vision_load_weight(blim_vision* v, const char* name, const void* data, int32_t dtype, int32_t on_device);
int blim_vision_init_synthetic_weights(blim_vision* v, uint64_t seed);
/* HOST f32 [num_frames * (image_size/patch_size)^2, hidden]: the sinusoidal position table (vision_tower_builder.py:222-269),
 * computed by the host (blim_amd/vision.py:pos_embed). */
int blim_vision_set_pos_embed(blim_vision* v, const float* table_host);
int blim_vision_ready(const blim_vision* v);
/* frames: device 16-bit (compute dtype) [n_clips, num_frames, 3, S, S], normalised pixels.  out_feat (may be NULL): f32
 * [n_clips, L, hidden] encoder output, L = num_frames * (S/patch)^2; out_tome (may be NULL): f32 [n_clips, tome_tokens, hidden]. */
int blim_vision_encode(blim_vision* v, const void* frames, int32_t n_clips, float* out_feat, float* out_tome, void* stream);
/* The encoder's attention alone (tests / bench): qkv 16-bit [n_clips * L, 3 * heads * 64] = [q | k | v] per token, heads of 64; out 16-bit
 * [n_clips * L, heads * 64] = softmax(q k^T / 8) v within each clip, non-causal (vision_tower_builder.py:100-128). */
int blim_vit_attention(const void* qkv, int32_t n_clips, int32_t L, int32_t heads, int32_t dtype16, void* out, void* stream);
/* ToMe alone: x f32 [b, p, c] (c = heads * 64) -> out f32 [b, target, c]; bipartite soft matching + size-weighted merge,
 * mm_projector_builder.py:6-130. */
int blim_tome_merge(blim_vision* v, const float* x, int32_t b, int32_t p, int32_t c, int32_t heads, int32_t target, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fine-tuning step: what training_utils.py:57-95 does per batch -- the reference's two decoder forwards (VTG rows, TVG rows;
 * here one forward over a packed batch holding both), loss = vtg_loss + tvg_loss, backward into the LoRA adapters of main.py:96-101 (projector mlp / tvg_mlp Linear 0 and 2, every
 * q/k/v/o_proj, lm_head; y = W x + b + alpha/r * B A dropout(x)) and the fp32 visual_head (main.py:104-107), AdamW (main.py:147).
 * Base weights stay frozen in the engine.  The trainable tensors live in ONE flat f32 device buffer owned by the caller (so that the
 * host can all-reduce the matching gradient buffer over RCCL in one call, as DistributedDataParallel does for the reference,
 * main.py:141-143); tensor order: blim_amd/lora.py:trainable_names, every tensor starting on a 64-element boundary. */
typedef struct blim_trainer blim_trainer;
typedef struct blim_train_config {
    int32_t lora_r;       /* main.py --lora_r (8); <= 16 */
    float lora_alpha;     /* --lora_alpha (32) */
    float lora_dropout;   /* --lora_drop (0.05): dropout on the adapters' input, counter-based mask per (step seed, adapter, token, column) */
} blim_train_config;
/* number of f32 elements of the flat parameter / gradient / moment buffers */
int64_t blim_train_flat_size(const blim_engine* e, int32_t lora_r);
/* name = "<weight>:A" | "<weight>:B" | "visual_head" (weights: mlp.{0,2}.w, tvg_mlp.{0,2}.w, lm_head, layers.N.{q,k,v,o}_proj.w) */
int blim_train_param_offset(const blim_engine* e, int32_t lora_r, const char* name, int64_t* offset, int64_t* rows, int64_t* cols);
/* params / grads: device f32 [blim_train_flat_size].  Builds the training copies of the frozen weights (K-augmented copies that
 * carry the adapters' B matrices as 64 extra K columns, and transposed copies for the input-gradient GEMMs). */
/* (An fp8 engine accepts a trainer for loading and MERGING adapters only -- evaluating a fine-tuned checkpoint in fp8 mode; blim_train_step refuses it.) */
int blim_train_create(blim_engine* e, const blim_train_config* cfg, float* params, float* grads, blim_trainer** out);
void blim_train_destroy(blim_trainer* t);
/* after `params` changed (load, optimizer step): refresh the 16-bit copies the forward reads */
int blim_train_sync_params(blim_trainer* t, void* stream);
/* write W + alpha/r * B A (and visual_head) into the ENGINE's scoring weights: evaluation between epochs (main.py:166) sees the
 * fine-tuned model; always merged from the pristine base, so it can be called repeatedly.  BLIM_ERR_STATE while the engine holds adapters apart
 * (blim_load_adapter): one or the other. */
int blim_train_merge(blim_trainer* t, void* stream);
typedef struct blim_train_batch {
    const blim_batch* batch;      /* ONE packed batch holding the VTG rows and the TVG rows (either part may be absent), no shared prefixes */
    const int32_t* src_index;     /* [T] token id >= 0, or -(f + 1): f < n_feat_rows = row f of the `mlp` projection (a VTG row's video token),
                                   * f >= n_feat_rows = clip mean f - n_feat_rows of the `tvg_mlp` projection (a TVG row's clip token) */
    const void* feats;            /* 16-bit [n_feat_rows, mm_hidden]: raw features of the batch's videos, video-major */
    int64_t n_feat_rows;
    int32_t tok_per_clip;         /* rows averaged per clip for the TVG tokens (modeling_videochat_flash.py:243) */
    int32_t max_seq_len;          /* longest row of the batch */
    const int32_t* rows;          /* VTG: [n_rows] token rows whose next-token label is scored */
    const int32_t* labels;        /* VTG: [n_rows] target token ids */
    int64_t n_rows;
    const int32_t* tvg_rows;      /* TVG: [n_samples * num_clips] rows predicting clip c (training_utils.py:73) */
    const int32_t* tvg_labels;    /* TVG: [n_samples] index of the sample's video in the vocabulary */
    int64_t n_tvg_rows;
    const void* vocab;            /* TVG: 16-bit clip-major [num_clips][n_vocab][mm_hidden] */
    int32_t n_vocab;
    float grad_scale;             /* upstream gradient of both losses (AMP loss scale / accum_iter) */
    uint64_t dropout_seed;
} blim_train_batch;
/* training_utils.py:57-85 for one batch: forward of both kinds of rows through the decoder in one pass, loss = vtg_loss + tvg_loss, backward;
 * gradients ACCUMULATE into `grads` (scaled by grad_scale); loss_sums[0] += the summed negative log-likelihood over the n_rows VTG label rows,
 * loss_sums[1] += the same over the n_tvg_rows TVG rows (device f32 [2]; the mean losses of training_utils.py:68 / :79 divide by the row counts) */
/* Reproducible bit for bit from run to run, like the reference's autograd on these shapes: every split reduction (adapter gradients over time
 * splits, du over column slices, the loss sums, the gradient norm) is summed from per-split partials in a fixed order -- no float atomics. */
int blim_train_step(blim_trainer* t, const blim_train_batch* b, float* loss_sums, void* stream);
/* stats[0] += sum((g * inv_scale)^2) over the flat gradient buffer, stats[1] = 1 if any element is inf / nan (device f32 [2]) */
int blim_train_grad_stats(blim_trainer* t, float inv_scale, float* stats, void* stream);
/* torch.optim.AdamW over the flat buffers (g = grad * inv_scale; decoupled weight decay on every tensor: all are 2-D, timm's
 * param_groups_weight_decay exempts only 1-D tensors); step counts from 1; calls blim_train_sync_params */
int blim_train_adamw(blim_trainer* t, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay, float inv_scale,
                     int32_t step, void* stream);
/* bring-up aid: copies a saved activation of the last forward ("res<l>" f32 [T,H], "qkv<l>", "attn<l>", "gu<l>", "dres" f32 [T,H]) */
int blim_train_debug_read(blim_trainer* t, const char* which, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BLIM_H_ */
