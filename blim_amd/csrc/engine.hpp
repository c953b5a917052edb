// Engine state shared by engine.hip (scoring path) and train.hip (fine-tuning step).
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/blim.h"
#include "adapters.hpp"
#include "attention.hpp"
#include "common.hpp"
#include "gemm.hpp"
#include "kernels.hpp"

#define TRY(expr)                 \
    do {                          \
        int _rc = (expr);         \
        if (_rc != BLIM_OK) return _rc; \
    } while (0)

// ---------------------------------------------------------------------------- timing classes
enum TimeClass { TC_GEMM_QKV = 0, TC_ATTN, TC_GEMM_O, TC_GEMM_GATEUP, TC_GEMM_DOWN, TC_NORM, TC_LMHEAD_LSE, TC_GEMM_OTHER, TC_MISC, TC_QUANT, TC_COUNT };
struct TimedSpan { hipEvent_t a, b; int cls; double flops; };

// ---------------------------------------------------------------------------- engine
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct LayerW {
    float* norm1 = nullptr; float* norm2 = nullptr;
    bf16_t* wqkv = nullptr; float* bqkv = nullptr;
    bf16_t* wo = nullptr; bf16_t* wgu = nullptr; bf16_t* wd = nullptr;
    // fp8 mode: e4m3 copies of the four matrices (same stored row order) + one f32 scale per stored row
    uint8_t* wqkv8 = nullptr; uint8_t* wo8 = nullptr; uint8_t* wgu8 = nullptr; uint8_t* wd8 = nullptr;
    float* sqkv = nullptr; float* so = nullptr; float* sgu = nullptr; float* sd = nullptr;
    // option "precise_lo6" (fp16 engines): e2m3 tile images of the four matrices (kernels.hpp: launch_f6_tiles, W side) -- the W operand of the compensated GEMMs'
    // second pass (gemm.hpp: W6)
    uint8_t* wqkv6 = nullptr; uint8_t* wo6 = nullptr; uint8_t* wgu6 = nullptr; uint8_t* wd6 = nullptr;
};

// LoRA adapters kept apart (adapters.hpp): the f32 matrices as loaded + their 16-bit MFMA operand
struct AdapterW { float* A = nullptr; float* B = nullptr; uint16_t* A16 = nullptr; int n_in = 0, n_out = 0; };
struct LayerAd { AdapterW ad[4]; uint16_t* wqkv_aug = nullptr; uint16_t* wo_aug = nullptr;           // ad: q, k, v, o
                 uint8_t* wqkv_aug6 = nullptr; uint8_t* wo_aug6 = nullptr; };   // option "precise_lo6": e2m3 tile images of the augmented matrices

struct blim_engine {
    blim_config c;
    int hd = 128;
    int qkv_n = 0;
    std::vector<LayerW> L;
    bf16_t* embed = nullptr; bf16_t* lm_head = nullptr; bf16_t* visual_head = nullptr;
    float* final_norm = nullptr;
    bf16_t* mlp_w0[2] = {nullptr, nullptr}; float* mlp_b0[2] = {nullptr, nullptr};
    bf16_t* mlp_w2[2] = {nullptr, nullptr}; float* mlp_b2[2] = {nullptr, nullptr};
    float* rope_cos = nullptr; float* rope_sin = nullptr;
    bool f8 = false;              // BLIM_COMPUTE_F8: c.compute_dtype is then F16 (the 16-bit side of the mode)
    bool f8_ready = false;        // fp8 copies are current
    int f8_mask = 31;             // which GEMMs run in fp8 (option "f8_mask"): 1 qkv, 2 o_proj, 4 gate|up, 8 down, 16 lm_head
    uint8_t* lm_head8 = nullptr; float* s_lm = nullptr;
    std::map<std::string, bool> loaded;
    std::vector<void*> owned;
    // workspaces
    DevBuf resid, xn, qkv, attn, act, hsel, lse_part, lab_logit, logprob, stage, proj_tmp, vh, tvg_logits, dense_idx;
    DevBuf rope_rows;                     // [T, 128] cos | sin of every token's position (per batch)
    DevBuf resid_live;                    // f32 [R, H]: the residual rows the caller reads, carried through the LAST layer's o_proj / MLP alone (run_layers)
    bool prune_last = true;               // option "prune_last"
    DevBuf x8, a8, act8, hsel8, rscale;   // fp8 mode: quantised GEMM inputs and their per-row scales
    DevBuf attn_mx;                       // fp8 mode: E8M0 scale per (token, head), written by the attention kernel (attention.hpp: out_mx)
    DevBuf act_mx;                        // fp8 mode: E8M0 scale per (token, 128 SwiGLU outputs), written by the gate|up epilogue (gemm.hpp: out_mx)
    int f8_fuse = 1;                      // option "f8_fuse": quantise the SwiGLU output inside the gate|up epilogue (0: separate quant_rows pass)
    // options / timing
    int attn_tr = 1;
    // compensated ("precise") mode, option "precise" (fp16 engines): every 16-bit activation that feeds a GEMM or the attention's
    // P.V product travels as hi + lo (lo = f16(x - f32(hi))), the GEMMs walk K twice ([hi | lo] against the same weights).  About
    // 21 significant bits of the activations reach the f32 accumulators; costs 2x the GEMM flops, so the host turns it on for the
    // cheap TVG calls only (their scores are ~10x smaller in magnitude than the VTG ones: DESIGN.md section 4).
    bool precise = false;
    bool precise_mlp = true;       // option "precise_mlp": compensate the MLP branch too (87 % of the flops, ~20 % of the error variance)
    bool masked_query_zero = false; // option "masked_query_zero" (PARITY-UNPINNED): query positions the key mask hides write a zero attention output -- what the reference's
                                   // flash-attention-2 class does (modeling_qwen2_flash.py:526-563: dropped before flash_attn_varlen_func, zero-padded back) where its
                                   // eager / SDPA classes, the pinned semantics, compute such rows like any other
    bool precise_embeds = false;   // option "precise_embeds": in precise mode the INPUT embeddings (blim_assemble output, blim_decode / blim_score_* input) and
                                   // the projector outputs feeding them are [hi | lo] rows of width 2H too (the fused TVG path; the literal
                                   // forward() keeps the reference's [B, L, H] embeddings)
    bool timing = false;
    std::vector<TimedSpan> spans;
    // ---- LoRA adapters kept apart (blim_load_adapter; adapters.hpp).  aug = extra K columns of every adapted Linear's operands (0 = none loaded);
    // the augmented weight copies [W | B_hi | B_lo | 0] are (re)built lazily from the placed base weights (build_aug), so base weights and
    // adapters may be loaded in any order.
    int lora_r = 0; float lora_scale = 0.f; int aug = 0;
    std::vector<LayerAd> AD; AdapterW ad_lm, ad_mlp[2][2];       // ad_mlp[mlp | tvg_mlp][Linear 0 | Linear 2]
    uint16_t* lm_aug = nullptr; uint16_t* w0_aug[2] = {nullptr, nullptr}; uint16_t* w2_aug[2] = {nullptr, nullptr};
    bool aug_ready = false;
    // option "precise_lo6" (fp16 engines): in the compensated modes the second walk over K -- the product with the activations' LO parts -- runs on the block-scaled
    // MFMA with e2m3 operands at four times the 16-bit rate (gemm.hip, phase 2), against e2m3 tile images of the decoder weights and the head, built lazily
    // (finalize_lo6: + 0.78 byte per decoder / head weight, 5.9 GB at 7B)
    bool lo6 = false, lo6_ready = false, lo6_fuse = true; int lo6_fuse_mask = 3;   // mask (debug, env BLIM_LO6_FUSED_MASK): 1 SwiGLU epilogue, 2 RMSNorm         // lo6_fuse: the SwiGLU epilogue writes the down GEMM's e2m3 input tiles itself (env BLIM_LO6_FUSED_TILES=0: a pass over its lo rows does)
    uint8_t* lm6 = nullptr;                                      // lm_head (or its augmented copy) as e2m3 tiles; K = lm6_k
    int lm6_k = 0;
    const void* lm6_src = nullptr;                               // the matrix lm6 was derived from
    std::set<const void*> c6_dirty;                              // base matrices (re)placed since their e2m3 image was built
    DevBuf a6, a6b, h6;                                               // e2m3 tile images of the current GEMM input's lo part / of the scored rows' lo parts
    std::set<std::string> merged_pending;                        // after blim_train_merge: the adapted weights that still hold W + s B A (lora_merged stays set until all are re-placed)
    bool lora_merged = false;                                    // blim_train_merge wrote W + (alpha / r) B A into the base weights: adapters apart on top would apply the update twice
    std::vector<void*> aug_owned;                                // the augmented copies + A16 tables (freed on rebuild)
    std::vector<void*> ad_owned;                                 // the f32 A / B matrices
    DevBuf feats_aug, hid_aug;                                   // staging: caller-provided rows copied into augmented rows
    // ---- registered video vocabulary (blim_set_video_vocab; modeling_videochat_flash.py:589-590): clip-major [C][N][M] as f32-derived 16-bit operands --
    // vocab3 rows [hi | lo | hi] (3 M wide) for the three-term compensated TVG logits, vocab1 rows = hi alone for plain calls
    DevBuf vocab3, vocab1, vh3;
    int n_vocab = 0;
    // visual_head (a full fp32 tensor of the resume file, main.py:104-107) as [hi | lo | hi] rows of width 3 H for the three-term product in compensated calls
    DevBuf visual_head3, hs3;
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct SpanGuard {
    blim_engine* e; hipStream_t s; int idx = -1;
    SpanGuard(blim_engine* e_, hipStream_t s_, int cls, double flops) : e(e_), s(s_) {
        if (!e->timing) return;
        TimedSpan t; t.cls = cls; t.flops = flops;
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
        hipEventRecord(t.a, s);
        e->spans.push_back(t);
        idx = (int)e->spans.size() - 1;
    }
    ~SpanGuard() { if (idx >= 0) hipEventRecord(e->spans[idx].b, s); }
};


int dev_alloc(blim_engine* e, void** p, size_t bytes);
int ensure(DevBuf& b, size_t bytes);   // grow-only workspace
GemmParams gp(int dt, const void* A, int64_t lda, const void* W, int64_t M, int N, int K, void* C, int64_t ldc);
int engine_rope_rows(blim_engine* e, const blim_batch* b, hipStream_t s, float** out, int64_t* stride);
int check_batch(const blim_batch* b);
int engine_set_visual_head3(blim_engine* e, const void* dev_src, int dtype, hipStream_t s);   // dev_src [M, H]: BLIM_DTYPE_F32 (hi + lo kept) or BLIM_DTYPE_BF16 (lo = 0)
