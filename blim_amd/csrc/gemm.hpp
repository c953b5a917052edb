// bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  with fused epilogues (gfx950).
// Covers K1 (projector), K4+K5 (QKV+bias+RoPE), K7 (o_proj+residual), K8 (SwiGLU, down+residual),
// K10+K11 (lm_head + log-sum-exp + label gather), K13/K14 (visual head, video-vocabulary logits)
// of SURVEY.md section 2.3.
#pragma once
#include "common.hpp"

enum GemmEpi : int {
    EPI_BF16 = 0,    // C bf16 [M,ldc] = act(acc + bias)
    EPI_F32 = 1,     // C f32  [M,ldc] = acc * scale
    EPI_RESID = 2,   // C f32  [M,ldc] = (resid_in ? resid_in : C) + acc (+ bias)   (residual stream)
    EPI_QKV = 3,     // C bf16 [M,ldc] = rope(acc + bias) for cols < rope_cols, acc + bias otherwise;
                     //   W rows of every q/k head are stored pair-interleaved (see qkv_perm_row)
    EPI_SWIGLU = 4,  // C bf16 [M, N/2] = silu(gate) * up;  W rows interleaved 16 gate / 16 up
    EPI_LSE = 5,     // per (row, 256-col tile): (max, sum exp) partials + the label's logit
};

struct GemmParams {
    int dtype;  // DT_BF16 / DT_F16: format of A, W and of 16-bit outputs; DT_F8: A, W are e4m3 bytes (K-step 128), 16-bit outputs fp16
    const float* row_scale;  // DT_F8: [M] dequantisation scale of every A row
    const float* col_scale;  // DT_F8: [N] dequantisation scale of every W row (in W's stored row order)
    const bf16_t* A;
    int64_t lda;
    const bf16_t* W;  // [N, K], row stride K
    int M, N, K;
    void* C;
    int64_t ldc;
    const float* bias;  // [N] (in W's row order) or nullptr
    const float* resid_in;  // EPI_RESID: C = resid_in + acc (+ bias); nullptr = in place (C += acc).  Same row stride as C.
    uint16_t* swiglu_gu;    // EPI_BF16, fine-tuning backward: when non-null the tile (d act, N = I columns) is not stored; the saved gate | up
    int64_t swiglu_ld;      //   pre-activations [M, swiglu_ld] (16 gate / 16 up columns interleaved) become [d gate | d up] in place
    uint16_t* swiglu_act;   // EPI_BF16, fine-tuning forward: when non-null the tile is the gate | up pre-activations (stored to C as usual) and
    int64_t swiglu_act_ld;  //   act = silu(gate) * up [M, N / 2] is written here as well (row stride swiglu_act_ld)
    int act;            // EPI_BF16: 0 none, 1 exact-erf GELU
    float scale;        // EPI_F32
    // EPI_QKV
    int rope_cols;           // (num_heads + num_kv_heads) * 128
    const float* rope_rows;  // cos / sin of every ROW's position, gathered once per batch (engine.hip: rope_rows_kernel), chunk-major:
                             // [8 = {cos, sin} x 4 groups of 16 dims][rope_stride rows][16]: a fragment's 16 consecutive rows are 1 KB contiguous
    int64_t rope_stride;     // rows per chunk of that table (>= M)
    // EPI_LSE
    const int32_t* labels;   // [M] target column per row (or < 0)
    float2* lse_part;        // [M, ceil(N/256)] (max, sumexp)
    float* label_logit;      // [M]
    // compensated ("precise") mode, fp16 engines: the A operand is [hi | lo] along K (lo = f16(x - f32(hi)): ~21 significant bits of
    // the activation reach the f32 accumulator) -- K counts both halves and W's K index wraps after w_wrap_k elements;
    // 16-bit outputs are written as hi at C and lo at C + lo_off elements (0 = plain)
    int w_wrap_k;
    int64_t lo_off;
    // fp8 mode, fused quantisation: EPI_SWIGLU with out8 != nullptr writes its 128 output columns per tile as e4m3
    // bytes (C = out8, ldc in bytes) plus ONE E8M0 byte per (row, tile column) = per (row, 128-deep K-step of the consuming GEMM) into
    // out_mx; a DT_F8 GEMM with a_mx != nullptr feeds that byte to the block-scaled MFMA as the A operand's scale (all four 32-blocks of
    // the K-step share it).  Table layout: [K-step][256-row tile][(wm * 16 + fr) * 8 + mi] with row-in-tile = 128 wm + 16 mi + fr, i.e. the
    // eight bytes a lane needs for its eight 16-row fragments are one 8-byte load; mx_stride = bytes per K-step = 256 * row tiles.
    uint8_t* out_mx;
    const uint8_t* a_mx;
    int64_t mx_stride;
    // fp16 engines, compensated modes with an e2m3 second pass ("lo6", gemm.hip phase 2; round 5, replaces round 4's e4m3 pass).  A / lda / K / W describe the plain
    // 16-bit product of the hi parts (w_wrap_k = 0); A6 / W6 are the e2m3 forms of the A operand's LO part and of W (kernels.hpp: launch_f6_tiles), K6 values per
    // row (K6 % 128 == 0), stored as the LDS IMAGE of the second pass's operand tiles: one block of F6_TILE_BYTES per (256-row tile, 128-value K-step), tile-major
    // ([row tile][K-step]), so that a tile is staged by a lane-linear LDS-DMA copy of 25 KiB.  Inside a block, for each 16-row fragment group fb (16 of them):
    //   [fb * 1536 + lane * 16, + 16)          bytes 0-15 of the 24 packed e2m3 bytes of (row lane & 15, values 32 g .. 32 g + 31 of the step, g = lane >> 4)
    //   [fb * 1536 + 1024 + lane * 8, + 8)     bytes 16-23 of the same 32 values
    // -- exactly what MFMA lane `lane` of that fragment needs, so a fragment is one ds_read_b128 + one ds_read_b64 of consecutive lanes (no bank conflict by
    // construction) -- and at byte 24576 the E8M0 scale bytes of the tile's 256 x 4 blocks, ordered so that a lane reads the scales of all its fragments at once:
    //   A side: [((row >> 7) * 4 + g) * 128 + (row & 15) * 8 + ((row >> 4) & 7)]     (8 bytes per lane: its eight 16-row fragments of a wave's 128 rows)
    //   W side: [((row >> 6) * 4 + g) * 64 + (row & 15) * 4 + ((row >> 4) & 3)]      (4 bytes per lane: its four fragments of a wave's 64 columns)
    // Value j of a 32-value block sits at bits [6 j, 6 j + 6) of its 24 bytes; a block's scale is the smallest power of two with max|x| / scale <= 7.5.
    const uint8_t* A6;
    const uint8_t* W6;
    int K6;
    // EPI_SWIGLU with split (hi | lo) outputs: out6 != nullptr -> the epilogue ALSO writes the lo part of its output as e2m3 operand tiles -- the A6 operand of the
    // GEMM that consumes it (the tile's 128 output columns are one K-step of that GEMM, so workgroup (tm, tn) writes tile [tm][tn] whole: bit-identical to
    // launch_f6_tiles on the lo rows) -- and does NOT store the lo half at C + lo_off: the rows' lo parts are neither written nor re-read.  N % 256 == 0.
    uint8_t* out6;
    int f16_saturate;        // fp16 outputs: saturate to +-65504 instead of +-inf (common.hpp: f16_saturate_on).  engine.hip's gp() sets it; the one
                             // 16-bit GRADIENT store of the trainer clears it (the loss scaler must see an overflow as inf)
    int group_m;             // M-tiles per band of the tile order (8; BLIM_GEMM_GROUP_M)
    int tile_map;            // 1: 32-tile groups round-robin over the XCDs (default), 0: XCD-contiguous chunks (BLIM_GEMM_TILE_MAP)
    int debug_skip_epilogue; // timing aid only (set from BLIM_GEMM_SKIP_EPI)
    unsigned long long* debug_stamps;  // timing aid: [n_workgroups][8] s_memrealtime at {entry, main loop start, main loop end, exit, C staged in LDS, stores issued}
};
void gemm_set_debug_stamps(unsigned long long* buf);

// 256x256 tiles x 128 bytes of K per step (64 16-bit / 128 fp8 elements), 512 threads.  K % 64 == 0 (fp8: % 128), lda % 8 == 0 (fp8: % 16).  Rows/cols beyond M/N are clamped on
// load and masked on store, so neither A nor W needs padding.
int launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream);

// Row permutation used for q/k heads so that RoPE partners (d, d+64) land in the same lane:
// stored row c' (0..127 within a head) holds natural row d = 16*(c'>>5) + (c'&15) + 64*((c'>>4)&1).
static inline int qkv_perm_row(int cprime) { return 16 * (cprime >> 5) + (cprime & 15) + 64 * ((cprime >> 4) & 1); }
// SwiGLU interleave: stored row r of the fused [2I, K] matrix: group g = r>>5, t = r&31:
// t < 16 -> gate row 16g+t, else up row 16g+(t-16).
