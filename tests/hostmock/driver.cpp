// TEST INFRASTRUCTURE (tests/test_host_sanitizers.py): drives the HOST side of the C ABI -- csrc/engine.hip, adapters.hip, train.hip compiled as plain C++ against the
// mock HIP runtime of this directory -- under AddressSanitizer + UBSan.  Kernels do nothing here; what runs is the bookkeeping: create / destroy, weight and adapter
// tables (every name, partial sets, reloads), the lazily built derived copies (augmented weights, the combined 16-bit | e2m3 copies, fp8 copies), workspace growth,
// every option key, the scoring / decode / forward entry points' host logic in every numeric mode, the trainer's life cycle, and the error paths (bad arguments,
// wrong state, out of device memory).  Exit code 0 = every expectation met; the sanitizers abort on their own findings.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/blim.h"

extern "C" size_t mock_hip_mem_limit, mock_hip_mem_in_use;
extern "C" int mock_hip_device_count;
extern "C" size_t mock_hip_live_allocations();

static int failures = 0;
#define EXPECT(cond)                                                                              \
    do {                                                                                          \
        if (!(cond)) { fprintf(stderr, "EXPECT failed: %s (%s:%d) last error: %s\n", #cond, __FILE__, __LINE__, blim_last_error()); ++failures; } \
    } while (0)

static blim_config cfg_of(int dtype, int H = 256, int I = 512, int layers = 2) {
    blim_config c;
    memset(&c, 0, sizeof c);
    c.vocab_size = 1024; c.hidden_size = H; c.intermediate_size = I; c.num_layers = layers; c.num_heads = 2; c.num_kv_heads = 1;
    c.mm_hidden_size = 64; c.num_clips = 4; c.max_positions = 128; c.compute_dtype = dtype; c.rms_eps = 1e-6f; c.rope_theta = 1e6f;
    return c;
}
static std::vector<std::string> weight_names(int layers) {
    std::vector<std::string> n = {"embed_tokens", "final_norm", "lm_head", "visual_head"};
    for (const char* p : {"mlp", "tvg_mlp"}) for (const char* t : {"0.w", "0.b", "2.w", "2.b"}) n.push_back(std::string(p) + "." + t);
    for (int i = 0; i < layers; ++i)
        for (const char* t : {"input_norm", "post_norm", "q_proj.w", "k_proj.w", "v_proj.w", "q_proj.b", "k_proj.b", "v_proj.b", "o_proj.w", "gate_proj.w", "up_proj.w", "down_proj.w"})
            n.push_back("layers." + std::to_string(i) + "." + t);
    return n;
}

struct Batch {
    std::vector<int32_t> pos, seq_start, seq_len, pfx_start, pfx_len, blk_seq, blk_q0, own;
    std::vector<uint8_t> vis;
    blim_batch b;
    Batch(int n_seq, int len, bool with_own) {
        for (int s = 0; s < n_seq; ++s) {
            seq_start.push_back(s * len); seq_len.push_back(len); pfx_start.push_back(0); pfx_len.push_back(s ? len : 0);
            for (int q = 0; q < len; q += 32) { blk_seq.push_back(s); blk_q0.push_back(q); }
            for (int i = 0; i < len; ++i) { pos.push_back(i); vis.push_back(1); own.push_back(0); }
        }
        b.n_tokens = n_seq * len; b.n_seqs = n_seq; b.n_blocks = (int32_t)blk_seq.size();
        b.positions = pos.data(); b.key_visible = vis.data(); b.seq_start = seq_start.data(); b.seq_len = seq_len.data(); b.pfx_start = pfx_start.data(); b.pfx_len = pfx_len.data();
        b.blk_seq = blk_seq.data(); b.blk_q0 = blk_q0.data(); b.own_start = with_own ? own.data() : nullptr;
    }
};

static void load_all(blim_engine* e, const blim_config& c, int dtype_code) {
    std::vector<float> big((size_t)c.vocab_size * c.hidden_size, 0.01f);
    for (const std::string& n : weight_names(c.num_layers)) EXPECT(blim_load_weight(e, n.c_str(), big.data(), dtype_code, 0) == 0);
}

// every scoring-side entry point once, on host buffers standing in for device buffers
static void score_calls(blim_engine* e, const blim_config& c, int n_seq, int len, bool with_own) {
    Batch B(n_seq, len, with_own);
    const int64_t T = B.b.n_tokens;
    const int H = c.hidden_size, wide = 2;
    std::vector<uint16_t> embeds((size_t)T * H * wide, 0), hid((size_t)T * H * wide, 0), feats((size_t)64 * 64 * wide, 0), proj((size_t)64 * H * wide, 0);
    std::vector<float> f32((size_t)T * std::max(H, c.vocab_size), 0.f), scores(64, 0.f), logits((size_t)T * c.vocab_size, 0.f);
    std::vector<int32_t> rows, labels, row_start = {0}, src((size_t)T, 1), vlab(16, 0);
    for (int s = 0; s < n_seq; ++s) { for (int i = 0; i < 4; ++i) { rows.push_back(s * len + len - 5 + i); labels.push_back(7); } row_start.push_back((int32_t)rows.size()); }
    EXPECT(blim_project_video(e, feats.data(), 64, 0, proj.data(), nullptr) == 0);
    EXPECT(blim_project_video(e, feats.data(), 64, 1, proj.data(), nullptr) == 0);
    EXPECT(blim_project_video(e, feats.data(), 64, 2, proj.data(), nullptr) == BLIM_ERR_ARG);
    EXPECT(blim_group_mean(e, proj.data(), 4, 16, hid.data(), nullptr) == 0);
    EXPECT(blim_assemble(e, src.data(), T, proj.data(), embeds.data(), nullptr) == 0);
    EXPECT(blim_decode(e, &B.b, embeds.data(), nullptr, 0, hid.data(), f32.data(), nullptr) == 0);
    EXPECT(blim_decode(e, &B.b, embeds.data(), rows.data(), (int64_t)rows.size(), hid.data(), nullptr, nullptr) == 0);
    EXPECT(blim_score_vtg(e, &B.b, embeds.data(), rows.data(), labels.data(), (int64_t)rows.size(), row_start.data(), n_seq, scores.data(), nullptr) == 0);
    std::vector<float> vocab_f32((size_t)16 * c.num_clips * c.mm_hidden_size, 0.f);
    std::vector<uint16_t> vocab16((size_t)16 * c.num_clips * c.mm_hidden_size, 0);
    EXPECT(blim_set_video_vocab(e, vocab_f32.data(), 16, nullptr) == 0 || c.compute_dtype == BLIM_COMPUTE_F8);
    std::vector<int32_t> trows;
    for (int s = 0; s < n_seq; ++s) for (int i = 0; i < c.num_clips; ++i) trows.push_back(s * len + len - 6 + i);
    EXPECT(blim_score_tvg(e, &B.b, embeds.data(), trows.data(), vocab16.data(), 16, vlab.data(), n_seq, scores.data(), nullptr) == 0);
    std::vector<uint8_t> mask((size_t)2 * 40, 1);
    std::vector<float> hidden_f32((size_t)2 * 40 * H, 0.f), lg((size_t)2 * 40 * c.vocab_size, 0.f);
    EXPECT(blim_forward(e, embeds.data(), mask.data(), 2, 40, lg.data(), hidden_f32.data(), nullptr) == 0);
    EXPECT(blim_vtg_logprobs(e, hid.data(), labels.data(), (int64_t)labels.size(), f32.data(), nullptr) == 0);
    EXPECT(blim_lm_head(e, hid.data(), 8, logits.data(), nullptr) == 0);
    EXPECT(blim_visual_head(e, hid.data(), 8, proj.data(), nullptr) == 0);
    EXPECT(blim_segment_mean(e, f32.data(), row_start.data(), n_seq, 0, scores.data(), nullptr) == 0);
}

int main() {
    EXPECT(blim_abi_version() == BLIM_ABI_VERSION);
    blim_engine* e = nullptr;
    // ---- creation: bad configurations, no device
    {
        blim_config c = cfg_of(BLIM_COMPUTE_F16);
        EXPECT(blim_create(nullptr, &e) == BLIM_ERR_ARG && blim_create(&c, nullptr) == BLIM_ERR_ARG);
        blim_config bad = c; bad.hidden_size = 192;                                    // head_dim != 128
        EXPECT(blim_create(&bad, &e) == BLIM_ERR_ARG && strstr(blim_last_error(), "head_dim"));
        bad = c; bad.compute_dtype = 7;
        EXPECT(blim_create(&bad, &e) == BLIM_ERR_ARG);
        bad = c; bad.num_layers = 0;
        EXPECT(blim_create(&bad, &e) != 0);
        mock_hip_device_count = 0;
        EXPECT(blim_create(&c, &e) != 0);                                              // the product path fails loudly without a device
        mock_hip_device_count = 1;
        mock_hip_mem_limit = 1 << 20;                                                  // out of device memory half way through the weight allocations
        EXPECT(blim_create(&c, &e) != 0);
        mock_hip_mem_limit = 0;
    }
    EXPECT(mock_hip_live_allocations() == 0);                                          // failed creations release what they had taken
    for (int dtype : {BLIM_COMPUTE_F16, BLIM_COMPUTE_BF16, BLIM_COMPUTE_F8}) {
        const blim_config c = cfg_of(dtype);
        EXPECT(blim_create(&c, &e) == 0 && e);
        if (!e) continue;
        // ---- weights: state before loading, unknown names, f32 and bf16 sources, host and "device" pointers, synthetic fill, reloads
        EXPECT(blim_weights_ready(e) == BLIM_ERR_STATE);
        {
            Batch B(1, 8, false);
            std::vector<uint16_t> emb((size_t)8 * c.hidden_size, 0), out((size_t)8 * c.hidden_size, 0);
            EXPECT(blim_decode(e, &B.b, emb.data(), nullptr, 0, out.data(), nullptr, nullptr) == BLIM_ERR_STATE && strstr(blim_last_error(), "not loaded"));
        }
        std::vector<float> z(16, 0.f);
        EXPECT(blim_load_weight(e, "layers.0.nonsense", z.data(), BLIM_DTYPE_F32, 0) == BLIM_ERR_ARG && strstr(blim_last_error(), "unknown weight"));
        EXPECT(blim_load_weight(e, "layers.9.q_proj.w", z.data(), BLIM_DTYPE_F32, 0) == BLIM_ERR_ARG);
        EXPECT(blim_load_weight(e, "lm_head", z.data(), 5, 0) == BLIM_ERR_ARG && blim_load_weight(e, nullptr, z.data(), 0, 0) == BLIM_ERR_ARG);
        load_all(e, c, BLIM_DTYPE_F32);
        EXPECT(blim_weights_ready(e) == 0);
        load_all(e, c, BLIM_DTYPE_BF16);
        EXPECT(blim_init_synthetic_weights(e, 3) == 0);
        // ---- options: every key, both values, refusals
        for (const char* k : {"precise", "precise_embeds", "precise_mlp", "precise_lo6", "prune_last", "f8_fuse", "f8_mask", "attn_tr"}) {
            for (int v : {0, 1}) {
                const int rc = blim_set_option(e, k, v);
                EXPECT(rc == 0 || rc == BLIM_ERR_ARG || rc == BLIM_ERR_STATE);
            }
        }
        EXPECT(blim_set_option(e, "no_such_option", 1) == BLIM_ERR_ARG && blim_set_option(e, nullptr, 1) == BLIM_ERR_ARG);
        if (dtype == BLIM_COMPUTE_F8) EXPECT(blim_set_option(e, "precise", 1) != 0 && blim_set_option(e, "precise_lo6", 1) != 0);
        const bool lo6_dims = c.hidden_size % 128 == 0 && c.intermediate_size % 128 == 0;
        // bf16 engines take the e2m3 second pass as an opt-in since round 6 (same size rule as fp16 engines); left off again for the calls below
        if (dtype == BLIM_COMPUTE_BF16) { EXPECT(blim_set_option(e, "precise_lo6", 1) == (lo6_dims ? 0 : BLIM_ERR_ARG)); EXPECT(blim_set_option(e, "precise_lo6", 0) == 0); }
        EXPECT(blim_set_option(e, "prune_last", 1) == 0 && blim_set_option(e, "f8_fuse", 1) == 0 && blim_set_option(e, "f8_mask", 31) == 0);
        // ---- workspaces: growth in steps, then the calls in every numeric mode
        for (int64_t t : {64, 300, 5000, 200}) EXPECT(blim_reserve(e, t, t / 2 + 1) == 0);
        EXPECT(blim_reserve(e, -1, 0) == BLIM_ERR_ARG);
        EXPECT(blim_timing_enable(e, 1) == 0);
        const bool can_precise = dtype != BLIM_COMPUTE_F8;
        for (int precise = 0; precise <= (can_precise ? 1 : 0); ++precise) {
            EXPECT(blim_set_option(e, "precise", precise) == 0);
            for (int mlp = 0; mlp <= precise; ++mlp) {
                EXPECT(blim_set_option(e, "precise_mlp", mlp) == 0);
                if (precise) EXPECT(blim_set_option(e, "precise_embeds", mlp) == 0);
                const bool lo6_engine = dtype == BLIM_COMPUTE_F16 || (dtype == BLIM_COMPUTE_BF16 && lo6_dims);
                for (int lo6 = 0; lo6 <= ((precise && lo6_engine) ? 1 : 0); ++lo6) {
                    if (lo6_engine) EXPECT(blim_set_option(e, "precise_lo6", lo6) == 0);
                    score_calls(e, c, 3, 40, false);
                    score_calls(e, c, 2, 70, true);
                }
            }
        }
        {
            std::vector<double> ms(blim_timing_num_classes()), fl(blim_timing_num_classes());
            std::vector<int64_t> calls(blim_timing_num_classes());
            EXPECT(blim_timing_report(e, ms.data(), calls.data(), fl.data()) == 0);
            for (int i = 0; i < blim_timing_num_classes(); ++i) EXPECT(blim_timing_class_name(i) != nullptr);
            EXPECT(blim_timing_enable(e, 0) == 0);
        }
        // ---- a position beyond the RoPE table
        {
            Batch B(1, 8, false);
            B.pos[3] = 4000;
            std::vector<uint16_t> emb((size_t)8 * c.hidden_size * 2, 0), out((size_t)8 * c.hidden_size * 2, 0);
            const int rc = blim_decode(e, &B.b, emb.data(), nullptr, 0, out.data(), nullptr, nullptr);
            EXPECT(rc == 0 || rc == BLIM_ERR_ARG);          // device data: the library may not see it (blim.h); it must not fault on the host
            blim_batch bad = B.b; bad.n_tokens = 0;
            EXPECT(blim_decode(e, &bad, emb.data(), nullptr, 0, out.data(), nullptr, nullptr) == BLIM_ERR_ARG);
            bad = B.b; bad.seq_len = nullptr;
            EXPECT(blim_decode(e, &bad, emb.data(), nullptr, 0, out.data(), nullptr, nullptr) == BLIM_ERR_ARG);
        }
        // ---- adapters apart: ranks 4 / 8 / 16, partial sets, a rank change refused, reload over a live set, clear; derived copies rebuilt lazily in between
        const int H = c.hidden_size, M = c.mm_hidden_size, V = c.vocab_size, qn = c.num_heads * 128, kn = c.num_kv_heads * 128;
        std::vector<float> A((size_t)16 * std::max(H, M), 0.01f), Bm((size_t)std::max(V, H) * 16, 0.02f), Wbig((size_t)c.vocab_size * H, 0.03f);
        for (int r : {4, 8, 16}) {
            EXPECT(blim_clear_adapters(e) == 0 && blim_num_adapters(e) == 0);
            EXPECT(blim_load_adapter(e, "layers.0.q_proj.w", A.data(), Bm.data(), r, 32.f) == 0);
            EXPECT(blim_load_adapter(e, "layers.0.q_proj.w", A.data(), Bm.data(), r == 4 ? 8 : 4, 32.f) != 0);                     // one rank / scale per engine
            EXPECT(blim_load_adapter(e, "layers.0.nonsense", A.data(), Bm.data(), r, 32.f) == BLIM_ERR_ARG);
            EXPECT(blim_load_adapter(e, "layers.0.q_proj.w", nullptr, Bm.data(), r, 32.f) == BLIM_ERR_ARG && blim_load_adapter(e, "lm_head", A.data(), Bm.data(), 17, 32.f) == BLIM_ERR_ARG);
            if (can_precise) { EXPECT(blim_set_option(e, "precise", 1) == 0); }
            score_calls(e, c, 2, 40, false);                                                     // a partial set: one adapter
            for (int l = 0; l < c.num_layers; ++l)
                for (const char* t : {"q_proj.w", "k_proj.w", "v_proj.w", "o_proj.w"}) EXPECT(blim_load_adapter(e, ("layers." + std::to_string(l) + "." + t).c_str(), A.data(), Bm.data(), r, 32.f) == 0);
            for (const char* n : {"lm_head", "mlp.0.w", "mlp.2.w", "tvg_mlp.0.w", "tvg_mlp.2.w"}) EXPECT(blim_load_adapter(e, n, A.data(), Bm.data(), r, 32.f) == 0);
            EXPECT(blim_num_adapters(e) == 4 * c.num_layers + 5);
            score_calls(e, c, 2, 40, true);
            EXPECT(blim_load_weight(e, "layers.1.down_proj.w", Wbig.data(), BLIM_DTYPE_F32, 0) == 0);     // a base weight replaced under live adapters
            EXPECT(blim_load_adapter(e, "lm_head", A.data(), Bm.data(), r, 32.f) == 0);                   // ... and an adapter reloaded
            if (can_precise) { EXPECT(blim_set_option(e, "precise", 0) == 0); }
            score_calls(e, c, 2, 40, false);
        }
        EXPECT(blim_clear_adapters(e) == 0 && blim_num_adapters(e) == 0);
        (void)qn; (void)kn;
        // ---- out of device memory while a call grows its workspace / builds its derived copies: an error code, and the engine stays usable
        if (dtype == BLIM_COMPUTE_F16) {
            EXPECT(blim_set_option(e, "precise", 1) == 0 && blim_set_option(e, "precise_lo6", 1) == 0);
            EXPECT(blim_load_weight(e, "layers.0.o_proj.w", Wbig.data(), BLIM_DTYPE_F32, 0) == 0);
            mock_hip_mem_limit = mock_hip_mem_in_use + 4096;
            Batch B(60, 120, false);                                                      // 7,200 tokens: more than any call before, the workspaces must grow
            std::vector<uint16_t> emb((size_t)7200 * c.hidden_size * 2, 0), out((size_t)7200 * c.hidden_size * 2, 0);
            EXPECT(blim_decode(e, &B.b, emb.data(), nullptr, 0, out.data(), nullptr, nullptr) != 0);
            mock_hip_mem_limit = 0;
            EXPECT(blim_decode(e, &B.b, emb.data(), nullptr, 0, out.data(), nullptr, nullptr) == 0);
            EXPECT(blim_set_option(e, "precise", 0) == 0);
        }
        // ---- the trainer's life cycle on this engine
        {
            blim_train_config tc; tc.lora_r = 8; tc.lora_alpha = 32.f; tc.lora_dropout = 0.05f;
            const int64_t n = blim_train_flat_size(e, tc.lora_r);
            EXPECT(n > 0 && blim_train_flat_size(e, 17) < 0);
            int64_t off = -1, rows = 0, cols = 0;
            EXPECT(blim_train_param_offset(e, 8, "lm_head:B", &off, &rows, &cols) == 0 && rows == V && cols == 8 && off >= 0 && off + rows * cols <= n);
            EXPECT(blim_train_param_offset(e, 8, "layers.1.k_proj.w:A", &off, &rows, &cols) == 0 && rows == 8 && cols == H);
            EXPECT(blim_train_param_offset(e, 8, "visual_head", &off, &rows, &cols) == 0 && rows == M && cols == H);
            EXPECT(blim_train_param_offset(e, 8, "layers.7.k_proj.w:A", &off, &rows, &cols) != 0 && blim_train_param_offset(e, 8, "lm_head:C", &off, &rows, &cols) != 0);
            std::vector<float> params((size_t)std::max<int64_t>(n, 1), 0.f), grads(params.size(), 0.f), m1(params.size(), 0.f), m2(params.size(), 0.f);
            blim_trainer* t = nullptr;
            EXPECT(blim_train_create(e, nullptr, params.data(), grads.data(), &t) == BLIM_ERR_ARG);
            EXPECT(blim_train_create(e, &tc, params.data(), grads.data(), &t) == 0 && t);
            if (t) {
                EXPECT(blim_train_sync_params(t, nullptr) == 0);
                Batch B(3, 48, false);
                const int64_t T = B.b.n_tokens;
                std::vector<int32_t> src((size_t)T, 5), rows_v = {10, 11, 12, 60, 61}, labels = {1, 2, 3, 4, 5}, trows = {100, 101, 102, 103}, tlab = {2};
                std::vector<uint16_t> feats((size_t)256 * M, 0), vocab((size_t)c.num_clips * 8 * M, 0);
                src[3] = -1; src[4] = -2; src[100] = -(256 + 1);
                blim_train_batch tb; memset(&tb, 0, sizeof tb);
                tb.batch = &B.b; tb.src_index = src.data(); tb.feats = feats.data(); tb.n_feat_rows = 256; tb.tok_per_clip = 64; tb.max_seq_len = 48;
                tb.rows = rows_v.data(); tb.labels = labels.data(); tb.n_rows = 5; tb.tvg_rows = trows.data(); tb.tvg_labels = tlab.data(); tb.n_tvg_rows = 4;
                tb.vocab = vocab.data(); tb.n_vocab = 8; tb.grad_scale = 1024.f; tb.dropout_seed = 9;
                float loss[2] = {0.f, 0.f}, stats[2] = {0.f, 0.f};
                const int rc = blim_train_step(t, &tb, loss, nullptr);
                EXPECT(dtype == BLIM_COMPUTE_F8 ? rc != 0 : rc == 0);
                tb.n_tvg_rows = 0; tb.tvg_rows = nullptr;                                          // VTG rows alone
                if (dtype != BLIM_COMPUTE_F8) EXPECT(blim_train_step(t, &tb, loss, nullptr) == 0);
                EXPECT(blim_train_grad_stats(t, 1.f / 1024.f, stats, nullptr) == 0);
                EXPECT(blim_train_adamw(t, m1.data(), m2.data(), 1e-4f, 0.9f, 0.95f, 1e-8f, 0.02f, 1.f / 1024.f, 1, nullptr) == 0);
                EXPECT(blim_train_merge(t, nullptr) == 0);
                // a merged engine refuses adapters apart until EVERY adapted weight has been re-placed (one reloaded norm used to lift the guard)
                EXPECT(blim_load_adapter(e, "lm_head", A.data(), Bm.data(), 8, 32.f) == BLIM_ERR_STATE);
                EXPECT(blim_load_weight(e, "final_norm", Wbig.data(), BLIM_DTYPE_F32, 0) == 0);
                EXPECT(blim_load_adapter(e, "lm_head", A.data(), Bm.data(), 8, 32.f) == BLIM_ERR_STATE);
                EXPECT(blim_load_weight(e, "lm_head", Wbig.data(), BLIM_DTYPE_F32, 0) == 0);
                EXPECT(blim_load_adapter(e, "lm_head", A.data(), Bm.data(), 8, 32.f) == BLIM_ERR_STATE);      // q/k/v/o and the projector still merged
                load_all(e, c, BLIM_DTYPE_F32);
                EXPECT(blim_load_adapter(e, "lm_head", A.data(), Bm.data(), 8, 32.f) == 0);
                EXPECT(blim_train_merge(t, nullptr) == BLIM_ERR_STATE);                                       // adapters apart: one or the other
                EXPECT(blim_clear_adapters(e) == 0);
                score_calls(e, c, 2, 40, false);
                blim_train_destroy(t);
            }
            blim_train_destroy(nullptr);
        }
        blim_destroy(e);
        e = nullptr;
        EXPECT(mock_hip_live_allocations() == 0);                                      // destroy frees every device allocation of the engine and its trainer
    }
    blim_destroy(nullptr);
    if (failures) { fprintf(stderr, "%d expectation(s) failed\n", failures); return 1; }
    printf("host sanitizer drive: ok\n");
    return 0;
}
