// Fine-tuning step behind include/blim.h's blim_train_* (training_utils.py:57-95, main.py:96-150).
//
// The VTG rows and the TVG rows of a batch go through the decoder as ONE packed batch (forward and backward once; the two heads and the
// two projectors differ).  Forward = the scoring path's kernels with two changes: (1) every LoRA-adapted Linear runs on a K-AUGMENTED copy of its frozen
// weight, [W | B | 0] with 64 extra K columns, against activations [x | alpha/r * A drop(x) | 0] -- the rank-r update rides in the
// same MFMA accumulation as the base product, before bias / RoPE, at +1.8 % of the K loop; (2) activations the backward needs are
// kept per layer (288 GB of HBM: no recomputation): the f32 residual stream before each sub-block, the normalised QKV input with
// its adapter columns, q|k|v after RoPE, the attention output and its rows' log-sum-exp, and the gate|up pre-activations.
// Backward = input-gradient GEMMs on the same 256x256 MFMA kernel against TRANSPOSED copies of the frozen weights (built once),
// rank-r products for the adapters on the matrix cores, and the attention backward of train_kernels.hip (P re-materialised from the
// saved log-sum-exp).  SwiGLU has no pass of its own: forward and backward sit in the epilogues of the gate|up and d act GEMMs
// (gemm.hpp: swiglu_act / swiglu_gu).  Gradients w.r.t. 16-bit activations travel as the engine's 16-bit type scaled by the AMP loss
// scale (util/misc.py:232-252 NativeScaler), the residual-stream gradient stays f32 like the forward's residual stream.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <array>
#include <string>
#include <vector>

#include "engine.hpp"
#include "train.hpp"

#define AUG 64   // extra K columns of an augmented weight / activation row

// The trainer's GEMMs and norms do NOT saturate their fp16 stores (the scoring path's do: common.hpp f16_saturate_on): an activation beyond 65504 must become inf,
// reach the loss and the gradients, and make the AMP scaler skip the step and halve the scale -- what the reference's autocast + GradScaler do (util/misc.py:232-259).
static GemmParams gpt(int dt, const void* A, int64_t lda, const void* W, int64_t M, int N, int K, void* C, int64_t ldc) {
    GemmParams p = gp(dt, A, lda, W, M, N, K, C, ldc);
    p.f16_saturate = 0;
    return p;
}

struct Adapter { int64_t offA = 0, offB = 0; int n_in = 0, n_out = 0; uint16_t* Bt16 = nullptr; int64_t ldb = 0; uint16_t* A16 = nullptr; };   // Bt16: 16-bit B^T [16, ldb] (launch_lora_du); A16: 16-bit A [16, n_in] (launch_lora_down)

struct TrainLayerW { uint16_t* wqkv_aug = nullptr; uint16_t* wo_aug = nullptr; uint16_t* wqkvT = nullptr; uint16_t* woT = nullptr; uint16_t* wguT = nullptr; uint16_t* wdT = nullptr; };

struct Layout {
    Adapter mlp[2][2];                        // [mlp | tvg_mlp][Linear 0 | Linear 2]
    Adapter lm;
    std::vector<std::array<Adapter, 4>> layer;   // q, k, v, o
    int64_t off_vh = 0, total = 0;
};

static Layout make_layout(const blim_config& c, int r) {
    Layout L;
    int64_t off = 0;
    auto put = [&](Adapter& a, int n_out, int n_in) {
        a.n_in = n_in; a.n_out = n_out;
        a.offA = off; off += round_up((int64_t)r * n_in, 64);
        a.offB = off; off += round_up((int64_t)n_out * r, 64);
    };
    const int H = c.hidden_size, M = c.mm_hidden_size, qn = c.num_heads * 128, kn = c.num_kv_heads * 128;
    for (int w = 0; w < 2; ++w) { put(L.mlp[w][0], H, M); put(L.mlp[w][1], H, H); }      // blim_amd/checkpoint.py:expected_adapters order
    put(L.lm, c.vocab_size, H);
    L.layer.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) { put(L.layer[l][0], qn, H); put(L.layer[l][1], kn, H); put(L.layer[l][2], kn, H); put(L.layer[l][3], H, H); }
    L.off_vh = off; off += round_up((int64_t)M * H, 64);
    L.total = off;
    return L;
}

struct blim_trainer {
    blim_engine* e = nullptr;
    int r = 8; float s = 4.0f; float p_drop = 0.f;
    float* params = nullptr; float* grads = nullptr;
    Layout lay;
    std::vector<TrainLayerW> L;
    uint16_t* lm_aug = nullptr; uint16_t* lmT = nullptr; int Vp = 0;
    uint16_t* w0_aug[2] = {nullptr, nullptr}; uint16_t* w2_aug[2] = {nullptr, nullptr}; uint16_t* w2T[2] = {nullptr, nullptr};
    uint16_t* vh16 = nullptr;
    std::vector<void*> owned;
    // saved activations of the last forward (per layer, strided by tokens) and workspaces
    DevBuf sv_res, sv_mid, sv_xn1, sv_qkv, sv_attn, sv_gu, sv_lse;
    DevBuf red_scratch;   // partials of the split reductions (adapter gradients, du, per-row losses): summed in a fixed order, never by float atomics
    DevBuf xn2, act, dres, dy16, dtmp32, dqkv32, dqkv16, du, attn_D, P16, dS16, logits, dlog16, hsel, hsel_t, dhsel;
    DevBuf feats_aug[2], pre16[2], h16[2], vid16, proj16, embeds, dout16[2], dh32, vh32, vhb16, dl32, dvh;      // [2]: projector mlp / tvg_mlp
    int64_t last_T = 0;
};

static std::vector<Adapter*> all_adapters(blim_trainer* t) {
    std::vector<Adapter*> v;
    for (int w = 0; w < 2; ++w) for (int i = 0; i < 2; ++i) v.push_back(&t->lay.mlp[w][i]);
    v.push_back(&t->lay.lm);
    for (auto& l : t->lay.layer) for (auto& a : l) v.push_back(&a);
    return v;
}

static int talloc(blim_trainer* t, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    t->owned.push_back(*p);
    return BLIM_OK;
}

extern "C" int64_t blim_train_flat_size(const blim_engine* e, int32_t lora_r) {
    if (!e || lora_r <= 0 || lora_r > 16) return -1;          // (blim_train_create / blim_load_adapter refuse r > 16: no layout for it either)
    return make_layout(e->c, lora_r).total;
}

extern "C" int blim_train_param_offset(const blim_engine* e, int32_t lora_r, const char* name, int64_t* offset, int64_t* rows, int64_t* cols) {
    ARG_CHECK(e && name && offset && rows && cols && lora_r > 0 && lora_r <= 16);
    const Layout L = make_layout(e->c, lora_r);
    const std::string n(name);
    if (n == "visual_head") { *offset = L.off_vh; *rows = e->c.mm_hidden_size; *cols = e->c.hidden_size; return BLIM_OK; }
    const size_t colon = n.rfind(':');
    if (colon == std::string::npos || colon + 2 != n.size() || (n[colon + 1] != 'A' && n[colon + 1] != 'B')) { blim_set_error("unknown trainable '%s'", name); return BLIM_ERR_ARG; }
    const std::string w = n.substr(0, colon);
    const bool isA = n[colon + 1] == 'A';
    const Adapter* a = nullptr;
    int li = -1; char rest[64] = "";
    if (w == "mlp.0.w") a = &L.mlp[0][0]; else if (w == "mlp.2.w") a = &L.mlp[0][1];
    else if (w == "tvg_mlp.0.w") a = &L.mlp[1][0]; else if (w == "tvg_mlp.2.w") a = &L.mlp[1][1];
    else if (w == "lm_head") a = &L.lm;
    else if (sscanf(w.c_str(), "layers.%d.%63s", &li, rest) == 2 && li >= 0 && li < e->c.num_layers) {
        const std::string q(rest);
        if (q == "q_proj.w") a = &L.layer[li][0]; else if (q == "k_proj.w") a = &L.layer[li][1];
        else if (q == "v_proj.w") a = &L.layer[li][2]; else if (q == "o_proj.w") a = &L.layer[li][3];
    }
    if (!a) { blim_set_error("unknown trainable '%s'", name); return BLIM_ERR_ARG; }
    *offset = isA ? a->offA : a->offB;
    *rows = isA ? lora_r : a->n_out;
    *cols = isA ? a->n_in : lora_r;
    return BLIM_OK;
}

// [N, K] (row stride K) -> [N, K + AUG] with the extra columns zero
static int make_aug(blim_trainer* t, uint16_t** dst, const void* src, int64_t N, int K) {
    TRY(talloc(t, (void**)dst, (size_t)N * (K + AUG) * 2));
    HIP_TRY(hipMemset2D((char*)*dst + (size_t)K * 2, (size_t)(K + AUG) * 2, 0, AUG * 2, N));
    HIP_TRY(hipMemcpy2D(*dst, (size_t)(K + AUG) * 2, src, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice));
    return BLIM_OK;
}
static int make_T(blim_trainer* t, uint16_t** dst, int64_t ldd, const void* src, int64_t N, int K, int mode, int rope_rows) {
    TRY(talloc(t, (void**)dst, (size_t)K * ldd * 2));
    if (ldd > N) HIP_TRY(hipMemset(*dst, 0, (size_t)K * ldd * 2));
    return launch_transpose16(*dst, ldd, (const uint16_t*)src, K, N, K, mode, rope_rows, 0);
}

extern "C" int blim_train_create(blim_engine* e, const blim_train_config* cfg, float* params, float* grads, blim_trainer** out) {
    ARG_CHECK(e && cfg && params && grads && out);
    ARG_CHECK(cfg->lora_r > 0 && cfg->lora_r <= 16 && cfg->lora_alpha > 0.f && cfg->lora_dropout >= 0.f && cfg->lora_dropout < 1.f);
    TRY(blim_weights_ready(e));
    // (an fp8 engine keeps its 16-bit matrices: a trainer on it can load and MERGE adapters -- evaluation of a fine-tuned checkpoint in fp8 mode --
    // but blim_train_step refuses to run)
    const blim_config& c = e->c;
    const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, M = c.mm_hidden_size;
    const int rope_rows = (c.num_heads + c.num_kv_heads) * 128;
    blim_trainer* t = new blim_trainer();
    t->e = e; t->r = cfg->lora_r; t->s = cfg->lora_alpha / (float)cfg->lora_r; t->p_drop = cfg->lora_dropout;
    t->params = params; t->grads = grads;
    t->lay = make_layout(c, t->r);
    t->L.resize(c.num_layers);
    t->Vp = (int)round_up(V, 64);
    int rc = BLIM_OK;
#define T_(expr) do { if (rc == BLIM_OK) rc = (expr); } while (0)
    for (Adapter* a : all_adapters(t)) {
        a->ldb = round_up(a->n_out, 64);
        T_(talloc(t, (void**)&a->Bt16, (size_t)16 * a->ldb * 2));
        if (rc == BLIM_OK && hipMemset(a->Bt16, 0, (size_t)16 * a->ldb * 2) != hipSuccess) rc = BLIM_ERR_HIP;
        T_(talloc(t, (void**)&a->A16, (size_t)16 * a->n_in * 2));
        if (rc == BLIM_OK && hipMemset(a->A16, 0, (size_t)16 * a->n_in * 2) != hipSuccess) rc = BLIM_ERR_HIP;
    }
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerW& w = e->L[l]; TrainLayerW& x = t->L[l];
        T_(make_aug(t, &x.wqkv_aug, w.wqkv, e->qkv_n, H));
        T_(make_aug(t, &x.wo_aug, w.wo, H, H));
        T_(make_T(t, &x.wqkvT, e->qkv_n, w.wqkv, e->qkv_n, H, 1, rope_rows));
        T_(make_T(t, &x.woT, H, w.wo, H, H, 0, 0));
        T_(make_T(t, &x.wguT, 2 * (int64_t)I, w.wgu, 2 * (int64_t)I, H, 0, 0));
        T_(make_T(t, &x.wdT, H, w.wd, H, I, 0, 0));
    }
    T_(make_aug(t, &t->lm_aug, e->lm_head, V, H));
    T_(make_T(t, &t->lmT, t->Vp, e->lm_head, V, H, 0, 0));
    for (int w = 0; w < 2; ++w) {
        T_(make_aug(t, &t->w0_aug[w], e->mlp_w0[w], H, M));
        T_(make_aug(t, &t->w2_aug[w], e->mlp_w2[w], H, H));
        T_(make_T(t, &t->w2T[w], H, e->mlp_w2[w], H, H, 0, 0));
    }
    T_(talloc(t, (void**)&t->vh16, (size_t)M * H * 2));
#undef T_
    if (rc == BLIM_OK && hipDeviceSynchronize() != hipSuccess) { blim_set_error("training weight copies failed"); rc = BLIM_ERR_HIP; }
    if (rc == BLIM_OK) rc = blim_train_sync_params(t, nullptr);
    if (rc == BLIM_OK && hipDeviceSynchronize() != hipSuccess) { blim_set_error("blim_train_sync_params failed"); rc = BLIM_ERR_HIP; }
    if (rc != BLIM_OK) { blim_train_destroy(t); return rc; }
    *out = t;
    return BLIM_OK;
}

extern "C" void blim_train_destroy(blim_trainer* t) {
    if (!t) return;
    hipDeviceSynchronize();
    for (void* p : t->owned) hipFree(p);
    DevBuf* bufs[] = {&t->red_scratch, &t->sv_res, &t->sv_mid, &t->sv_xn1, &t->sv_qkv, &t->sv_attn, &t->sv_gu, &t->sv_lse, &t->xn2, &t->act, &t->dres, &t->dy16, &t->dtmp32, &t->dqkv32, &t->dqkv16,
                      &t->du, &t->attn_D, &t->P16, &t->dS16, &t->logits, &t->dlog16, &t->hsel, &t->hsel_t, &t->dhsel, &t->feats_aug[0], &t->feats_aug[1], &t->pre16[0], &t->pre16[1], &t->h16[0], &t->h16[1],
                      &t->vid16, &t->proj16, &t->embeds, &t->dout16[0], &t->dout16[1], &t->dh32, &t->vh32, &t->vhb16, &t->dl32, &t->dvh};
    for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
    delete t;
}

extern "C" int blim_train_sync_params(blim_trainer* t, void* stream) {
    ARG_CHECK(t);
    hipStream_t s = (hipStream_t)stream;
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, M = c.mm_hidden_size, dt = c.compute_dtype, r = t->r;
    const int64_t qn = (int64_t)c.num_heads * 128, kn = (int64_t)c.num_kv_heads * 128;
    const float* P = t->params;
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& a = t->lay.layer[l];
        TRY(launch_lora_b_to_aug(t->L[l].wqkv_aug, H + AUG, 0, H, P + a[0].offB, (int)qn, r, 1, dt, s));
        TRY(launch_lora_b_to_aug(t->L[l].wqkv_aug, H + AUG, qn, H + r, P + a[1].offB, (int)kn, r, 1, dt, s));
        TRY(launch_lora_b_to_aug(t->L[l].wqkv_aug, H + AUG, qn + kn, H + 2 * r, P + a[2].offB, (int)kn, r, 0, dt, s));
        TRY(launch_lora_b_to_aug(t->L[l].wo_aug, H + AUG, 0, H, P + a[3].offB, H, r, 0, dt, s));
    }
    TRY(launch_lora_b_to_aug(t->lm_aug, H + AUG, 0, H, P + t->lay.lm.offB, c.vocab_size, r, 0, dt, s));
    for (int w = 0; w < 2; ++w) {
        TRY(launch_lora_b_to_aug(t->w0_aug[w], M + AUG, 0, M, P + t->lay.mlp[w][0].offB, H, r, 0, dt, s));
        TRY(launch_lora_b_to_aug(t->w2_aug[w], H + AUG, 0, H, P + t->lay.mlp[w][1].offB, H, r, 0, dt, s));
    }
    for (Adapter* a : all_adapters(t)) {
        TRY(launch_lora_bt(a->Bt16, a->ldb, P + a->offB, a->n_out, r, dt, s));
        TRY(launch_lora_a16(a->A16, P + a->offA, a->n_in, r, dt, s));
    }
    return launch_f32_to_16(t->vh16, H, P + t->lay.off_vh, H, M, H, 1.0f, dt, s);
}

extern "C" int blim_train_merge(blim_trainer* t, void* stream) {
    ARG_CHECK(t);
    hipStream_t s = (hipStream_t)stream;
    blim_engine* e = t->e;
    if (e->aug) {
        blim_set_error("blim_train_merge: the engine holds adapters apart (blim_load_adapter); merging into its base weights as well would apply the update twice -- blim_clear_adapters first");
        return BLIM_ERR_STATE;
    }
    const blim_config& c = e->c;
    const int H = c.hidden_size, M = c.mm_hidden_size, dt = c.compute_dtype, r = t->r;
    const int64_t qn = (int64_t)c.num_heads * 128, kn = (int64_t)c.num_kv_heads * 128;
    const float* P = t->params;
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& a = t->lay.layer[l];
        const int64_t row0[3] = {0, qn, qn + kn};
        const int n_out[3] = {(int)qn, (int)kn, (int)kn};
        for (int j = 0; j < 3; ++j)
            TRY(launch_lora_merge(e->L[l].wqkv, t->L[l].wqkv_aug, H + AUG, row0[j], P + a[j].offB, P + a[j].offA, n_out[j], H, r, t->s, j < 2 ? 1 : 0, dt, s));
        TRY(launch_lora_merge(e->L[l].wo, t->L[l].wo_aug, H + AUG, 0, P + a[3].offB, P + a[3].offA, H, H, r, t->s, 0, dt, s));
    }
    TRY(launch_lora_merge(e->lm_head, t->lm_aug, H + AUG, 0, P + t->lay.lm.offB, P + t->lay.lm.offA, c.vocab_size, H, r, t->s, 0, dt, s));
    for (int w = 0; w < 2; ++w) {
        TRY(launch_lora_merge(e->mlp_w0[w], t->w0_aug[w], M + AUG, 0, P + t->lay.mlp[w][0].offB, P + t->lay.mlp[w][0].offA, H, M, r, t->s, 0, dt, s));
        TRY(launch_lora_merge(e->mlp_w2[w], t->w2_aug[w], H + AUG, 0, P + t->lay.mlp[w][1].offB, P + t->lay.mlp[w][1].offA, H, H, r, t->s, 0, dt, s));
    }
    TRY(launch_f32_to_16(e->visual_head, H, P + t->lay.off_vh, H, M, H, 1.0f, dt, s));
    TRY(engine_set_visual_head3(e, P + t->lay.off_vh, BLIM_DTYPE_F32, s));      // the scoring path's hi + lo copy of the head
    e->f8_ready = false;
    e->lo6_ready = false;          // the combined 16-bit | e2m3 copies of the compensated modes' second pass follow the merged weights
    e->merged_pending.clear();
    for (int l = 0; l < c.num_layers; ++l) {
        e->c6_dirty.insert(e->L[l].wqkv); e->c6_dirty.insert(e->L[l].wo);
        for (const char* n : {"q_proj.w", "k_proj.w", "v_proj.w", "o_proj.w"}) e->merged_pending.insert("layers." + std::to_string(l) + "." + n);
    }
    e->c6_dirty.insert(e->lm_head);
    for (const char* n : {"lm_head", "mlp.0.w", "mlp.2.w", "tvg_mlp.0.w", "tvg_mlp.2.w"}) e->merged_pending.insert(n);
    e->lora_merged = true;
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- LoRA helpers
// x16 [n, ldx] carries u~ = s * A drop(x) in columns K + col ..; gradients of one adapter from dy16 [n, ldy] (columns of this projection)
static int lora_backward(blim_trainer* t, const Adapter& a, const uint16_t* dy16, int64_t ldy, const uint16_t* x16, int64_t ldx, int K, int col, int64_t n, float* du,
                         uint64_t seed, uint32_t site, hipStream_t s) {
    const int dt = t->e->c.compute_dtype;
    const int Np = (int)round_up(a.n_out, 16);
    // partial sums of the split reductions (summed in a fixed order by the launchers: no float atomics anywhere in the backward)
    TRY(ensure(t->red_scratch, std::max(std::max(lora_wgrad_scratch_bytes(n, a.n_out, t->r), lora_wgrad_scratch_bytes(n, K, t->r)), lora_du_scratch_bytes(n, Np, t->r))));
    float* scr = (float*)t->red_scratch.p;
    TRY(launch_lora_dB(t->grads + a.offB, dy16, ldy, x16 + K + col, ldx, n, a.n_out, t->r, dt, scr, s));
    TRY(launch_lora_du(du, dy16, ldy, a.Bt16, a.ldb, n, Np, t->r, t->s, dt, scr, s));     // dy columns beyond n_out (lm_head: up to Vp) are zero
    return launch_lora_dA(t->grads + a.offA, du, x16, ldx, n, K, t->r, t->p_drop, seed, site, dt, scr, s);
}
static int lora_dx1(float* dx, int64_t ldd, const float* du, const float* A, int64_t n, int K, int r, float p, uint64_t seed, uint32_t site, hipStream_t s) {
    LoraDxArgs a; a.n = 1; a.du[0] = du; a.A[0] = A; a.du[1] = a.du[2] = nullptr; a.A[1] = a.A[2] = nullptr;
    return launch_lora_dx(dx, ldd, a, n, K, r, p, seed, site, s);
}
// buffers of augmented rows [.., K + AUG]: the forward writes columns [0, K) and the adapters' u~ columns; the padding columns after
// them must be zero and nothing ever writes them, so they are cleared when the buffer is (re)allocated, not per step
// (cleared on the step's own stream: a memset on the null stream is not ordered against a non-blocking caller stream)
static int ensure_z(DevBuf& b, size_t bytes, hipStream_t s) {
    if (b.bytes >= bytes) return BLIM_OK;
    TRY(ensure(b, bytes));
    HIP_TRY(hipMemsetAsync(b.p, 0, b.bytes, s));
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- forward
// One projector (mm_projector_builder.py:88-93 with the adapters of main.py:96-98): which = 0 `mlp` -> rows [0, F) of vid16 (the VTG rows'
// video tokens), which = 1 `tvg_mlp` + mean over each clip's tokens (modeling_videochat_flash.py:243) -> rows [F, F + F / tok) (the TVG rows'
// clip tokens).  Both keep their own saved activations: the two backward passes run after the shared decoder backward.
static int projector_forward(blim_trainer* t, const blim_train_batch* b, int which, hipStream_t s) {
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, M = c.mm_hidden_size, dt = c.compute_dtype, r = t->r;
    const int Ha = H + AUG, Ma = M + AUG;
    const int64_t F = b->n_feat_rows;
    TRY(ensure_z(t->feats_aug[which], (size_t)F * Ma * 2, s)); TRY(ensure(t->pre16[which], (size_t)F * H * 2)); TRY(ensure_z(t->h16[which], (size_t)F * Ha * 2, s));
    uint16_t* fa = (uint16_t*)t->feats_aug[which].p; uint16_t* pre = (uint16_t*)t->pre16[which].p; uint16_t* h16 = (uint16_t*)t->h16[which].p;
    uint16_t* vid = (uint16_t*)t->vid16.p;
    HIP_TRY(hipMemcpy2DAsync(fa, (size_t)Ma * 2, b->feats, (size_t)M * 2, (size_t)M * 2, F, hipMemcpyDeviceToDevice, s));
    LoraDownArgs la; la.n = 1; la.A16[1] = la.A16[2] = nullptr;
    la.A16[0] = t->lay.mlp[which][0].A16;
    TRY(launch_lora_down(fa, Ma, F, M, la, r, t->s, t->p_drop, b->dropout_seed, 1000 + 2 * which, dt, s));
    {
        GemmParams p = gpt(dt, fa, Ma, t->w0_aug[which], F, H, Ma, pre, H);
        p.bias = e->mlp_b0[which];
        TRY(launch_gemm(EPI_BF16, p, s));
    }
    TRY(launch_gelu_fwd(h16, Ha, pre, F, H, dt, s));
    la.A16[0] = t->lay.mlp[which][1].A16;
    TRY(launch_lora_down(h16, Ha, F, H, la, r, t->s, t->p_drop, b->dropout_seed, 1001 + 2 * which, dt, s));
    uint16_t* proj = which == 0 ? vid : (uint16_t*)t->proj16.p;
    {
        GemmParams p = gpt(dt, h16, Ha, t->w2_aug[which], F, H, Ha, proj, H);
        p.bias = e->mlp_b2[which];
        TRY(launch_gemm(EPI_BF16, p, s));
    }
    if (which == 1) TRY(launch_group_mean((bf16_t*)(vid + F * H), (const bf16_t*)proj, F / b->tok_per_clip, b->tok_per_clip, H, dt, s));
    return BLIM_OK;
}

// both projectors + sequence assembly + the decoder over ONE packed batch holding the VTG rows and the TVG rows (the TVG rows are a
// tenth of the tokens: as a pass of their own they ran at a third of the VTG pass's efficiency), keeping activations
static int train_forward(blim_trainer* t, const blim_train_batch* b, hipStream_t s) {
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, I = c.intermediate_size, dt = c.compute_dtype, r = t->r;
    const int Ha = H + AUG, qn = e->qkv_n;
    const int64_t T = b->batch->n_tokens, F = b->n_feat_rows;
    const int NL = c.num_layers;
    ARG_CHECK(F > 0 && T > 0 && b->tok_per_clip > 0 && F % b->tok_per_clip == 0);
    TRY(ensure(t->vid16, (size_t)(F + F / b->tok_per_clip) * H * 2)); TRY(ensure(t->proj16, (size_t)F * H * 2)); TRY(ensure(t->embeds, (size_t)T * H * 2));
    if (b->n_rows > 0) TRY(projector_forward(t, b, 0, s));
    if (b->n_tvg_rows > 0) TRY(projector_forward(t, b, 1, s));
    TRY(launch_assemble((bf16_t*)t->embeds.p, b->src_index, T, H, e->embed, (const bf16_t*)t->vid16.p, s));
    // ---- decoder
    TRY(ensure(t->sv_res, (size_t)(NL + 1) * T * H * 4)); TRY(ensure(t->sv_mid, (size_t)NL * T * H * 4));
    TRY(ensure_z(t->sv_xn1, (size_t)NL * T * Ha * 2, s)); TRY(ensure(t->sv_qkv, (size_t)NL * T * qn * 2)); TRY(ensure_z(t->sv_attn, (size_t)NL * T * Ha * 2, s));
    TRY(ensure(t->sv_gu, (size_t)NL * T * 2 * I * 2));
    TRY(ensure(t->sv_lse, (size_t)NL * T * c.num_heads * 4));
    TRY(ensure(t->xn2, (size_t)T * H * 2)); TRY(ensure(t->act, (size_t)T * I * 2));
    float* res = (float*)t->sv_res.p; float* mid = (float*)t->sv_mid.p;
    t->last_T = T;
    TRY(launch_h16_to_f32(res, (const bf16_t*)t->embeds.p, T * H, dt, s));
    float* rope_rows = nullptr; int64_t rope_stride = 0;
    TRY(engine_rope_rows(e, b->batch, s, &rope_rows, &rope_stride));
    for (int li = 0; li < NL; ++li) {
        const LayerW& l = e->L[li]; const TrainLayerW& x = t->L[li]; const auto& ad = t->lay.layer[li];
        float* x_in = res + (int64_t)li * T * H; float* x_mid = mid + (int64_t)li * T * H; float* x_out = res + (int64_t)(li + 1) * T * H;
        uint16_t* xn1 = (uint16_t*)t->sv_xn1.p + (int64_t)li * T * Ha; uint16_t* qkv = (uint16_t*)t->sv_qkv.p + (int64_t)li * T * qn;
        uint16_t* attn = (uint16_t*)t->sv_attn.p + (int64_t)li * T * Ha; uint16_t* gu = (uint16_t*)t->sv_gu.p + (int64_t)li * T * 2 * I;
        TRY(launch_rmsnorm(x_in, H, nullptr, T, H, l.norm1, c.rms_eps, (bf16_t*)xn1, dt, nullptr, s, 0, Ha, nullptr, false));
        LoraDownArgs q3; q3.n = 3; for (int j = 0; j < 3; ++j) q3.A16[j] = ad[j].A16;
        TRY(launch_lora_down(xn1, Ha, T, H, q3, r, t->s, t->p_drop, b->dropout_seed, 8 * li, dt, s));
        {
            GemmParams p = gpt(dt, xn1, Ha, x.wqkv_aug, T, qn, Ha, qkv, qn);
            p.bias = l.bqkv; p.rope_cols = (c.num_heads + c.num_kv_heads) * 128; p.rope_rows = rope_rows; p.rope_stride = rope_stride;
            TRY(launch_gemm(EPI_QKV, p, s));
        }
        {
            AttnParams a;
            memset(&a, 0, sizeof(a));
            a.dtype = dt; a.qkv = (const bf16_t*)qkv; a.ldq = qn; a.num_heads = c.num_heads; a.num_kv_heads = c.num_kv_heads;
            a.key_visible = b->batch->key_visible; a.seq_start = b->batch->seq_start; a.seq_len = b->batch->seq_len; a.pfx_start = b->batch->pfx_start; a.pfx_len = b->batch->pfx_len;
            a.blk_seq = b->batch->blk_seq; a.blk_q0 = b->batch->blk_q0; a.n_blocks = b->batch->n_blocks; a.out = (bf16_t*)attn; a.ldo = Ha; a.scale = 0.08838834764831845f;
            a.lse_out = (float*)t->sv_lse.p + (int64_t)li * T * c.num_heads;
            TRY(launch_attention(a, e->attn_tr, s));
        }
        LoraDownArgs o1; o1.n = 1; o1.A16[0] = ad[3].A16; o1.A16[1] = o1.A16[2] = nullptr;
        TRY(launch_lora_down(attn, Ha, T, H, o1, r, t->s, t->p_drop, b->dropout_seed, 8 * li + 3, dt, s));
        {
            GemmParams p = gpt(dt, attn, Ha, x.wo_aug, T, H, Ha, x_mid, H);
            p.resid_in = x_in;                                             // x_mid = x_in + o_proj(attn): the layer input stays intact for the backward
            TRY(launch_gemm(EPI_RESID, p, s));
        }
        TRY(launch_rmsnorm(x_mid, H, nullptr, T, H, l.norm2, c.rms_eps, (bf16_t*)t->xn2.p, dt, nullptr, s, 0, 0, nullptr, false));
        {   // gate | up pre-activations kept for the backward; act = silu(gate) * up formed in the same epilogue
            GemmParams p = gpt(dt, t->xn2.p, H, l.wgu, T, 2 * I, H, gu, 2 * (int64_t)I);
            p.swiglu_act = (uint16_t*)t->act.p; p.swiglu_act_ld = I;
            TRY(launch_gemm(EPI_BF16, p, s));
        }
        {
            GemmParams p = gpt(dt, t->act.p, I, l.wd, T, H, I, x_out, H);
            p.resid_in = x_mid;
            TRY(launch_gemm(EPI_RESID, p, s));
        }
    }
    return BLIM_OK;
}

// ---------------------------------------------------------------------------- backward
// dres = gradient w.r.t. the last layer's output on entry, w.r.t. the input embeddings on exit
static int train_backward_layers(blim_trainer* t, const blim_train_batch* b, hipStream_t s) {
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, I = c.intermediate_size, dt = c.compute_dtype, r = t->r;
    const int Ha = H + AUG, qn = e->qkv_n;
    const int64_t T = t->last_T;
    const int64_t qcols[3] = {0, (int64_t)c.num_heads * 128, (int64_t)(c.num_heads + c.num_kv_heads) * 128};
    TRY(ensure(t->dy16, (size_t)T * H * 2)); TRY(ensure(t->dtmp32, (size_t)T * H * 4)); TRY(ensure(t->dqkv32, (size_t)T * qn * 4)); TRY(ensure(t->dqkv16, (size_t)T * qn * 2));
    TRY(ensure(t->du, (size_t)T * 3 * 16 * 4));
    const int64_t Lm = attn_bwd_lm(b->max_seq_len);
    const size_t mats = (size_t)b->batch->n_seqs * c.num_heads * Lm * Lm;
    TRY(ensure(t->attn_D, (size_t)T * c.num_heads * 4)); TRY(ensure(t->P16, mats * 2)); TRY(ensure(t->dS16, mats * 2));
    float* dres = (float*)t->dres.p; uint16_t* dy16 = (uint16_t*)t->dy16.p; float* dtmp = (float*)t->dtmp32.p;
    float* dqkv32 = (float*)t->dqkv32.p; uint16_t* dqkv16 = (uint16_t*)t->dqkv16.p; float* du = (float*)t->du.p;
    uint16_t* dattn16 = (uint16_t*)t->xn2.p;      // [T, H] 16-bit scratch (the forward's normalised MLP input is not needed any more)
    float* res = (float*)t->sv_res.p; float* mid = (float*)t->sv_mid.p;
    for (int li = c.num_layers - 1; li >= 0; --li) {
        const LayerW& l = e->L[li]; const TrainLayerW& x = t->L[li]; const auto& ad = t->lay.layer[li];
        float* x_in = res + (int64_t)li * T * H; float* x_mid = mid + (int64_t)li * T * H;
        uint16_t* xn1 = (uint16_t*)t->sv_xn1.p + (int64_t)li * T * Ha; uint16_t* qkv = (uint16_t*)t->sv_qkv.p + (int64_t)li * T * qn;
        uint16_t* attn = (uint16_t*)t->sv_attn.p + (int64_t)li * T * Ha; uint16_t* gu = (uint16_t*)t->sv_gu.p + (int64_t)li * T * 2 * I;
        // ---- MLP block: x_out = x_mid + down(silu(gate(n2)) * up(n2)), n2 = rmsnorm(x_mid); dy16 = 16-bit(dres) comes from the previous RMSNorm backward
        if (li == c.num_layers - 1) TRY(launch_f32_to_16(dy16, H, dres, H, T, H, 1.0f, dt, s));
        { GemmParams p = gpt(dt, dy16, H, x.wdT, T, I, H, t->act.p, I); p.swiglu_gu = gu; p.swiglu_ld = 2 * (int64_t)I; TRY(launch_gemm(EPI_BF16, p, s)); }   // d act = dy . Wd, and in the
                                                                                                                         // epilogue gu <- [d gate | d up] (no d act round trip)
        { GemmParams p = gpt(dt, gu, 2 * (int64_t)I, x.wguT, T, H, 2 * I, dtmp, H); TRY(launch_gemm(EPI_F32, p, s)); }    // d n2
        TRY(launch_rmsnorm_bwd(dres, dtmp, x_mid, nullptr, T, H, l.norm2, c.rms_eps, 1, dy16, dt, s));                   // dres = d x_mid (+ its 16-bit copy)
        // ---- attention block: x_mid = x_in + o_proj(attn), attn = Attention(rope(qkv(n1))), n1 = rmsnorm(x_in)
        TRY(lora_backward(t, ad[3], dy16, H, attn, Ha, H, 0, T, du, b->dropout_seed, 8 * li + 3, s));
        { GemmParams p = gpt(dt, dy16, H, x.woT, T, H, H, dtmp, H); TRY(launch_gemm(EPI_F32, p, s)); }                    // d attn (base path)
        {   // d attn = base path + the o_proj adapter's input gradient, written straight as the attention backward's 16-bit operand
            LoraDxArgs a1; a1.n = 1; a1.du[0] = du; a1.A[0] = t->params + ad[3].offA; a1.du[1] = a1.du[2] = nullptr; a1.A[1] = a1.A[2] = nullptr;
            TRY(launch_lora_dx(dtmp, H, a1, T, H, r, t->p_drop, b->dropout_seed, 8 * li + 3, s, dattn16, H, dt));
        }
        {
            AttnBwdParams a;
            memset(&a, 0, sizeof(a));
            a.dtype = dt; a.qkv = qkv; a.ldq = qn; a.dout = dattn16; a.ldo = H; a.o16 = attn; a.ldo16 = Ha; a.num_heads = c.num_heads; a.num_kv_heads = c.num_kv_heads;
            a.lse = (const float*)t->sv_lse.p + (int64_t)li * T * c.num_heads;
            a.key_visible = b->batch->key_visible; a.seq_start = b->batch->seq_start; a.seq_len = b->batch->seq_len; a.n_seqs = b->batch->n_seqs; a.max_len = b->max_seq_len;
            a.scale = 0.08838834764831845f; a.D = (float*)t->attn_D.p; a.P16 = (uint16_t*)t->P16.p; a.dS16 = (uint16_t*)t->dS16.p; a.dqkv = dqkv32;
            TRY(launch_attention_bwd(a, T, s));
        }
        TRY(launch_rope_bwd(dqkv16, dqkv32, T, qn, (c.num_heads + c.num_kv_heads) * 128, b->batch->positions, e->rope_cos, e->rope_sin, c.max_positions, dt, s));
        for (int j = 0; j < 3; ++j)      // (a single pass over x for the three dA was tried: slower, 48 accumulators per lane)
            TRY(lora_backward(t, ad[j], dqkv16 + qcols[j], qn, xn1, Ha, H, j * r, T, du + (int64_t)j * T * r, b->dropout_seed, 8 * li + j, s));
        { GemmParams p = gpt(dt, dqkv16, qn, x.wqkvT, T, H, qn, dtmp, H); TRY(launch_gemm(EPI_F32, p, s)); }              // d n1 (base path)
        {   // dres = d x_in; the three adapters' input gradients join dtmp inside the RMSNorm backward
            LoraDxArgs a3; a3.n = 3;
            for (int j = 0; j < 3; ++j) { a3.du[j] = du + (int64_t)j * T * r; a3.A[j] = t->params + ad[j].offA; }
            if (H <= 4096) TRY(launch_rmsnorm_bwd(dres, dtmp, x_in, nullptr, T, H, l.norm1, c.rms_eps, 1, li > 0 ? dy16 : nullptr, dt, s, &a3, r, t->p_drop, b->dropout_seed, 8 * li));
            else {
                TRY(launch_lora_dx(dtmp, H, a3, T, H, r, t->p_drop, b->dropout_seed, 8 * li, s));
                TRY(launch_rmsnorm_bwd(dres, dtmp, x_in, nullptr, T, H, l.norm1, c.rms_eps, 1, li > 0 ? dy16 : nullptr, dt, s));
            }
        }
    }
    return BLIM_OK;
}

static int train_backward_projector(blim_trainer* t, const blim_train_batch* b, int which, hipStream_t s) {
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, M = c.mm_hidden_size, dt = c.compute_dtype, r = t->r;
    const int Ha = H + AUG, Ma = M + AUG;
    const int64_t T = t->last_T, F = b->n_feat_rows;
    TRY(ensure(t->dh32, (size_t)F * H * 4)); TRY(ensure(t->du, (size_t)std::max<int64_t>(F, T) * 3 * 16 * 4));
    uint16_t* dout = (uint16_t*)t->dout16[which].p; float* dh = (float*)t->dh32.p; float* du = (float*)t->du.p;
    const Adapter& a2 = t->lay.mlp[which][1]; const Adapter& a0 = t->lay.mlp[which][0];
    TRY(lora_backward(t, a2, dout, H, (const uint16_t*)t->h16[which].p, Ha, H, 0, F, du, b->dropout_seed, 1001 + 2 * which, s));
    { GemmParams p = gpt(dt, dout, H, t->w2T[which], F, H, H, dh, H); TRY(launch_gemm(EPI_F32, p, s)); }
    TRY(lora_dx1(dh, H, du, t->params + a2.offA, F, H, r, t->p_drop, b->dropout_seed, 1001 + 2 * which, s));
    TRY(launch_gelu_bwd(dout, dh, (const uint16_t*)t->pre16[which].p, F, H, dt, s));                                    // dout <- d pre-activation
    return lora_backward(t, a0, dout, H, (const uint16_t*)t->feats_aug[which].p, Ma, M, 0, F, du, b->dropout_seed, 1000 + 2 * which, s);
}

static int check_train_batch(const blim_trainer* t, const blim_train_batch* b) {
    ARG_CHECK(t && b && b->batch && b->src_index && b->feats && b->max_seq_len > 0 && (b->n_rows > 0 || b->n_tvg_rows > 0));
    ARG_CHECK(b->n_rows == 0 || (b->rows && b->labels));
    ARG_CHECK(b->n_tvg_rows == 0 || (b->tvg_rows && b->tvg_labels && b->vocab && b->n_vocab > 0 && b->n_tvg_rows % t->e->c.num_clips == 0));
    TRY(check_batch(b->batch));
    return BLIM_OK;
}

// VTG head: final norm at the scored rows, lm_head (+ adapter), cross-entropy (training_utils.py:23-32: mean over the label tokens), and
// back to the residual-stream gradient of those rows (dres rows are disjoint from the TVG head's)
static int vtg_head(blim_trainer* t, const blim_train_batch* b, float* loss_sum, hipStream_t s) {
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, V = c.vocab_size, dt = c.compute_dtype, r = t->r, Ha = H + AUG, Vp = t->Vp;
    const int64_t T = t->last_T, R = b->n_rows;
    const float* x_final = (const float*)t->sv_res.p + (int64_t)c.num_layers * T * H;
    TRY(ensure_z(t->hsel, (size_t)R * Ha * 2, s)); TRY(ensure(t->logits, (size_t)R * Vp * 4)); TRY(ensure(t->dlog16, (size_t)R * Vp * 2)); TRY(ensure(t->dhsel, (size_t)R * H * 4));
    TRY(ensure(t->du, (size_t)std::max<int64_t>(R, T) * 3 * 16 * 4));
    uint16_t* hsel = (uint16_t*)t->hsel.p; float* logits = (float*)t->logits.p; uint16_t* dlog = (uint16_t*)t->dlog16.p; float* dhsel = (float*)t->dhsel.p; float* du = (float*)t->du.p;
    TRY(launch_rmsnorm(x_final, H, b->rows, R, H, e->final_norm, c.rms_eps, (bf16_t*)hsel, dt, nullptr, s, T, Ha, nullptr, false));
    LoraDownArgs la; la.n = 1; la.A16[0] = t->lay.lm.A16; la.A16[1] = la.A16[2] = nullptr;
    TRY(launch_lora_down(hsel, Ha, R, H, la, r, t->s, t->p_drop, b->dropout_seed, 2000, dt, s));
    { GemmParams p = gpt(dt, hsel, Ha, t->lm_aug, R, V, Ha, logits, Vp); TRY(launch_gemm(EPI_F32, p, s)); }
    TRY(ensure(t->red_scratch, (size_t)R * 4));
    TRY(launch_ce_fwd_bwd(logits, Vp, V, b->labels, 1, R, b->grad_scale / (float)R, dlog, nullptr, Vp, loss_sum, dt, (float*)t->red_scratch.p, s));
    TRY(lora_backward(t, t->lay.lm, dlog, Vp, hsel, Ha, H, 0, R, du, b->dropout_seed, 2000, s));
    { GemmParams p = gpt(dt, dlog, Vp, t->lmT, R, H, Vp, dhsel, H); TRY(launch_gemm(EPI_F32, p, s)); }
    TRY(lora_dx1(dhsel, H, du, t->params + t->lay.lm.offA, R, H, r, t->p_drop, b->dropout_seed, 2000, s));
    return launch_rmsnorm_bwd((float*)t->dres.p, dhsel, x_final, b->rows, R, H, e->final_norm, c.rms_eps, 0, nullptr, dt, s);
}

// TVG head (training_utils.py:71-79): hidden at the 4 positions before <|im_end|> -> visual_head -> . video_vocab / sqrt(M) -> CE over the N videos
static int tvg_head(blim_trainer* t, const blim_train_batch* b, float* loss_sum, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    blim_engine* e = t->e;
    const blim_config& c = e->c;
    const int H = c.hidden_size, M = c.mm_hidden_size, dt = c.compute_dtype, C = c.num_clips, N = b->n_vocab;
    const int BC = (int)b->n_tvg_rows, B = BC / C;
    const int64_t T = t->last_T;
    const float* x_final = (const float*)t->sv_res.p + (int64_t)c.num_layers * T * H;
    TRY(ensure(t->hsel_t, (size_t)BC * H * 2)); TRY(ensure(t->vh32, (size_t)BC * M * 4)); TRY(ensure(t->vhb16, (size_t)BC * M * 2)); TRY(ensure(t->logits, (size_t)BC * N * 4));
    TRY(ensure(t->dl32, (size_t)BC * N * 4)); TRY(ensure(t->dvh, (size_t)BC * M * 4)); TRY(ensure(t->dhsel, (size_t)BC * H * 4));
    uint16_t* hsel = (uint16_t*)t->hsel_t.p; float* vh32 = (float*)t->vh32.p; uint16_t* vhb = (uint16_t*)t->vhb16.p; float* logits = (float*)t->logits.p;
    float* dl = (float*)t->dl32.p; float* dvh = (float*)t->dvh.p; float* dhsel = (float*)t->dhsel.p;
    TRY(launch_rmsnorm(x_final, H, b->tvg_rows, BC, H, e->final_norm, c.rms_eps, (bf16_t*)hsel, dt, nullptr, s, T, 0, nullptr, false));
    { GemmParams p = gpt(dt, hsel, H, t->vh16, BC, M, H, vh32, M); TRY(launch_gemm(EPI_F32, p, s)); }
    TRY(launch_f32_to_16(vhb, M, vh32, M, BC, M, 1.0f, dt, s));
    TRY(blim_tvg_logits(e, vhb, b->vocab, N, B, logits, stream));
    TRY(ensure(t->red_scratch, (size_t)BC * 4));
    TRY(launch_ce_fwd_bwd(logits, N, N, b->tvg_labels, C, BC, b->grad_scale / (float)BC, nullptr, dl, N, loss_sum, dt, (float*)t->red_scratch.p, s));
    TRY(launch_tvg_dvh(dvh, dl, (const uint16_t*)b->vocab, BC, C, N, M, 1.0f / sqrtf((float)M), dt, s));
    TRY(launch_outer_acc(t->grads + t->lay.off_vh, dvh, hsel, H, BC, M, H, dt, s));
    TRY(launch_rows_matmul(dhsel, dvh, t->params + t->lay.off_vh, BC, M, H, s));
    return launch_rmsnorm_bwd((float*)t->dres.p, dhsel, x_final, b->tvg_rows, BC, H, e->final_norm, c.rms_eps, 0, nullptr, dt, s);
}

extern "C" int blim_train_step(blim_trainer* t, const blim_train_batch* b, float* loss_sums, void* stream) {
    if (t && t->e && t->e->f8) { blim_set_error("training needs a 16-bit engine (fp16, as the reference's autocast, or bf16), not fp8"); return BLIM_ERR_STATE; }
    ARG_CHECK(loss_sums);
    TRY(check_train_batch(t, b));
    hipStream_t s = (hipStream_t)stream;
    const blim_config& c = t->e->c;
    const int H = c.hidden_size, dt = c.compute_dtype;
    TRY(train_forward(t, b, s));
    const int64_t T = t->last_T, F = b->n_feat_rows;
    TRY(ensure(t->dres, (size_t)T * H * 4));
    HIP_TRY(hipMemsetAsync(t->dres.p, 0, (size_t)T * H * 4, s));
    if (b->n_rows > 0) TRY(vtg_head(t, b, loss_sums, s));
    if (b->n_tvg_rows > 0) TRY(tvg_head(t, b, loss_sums + 1, stream));
    TRY(train_backward_layers(t, b, s));
    // d embeds -> the two projectors' output gradients (VTG video tokens: rows [0, F) of vid16; TVG clip tokens: rows F.. = clip means)
    TRY(ensure(t->dout16[0], (size_t)F * H * 2)); TRY(ensure(t->dout16[1], (size_t)F * H * 2));
    if (b->n_rows > 0) HIP_TRY(hipMemsetAsync(t->dout16[0].p, 0, (size_t)F * H * 2, s));
    if (b->n_tvg_rows > 0) HIP_TRY(hipMemsetAsync(t->dout16[1].p, 0, (size_t)F * H * 2, s));
    TRY(launch_feat_grad((uint16_t*)t->dout16[0].p, (uint16_t*)t->dout16[1].p, (const float*)t->dres.p, b->src_index, T, H, F, b->tok_per_clip, dt, s));
    if (b->n_rows > 0) TRY(train_backward_projector(t, b, 0, s));
    if (b->n_tvg_rows > 0) TRY(train_backward_projector(t, b, 1, s));
    return BLIM_OK;
}

extern "C" int blim_train_grad_stats(blim_trainer* t, float inv_scale, float* stats, void* stream) {
    ARG_CHECK(t && stats);
    TRY(ensure(t->red_scratch, 1024 * 4));
    return launch_grad_stats(t->grads, t->lay.total, inv_scale, stats, (float*)t->red_scratch.p, (hipStream_t)stream);
}

extern "C" int blim_train_adamw(blim_trainer* t, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay, float inv_scale,
                                int32_t step, void* stream) {
    ARG_CHECK(t && exp_avg && exp_avg_sq && step >= 1);
    const float c1 = 1.0f - powf(beta1, (float)step), c2 = 1.0f - powf(beta2, (float)step);
    TRY(launch_adamw(t->params, t->grads, exp_avg, exp_avg_sq, t->lay.total, lr, beta1, beta2, eps, weight_decay, inv_scale, c1, c2, (hipStream_t)stream));
    return blim_train_sync_params(t, stream);
}

extern "C" int blim_train_debug_read(blim_trainer* t, const char* which, void* dst, int64_t bytes, void* stream) {
    ARG_CHECK(t && which && dst && bytes > 0);
    const blim_config& c = t->e->c;
    const int H = c.hidden_size, I = c.intermediate_size, qn = t->e->qkv_n, Ha = H + AUG;
    const int64_t T = t->last_T;
    const std::string w(which);
    const void* src = nullptr;
    int li = 0;
    if (w == "dres") src = t->dres.p;
    else if (w == "embeds") src = t->embeds.p;
    else if (w == "logits") src = t->logits.p;
    else if (sscanf(which, "res%d", &li) == 1 && li >= 0 && li <= c.num_layers) src = (const float*)t->sv_res.p + (int64_t)li * T * H;
    else if (sscanf(which, "mid%d", &li) == 1 && li >= 0 && li < c.num_layers) src = (const float*)t->sv_mid.p + (int64_t)li * T * H;
    else if (sscanf(which, "xn%d", &li) == 1 && li >= 0 && li < c.num_layers) src = (const uint16_t*)t->sv_xn1.p + (int64_t)li * T * Ha;
    else if (sscanf(which, "qkv%d", &li) == 1 && li >= 0 && li < c.num_layers) src = (const uint16_t*)t->sv_qkv.p + (int64_t)li * T * qn;
    else if (sscanf(which, "attn%d", &li) == 1 && li >= 0 && li < c.num_layers) src = (const uint16_t*)t->sv_attn.p + (int64_t)li * T * Ha;
    else if (sscanf(which, "gu%d", &li) == 1 && li >= 0 && li < c.num_layers) src = (const uint16_t*)t->sv_gu.p + (int64_t)li * T * 2 * I;
    if (!src) { blim_set_error("unknown trainer buffer '%s'", which); return BLIM_ERR_ARG; }
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return BLIM_OK;
}
