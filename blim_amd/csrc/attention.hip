// Flash-style GQA attention for gfx950, head_dim 128, packed tokens, shared-prefix KV reuse.
//
// One workgroup = one 32-query block of one sequence x one KV head; its G = num_heads/num_kv_heads
// waves are the G query heads that share that KV head, so a K/V tile is staged into LDS once per
// group.  Per 32-key tile each wave computes S^T = K.Q^T (keys on MFMA rows, queries on lanes:
// "swapped" product, v_mfma_f32_32x32x16_bf16), so a lane owns ONE query: its running max / sum are
// lane-local and only the two 32-lane halves are combined.  The S^T accumulator, exponentiated and
// packed to bf16, is directly the B operand of O^T = V^T.P^T; the V^T A-operand comes from the
// row-major V tile by the transposed LDS read ds_read_b64_tr_b16.
//
// LDS images (16-B chunk c of key row r):  K: chunk c ^ (r & 15)   (ds_read_b128 of 32 rows, conflict-free)
//                                          V: chunk c ^ ((r & 3) << 2)  (4 rows of a tr-read block on 4 bank quarters)
#include "attention.hpp"

#include <stdlib.h>

#define HD 128
#define KT 32  // keys per tile
#define QB 32  // queries per block

typedef __attribute__((ext_vector_type(4))) short s16x4;

// MAXC = 16-byte staging chunks per thread and tile = ceil(chunks per tile / threads); threads = 64 x (query heads per KV head).  The launch bound follows
// the head grouping (G >= 4: up to 512 threads; G = 2, 3: <= 192; G = 1: 64), so the few-thread variants -- whose threads each stage a large share of
// the tile -- get the registers of the waves that are not there instead of spilling (MHA, compensated: 450 VGPRs spilled under a 512-thread bound).
template <int MAXC, bool SPLIT> struct attn_bound { static constexpr int value = MAXC <= (SPLIT ? 8 : 4) ? 512 : MAXC <= (SPLIT ? 16 : 8) ? 256 : 64; };
template <bool USE_TR, int MAXC, int DT, bool SPLIT = false>
__global__ __launch_bounds__((attn_bound<MAXC, SPLIT>::value)) void attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t k_lds[(SPLIT ? 2 : 1) * KT * HD];   // SPLIT: K_hi tile, then K_lo tile
    __shared__ __attribute__((aligned(16))) bf16_t v_lds[(SPLIT ? 2 : 1) * KT * HD];   // SPLIT: V_hi tile, then V_lo tile
    __shared__ uint32_t vis_lds[KT / 4];  // 32 visibility bytes

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = p.num_heads / p.num_kv_heads;
    // Head groups (launch_attention, plain kernels with G >= 5): the G query heads of a KV head are split over `ngrp` workgroups of <= 4 waves, so that TWO
    // workgroups fit a CU (8 waves at ~200 VGPRs = 2 per SIMD) and one's barriers / load latency are covered by the other's work; each stages the K / V tiles itself.
    const int ngrp = (int)gridDim.y / p.num_kv_heads;
    const int kh = blockIdx.y / ngrp, hg = blockIdx.y - kh * ngrp;
    const int hpg = (G + ngrp - 1) / ngrp;               // heads per group
    const int wave_head = hg * hpg + wave;               // this wave's query head inside the KV group
    // Waves beyond the query heads of this group are HELPERS (compensated kernels with few heads per group, launch_attention): they take their share
    // of every tile's staging -- 2,048 16-byte chunks in the compensated mode, which G <= 6 waves had to hold as 8 - 32 staging registers each (7 - 64
    // spilled VGPRs) -- and the barriers, and compute nothing.
    const bool worker = wave < hpg && wave_head < G;
    const int head = kh * G + (worker ? wave_head : 0);
    const int blk = blockIdx.x;
    const int s = p.blk_seq[blk], q0 = p.blk_q0[blk];
    const int sstart = p.seq_start[s], slen = p.seq_len[s];
    const int plen = p.pfx_len[s];
    const int pstart = plen > 0 ? p.pfx_start[s] : 0;

    const int qi = lane & 31, hf = lane >> 5;
    const int qtok = sstart + min(q0 + qi, slen - 1);
    const int qstart = p.own_start ? p.own_start[qtok] : 0;      // segmented sequences: this query's first own key
    const int koff = p.num_heads * HD + kh * HD;
    const int voff = (p.num_heads + p.num_kv_heads) * HD + kh * HD;

    // Q fragments (B operand of S^T = K.Q^T): lane (q = lane&31, half hf) holds Q[q][16ks + 8hf .. +7]
    bf16x8 qf[8];
    {
        const bf16_t* qrow = p.qkv + (int64_t)qtok * p.ldq + head * HD + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
    }
    bf16x8 qf_lo[SPLIT ? 8 : 1];      // compensated mode: Q = Q_hi + Q_lo (lo parts p.v_lo_off columns further)
    if constexpr (SPLIT) {
        const bf16_t* qrow = p.qkv + (int64_t)qtok * p.ldq + p.v_lo_off + head * HD + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf_lo[ks] = *(const bf16x8*)(qrow + 16 * ks);
    }

    // Q has landed before the first K/V tile is requested: inside the loop the only vector-memory operations in flight are then the NEXT tile's
    // loads.  Without this the waitcnt pass could not tell, on the loop's back edge, that the Q loads were long complete and put counted vmcnt
    // waits in front of the S MFMAs -- which, counting in issue order, waited for the next tile's loads too: the prefetch it was meant to
    // overlap was serialised in front of the compute (a third of the kernel's time at the reference's shapes, tools/attn_classes.py).
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) expcnt(7) lgkmcnt(15)
    const int n_ptiles = (plen + KT - 1) / KT;
    const int own_keys = min(q0 + QB, slen);
    const int n_tiles = n_ptiles + (own_keys + KT - 1) / KT;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    const float NEG = -1.0e30f;
    float m_run = NEG, l_run = 0.f;
    const float c_log2 = p.scale * 1.4426950408889634f;

    // staging: 1024 16-B chunks per tile (512 K + 512 V; SPLIT: + 512 V_lo + 512 K_lo = 2048) spread over the workgroup's threads
    // (MAXC * nthreads >= 1024 / 2048: launch_attention picks MAXC from the group size)
    constexpr int NCHUNK = SPLIT ? 2048 : 1024;
    uint4 st[MAXC];
    uint8_t st_vis = 0;                      // this thread's key-visibility byte of the tile in flight (threads 0 .. 31)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) st[i] = make_uint4(0, 0, 0, 0);

    auto tile_desc = [&](int t, int& base_tok, int& k0, int& seg_len, bool& causal) __attribute__((always_inline)) {
        if (t < n_ptiles) { base_tok = pstart; k0 = t * KT; seg_len = plen; causal = false; }
        else { base_tok = sstart; k0 = (t - n_ptiles) * KT; seg_len = slen; causal = true; }
    };
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        int base_tok, k0, seg_len; bool causal;
        tile_desc(t, base_tok, k0, seg_len, causal);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < NCHUNK) {
                const int isv = idx >> 9, row = (idx >> 4) & 31, ch = idx & 15;   // isv: 0 K, 1 V (hi), 2 V_lo, 3 K_lo
                const int kk = min(k0 + row, seg_len - 1);
                st[i] = *(const uint4*)(p.qkv + (int64_t)(base_tok + kk) * p.ldq + (isv == 0 ? koff : isv == 1 ? voff : isv == 2 ? voff + p.v_lo_off : koff + p.v_lo_off) + 8 * ch);
            }
        }
        if (tid < KT) {                      // requested with the tile (read inside store_tile it was a dependent global load in front of a barrier, every tile)
            const int kk = k0 + tid;
            st_vis = (kk < seg_len) ? p.key_visible[base_tok + kk] : (uint8_t)0;
        }
    };
    auto store_tile = [&](int t) __attribute__((always_inline)) {
        int base_tok, k0, seg_len; bool causal;
        tile_desc(t, base_tok, k0, seg_len, causal);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < NCHUNK) {
                const int isv = idx >> 9, row = (idx >> 4) & 31, ch = idx & 15;
                if (isv == 1 || isv == 2) *(uint4*)(v_lds + (isv - 1) * KT * HD + row * HD + 8 * (ch ^ ((row & 3) << 2))) = st[i];
                else *(uint4*)(k_lds + (isv >> 1) * KT * HD + row * HD + 8 * (ch ^ (row & 15))) = st[i];
            }
        }
        if (tid < KT) ((uint8_t*)vis_lds)[tid] = st_vis ? 1 : 0;
    };

    load_tile(0);
    for (int t = 0; t < n_tiles; ++t) {
#if defined(ATTN_ABLATE) && ATTN_ABLATE == 1     // ablation builds only (timing; wrong results): the tile is staged once, every iteration computes on it
        if (t == 0) { __syncthreads(); store_tile(t); __syncthreads(); }
#else
        __syncthreads();  // everyone finished reading the previous tile
        store_tile(t);
        __syncthreads();
        if (t + 1 < n_tiles) load_tile(t + 1);
#endif
#if defined(ATTN_ABLATE) && ATTN_ABLATE == 2     // ... or: staging only, no MFMA / softmax work
        continue;
#endif
        if (!worker) continue;                       // (wave-uniform)

        int base_tok, k0, seg_len; bool causal;
        tile_desc(t, base_tok, k0, seg_len, causal);

        // ---- S^T = K . Q^T : A = K fragment (lane: key = lane&31, chunk 2ks+hf), B = qf[ks]
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        {
            const int key = lane & 31;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(k_lds + key * HD + 8 * ((2 * ks + hf) ^ (key & 15)));
                sacc = mfma32<DT>(kf, qf[ks], sacc);
                if constexpr (SPLIT) {   // (K_hi + K_lo).(Q_hi + Q_lo) without the lo.lo term
                    sacc = mfma32<DT>(kf, qf_lo[ks], sacc);
                    const bf16x8 kl = *(const bf16x8*)(k_lds + KT * HD + key * HD + 8 * ((2 * ks + hf) ^ (key & 15)));
                    sacc = mfma32<DT>(kl, qf[ks], sacc);
                }
            }
        }
        // ---- mask + online softmax; this lane's keys: krow(r) = (r&3) + 8*(r>>2) + 4*hf.  Bookkeeping in RAW score units (the
        // scale * log2(e) factor rides in the exponent's fma); the masks are applied only where a tile needs them (a tile that reaches the
        // causal diagonal or the segment end, or holds an invisible key -- a wave-uniform test), and the accumulators are rescaled only
        // when some lane's running maximum grew: this VALU work, not the MFMAs, is what bounds the kernel.
        float pv[16];
        const bool vis_all = (vis_lds[0] & vis_lds[1] & vis_lds[2] & vis_lds[3] & vis_lds[4] & vis_lds[5] & vis_lds[6] & vis_lds[7]) == 0x01010101u;
        if (!vis_all || (causal && (k0 + KT - 1 > q0 || p.own_start != nullptr))) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t vb = vis_lds[2 * g + hf];  // bytes for keys 8g + 4hf + 0..3
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const int kk = k0 + 8 * g + 4 * hf + j;
                    // branch-free on purpose (bitwise &, |; one select): with short-circuit operators and a conditional store the compiler built this
                    // as divergent control flow around whole-vector copies of the 16 score registers, and the result was wrong (tools/dbg_wide.py)
                    const bool ok = (((vb >> (8 * j)) & 0xFF) != 0) & (!causal | ((kk <= q0 + qi) & (kk >= qstart)));
                    sacc[r] = ok ? sacc[r] : NEG;
                }
            }
        }
        float tmax = NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[r]);
        tmax = xhalf_max(tmax);
        // Lazy reference maximum: m_run is the exponent's reference, not necessarily the running maximum.  It is moved (and the accumulators
        // rescaled) only when some query's tile maximum exceeds it by more than 2^8 in the exponent's units -- p then stays below 256, well inside
        // both 16-bit formats' range, and l_run / the accumulators are sums relative to the same reference, so the result is the same
        // softmax; what disappears is the 64-multiply rescale on nearly every tile of a row whose maximum creeps up (this VALU work, not
        // the MFMAs, bounds the kernel: ~300 vector instructions per tile and wave before this change against 16 MFMAs).
        float m_new = m_run;
        const bool move = tmax * c_log2 > m_run * c_log2 + 8.0f;                    // per QUERY: a query's reference never depends on its block neighbours
        if (__builtin_amdgcn_ballot_w64(move) != 0) {
            m_new = move ? fmaxf(m_run, tmax) : m_run;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_log2);   // first tile: exp2(-huge) = 0 on zero accumulators; queries that stay: exactly 1
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        // a row with no visible key so far keeps m = NEG: its masked scores must give p = 0, not exp2(NEG - NEG) = 1 -- one select per tile on the
        // reference (mc = 0 there: exp2(NEG * c) = 0) instead of one per element
        const float mc = m_new > 0.5f * NEG ? m_new * c_log2 : 0.f;
        float rsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sacc[r], c_log2, -mc));        // native v_exp_f32: arguments <= 8, results below 2^-126 flush to 0
            pv[r] = e;
            rsum += e;
        }
        // the two lane halves are added per TILE, not once after the loop: a merged sequence's segment lands in either half depending on its offset, and
        // a deferred sum would associate differently there -- scores would depend on the batch composition in the last bit (two-rank == one-rank test)
        l_run += xhalf_sum(rsum);
        m_run = m_new;

        // ---- P^T fragments: registers 8s..8s+7 are k-step s (k order: 16s + 8(j>>2) + 4hf + (j&3))
        bf16x8 pf[2], pf_lo[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t w[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a_ = pv[8 * s2 + 2 * j], b_ = pv[8 * s2 + 2 * j + 1];
                w[j] = pack2<DT>(a_, b_);
                if constexpr (SPLIT) wl[j] = pack2<DT>(a_ - from16<DT>(to16<DT>(a_)), b_ - from16<DT>(to16<DT>(b_)));
            }
            pf[s2] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            if constexpr (SPLIT) pf_lo[s2] = __builtin_bit_cast(bf16x8, make_uint4(wl[0], wl[1], wl[2], wl[3]));
        }
        // ---- O^T += V^T . P^T : A = V^T fragment (lane: d = 32db + (lane&31); keys 16s+4hf+{0..3} and +8)
#pragma unroll
        for (int vp = 0; vp < (SPLIT ? 2 : 1); ++vp) {
        const bf16_t* vt = v_lds + vp * KT * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 vf;
                if constexpr (USE_TR) {
                    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
                    const int dcol = 32 * db + 16 * g16 + 4 * (i16 & 3);  // first of this lane's 4 columns
                    const int ch = dcol >> 3, within = dcol & 7;
                    const int kr0 = 16 * s2 + 4 * hf + (i16 >> 2);
                    const int kr1 = kr0 + 8;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vt + kr0 * HD + 8 * (ch ^ ((kr0 & 3) << 2)) + within));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vt + kr1 * HD + 8 * (ch ^ ((kr1 & 3) << 2)) + within));
                    vf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                } else {
                    const int d = 32 * db + (lane & 31);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kr = 16 * s2 + 8 * (j >> 2) + 4 * hf + (j & 3);
                        vf[j] = (short)vt[kr * HD + 8 * ((d >> 3) ^ ((kr & 3) << 2)) + (d & 7)];
                    }
                }
                o[db] = mfma32<DT>(vf, pf[s2], o[db]);
                if constexpr (SPLIT) { if (vp == 0) o[db] = mfma32<DT>(vf, pf_lo[s2], o[db]); }   // V_hi.P_lo (V_lo.P_lo dropped)
            }
        }
        }
    }

    // ---- normalise and store: o[db][r] is O[q = lane&31][d = 32db + (r&3) + 8(r>>2) + 4hf]
    if (!worker) return;
    if (p.out8 != nullptr) {
        // fp8 mode: this wave's 128 outputs of a token are one K-step of the o_proj GEMM -> e4m3 with one power-of-two scale (E8M0):
        // the query's maximum is in two lanes (hf = 0, 1)
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float a = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[db][r] *= inv; a = fmaxf(a, fabsf(o[db][r])); }
        a = xhalf_max(a);
        int e = 0;
        if (a > 0.f) {
            int ex; const float mant = frexpf(a * (1.0f / FP8_MAX), &ex);
            e = (mant == 0.5f) ? ex - 1 : ex;
            if (ldexpf(a, -e) > FP8_MAX) e += 1;
            e = max(-127, min(127, e));
        }
        const float qs = ldexpf(1.0f, -e);
        if (q0 + qi < slen) {
            const int64_t tok = sstart + q0 + qi;
            uint8_t* orow = p.out8 + tok * p.ldo8 + head * HD;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint32_t*)(orow + 32 * db + 8 * g + 4 * hf) = pack_fp8x4(o[db][4 * g] * qs, o[db][4 * g + 1] * qs, o[db][4 * g + 2] * qs, o[db][4 * g + 3] * qs);
            if (hf == 0) {
                const int rl = (int)(tok & 255);
                p.out_mx[(int64_t)head * p.mx_stride + (tok >> 8) * 256 + ((rl >> 7) * 16 + (rl & 15)) * 8 + ((rl >> 4) & 7)] = (uint8_t)(e + 127);
            }
        }
        return;
    }
    if (q0 + qi < slen) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* orow = p.out + (int64_t)(sstart + q0 + qi) * p.ldo + head * HD;
        if (p.lse_out && hf == 0) p.lse_out[(int64_t)(sstart + q0 + qi) * p.num_heads + head] = l_run > 0.f ? m_run * p.scale + __logf(l_run) : 1.0e30f;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * db + 8 * g + 4 * hf;
                const float x0 = o[db][4 * g] * inv, x1 = o[db][4 * g + 1] * inv, x2 = o[db][4 * g + 2] * inv, x3 = o[db][4 * g + 3] * inv;
                *(uint2*)(orow + d) = make_uint2(pack2<DT>(x0, x1), pack2<DT>(x2, x3));
                if constexpr (SPLIT)
                    *(uint2*)(orow + p.out_lo_off + d) = make_uint2(pack2<DT>(x0 - from16<DT>(to16<DT>(x0)), x1 - from16<DT>(to16<DT>(x1))),
                                                                    pack2<DT>(x2 - from16<DT>(to16<DT>(x2)), x3 - from16<DT>(to16<DT>(x3))));
            }
    }
}

int launch_attention(const AttnParams& p, int use_tr_read, hipStream_t stream) {
    ARG_CHECK(p.qkv && p.out && p.key_visible && p.seq_start && p.seq_len && p.pfx_start && p.pfx_len && p.blk_seq && p.blk_q0);
    ARG_CHECK(p.n_blocks > 0 && p.num_kv_heads > 0 && p.num_heads % p.num_kv_heads == 0);
    ARG_CHECK(p.ldq % 8 == 0 && p.ldo % 4 == 0);
    const int G = p.num_heads / p.num_kv_heads;
    if (G > 8) { blim_set_error("attention: %d query heads per kv head > 8 unsupported", G); return BLIM_ERR_ARG; }
    const dim3 grid(p.n_blocks, p.num_kv_heads), block(64 * G);
    if (p.v_lo_off != 0 || p.out_lo_off != 0) {   // compensated mode (fp16 engines): transposed-read path only
        ARG_CHECK((p.dtype == DT_F16 || p.dtype == DT_BF16) && p.v_lo_off > 0 && p.out_lo_off > 0 && p.v_lo_off % 8 == 0 && p.out_lo_off % 4 == 0);
#define ATTN_SPLIT(MC, BLOCK)                                                                                          \
        do {                                                                                                         \
            if (p.dtype == DT_F16) hipLaunchKernelGGL((attn_kernel<true, MC, DT_F16, true>), grid, BLOCK, 0, stream, p);  \
            else hipLaunchKernelGGL((attn_kernel<true, MC, DT_BF16, true>), grid, BLOCK, 0, stream, p);               \
        } while (0)
        if (G >= 7) ATTN_SPLIT(5, block);           // 2,048 chunks / 448 (512) threads: five per thread, not eight (the 7B model's grouping)
        else ATTN_SPLIT(4, dim3(512));              // fewer heads per group: eight waves, G of them compute and the rest only help staging (four chunks per thread;
                                                    // as G waves with 8 / 16 / 32 staging chunks each these kernels spilled 7 / 11 / 64 VGPRs)
#undef ATTN_SPLIT
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { blim_set_error("attention launch failed: %s", hipGetErrorString(e2)); return BLIM_ERR_HIP; }
        return BLIM_OK;
    }
#define ATTN_LAUNCH(TR, MC)                                                                             \
    do {                                                                                            \
        if (p.dtype == DT_F16) hipLaunchKernelGGL((attn_kernel<TR, MC, DT_F16>), grid, block, 0, stream, p); \
        else hipLaunchKernelGGL((attn_kernel<TR, MC, DT_BF16>), grid, block, 0, stream, p);                \
    } while (0)
    static const int g_split = getenv("BLIM_ATTN_HEAD_GROUPS") ? atoi(getenv("BLIM_ATTN_HEAD_GROUPS")) : 2;
    if (G >= 5 && g_split > 1 && use_tr_read && !p.out8) {
        // two workgroups of ceil(G / 2) waves per (block, KV head): 1,024 chunks / 256 threads = four per thread
        const int hpg = (G + 1) / 2;
        const dim3 grid2(p.n_blocks, p.num_kv_heads * 2), block2(64 * hpg);
        if (p.dtype == DT_F16) hipLaunchKernelGGL((attn_kernel<true, 4, DT_F16>), grid2, block2, 0, stream, p);
        else hipLaunchKernelGGL((attn_kernel<true, 4, DT_BF16>), grid2, block2, 0, stream, p);
    } else if (G >= 4) { if (use_tr_read) ATTN_LAUNCH(true, 4); else ATTN_LAUNCH(false, 4); }
    else if (G >= 2) { if (use_tr_read) ATTN_LAUNCH(true, 8); else ATTN_LAUNCH(false, 8); }
    else { if (use_tr_read) ATTN_LAUNCH(true, 16); else ATTN_LAUNCH(false, 16); }
#undef ATTN_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { blim_set_error("attention launch failed: %s", hipGetErrorString(e)); return BLIM_ERR_HIP; }
    return BLIM_OK;
}
