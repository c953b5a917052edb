// bf16 MFMA GEMM for gfx950: 256x256x64 tiles, 8 waves (2 M x 4 N), each wave 128x64 of C as
// 8x4 fragments of v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous ([rows][K]); tiles go
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave-instruction) into a
// double buffer.  LDS image of a 1-KiB block (8 rows x 64 k):  [k-quarter q (32 B)][row 0..7][32 B],
// i.e. bank-row q holds bytes [32q, 32q+32) of all 8 rows.  A 16-row fragment read (ds_read_b128, lane
// = (row l&15, 16-B chunk l>>4)) then touches 16 distinct 16-B slots per hardware lane group:
// conflict-free, with the permutation applied on the per-lane global SOURCE address (the LDS-DMA
// destination is lane-linear).
#include "gemm.hpp"

#define BM 256
#define BN 256
#define BK 64
#define NTHREADS 512
#define TILE_BYTES (BM * BK * 2)           // 32 KiB per operand tile
#define BUF_BYTES (2 * TILE_BYTES)         // A + B
#define GROUP_M 8

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// 4x4 transpose inside each quad of lanes: in: lane c holds v[j] = X[row j][col c];
// out: lane c holds v[j] = X[row c][col j].
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, int lane) {
    const bool odd = lane & 1;
    float s0 = odd ? v0 : v1, s1 = odd ? v2 : v3;
    float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
    if (odd) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
    const bool hi = lane & 2;
    s0 = hi ? v0 : v2; s1 = hi ? v1 : v3;
    r0 = __shfl_xor(s0, 2); r1 = __shfl_xor(s1, 2);
    if (hi) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// PIPE 0: all 8 waves in lock step, two 64-KB LDS stages, one barrier per K-step (bring-up structure, kept for A/B runs).
// PIPE 1: ping-pong -- waves 0-3 and 4-7 (SIMD partners) alternate {LDS fragment reads} and {64 MFMAs} so each SIMD's
//         matrix pipe always has one wave computing; two 64-KB stages, LDS-DMA for K-step k+2 issued as one 64-KB burst.
// PIPE 2: ping-pong over a RING of five 32-KB operand slots (A or W tile of one K-step): one operand tile is issued per
//         half K-step (A(k+2) at the start of phase A(k), W(k+2) at the start of phase B(k)) and waited for with a
//         counted vmcnt, so the L2->LDS pipe never drains (measured: a drained 64-KB burst per step moves 44 GB/s per
//         CU, the same bytes issued as alternating 32-KB tiles 62 GB/s -- tools/dma_bench.hip).
template <int EPI, int PIPE, int DT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[PIPE == 2 ? 5 * TILE_BYTES : 2 * BUF_BYTES];  // 160 / 128 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- tile mapping: XCD-contiguous chunks (blocks b, b+8, ... share an XCD), then grouped M order
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int pid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int width = GROUP_M * ntn;
    const int first_m = (pid / width) * GROUP_M;
    const int gsz = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (pid % width) % gsz;
    const int tn = (pid % width) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- per-lane LDS-DMA sources: wave w stages blocks w, w+8, w+16, w+24 of A and of W.
    // 32-bit byte offsets from the (wave-uniform, scalar) operand bases: the K-step advance is a scalar add on the
    // base and the LDS-DMA uses the saddr + voffset form (no per-step vector address arithmetic).
    const int sq = lane >> 4, sr = (lane >> 1) & 7, sc = 2 * sq + (lane & 1);  // LDS slot lane -> (row sr, chunk sc)
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = wave + 8 * i;
        const int ra = min(row0 + 8 * b + sr, p.M - 1);
        const int rb = min(col0 + 8 * b + sr, p.N - 1);
        offA[i] = (uint32_t)(((int64_t)ra * p.lda + 8 * sc) * 2);
        offB[i] = (uint32_t)(((int64_t)rb * p.K + 8 * sc) * 2);
    }
    const char* baseA = (const char*)p.A;
    const char* baseW = (const char*)p.W;
    // one operand tile (32 KiB, 4 LDS-DMA per wave) of K-step kt into the LDS tile at byte offset `dst`
    auto stage_tile = [&](bool isW, int dst, int kt) __attribute__((always_inline)) {
        const char* g = (isW ? baseW : baseA) + (int64_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(g + (isW ? offB[i] : offA[i]), smem + dst + (wave + 8 * i) * 1024);
    };

    // ---- fragment read offsets (bytes) inside an operand tile
    const int fr = lane & 15, fc = lane >> 4;
    const int frag_off = (fr >> 3) * 1024 + (fr & 7) * 32 + (fc >> 1) * 256 + (fc & 1) * 16;
    const int a_off = (16 * wm) * 1024 + frag_off;   // + mi*2048 + ks*512
    const int b_off = (8 * wn) * 1024 + frag_off;    // + ni*2048 + ks*512

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    if constexpr (PIPE == 0) {
        stage_tile(false, 0, 0); stage_tile(true, TILE_BYTES, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) { stage_tile(false, (cur ^ 1) * BUF_BYTES, kt + 1); stage_tile(true, (cur ^ 1) * BUF_BYTES + TILE_BYTES, kt + 1); }
            const char* base = smem + cur * BUF_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[8], b[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8*)(base + TILE_BYTES + b_off + ni * 2048 + ks * 512);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) a[mi] = *(const bf16x8*)(base + a_off + mi * 2048 + ks * 512);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = mfma16<DT>(a[mi], b[ni], acc[mi][ni]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        const int grp = wave >> 2;  // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
        bf16x8 fa[2][8], fb[2][4];
        auto load_frags = [&](int offA_tile, int offB_tile) __attribute__((always_inline)) {
            const char* ba = smem + offA_tile + a_off;
            const char* bb = smem + offB_tile + b_off;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) fb[ks][ni] = *(const bf16x8*)(bb + ni * 2048 + ks * 512);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) fa[ks][mi] = *(const bf16x8*)(ba + mi * 2048 + ks * 512);
            }
        };
        auto compute = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = mfma16<DT>(fa[ks][mi], fb[ks][ni], acc[mi][ni]);
            __builtin_amdgcn_s_setprio(0);
        };
        // a phase boundary: own LDS reads complete (WAR on the tile about to be refilled), then rendezvous
#define PHASE_BARRIER()                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory")

        // Both variants run two straight-line loops (one per wave group) that execute the SAME barrier sequence:
        //   phase A(kt): group 0 computes step kt              | group 1 reads its fragments of step kt
        //   phase B(kt): group 0 reads its fragments of kt+1   | group 1 computes step kt
        if constexpr (PIPE == 1) {
            stage_tile(false, 0, 0); stage_tile(true, TILE_BYTES, 0);
            if (nk > 1) { stage_tile(false, BUF_BYTES, 1); stage_tile(true, BUF_BYTES + TILE_BYTES, 1); }
            if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // K-step 0 landed (8 LDS-DMA per wave per step)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PHASE_BARRIER();
            if (grp == 0) {
                load_frags(0, TILE_BYTES);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // K-step 1 landed
                PHASE_BARRIER();
                for (int kt = 0; kt < nk; ++kt) {
                    const int cur = (kt & 1) * BUF_BYTES, nxt = BUF_BYTES - cur;
                    compute();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // step kt+1 (issued one K-step ago) landed
                    PHASE_BARRIER();
                    if (kt + 2 < nk) { stage_tile(false, cur, kt + 2); stage_tile(true, cur + TILE_BYTES, kt + 2); }
                    if (kt + 1 < nk) load_frags(nxt, nxt + TILE_BYTES);
                    PHASE_BARRIER();
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PHASE_BARRIER();
                for (int kt = 0; kt < nk; ++kt) {
                    const int cur = (kt & 1) * BUF_BYTES;
                    load_frags(cur, cur + TILE_BYTES);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PHASE_BARRIER();
                    if (kt + 2 < nk) { stage_tile(false, cur, kt + 2); stage_tile(true, cur + TILE_BYTES, kt + 2); }
                    compute();
                    PHASE_BARRIER();
                }
            }
        } else {
            // ---- ring of five operand tiles.  Operand n = 2*step + isW lives in slot n % 5.
            // Issue order A0 W0 A1 W1 | A2 W2 A3 W3 ... : A(k+2) at the start of phase A(k) (into the slot W(k-1) left),
            // W(k+2) at the start of phase B(k) (into the slot A(k) left).  At the end of phase A(k) the step k+1 tiles
            // must have landed; the only younger LDS-DMA is A(k+2) (4 per wave) -> s_waitcnt vmcnt(4), or vmcnt(0) once
            // nothing younger is issued any more.
            constexpr int RING = 5 * TILE_BYTES;
            auto adv = [](int s, int j) { const int x = s + j * TILE_BYTES; return x >= RING ? x - RING : x; };
            stage_tile(false, 0, 0); stage_tile(true, TILE_BYTES, 0);
            if (nk > 1) { stage_tile(false, 2 * TILE_BYTES, 1); stage_tile(true, 3 * TILE_BYTES, 1); }
            if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // A0 W0 landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PHASE_BARRIER();
            int sa = 0;   // slot (byte offset) of A(kt); W(kt) = adv(sa,1), A(kt+1) = adv(sa,2), W(kt+1) = adv(sa,3), A(kt+2) = adv(sa,4), W(kt+2) = sa
            if (grp == 0) {
                load_frags(0, TILE_BYTES);
                PHASE_BARRIER();
                for (int kt = 0; kt < nk; ++kt) {
                    // phase A(kt)
                    if (kt + 2 < nk) stage_tile(false, adv(sa, 4), kt + 2);
                    compute();
                    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // A(kt+1) W(kt+1) landed
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PHASE_BARRIER();
                    // phase B(kt)
                    if (kt + 2 < nk) stage_tile(true, sa, kt + 2);
                    if (kt + 1 < nk) load_frags(adv(sa, 2), adv(sa, 3));
                    PHASE_BARRIER();
                    sa = adv(sa, 2);
                }
            } else {
                PHASE_BARRIER();
                for (int kt = 0; kt < nk; ++kt) {
                    if (kt + 2 < nk) stage_tile(false, adv(sa, 4), kt + 2);
                    load_frags(sa, adv(sa, 1));
                    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PHASE_BARRIER();
                    if (kt + 2 < nk) stage_tile(true, sa, kt + 2);
                    compute();
                    PHASE_BARRIER();
                    sa = adv(sa, 2);
                }
            }
        }
#undef PHASE_BARRIER
    }

    // =========================================================================== epilogues
    const int wrow0 = row0 + 128 * wm;  // wave's first row
    const int wcol0 = col0 + 64 * wn;   // wave's first column (in W's row order)

    if constexpr (EPI == EPI_LSE) {
        // accumulators as the MFMA leaves them: lane holds col (lane&15) of frag ni, rows 4*(lane>>4)+j of frag mi.
        float2* red = (float2*)smem;  // [4 wn][256 rows]
        const int q4 = lane >> 4;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rl = 128 * wm + 16 * mi + 4 * q4 + j;  // row inside the tile
                const int row = row0 + rl;
                const int lab = (row < p.M) ? p.labels[row] : -1;
                float v[4];
                float mx = -INFINITY;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int col = wcol0 + 16 * ni + fr;
                    v[ni] = (col < p.N) ? acc[mi][ni][j] : -INFINITY;
                    if (col == lab) p.label_logit[row] = v[ni];
                    mx = fmaxf(mx, v[ni]);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                float sm = 0.f;
                if (mx > -INFINITY) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) sm += __expf(v[ni] - mx);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                if (fr == 0) red[wn * 256 + rl] = make_float2(mx, sm);
            }
        }
        __syncthreads();
        if (tid < 256) {
            const int row = row0 + tid;
            if (row < p.M) {
                float2 a0 = red[tid], a1 = red[256 + tid], a2 = red[512 + tid], a3 = red[768 + tid];
                const float mx = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
                float sm = 0.f;
                if (mx > -INFINITY) {
                    sm = a0.y * __expf(a0.x - mx) + a1.y * __expf(a1.x - mx) + a2.y * __expf(a2.x - mx) + a3.y * __expf(a3.x - mx);
                }
                p.lse_part[(int64_t)row * ntn + tn] = make_float2(mx, sm);
            }
        }
        return;
    } else {
        const int tq = (lane & 15) >> 2;        // which 4-col group of the fragment this lane owns after the transpose
        const int rsub = 4 * (lane >> 4) + (lane & 3);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int row = wrow0 + 16 * mi + rsub;
            float t[4][4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                float v0 = acc[mi][ni][0], v1 = acc[mi][ni][1], v2 = acc[mi][ni][2], v3 = acc[mi][ni][3];
                quad_transpose(v0, v1, v2, v3, lane);
                t[ni][0] = v0; t[ni][1] = v1; t[ni][2] = v2; t[ni][3] = v3;
            }
            if (row >= p.M) continue;
            if constexpr (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_RESID) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int col = wcol0 + 16 * ni + 4 * tq;
                    if (col >= p.N) continue;
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = t[ni][j];
                    if constexpr (EPI != EPI_RESID) {
                        if (p.bias) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) x[j] += (col + j < p.N) ? p.bias[col + j] : 0.f;
                        }
                    }
                    if constexpr (EPI == EPI_BF16) {
                        if (p.act == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) x[j] = gelu_erf(x[j]);
                        }
                        bf16_t* out = (bf16_t*)p.C + (int64_t)row * p.ldc + col;
                        if (col + 3 < p.N) {
                            uint2 pk = make_uint2(pack2<DT>(x[0], x[1]), pack2<DT>(x[2], x[3]));
                            *(uint2*)out = pk;
                        } else {
                            for (int j = 0; j < 4 && col + j < p.N; ++j) out[j] = to16<DT>(x[j]);
                        }
                    } else if constexpr (EPI == EPI_F32) {
                        float* out = (float*)p.C + (int64_t)row * p.ldc + col;
                        if (col + 3 < p.N && (p.ldc & 3) == 0) {
                            *(float4*)out = make_float4(x[0] * p.scale, x[1] * p.scale, x[2] * p.scale, x[3] * p.scale);
                        } else {
                            for (int j = 0; j < 4 && col + j < p.N; ++j) out[j] = x[j] * p.scale;
                        }
                    } else {  // EPI_RESID
                        float* out = (float*)p.C + (int64_t)row * p.ldc + col;
                        if (col + 3 < p.N) {
                            float4 o = *(float4*)out;
                            o.x += x[0]; o.y += x[1]; o.z += x[2]; o.w += x[3];
                            *(float4*)out = o;
                        } else {
                            for (int j = 0; j < 4 && col + j < p.N; ++j) out[j] += x[j];
                        }
                    }
                }
            } else if constexpr (EPI == EPI_QKV) {
                // fragments (2p, 2p+1) hold RoPE partners d and d+64 for q/k heads; v heads are in natural order.
                const int head = wcol0 >> 7;
                if (wcol0 < p.rope_cols) {
                    const int pos = p.pos[row];
                    const int gbase = ((wcol0 & 127) >> 5);  // 0 or 2
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int cst = wcol0 + 32 * pr + 4 * tq;       // stored col of the lo element
                        const int d = 16 * (gbase + pr) + 4 * tq;      // natural d of the lo element (0..63)
                        const float4 cs = *(const float4*)(p.rope_cos + (int64_t)pos * 64 + d);
                        const float4 sn = *(const float4*)(p.rope_sin + (int64_t)pos * 64 + d);
                        const float c4[4] = {cs.x, cs.y, cs.z, cs.w}, s4[4] = {sn.x, sn.y, sn.z, sn.w};
                        float lo[4], hi[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x1 = t[2 * pr][j] + (p.bias ? p.bias[cst + j] : 0.f);
                            const float x2 = t[2 * pr + 1][j] + (p.bias ? p.bias[cst + 16 + j] : 0.f);
                            lo[j] = x1 * c4[j] - x2 * s4[j];
                            hi[j] = x2 * c4[j] + x1 * s4[j];
                        }
                        bf16_t* out = (bf16_t*)p.C + (int64_t)row * p.ldc + head * 128 + d;
                        *(uint2*)out = make_uint2(pack2<DT>(lo[0], lo[1]), pack2<DT>(lo[2], lo[3]));
                        *(uint2*)(out + 64) = make_uint2(pack2<DT>(hi[0], hi[1]), pack2<DT>(hi[2], hi[3]));
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const int col = wcol0 + 16 * ni + 4 * tq;
                        if (col >= p.N) continue;
                        float x[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[j] = t[ni][j] + (p.bias ? p.bias[col + j] : 0.f);
                        bf16_t* out = (bf16_t*)p.C + (int64_t)row * p.ldc + col;
                        *(uint2*)out = make_uint2(pack2<DT>(x[0], x[1]), pack2<DT>(x[2], x[3]));
                    }
                }
            } else if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int cst = wcol0 + 32 * pr;  // stored col of this 32-group
                    if (cst >= p.N) continue;
                    const int oc = (cst >> 5) * 16 + 4 * tq;  // output (intermediate) column
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = silu_f(t[2 * pr][j]) * t[2 * pr + 1][j];
                    bf16_t* out = (bf16_t*)p.C + (int64_t)row * p.ldc + oc;
                    *(uint2*)out = make_uint2(pack2<DT>(x[0], x[1]), pack2<DT>(x[2], x[3]));
                }
            }
        }
    }
}

#include <stdlib.h>
static int g_gemm_pipe = getenv("BLIM_GEMM_PIPE") ? atoi(getenv("BLIM_GEMM_PIPE")) : 2;
void gemm_set_pipe(int pipe) { g_gemm_pipe = pipe; }

template <int EPI, int PIPE>
static void launch_p(const GemmParams& p, dim3 grid, hipStream_t stream) {
    if (p.dtype == DT_F16) hipLaunchKernelGGL((gemm_kernel<EPI, PIPE, DT_F16>), grid, dim3(NTHREADS), 0, stream, p);
    else hipLaunchKernelGGL((gemm_kernel<EPI, PIPE, DT_BF16>), grid, dim3(NTHREADS), 0, stream, p);
}

template <int EPI>
static int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const dim3 grid(ntm * ntn);
#ifdef BLIM_GEMM_ALL_PIPES
    if (g_gemm_pipe == 0) launch_p<EPI, 0>(p, grid, stream);
    else if (g_gemm_pipe == 1) launch_p<EPI, 1>(p, grid, stream);
    else
#endif
    launch_p<EPI, 2>(p, grid, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        blim_set_error("gemm launch failed: %s", hipGetErrorString(e));
        return BLIM_ERR_HIP;
    }
    return BLIM_OK;
}

int launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
    ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0);
    ARG_CHECK(p.K % BK == 0);
    ARG_CHECK(p.lda % 8 == 0);
    ARG_CHECK(p.A && p.W);
    ARG_CHECK(p.dtype == DT_BF16 || p.dtype == DT_F16);
    ARG_CHECK((int64_t)p.M * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.K * 2 < (1ll << 32));  // 32-bit operand offsets
    switch (epi) {
        case EPI_BF16: ARG_CHECK(p.C && p.ldc % 4 == 0); return launch_t<EPI_BF16>(p, stream);
        case EPI_F32: ARG_CHECK(p.C); return launch_t<EPI_F32>(p, stream);
        case EPI_RESID: ARG_CHECK(p.C && p.ldc % 4 == 0); return launch_t<EPI_RESID>(p, stream);
        case EPI_QKV:
            ARG_CHECK(p.C && p.pos && p.rope_cos && p.rope_sin && p.N % 128 == 0 && p.rope_cols % 128 == 0 && p.ldc % 4 == 0);
            return launch_t<EPI_QKV>(p, stream);
        case EPI_SWIGLU: ARG_CHECK(p.C && p.N % 32 == 0 && p.ldc % 4 == 0); return launch_t<EPI_SWIGLU>(p, stream);
        case EPI_LSE: ARG_CHECK(p.labels && p.lse_part && p.label_logit); return launch_t<EPI_LSE>(p, stream);
    }
    blim_set_error("unknown epilogue %d", (int)epi);
    return BLIM_ERR_ARG;
}
