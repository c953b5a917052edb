// MFMA GEMM for gfx950 (bf16 / fp16 / fp8-e4m3 operands): 256x256 tiles x 128 bytes of K per step, 8 waves (2 M x 4 N), each
// wave 128x64 of C as 8x4 fragments of v_mfma_f32_16x16x32_{bf16,f16} (fp8: v_mfma_scale_f32_16x16x128_f8f6f4).  Both operands
// are K-contiguous ([rows][K]); tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per
// wave-instruction) into a ring of five 32-KiB slots.  LDS image of a 1-KiB block: natural 128-B rows, chunk c of row r at slot
// c ^ ((r >> 1) & 7) (r = row inside its 16-row fragment group): eight consecutive DMA lanes read one whole 128-B line, and the
// 16 lanes of every ds_read_b128 hardware lane group touch 16 different 4-bank groups.  The permutation is applied on the per-lane
// global SOURCE address (the LDS-DMA destination is lane-linear).
#include "gemm.hpp"
#include "kernels.hpp"
#include <type_traits>

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

// exact-erf GELU (nn.GELU() default, mm_projector_builder.py:89 / vision_tower_builder.py:47) with erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, one v_rcp + one v_exp + five fmas; libm's erff is ~40 instructions with branches and made the bias + GELU
// epilogue of the K = 1024 vision GEMMs as long as their main loop).  The result is rounded to 16 bits (2^-11 relative) right after.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup + Newton
// steps: ~10 VALU instructions per element, which made the SwiGLU epilogue VALU-bound: 5.4 us of C -> LDS per tile)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Main loop ("ring" structure; measured ladder in DESIGN.md section 3):
//  * waves 0-3 / 4-7 (the SIMD partners) ping-pong: while one group issues its 64 MFMAs of K-step k back to back, the other
//    reads its 24 LDS fragments of the next step, so each SIMD's matrix pipe always has one wave computing;
//  * LDS is a RING of five 32-KB operand slots (A or W tile of one K-step).  One operand tile is staged per half K-step
//    and waited for with a counted vmcnt, so the L2->LDS pipe never drains (a drained 64-KB burst per step moves
//    44 GB/s per CU, the same bytes as alternating 32-KB tiles 62 GB/s: tools/dma_bench.hip);
//  * only the fragment-READING group issues LDS-DMA (group 1 every A tile, group 0 every W tile, after its LDS reads):
//    the computing group issues nothing but MFMAs.
template <int EPI, int DT, bool SPLIT = false, bool MXA = false>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[5 * TILE_BYTES];  // 160 KiB ring / 136 KiB (two stages, or the staged C tile)

    // Compiler trap (round 6): the two-pass kernels live at the 256-register limit, and the bf16 instantiations -- the SAME source but for the MFMA opcode and without the
    // fp16 kernels' two s_setreg (MODE.FP16_OVFL) per tile -- came out with 450 - 630 spilled VGPRs inside the K loops and three quarters of the 16-bit MFMAs writing their
    // accumulators to other registers than they read.  The s_setreg instructions bound the register allocator's regions; with the same (for bf16: value-preserving, the bit is 0
    // and stays 0) instructions at the same two places the bf16 kernels allocate like the fp16 ones: 0 - 4 spills (tests/test_isa_guards.py).
    constexpr bool LO6_BF16_FENCE = MXA && DT == DT_BF16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int stamp_vb = blockIdx.x;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (p.debug_stamps && tid == 0) p.debug_stamps[(size_t)stamp_vb * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    // ---- persistent workgroups: grid = one workgroup per CU (a multiple of 8), each walks virtual block ids
    // vb = blockIdx.x, + gridDim.x, ...  (vb % 8 == blockIdx.x % 8: a workgroup keeps its XCD's chunk of the tile order).
    // No relaunch gap between tiles, and the next tile's first LDS-DMA overlaps the previous tile's store drain.
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
#pragma unroll 1
    for (int vb = blockIdx.x; vb < (p.tile_map ? ((nwg + 255) & ~255) : nwg); vb += gridDim.x) {
    // ---- tile mapping (blocks b, b+8, ... share an XCD), then grouped M order
    int pid;
    if (p.tile_map) {
        // groups of 32 consecutive tiles (8 M x 4 N) dealt round-robin to the XCDs: at any moment the eight XCDs work on 32
        // neighbouring columns of ONE 8-tile row band, so its A panels are fetched from HBM once and served to the other
        // seven XCDs by the memory-side cache (each XCD still sees the same 8 x 4 sharing in its own L2)
        const int xcd = vb & 7, k = vb >> 3;
        pid = ((k >> 5) * 8 + xcd) * 32 + (k & 31);
        if (pid >= nwg) continue;
    } else {
        // XCD-contiguous chunks of the tile order
        const int bid = vb, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    stamp_vb = pid;
    stamp(0);
    if constexpr (out16<DT>::value == DT_F16 || (LO6_BF16_FENCE)) { if (p.f16_saturate) f16_saturate_off(); }     // the MFMAs must see NaN / inf operands as such (common.hpp)
    const int width = p.group_m * ntn;
    const int first_m = (pid / width) * p.group_m;
    const int gsz = min(ntm - first_m, p.group_m);
    const int tm = first_m + (pid % width) % gsz;
    const int tn = (pid % width) / gsz;
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- per-lane LDS-DMA sources (see the main loop: each wave stages 8 blocks of ONE operand)
    // 32-bit byte offsets from the (wave-uniform, scalar) operand bases: the K-step advance is a scalar add on the
    // base and the LDS-DMA uses the saddr + voffset form (no per-step vector address arithmetic).
    // LDS image of a 1-KiB block = 8 rows x 128 B in natural row order, the eight 16-B chunks of row r stored at slot
    // c ^ g(r), g(r) = (r >> 1) & 7 with r the row's index inside its 16-row fragment group: (a) the eight consecutive lanes that
    // fill a row read one whole 128-B line (in permuted order) -- with lane PAIRS reading 32 B of a row each, as in the first
    // layout, the same bytes moved 25-30 % slower (tools/dma_bench.hip: 68 -> 90 GB/s per CU); (b) the 16 lanes of every
    // ds_read_b128 hardware lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) hit 16 different 4-bank groups.
    const int sr = lane >> 3;                                    // DMA lane -> row of the block; its chunk slot is lane & 7
    const char* baseA = (const char*)p.A;
    const char* baseW = (const char*)p.W;

    // ---- fragment read offsets (bytes) inside an operand tile
    // lane (row fr, fc): its two 16-B reads per K-step are the row's chunks fc and fc + 4 -- the two 32-deep MFMA steps of the
    // 16-bit forms (k = 8 fc .. and 32 + 8 fc ..), or the 32 e4m3 of one 128-deep fp8 MFMA (any k assignment works as long as the
    // A and W lanes use the same one).  Chunk c of row r sits at slot c ^ g(r): the second read is the first XOR 64 bytes.
    constexpr int ES = (DT == DT_F8) ? 1 : 2;            // operand element size
    constexpr bool MX8 = MXA && DT == DT_F8;             // fp8 GEMM whose A operand carries E8M0 block scales (a_mx)
    constexpr bool LO6 = MXA && DT != DT_F8;             // 16-bit GEMM followed, in the same accumulators, by an e2m3 pass over the A operand's LO part (see "phase 2")
    constexpr int ODT = out16<DT>::value;                // dtype of 16-bit outputs
    const int fr = lane & 15, fc = lane >> 4;
    const int fg = (fr >> 1) & 7;
    const int frag_off = (fr >> 3) * 1024 + (fr & 7) * 128 + 16 * ((fc ^ (fg & 3)) + 4 * (fg >> 2));
    const int a_off = (16 * wm) * 1024 + frag_off;   // + mi*2048; second read: ^ 64
    const int b_off = (8 * wn) * 1024 + frag_off;    // + ni*2048

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K * ES / (BK * 2);   // K-steps of 128 bytes per row
    const int nk6 = LO6 ? p.K6 / 128 : 0; // ... of the second pass (phase 2)
    // fp8: this thread's dequantisation scale (threads 0-255: the tile's rows, 256-511: its columns), requested before the K loop
    // so that its latency is not exposed in front of the epilogue
    float f8_scale = 0.f;
    if constexpr (DT == DT_F8) f8_scale = tid < 256 ? (p.row_scale ? p.row_scale[min(row0 + tid, p.M - 1)] : 1.0f) : p.col_scale[min(col0 + tid - 256, p.N - 1)];
    stamp(1);
    {
        const int grp = wave >> 2;  // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
        typedef int i32x8 __attribute__((ext_vector_type(8)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        bf16x8 fa[2][8], fb[2][4];        // 16-bit fragments: [32-deep step][16-row block]
        i32x8 fa8[8], fb8[4];             // fp8 fragments: 32 e4m3 per lane (the two 16-B reads land in the halves of one tuple)
        auto load_frags = [&](int offA_tile, int offB_tile) __attribute__((always_inline)) {
            const char* ba[2] = {smem + (offA_tile + a_off), smem + ((offA_tile + a_off) ^ 64)};
            const char* bb[2] = {smem + (offB_tile + b_off), smem + ((offB_tile + b_off) ^ 64)};
            if constexpr (DT == DT_F8) {
                auto rd = [](const char* q0, const char* q1) __attribute__((always_inline)) {
                    const i32x4 l = *(const i32x4*)q0, h = *(const i32x4*)q1;
                    return (i32x8){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
                };
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) fb8[ni] = rd(bb[0] + ni * 2048, bb[1] + ni * 2048);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) fa8[mi] = rd(ba[0] + mi * 2048, ba[1] + mi * 2048);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) fb[ks][ni] = *(const bf16x8*)(bb[ks] + ni * 2048);
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) fa[ks][mi] = *(const bf16x8*)(ba[ks] + mi * 2048);
                }
            }
        };
        // The W fragment is passed as the MFMA's first operand, so a 16x16 accumulator fragment holds C TRANSPOSED in the hardware
        // layout: lane l owns C row (l & 15) and the four consecutive C columns 4 (l >> 4) + {0..3} of fragment (mi, ni).  The
        // epilogues can then pack and stage 8 / 16 contiguous bytes per lane without any cross-lane transpose.
        // MXA: this lane's eight A-side E8M0 bytes of the current K-step (one per 16-row fragment) and of the next one
        uint2 mx_cur = make_uint2(0x7f7f7f7fu, 0x7f7f7f7fu), mx_nxt = mx_cur;
        const uint8_t* mx_base = nullptr;
        if constexpr (MX8) {
            int frm = fr;
            asm volatile("" : "+v"(frm));          // (per tile: the hoisted 64-bit lane address was spilled)
            mx_base = p.a_mx + (int64_t)tm * 256 + (wm * 16 + frm) * 8;
        }
        auto mx_request = [&](int kt) __attribute__((always_inline)) {
            if constexpr (MX8) mx_nxt = *(const uint2*)(mx_base + (int64_t)min(kt, nk - 1) * p.mx_stride);
        };
        auto compute = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_s_setprio(1);
            if constexpr (DT == DT_F8) {
                // e4m3 x e4m3 on the block-scaled MFMA: 32 MFMAs of 128-deep K per step.  W side: unit MX scales (0x7f; per-row f32 scales
                // in the epilogue).  A side: unit, or (MXA) byte mi & 3 of the lane's scale dword mi >> 2 -- opsel must be an immediate
#define F8_ROW(MI)                                                                                                                                  \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                                    \
                    acc[MI][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[ni], fa8[MI], acc[MI][ni], 0, 0, 0, 0x7f7f7f7f, (MX8 ? (MI & 3) : 0), \
                                                                                  (int)(MX8 ? ((MI) < 4 ? mx_cur.x : mx_cur.y) : 0x7f7f7f7fu));
                F8_ROW(0) F8_ROW(1) F8_ROW(2) F8_ROW(3) F8_ROW(4) F8_ROW(5) F8_ROW(6) F8_ROW(7)
#undef F8_ROW
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc[mi][ni] = mfma16<DT>(fb[ks][ni], fa[ks][mi], acc[mi][ni]);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        // (Measured dead end: issuing the last 4/8/16 MFMAs of a K-step behind the phase barrier, so that the rendezvous
        // latency is covered by queued MFMAs: 0 / -8 / -9 % -- two waves feeding one SIMD's matrix pipe at once costs more
        // than the bubble.)
        // a phase boundary: own LDS reads complete (WAR on the tile about to be refilled), then rendezvous
#define PHASE_BARRIER()                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory")

#ifdef GEMM_WAIT_PROF   // instrumented build only (tools/gemm_waits.py): where a wave's time goes inside one K-step
        unsigned long long wp_t[6], wp_acc[5] = {0, 0, 0, 0, 0};
#define WP_T(i) wp_t[i] = __builtin_readcyclecounter()
#define WP_ACC() do { for (int q_ = 0; q_ < 5; ++q_) wp_acc[q_] += wp_t[q_ + 1] - wp_t[q_]; } while (0)
#else
#define WP_T(i)
#define WP_ACC()
#endif
        // Two straight-line loops (one per wave group) that execute the SAME barrier sequence:
        //   phase A(kt): group 0 computes step kt              | group 1 reads its fragments of step kt, stages A(kt+2)
        //   phase B(kt): group 0 reads its fragments of kt+1, stages W(kt+2) | group 1 computes step kt
        // Operand n = 2*step + isW lives in ring slot n % 5: A(kt) at `sa`, W(kt) = adv(sa,1), A(kt+1) = adv(sa,2),
        // W(kt+1) = adv(sa,3), A(kt+2) = adv(sa,4) (the slot W(kt-1) left), W(kt+2) = sa (the slot A(kt) left).
        constexpr int RING = 5 * TILE_BYTES;
        auto adv = [](int s, int j) { const int x = s + j * TILE_BYTES; return x >= RING ? x - RING : x; };
        const int wg = wave & 3;
        const int sc = (lane & 7) ^ (4 * (wg & 1) + (sr >> 1));     // source chunk of this lane's slot: blocks wg + 4 i share (block & 1) = wg & 1
        uint32_t off8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = wg + 4 * i;   // 1-KiB block (8 rows) of the operand tile
            if (grp == 1) off8[i] = (uint32_t)((int64_t)min(row0 + 8 * b + sr, p.M - 1) * p.lda * ES + 16 * sc);
            else off8[i] = (uint32_t)((int64_t)min(col0 + 8 * b + sr, p.N - 1) * (p.w_wrap_k > 0 ? p.w_wrap_k : p.K) * ES + 16 * sc);   // W row stride
        }
        const char* gbase = grp == 1 ? baseA : baseW;
        // compensated mode: A is [hi | lo] along K and W is used twice -- the W group's K-step index wraps (scalar select)
        const int wrap = (grp == 0 && p.w_wrap_k > 0) ? p.w_wrap_k * ES / (BK * 2) : 0x7fffffff;
        auto stage8 = [&](int dst, int kt) __attribute__((always_inline)) {   // one operand tile share: 8 LDS-DMA per wave
#ifdef GEMM_ABLATE_DMA   // ablation builds only: 1 = stage the first two K-steps only, then compute on stale
                         // LDS; 2 = keep every LDS-DMA but re-read the first two K-steps (L2-resident, no fabric / HBM traffic)
            if (GEMM_ABLATE_DMA == 1 && kt >= 2) return;
            if (GEMM_ABLATE_DMA == 2) kt &= 1;
#endif
            const char* g = gbase + (int64_t)(kt >= wrap ? kt - wrap : kt) * (BK * 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // keep the lane offset a 32-bit VGPR at the point of use: the zero-extension then folds into the LDS-DMA's
                // saddr + voffset form (hoisted out of the K loop it becomes 8 VGPR pairs and the loop spills to scratch)
                uint32_t o = off8[i];
                asm volatile("" : "+v"(o));
                glds16(g + o, smem + dst + (wg + 4 * i) * 1024);
            }
        };
        // 16-bit loader: one LDS-DMA after every three fragment reads (instead of 24 reads, then 8 LDS-DMA): the DMA issue, the long
        // pole of the loading phase, overlaps the LDS reads (+0.4 - 1.3 % per GEMM, +0.9 % on the bench).
        auto frags_and_dma = [&](int offA_tile, int offB_tile, int dst, int kt, bool on) __attribute__((always_inline)) {
            const char* ba[2] = {smem + (offA_tile + a_off), smem + ((offA_tile + a_off) ^ 64)};
            const char* bb[2] = {smem + (offB_tile + b_off), smem + ((offB_tile + b_off) ^ 64)};
#ifdef GEMM_ABLATE_DMA
            if (GEMM_ABLATE_DMA == 1 && kt >= 2) on = false;
            if (GEMM_ABLATE_DMA == 2) kt &= 1;
#endif
#if defined(GEMM_ABLATE_WREAD) && GEMM_ABLATE_WREAD == 2   // ... and no LDS-DMA of W either after the prologue (group 0 stages W)
            if (grp == 0 && kt >= 4) on = false;
#endif
            const char* g = gbase + (int64_t)(kt >= wrap ? kt - wrap : kt) * (BK * 2);
            auto dma1 = [&](int q) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                if (on) { uint32_t o = off8[q]; asm volatile("" : "+v"(o)); glds16(g + o, smem + dst + (wg + 4 * q) * 1024); }
                __builtin_amdgcn_sched_barrier(0);
            };
            {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int j = 3 * q; j < 3 * q + 3; ++j) {
                        const int ks = j / 12, r = j % 12;
#ifdef GEMM_ABLATE_WREAD   // ablation builds only: the W fragments are NOT read from LDS -- an upper bound on what
                           // fetching them global -> VGPR instead of through LDS could buy (the loads themselves not even charged)
                        if (r < 4) { }   // (no read; the MFMAs below take copies of A fragments: keeping stale W fragments live across the back edge spills)
#else
                        if (r < 4) fb[ks][r] = *(const bf16x8*)(bb[ks] + r * 2048);
#endif
                        else fa[ks][r - 4] = *(const bf16x8*)(ba[ks] + (r - 4) * 2048);
                    }
                    dma1(q);
                }
#ifdef GEMM_ABLATE_WREAD
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fb[ks][r] = fa[ks][r + 4];
#endif
            }
        };
        // ---- phase 2 (LO6): acc += (lo part of the A operand) . W^T on the block-scaled MFMA with e2m3 operands, at FOUR times the 16-bit rate (gfx950 issues fp6
        // at the fp4 rate: tools/mfma_f6_probe.hip, 7.5 against 4.1 PFLOP/s for e4m3 on random operands).  The compensated modes' second walk over K carries
        // x_lo = x - f32(x_hi), 2^-11 of x in fp16: the product W . x_lo only needs a few percent of relative accuracy to remove > 95 % of the rounding noise of x_hi,
        // and e2m3 with one E8M0 scale per 32 values delivers what round 4's e4m3 did (profiles/r05_lo_format_emulation.txt: 16,000-entry populations on the
        // trained-like weight sets).  With the MFMAs at a quarter of the 16-bit time the pass is bound by what it moves, so it moves as little as the format allows:
        // the operands come as ready-made LDS images of dense tiles (gemm.hpp: 25 KiB per 256 rows x 128 values instead of the 32 KiB of a byte per value), staged by
        // lane-linear LDS-DMA copies into a ring of SIX slots (three per operand: every tile is requested two barriers before its first read), read as one
        // ds_read_b128 + one ds_read_b64 per fragment plus one scale read per operand and step.  The schedule is behind the 16-bit loops ("phase 2, all eight waves
        // alike"); the forms it replaced -- both wave groups reading, then both computing (1.5 us per step); a half-step-shifted ping-pong of the two groups (1.1 us) --
        // and what bounds it are in profiles/r05_p2_schedule_study.md.
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        const char* g6A = LO6 ? (const char*)p.A6 + (int64_t)tm * nk6 * F6_TILE_BYTES : nullptr;
        const char* g6W = LO6 ? (const char*)p.W6 + (int64_t)tn * nk6 * F6_TILE_BYTES : nullptr;
        uint2 sc6a = make_uint2(0, 0);                                // this lane's eight A-side scale bytes of the step (one per 16-row fragment)
        uint32_t sc6w = 0;                                            // ... and its four W-side ones
        // piece q (0-6) of this wave's share of one operand tile (K-step kd into ring slot `dslot`): six KiB-blocks of e2m3 and a quarter of the scale KiB
        // (the scale KiB as 16-byte pieces of the first 16 lanes, 256 B per wave: the 4-byte form of the LDS-DMA is tracked by the compiler's wait-count pass as an
        // LDS store any LDS read may alias, and it put an s_waitcnt vmcnt(0) in front of the next step's fragment reads)
        auto dma6 = [&](int q, int dslot, int kd, bool isW, bool on) __attribute__((always_inline)) {
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 == 3   // ablation build: only the tiles requested in front of the pass (K-steps 0 - 2) are staged, then it runs on stale tiles
            on = on && kd < 3;
#endif
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 >= 4   // ablation builds (round 6, profiles/r06_lo4_bound.md): the bytes of an e2m1 image instead of an e2m3 one (17 of 25 KiB: pieces 4, 5 not
            on = on && !((q == 4 || q == 5) && (isW || GEMM_ABLATE_P2 == 5));   // requested, fragments read as ONE ds_read_b128) -- 4 = the W operand, 5 = both.  Timing only, wrong results
#endif
            if (on) {
                const char* g = (isW ? g6W : g6A) + (int64_t)kd * F6_TILE_BYTES;
                char* d = smem + dslot * F6_TILE_BYTES;
                uint32_t o16 = (uint32_t)lane * 16u;                 // (unsigned 32-bit lane offset + scalar base: the saddr + voffset form of the LDS-DMA, as in the 16-bit loop)
                asm volatile("" : "+v"(o16));
                if (q < 6) glds16(g + (wg + 4 * q) * 1024 + o16, d + (wg + 4 * q) * 1024);
                else if (lane < 16) glds16(g + 24576 + wg * 256 + o16, d + 24576 + wg * 256);
            }
        };
        // one whole operand tile of K-step k into ring slot `slot` (0-5) by the four waves of ONE group: 7 pieces per wave
        auto stage6 = [&](int slot, int k, bool isW) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 7; ++q) dma6(q, slot, k, isW, true);
        };
        // one 16-row fragment: ds_read_b128 + ds_read_b64 (qa = tile + lane * 16, qb = tile + 1024 + lane * 8, formed once per step; off = 1536 x fragment index)
        auto rd6 = [&](const char* qa, const char* qb, int off) __attribute__((always_inline)) {
            const i32x4 l = *(const i32x4*)(qa + off);
            const i32x2 h = *(const i32x2*)(qb + off);
            return (i32x8){l[0], l[1], l[2], l[3], h[0], h[1], 0, 0};
        };
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 >= 4
        auto rd6_4 = [&](const char* qa, const char* qb, int off) __attribute__((always_inline)) {
            const i32x4 l = *(const i32x4*)(qa + off);
            return (i32x8){l[0], l[1], l[2], l[3], l[0], l[1], 0, 0};
        };
#define RD6W rd6_4
#if GEMM_ABLATE_P2 == 5
#define RD6A rd6_4
#else
#define RD6A rd6
#endif
#else
#define RD6W rd6
#define RD6A rd6
#endif
        // (scales read through ext-vector types like the fragments: behind loads typed uint32_t / uint2 the compiler's wait-count pass put an s_waitcnt vmcnt(0) --
        // every outstanding LDS-DMA -- in front of each step's fragment reads)
        typedef int i32x1 __attribute__((ext_vector_type(1)));
        auto rd6_sw = [&](const char* tb) __attribute__((always_inline)) {
            return (uint32_t)(*(const i32x1*)(tb + 24576 + wn * 256 + lane * 4))[0];           // ((wn * 4 + g) * 16 + r) * 4 = wn * 256 + lane * 4
        };
        auto rd6_sa = [&](const char* ta) __attribute__((always_inline)) {
            const i32x2 v = *(const i32x2*)(ta + 24576 + wm * 512 + lane * 8);                 // ((wm * 4 + g) * 16 + r) * 8 = wm * 512 + lane * 8
            return make_uint2((uint32_t)v[0], (uint32_t)v[1]);
        };
        int sa = 0;
        mx_request(0);
        if (grp == 0) {
            stage8(TILE_BYTES, 0);                                   // W0
            if (nk > 1) stage8(3 * TILE_BYTES, 1);                  // W1
            if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PHASE_BARRIER();                                         // A0 W0 landed (every wave waited for its own DMA)
            if constexpr (MX8) mx_cur = mx_nxt;                      // requested at tile entry, in front of every LDS-DMA
            if constexpr (DT == DT_F8) {
                // same barrier sequence, loop rotated so that a step's fragments are read and consumed inside one iteration:
                // the fp8 fragments are 8-register tuples assembled from two 16-B reads, and carried across the back edge the
                // register allocator keeps a second copy of all twelve (spilling ~200 VGPRs)
                int sp = 0;                                          // slot of A(k-1)
                for (int k = 0; k < nk; ++k) {
                    mx_request(k + 1);                               // (MXA) next step's A scales: in front of the LDS-DMA in the vmcnt order
                    load_frags(sa, adv(sa, 1));
                    if (k >= 1 && k + 1 < nk) stage8(sp, k + 1);     // W(k+1) into the slot A(k-1) left
                    PHASE_BARRIER();
                    compute();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of W(k+1) landed
                    if constexpr (MX8) mx_cur = mx_nxt;
                    PHASE_BARRIER();
                    sp = sa; sa = adv(sa, 2);
                }
                PHASE_BARRIER();
            } else {
            load_frags(0, TILE_BYTES);
            PHASE_BARRIER();
            for (int kt = 0; kt < nk; ++kt) {
                WP_T(0);
                compute();
                WP_T(1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my share of W(kt+1) landed
                WP_T(2);
                PHASE_BARRIER();
                WP_T(3);
                if constexpr (LO6) { if (kt + 1 == nk) break; }         // (the last interval of the two-pass kernels: below)
                if (kt + 1 < nk) frags_and_dma(adv(sa, 2), adv(sa, 3), sa, kt + 2, kt + 2 < nk);   // fragments of kt+1, W(kt+2)
#ifdef GEMM_WAIT_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                WP_T(4);
                PHASE_BARRIER();
                WP_T(5);
                WP_ACC();
                sa = adv(sa, 2);
            }
            // last 16-bit interval: this group has nothing left to read while the A group computes from registers, and no wave reads LDS any more
            if constexpr (LO6) {
                // the second pass's first three K-steps (both operands) are requested HERE, under the A group's last 64 MFMAs
                stage6(0, 0, false); stage6(1, 0, true);
                if (nk6 > 1) { stage6(2, 1, false); stage6(3, 1, true); }
                if (nk6 > 2) { stage6(4, 2, false); stage6(5, 2, true); }
                PHASE_BARRIER();
            }
            }
        } else {
            stage8(0, 0);                                            // A0
            if (nk > 1) stage8(2 * TILE_BYTES, 1);                  // A1
            if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PHASE_BARRIER();
            if constexpr (MX8) mx_cur = mx_nxt;
            PHASE_BARRIER();
            for (int kt = 0; kt < nk; ++kt) {
                WP_T(0);
                if constexpr (DT == DT_F8) {
                    mx_request(kt + 1);
                    load_frags(sa, adv(sa, 1));
                    if (kt + 2 < nk) stage8(adv(sa, 4), kt + 2);
                } else frags_and_dma(sa, adv(sa, 1), adv(sa, 4), kt + 2, kt + 2 < nk);          // fragments of kt, A(kt+2)
#ifdef GEMM_WAIT_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                WP_T(1);
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // my share of A(kt+1) landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WP_T(2);
                PHASE_BARRIER();
                WP_T(3);
                compute();
                if constexpr (MX8) mx_cur = mx_nxt;                  // older than this iteration's LDS-DMA: landed by the vmcnt(8) above
                WP_T(4);
                PHASE_BARRIER();
                WP_T(5);
                WP_ACC();
                sa = adv(sa, 2);
            }
        }
        if constexpr (LO6) {
            // ---- phase 2, all eight waves alike.  Every fragment register is refilled IN PLACE between its last MFMA and its next one, and the 32 MFMAs of a step
            // are walked in four quadrants (4 A fragments x 2 W fragments) whose order alternates from step to step, so that the refills spread over the whole step
            // and each has at least eight MFMAs (128 cycles) to arrive:
            //   step (CA, CB):  QA rows 0-3 x cols CA   | reads fw[CB] fw[CB+1] fa[4] fa[5] of the step's own tile
            //                   QB rows 0-3 x cols CB   | reads fa[6] fa[7]                 (the last reads of this tile)
            //                   -- barrier: every wave has left this tile; the next tile has landed everywhere
            //                   QC rows 4-7 x cols CB   | reads fa[0] fa[1] of the NEXT tile
            //                   QD rows 4-7 x cols CA   | reads fa[2] fa[3], the scales, fw[CB] fw[CB+1] of the next tile      -> next step: (CB, CA)
            // LDS reads and LDS-DMA requests issue in the shadow of the wave's own MFMAs and the SIMD's other wave fills the matrix pipe whenever this one waits.
            // (All of a step's refills behind its last column and its LDS-DMA behind the first -- 18 reads in one burst of all eight waves, seven requests in
            // another -- took the same 1.1 us per step as the two-group ping-pong before it: tools/gemm_waits_lo6.py.)
            // Ring: A(k) in slot 2 (k % 3), W(k) in 2 (k % 3) + 1.  Between the barriers of steps k and k+1 the waves request tile k+3 into the slots tile k left
            // (W group: W tiles, A group: A tiles; 7 LDS-DMA per wave).  Tiles 0 - 2 were requested by the W group under the last 16-bit MFMAs.
            // (The compiler's wait-count pass answers every LDS-read dependency of this loop with s_waitcnt lgkmcnt(0): it counts the LDS-DMA builtins as accesses to
            // both memories and, behind asm waits, as pending for ever.  With the pass's LDS-DMA and vmcnt waits as asm statements and one modelled vmcnt(0) between
            // the passes it emits counted lgkmcnt(N) waits -- measured +0.7 %, and gone again once every asm request saved and restored M0, which the compiler
            // reserves for itself: profiles/r05_p2_schedule_study.md.  Kept: the builtins.)
#define P2_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P2_BARRIER() PHASE_BARRIER()      // = s_waitcnt lgkmcnt(0) + s_barrier: a wave's LDS reads of the tile it leaves have RETURNED before any wave may overwrite its slots
            if (grp == 0) {
                if (nk6 > 2) P2_VMCNT(28);                               // my shares of A(0) W(0) landed
                else if (nk6 > 1) P2_VMCNT(14);
                else P2_VMCNT(0);
            }
            P2_BARRIER();
            const int qoa = 8 * wm * 1536 + lane * 16, qoa8 = 8 * wm * 1536 + 1024 + lane * 8;       // this lane's byte offsets inside an A / a W tile image
            const int qob = F6_TILE_BYTES + 4 * wn * 1536 + lane * 16, qob8 = F6_TILE_BYTES + 4 * wn * 1536 + 1024 + lane * 8;
            sc6w = rd6_sw(smem + F6_TILE_BYTES); sc6a = rd6_sa(smem);
            fb8[0] = RD6W(smem + qob, smem + qob8, 0); fb8[1] = RD6W(smem + qob, smem + qob8, 1536);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa8[i] = RD6A(smem + qoa, smem + qoa8, i * 1536);
            int s3 = 0;                                                  // k % 3
#ifdef GEMM_WAIT_PROF   // (instrumented build: the sums below are phase 2's alone -- QA + QB | vmcnt | barrier | QC | QD)
            for (int q_ = 0; q_ < 5; ++q_) wp_acc[q_] = 0;
#endif
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 == 1   // ablation builds (make ablate_p2; timing only, wrong results): 1 = no MFMAs in phase 2, 2 = no fragment refills, 3 = no LDS-DMA
#define L6_MMA(MI, NI)
#else
#define L6_MMA(MI, NI) acc[MI][NI] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[NI], fa8[MI], acc[MI][NI], 2, 2, NI, (int)sc6w, (MI & 3), (int)((MI) < 4 ? sc6a.x : sc6a.y));   // cbsz = blgp = 2: e2m3
#endif
#define L6_SB() __builtin_amdgcn_sched_barrier(0)
            auto step6 = [&](int k, auto ca_, auto cb_) __attribute__((always_inline)) {
                constexpr int CA = decltype(ca_)::value, CB = decltype(cb_)::value;
                const int s3n = s3 == 2 ? 0 : s3 + 1, s3p = s3 == 0 ? 2 : s3 - 1;      // slots of tile k+1 / of tiles k-1 and k+2
                const char* t0 = smem + (2 * s3) * F6_TILE_BYTES;                        // this step's tile (A image; the W image follows it)
                const char* t1 = smem + (2 * s3n) * F6_TILE_BYTES;                       // the next step's
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 == 2
                const bool rl = false, own = false;
#else
                const bool rl = k + 1 < nk6, own = true;                 // there is a next step: refill from its tile
#endif
                const bool on2 = k >= 1 && k + 2 < nk6;                  // pieces 4-6 of tile k+2 (0-3: requested in the previous step) into the slots of tile k-1
                const bool on3 = k + 3 < nk6;                            // pieces 0-3 of tile k+3 into the slots of tile k, behind the barrier
                const int d2 = 2 * s3p + (grp == 0 ? 1 : 0), d3 = 2 * s3 + (grp == 0 ? 1 : 0);
                const bool isW = grp == 0;
                uint2 sc6a_n = sc6a; uint32_t sc6w_n = sc6w;
                WP_T(0);
                __builtin_amdgcn_s_setprio(1);
                // QA
                L6_MMA(0, CA) L6_SB(); if (own) fb8[CB] = RD6W(t0 + qob, t0 + qob8, CB * 1536); L6_SB();
                L6_MMA(1, CA) L6_SB(); dma6(4, d2, k + 2, isW, on2); L6_SB();
                L6_MMA(2, CA) L6_SB(); if (own) fb8[CB + 1] = RD6W(t0 + qob, t0 + qob8, (CB + 1) * 1536); L6_SB();
                L6_MMA(3, CA) L6_SB();
                L6_MMA(0, CA + 1) L6_SB(); if (own) fa8[4] = RD6A(t0 + qoa, t0 + qoa8, 4 * 1536); L6_SB();
                L6_MMA(1, CA + 1) L6_SB(); dma6(5, d2, k + 2, isW, on2); L6_SB();
                L6_MMA(2, CA + 1) L6_SB(); if (own) fa8[5] = RD6A(t0 + qoa, t0 + qoa8, 5 * 1536); L6_SB();
                L6_MMA(3, CA + 1) L6_SB();
                // QB
                L6_MMA(0, CB) L6_SB(); if (own) fa8[6] = RD6A(t0 + qoa, t0 + qoa8, 6 * 1536); L6_SB();
                L6_MMA(1, CB) L6_SB(); if (own) fa8[7] = RD6A(t0 + qoa, t0 + qoa8, 7 * 1536); L6_SB();
                L6_MMA(2, CB) L6_SB(); dma6(6, d2, k + 2, isW, on2); L6_SB();
                L6_MMA(3, CB)
                L6_MMA(0, CB + 1) L6_MMA(1, CB + 1) L6_MMA(2, CB + 1) L6_MMA(3, CB + 1)
                L6_SB();
                WP_T(1);
                // every read of tile k has been requested (and is awaited in front of the barrier: P2_BARRIER's own lgkmcnt(0)).  Tile k+1: my pieces of it were requested two barriers ago --
                // only the seven of tile k+2 may stay in flight (k = 0: the W group's fourteen of A(2) W(2))
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 == 6   // ablation build (round 6: is the pass bound by LDS-DMA LATENCY?): the waits let one more tile stay in flight -- tile k+1 is read without
                                                     // having been awaited (timing only, wrong results).  If the step got much shorter, a deeper ring would be the lever
                if (k == 0) { if (nk6 > 2) P2_VMCNT(21); else P2_VMCNT(0); }
                else if (k + 2 < nk6) P2_VMCNT(14);
                else P2_VMCNT(0);
#else
                if (k == 0) { if (nk6 > 2) P2_VMCNT(14); else P2_VMCNT(0); }
                else if (k + 2 < nk6) P2_VMCNT(7);
                else P2_VMCNT(0);
#endif
                WP_T(2);
#if defined(GEMM_ABLATE_P2) && GEMM_ABLATE_P2 == 7   // ablation build (round 6: is it the per-step rendezvous of the eight waves?): a barrier every OTHER step only (timing only, wrong results)
                if (k & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } else { P2_BARRIER(); }
#else
                P2_BARRIER();
#endif
                WP_T(3);
                // QC
                L6_MMA(4, CB) L6_SB(); if (rl) fa8[0] = RD6A(t1 + qoa, t1 + qoa8, 0); L6_SB();
                L6_MMA(5, CB) L6_SB(); dma6(0, d3, k + 3, isW, on3); L6_SB();
                L6_MMA(6, CB) L6_SB(); if (rl) fa8[1] = RD6A(t1 + qoa, t1 + qoa8, 1536); L6_SB();
                L6_MMA(7, CB) L6_SB(); dma6(1, d3, k + 3, isW, on3); L6_SB();
                L6_MMA(4, CB + 1) L6_SB(); if (rl) { sc6w_n = rd6_sw(t1 + F6_TILE_BYTES); sc6a_n = rd6_sa(t1); } L6_SB();
                L6_MMA(5, CB + 1) L6_MMA(6, CB + 1) L6_MMA(7, CB + 1)
                L6_SB();
                WP_T(4);
                // QD (fw[CB], fw[CB + 1] are free: their refills for the next step go first)
                if (rl) fb8[CB] = RD6W(t1 + qob, t1 + qob8, CB * 1536);
                L6_SB();
                L6_MMA(4, CA) L6_SB(); if (rl) fb8[CB + 1] = RD6W(t1 + qob, t1 + qob8, (CB + 1) * 1536); L6_SB();
                L6_MMA(5, CA) L6_SB(); dma6(2, d3, k + 3, isW, on3); L6_SB();
                L6_MMA(6, CA) L6_SB(); if (rl) fa8[2] = RD6A(t1 + qoa, t1 + qoa8, 2 * 1536); L6_SB();
                L6_MMA(7, CA) L6_SB(); dma6(3, d3, k + 3, isW, on3); L6_SB();
                L6_MMA(4, CA + 1) L6_SB(); if (rl) fa8[3] = RD6A(t1 + qoa, t1 + qoa8, 3 * 1536); L6_SB();
                L6_MMA(5, CA + 1) L6_MMA(6, CA + 1) L6_MMA(7, CA + 1)
                L6_SB();
                __builtin_amdgcn_s_setprio(0);
                sc6a = sc6a_n; sc6w = sc6w_n;
                WP_T(5);
                WP_ACC();
                s3 = s3n;
            };
            {
                int k = 0;
                for (; k + 1 < nk6; k += 2) {
                    step6(k, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                    step6(k + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
                }
                if (k < nk6) step6(k, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            }
#undef L6_SB
#undef RD6W
#undef RD6A
#undef P2_BARRIER
#undef P2_VMCNT
#undef L6_MMA
        }
#undef PHASE_BARRIER
#ifdef GEMM_WAIT_PROF
        if (p.debug_stamps && lane == 0 && (wave & 3) == 0)
            for (int q_ = 0; q_ < 5; ++q_) p.debug_stamps[((size_t)nwg + (size_t)stamp_vb * 2 + grp) * 8 + q_] = wp_acc[q_];
#endif
    }

    stamp(2);
    // =========================================================================== epilogues
    int tid_e = tid;                        // the epilogue's copy of the thread index, formed per tile: its derived indices (row / segment / scale slots) are kernel
    asm volatile("" : "+v"(tid_e));         // invariants that LICM hoists out of the persistent loop and, in the fp8 kernels, spills across the K loop
    if constexpr (out16<DT>::value == DT_F16) { if (p.f16_saturate) f16_saturate_on(); }      // fp16 stores of this tile saturate instead of overflowing to inf
    else if constexpr (LO6_BF16_FENCE) { if (p.f16_saturate) f16_saturate_off(); }
    if (p.debug_skip_epilogue) {   // timing aid (tools/gemm_k_sweep.py): keep the accumulators live, store nothing
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        __syncthreads();
        continue;
    }
    const int wrow0 = row0 + 128 * wm;  // wave's first row
    const int wcol0 = col0 + 64 * wn;   // wave's first column (in W's row order)
    if constexpr (DT == DT_F8) {
        // dequantise: acc[row][col] *= row_scale[row] * col_scale[col].  The tile's 256 + 256 scales go through LDS (free now).
        float* sc = (float*)smem;
        sc[tid_e] = f8_scale;
        __syncthreads();
        float4 cs[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) cs[ni] = *(const float4*)(sc + 256 + 64 * wn + 16 * ni + 4 * (lane >> 4));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const float rs = sc[128 * wm + 16 * mi + (lane & 15)];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[mi][ni][0] *= rs * cs[ni].x; acc[mi][ni][1] *= rs * cs[ni].y; acc[mi][ni][2] *= rs * cs[ni].z; acc[mi][ni][3] *= rs * cs[ni].w;
            }
        }
        __syncthreads();
    }

    if constexpr (EPI == EPI_LSE) {
        // lane (r = lane & 15, q = lane >> 4) holds row 16 mi + r of the wave's tile and its columns 16 ni + 4 q + {0..3}: sixteen
        // logits of one row per mi; the row's other 48 columns of this wave sit in lanes r + 16, r + 32, r + 48.
        float2* red = (float2*)smem;  // [4 wn][256 rows]
        const int q4 = lane >> 4;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int rl = 128 * wm + 16 * mi + fr;          // row inside the tile
            const int row = row0 + rl;
            const int lab = (row < p.M) ? p.labels[row] : -1;
            float v[4][4];
            float mx = -INFINITY;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wcol0 + 16 * ni + 4 * q4 + j;
                    v[ni][j] = (col < p.N) ? acc[mi][ni][j] : -INFINITY;
                    if (col == lab) p.label_logit[row] = v[ni][j];
                    mx = fmaxf(mx, v[ni][j]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sm = 0.f;
            if (mx > -INFINITY) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sm += __expf(v[ni][j] - mx);
            }
            sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
            if (q4 == 0) red[wn * 256 + rl] = make_float2(mx, sm);
        }
        __syncthreads();
        if (tid_e < 256) {
            const int row = row0 + tid_e;
            if (row < p.M) {
                float2 a0 = red[tid_e], a1 = red[256 + tid_e], a2 = red[512 + tid_e], a3 = red[768 + tid_e];
                const float mx = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
                float sm = 0.f;
                if (mx > -INFINITY) {
                    sm = a0.y * __expf(a0.x - mx) + a1.y * __expf(a1.x - mx) + a2.y * __expf(a2.x - mx) + a3.y * __expf(a3.x - mx);
                }
                p.lse_part[(int64_t)row * ntn + tn] = make_float2(mx, sm);
            }
        }
        stamp(3);
        __syncthreads();   // `red` (LDS) is reused by the next tile's ring
        continue;
    } else {
        // ---- C tile staged through LDS so that global stores are whole rows (512 B / 1 KiB per row), 16 B per lane.
        // (Direct stores from the MFMA layout touch 32-B row segments: measured 10-13 us per tile, 12 % of a K=3584 tile.)
        // (formed per tile: as kernel invariants they and everything derived from them -- bias / RoPE-table / LDS addresses -- were hoisted out of the
        // persistent loop and, in the fullest kernels, spilled across the K loop)
        int tq = lane >> 4;                     // which 4-col group of a fragment this lane owns
        int rsub = lane & 15;                   // which of its 16 rows
        asm volatile("" : "+v"(tq), "+v"(rsub));
        __syncthreads();                        // every wave is out of the main loop: LDS is reusable
        if constexpr (EPI == EPI_BF16 || EPI == EPI_QKV || EPI == EPI_SWIGLU) {
            constexpr int NC = (EPI == EPI_SWIGLU) ? 128 : 256;   // output columns of this tile
            constexpr int RS = NC * 2 + 16;                       // LDS row stride (bytes), = 16 mod 256: the 16 rows x 2 column groups of a
                                                                  // half-wave's ds_write_b64 cover all 64 banks once
            if constexpr (EPI == EPI_SWIGLU && DT == DT_F8) {
                if (p.out_mx != nullptr) {
                    // fp8 mode, fused quantisation of the SwiGLU output: the tile's 128 output columns are exactly one 128-deep K-step of
                    // the down GEMM, so each row gets ONE power-of-two (E8M0) scale per tile: 2^e with e minimal such that
                    // amax * 2^-e <= 448.  x = silu(g) * u replaces the gate accumulators in place; row maxima go lane -> wave (shuffles
                    // over the four column groups) -> tile (LDS, four waves per row); bytes are staged as rows of 128 B.
                    float* red = (float*)smem;                              // [4 wn][256 rows]
                    uint8_t* st8 = (uint8_t*)smem + 8192;                   // [256 rows][144 B]
                    constexpr int RS8 = 144;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        float m = 0.f;
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float x = silu_f(acc[mi][2 * pr][j]) * acc[mi][2 * pr + 1][j];
                                acc[mi][2 * pr][j] = x;
                                m = fmaxf(m, fabsf(x));
                            }
                        m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
                        if (tq == 0) red[wn * 256 + 128 * wm + 16 * mi + rsub] = m;
                    }
                    __syncthreads();
                    float* qinv = (float*)(smem + 4096);                    // [256 rows] 2^-e
                    if (tid_e < 256) {                                        // one thread per row: the scale, once
                        const float a = fmaxf(fmaxf(red[tid_e], red[256 + tid_e]), fmaxf(red[512 + tid_e], red[768 + tid_e]));
                        int e = 0;
                        if (a > 0.f) {
                            int ex; const float mant = frexpf(a * (1.0f / FP8_MAX), &ex);
                            e = (mant == 0.5f) ? ex - 1 : ex;
                            if (ldexpf(a, -e) > FP8_MAX) e += 1;
                            e = max(-127, min(127, e));
                        }
                        qinv[tid_e] = ldexpf(1.0f, -e);
                        p.out_mx[(int64_t)tn * p.mx_stride + (int64_t)tm * 256 + ((tid_e >> 7) * 16 + (tid_e & 15)) * 8 + ((tid_e >> 4) & 7)] = (uint8_t)(e + 127);
                    }
                    __syncthreads();
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const int rl = 128 * wm + 16 * mi + rsub;
                        const float inv = qinv[rl];
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
                            *(uint32_t*)(st8 + rl * RS8 + (2 * wn + pr) * 16 + 4 * tq) =
                                pack_fp8x4(acc[mi][2 * pr][0] * inv, acc[mi][2 * pr][1] * inv, acc[mi][2 * pr][2] * inv, acc[mi][2 * pr][3] * inv);
                    }
                    __syncthreads();
                    const int64_t ob0 = col0 / 2;                           // first output byte column of the tile
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int chunk = tid_e + NTHREADS * i, rl = chunk >> 3, seg = chunk & 7;
                        const int row = row0 + rl;
                        if (row < p.M && ob0 + seg * 16 + 15 < p.N / 2)
                            *(uint4*)((uint8_t*)p.C + (int64_t)row * p.ldc + ob0 + seg * 16) = *(const uint4*)(st8 + rl * RS8 + seg * 16);
                    }
                    stamp(3);
                    __syncthreads();
                    continue;
                }
            }
            // compensated mode (SPLIT): every output leaves as hi = f16(x) at C and lo = f16(x - f32(hi)) at C + lo_off.  Both halves are formed in ONE sweep
            // over a wave's accumulators, which die as they are consumed, exactly as in the plain kernels, and are staged side by side in LDS -- the hi tile
            // and, LO_OFF bytes further, the lo tile -- which fits for 128 rows at a time: two passes, pass hp covering every wave's row groups
            // 4 hp .. 4 hp + 3 (tile rows 128 wm + 64 hp + [0, 64)).  The pass loop is NOT unrolled (one copy of the sweep: the epilogue's code must not push
            // the K loop out of the instruction cache -- unrolled, the SwiGLU form cost 3 us per tile); pass 1 first moves accumulator groups 4 - 7 into
            // 0 - 3 (64 register moves), so that the sweep indexes its registers with compile-time constants.  (Round 3 swept all accumulators twice, once
            // per output half, keeping the 128 of them live across the first sweep and its stores: 12 - 60 spilled VGPRs in the QKV / plain-output kernels
            // that every TVG call and the bf16 parity mode run.)
            constexpr int PROWS = SPLIT ? 128 : 256;              // rows staged per pass
            constexpr int LO_OFF = PROWS * RS;                    // byte offset of the lo tile in LDS (SPLIT)
            static_assert(!SPLIT || 2 * LO_OFF <= 5 * TILE_BYTES, "hi + lo tiles of a half must fit the ring");
            auto ST = [&](char* o_, float a_, float b_, float c_, float d_) __attribute__((always_inline)) {
                const uint2 hi_ = make_uint2(pack2<ODT>(a_, b_), pack2<ODT>(c_, d_));
                *(uint2*)o_ = hi_;
                if constexpr (SPLIT) {
                    const uint16_t* h16_ = (const uint16_t*)&hi_;
                    *(uint2*)(o_ + LO_OFF) = make_uint2(pack2<ODT>(a_ - from16<ODT>(h16_[0]), b_ - from16<ODT>(h16_[1])), pack2<ODT>(c_ - from16<ODT>(h16_[2]), d_ - from16<ODT>(h16_[3])));
                }
            };
            constexpr int NMI = SPLIT ? 4 : 8;                    // row groups swept per pass
#pragma unroll 1
            for (int hp = 0; hp < (SPLIT ? 2 : 1); ++hp) {
            const int mg0 = SPLIT ? 4 * hp : 0;                   // tile row group of this pass's accumulator group 0
            const int lrow0 = SPLIT ? 64 * wm : 128 * wm;         // staged row of (mi, rsub) = lrow0 + 16 mi + rsub
            if (SPLIT && hp) {
                __syncthreads();                                  // pass 0's LDS reads are complete
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = acc[(mi + 4) & 7][ni];
            }
            {
            float4 qkv_bias[4];                               // EPI_QKV: this lane's 16 bias values (the same for all eight mi)
            if constexpr (EPI == EPI_QKV) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) qkv_bias[ni] = *(const float4*)(p.bias + wcol0 + 16 * ni + 4 * tq);
            }
            float4 bf16_bias[4];                              // EPI_BF16: likewise (zeros without a bias / beyond N)
            if constexpr (EPI == EPI_BF16) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int col = wcol0 + 16 * ni + 4 * tq;
                    bf16_bias[ni] = (p.bias && col + 3 < p.N) ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias && col < p.N && col + 3 >= p.N) {   // ragged last columns (static component indices: a dynamic one sends the vector to scratch)
                        bf16_bias[ni].x = p.bias[col];
                        if (col + 1 < p.N) bf16_bias[ni].y = p.bias[col + 1];
                        if (col + 2 < p.N) bf16_bias[ni].z = p.bias[col + 2];
                    }
                }
            }
            if constexpr (EPI == EPI_QKV) {
                // fragments (2p, 2p+1) hold RoPE partners d and d+64 for q/k heads; v heads are in natural order.  The wave's 64 columns
                // are either all rotated or all plain (rope_cols % 64 == 0): one wave-uniform branch OUTSIDE the row loop, and the
                // cos/sin rows (p.rope_rows: gathered per token once per batch) of row group mi+1 are requested before group mi is
                // rotated -- as one branchy loop this was a chain of 24 dependent global loads per tile (8.6 us of a 10 us epilogue).
                if (wcol0 < p.rope_cols) {
                    const int hl = wn >> 1;                    // head inside the tile
                    const int gbase = (wn & 1) * 2;            // first 32-group of this wave inside the head
                    const int d0 = 16 * gbase + 4 * tq;        // natural d of pair 0's lo element; pair 1: + 16
                    float4 cs[2][2], sn[2][2];                 // [buffer][pair]
                    // table layout (engine.hip rope_rows_kernel): [8 chunks = {cos, sin} x 4 groups of 16 dims][rope_stride rows][16 floats] -- the 16
                    // consecutive rows of a fragment are 1 KB contiguous per chunk, so a wave's load touches 8 full lines (row-major [row][128]
                    // it touched 16 half-used lines per instruction: 4,096 line requests per tile, the whole 8 us of this epilogue)
                    const int64_t cstride = p.rope_stride * 16;            // floats per chunk
                    auto request = [&](int mi, int buf) __attribute__((always_inline)) {
                        const int row = min(row0 + 128 * wm + 16 * (mg0 + mi) + rsub, p.M - 1);
                        const float* base = p.rope_rows + (int64_t)row * 16 + 4 * tq;
                        cs[buf][0] = *(const float4*)(base + (gbase + 0) * cstride); cs[buf][1] = *(const float4*)(base + (gbase + 1) * cstride);
                        sn[buf][0] = *(const float4*)(base + (4 + gbase + 0) * cstride); sn[buf][1] = *(const float4*)(base + (4 + gbase + 1) * cstride);
                    };
                    request(0, 0);
#pragma unroll
                    for (int mi = 0; mi < NMI; ++mi) {
                        if (mi + 1 < NMI) request(mi + 1, (mi + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);     // keep the next group's loads in front of this group's arithmetic
                        char* lrow = smem + (lrow0 + 16 * mi + rsub) * RS;
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const float4 c = cs[mi & 1][pr], sv = sn[mi & 1][pr];
                            const float c4[4] = {c.x, c.y, c.z, c.w}, s4[4] = {sv.x, sv.y, sv.z, sv.w};
                            const float4 b0 = qkv_bias[2 * pr], b1 = qkv_bias[2 * pr + 1];
                            const float bl[4] = {b0.x, b0.y, b0.z, b0.w}, bh[4] = {b1.x, b1.y, b1.z, b1.w};
                            float lo[4], hi[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float x1 = acc[mi][2 * pr][j] + bl[j];
                                const float x2 = acc[mi][2 * pr + 1][j] + bh[j];
                                lo[j] = x1 * c4[j] - x2 * s4[j];
                                hi[j] = x2 * c4[j] + x1 * s4[j];
                            }
                            char* o = lrow + (hl * 128 + d0 + 16 * pr) * 2;
                            ST(o, lo[0], lo[1], lo[2], lo[3]);
                            ST(o + 128, hi[0], hi[1], hi[2], hi[3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < NMI; ++mi) {
                        char* lrow = smem + (lrow0 + 16 * mi + rsub) * RS;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const float4 bv = qkv_bias[ni];
                            ST(lrow + (64 * wn + 16 * ni + 4 * tq) * 2, acc[mi][ni][0] + bv.x, acc[mi][ni][1] + bv.y, acc[mi][ni][2] + bv.z, acc[mi][ni][3] + bv.w);
                        }
                    }
                }
            } else
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) {
                const int rl = lrow0 + 16 * mi + rsub;            // row inside the staged block
                float t[4][4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) { t[ni][0] = acc[mi][ni][0]; t[ni][1] = acc[mi][ni][1]; t[ni][2] = acc[mi][ni][2]; t[ni][3] = acc[mi][ni][3]; }
                char* lrow = smem + rl * RS;
                if constexpr (EPI == EPI_BF16) {
                    if (p.bias == nullptr && p.act == 0) {         // wave-uniform fast path: convert and stage
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            ST(lrow + (64 * wn + 16 * ni + 4 * tq) * 2, t[ni][0], t[ni][1], t[ni][2], t[ni][3]);
                    } else if (p.act == 1) {                       // bias (hoisted: bf16_bias) + GELU, wave-uniform branch, unrolled
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const float4 b = bf16_bias[ni];
                            ST(lrow + (64 * wn + 16 * ni + 4 * tq) * 2, gelu_erf(t[ni][0] + b.x), gelu_erf(t[ni][1] + b.y), gelu_erf(t[ni][2] + b.z), gelu_erf(t[ni][3] + b.w));
                        }
                    } else {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const float4 b = bf16_bias[ni];
                            ST(lrow + (64 * wn + 16 * ni + 4 * tq) * 2, t[ni][0] + b.x, t[ni][1] + b.y, t[ni][2] + b.z, t[ni][3] + b.w);
                        }
                    }
                } else {  // EPI_SWIGLU: fragments (2p, 2p+1) = gate / up of the same 16 intermediate columns
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int cl = (2 * wn + pr) * 16 + 4 * tq;
                        float x[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[j] = silu_f(t[2 * pr][j]) * t[2 * pr + 1][j];
                        ST(lrow + cl * 2, x[0], x[1], x[2], x[3]);
                    }
                }
            }
            }   // staging by the waves of this pass
            __syncthreads();
            stamp(4);
            bool lo_as_tiles = false;
            if constexpr (EPI == EPI_SWIGLU && SPLIT && LO6) {
                if (p.out6 != nullptr) {
                    // the lo half of this pass's 128 staged rows as e2m3 operand tiles of the consuming GEMM (gemm.hpp: out6): thread = (row r of a 16-row fragment
                    // group, one of its four 32-value blocks); a wave's stores cover one group's KiB + half KiB of the image
                    lo_as_tiles = true;
                    const int r = tid_e & 15, g = (tid_e >> 4) & 3, fl = tid_e >> 6;            // fl: 4 (staged wm) + mi
                    const int ls = 64 * (fl >> 2) + 16 * (fl & 3) + r;                          // staged row
                    const int rl = 128 * (fl >> 2) + 64 * hp + 16 * (fl & 3) + r;               // tile row
                    F6Block q;
                    if (row0 + rl < p.M) {
                        float f[32];
                        const char* src = smem + LO_OFF + ls * RS + g * 64;
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const uint4 v = *(const uint4*)(src + 16 * c4);
                            const uint16_t* e16 = (const uint16_t*)&v;
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[8 * c4 + j] = from16<ODT>(e16[j]);
                        }
                        q = e2m3_block(f);
                    } else { q.d[0] = q.d[1] = q.d[2] = q.d[3] = q.d[4] = q.d[5] = 0u; q.e8 = 0u; }
                    uint8_t* t = p.out6 + ((int64_t)tm * ntn + tn) * F6_TILE_BYTES;
                    const int fb = rl >> 4;
                    *(uint4*)(t + fb * 1536 + g * 256 + r * 16) = make_uint4(q.d[0], q.d[1], q.d[2], q.d[3]);
                    *(uint2*)(t + fb * 1536 + 1024 + g * 128 + r * 8) = make_uint2(q.d[4], q.d[5]);
                    t[24576 + ((rl >> 7) * 4 + g) * 128 + (rl & 15) * 8 + ((rl >> 4) & 7)] = (uint8_t)q.e8;
                }
            }
            constexpr int LPR = NC * 2 / 16;                      // lanes (16 B each) per output row
            const int n_out = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
            const int oc0 = (EPI == EPI_SWIGLU) ? col0 / 2 : col0;
            const int seg = tid_e % LPR;
            const int oc = oc0 + 8 * seg;
            constexpr int RPP = NTHREADS / LPR;                   // rows per pass of the workgroup
            bool fused_done = false;
            if constexpr (EPI == EPI_BF16 && !SPLIT) {            // (the trainer's fused SwiGLU forms never meet split outputs: keeps the compensated kernels' register pressure down)
                if (p.swiglu_gu != nullptr) {
                    // fine-tuning backward (train.hip): this tile is d act = dy . Wd; instead of storing it, turn the saved gate | up
                    // pre-activations (16 gate / 16 up columns interleaved, the fused matrix's stored row order) into [d gate | d up] in place
                    constexpr int ODT = out16<DT>::value;
                    int lds0 = (tid_e / LPR) * RS + seg * 16;       // formed per tile (hoisted out of the persistent loop it was spilled across the K loop)
                    asm volatile("" : "+v"(lds0));
#pragma unroll 2
                    for (int rl = tid_e / LPR; rl < 256; rl += RPP) {
                        const int row = row0 + rl;
                        if (row >= p.M || oc >= n_out) continue;
                        const uint4 dv = *(const uint4*)(smem + lds0 + (rl - tid_e / LPR) * RS);
                        uint16_t* gp = p.swiglu_gu + (int64_t)row * p.swiglu_ld + 32 * (oc >> 4) + (oc & 15);
                        const uint4 gr = *(const uint4*)gp, ur = *(const uint4*)(gp + 16);
                        const uint16_t* dh = (const uint16_t*)&dv; const uint16_t* gh = (const uint16_t*)&gr; const uint16_t* uh = (const uint16_t*)&ur;
                        uint4 og, ou;
                        uint16_t* pg = (uint16_t*)&og; uint16_t* pu = (uint16_t*)&ou;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float g = from16<ODT>(gh[e]), u = from16<ODT>(uh[e]), d = from16<ODT>(dh[e]);
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
                            pg[e] = to16<ODT>(d * u * sg * (1.0f + g * (1.0f - sg)));
                            pu[e] = to16<ODT>(d * g * sg);
                        }
                        *(uint4*)gp = og;
                        *(uint4*)(gp + 16) = ou;
                    }
                    fused_done = true;
                }
            }
            // staged row rl of this pass -> tile row: plain: rl; SPLIT: staged rows [64 wm', 64 wm' + 64) are tile rows 128 wm' + 64 hp + [0, 64)
            auto tile_row = [&](int rl_) __attribute__((always_inline)) { return SPLIT ? 128 * (rl_ >> 6) + 64 * hp + (rl_ & 63) : rl_; };
            if (fused_done) {
            } else if (row0 + 256 <= p.M && oc0 + NC <= n_out && (p.ldc & 7) == 0) {   // interior tile: all LDS reads, then all stores, no branches
                constexpr int NV = PROWS / RPP;
                uint4 v[NV];
                static_assert(64 % RPP == 0, "a thread's rows i * RPP must not straddle a 64-row staging block");
                const char* lsrc = smem + (tid_e / LPR) * RS + seg * 16;
                bf16_t* out = (bf16_t*)p.C + (int64_t)(row0 + tid_e / LPR) * p.ldc + oc;       // + tile_row(i * RPP) rows: a compile-time row count per i
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = *(const uint4*)(lsrc + i * RPP * RS);
#pragma unroll
                for (int i = 0; i < NV; ++i) *(uint4*)(out + (int64_t)tile_row(i * RPP) * p.ldc) = v[i];
                if (SPLIT && !lo_as_tiles) {                      // ... and the lo tile of the same rows
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NV; ++i) v[i] = *(const uint4*)(lsrc + LO_OFF + i * RPP * RS);
#pragma unroll
                    for (int i = 0; i < NV; ++i) *(uint4*)(out + p.lo_off + (int64_t)tile_row(i * RPP) * p.ldc) = v[i];
                }
                if constexpr (EPI == EPI_BF16 && !SPLIT) {
                    // fine-tuning forward (train.hip): the tile is the gate | up pre-activations (kept for the backward); the lanes holding a
                    // gate chunk also form act = silu(gate) * up from the up chunk two 16-byte chunks further in the same LDS row
                    if (p.swiglu_act != nullptr && (seg & 3) < 2) {
                        constexpr int ODT = out16<DT>::value;
                        uint16_t* aout = p.swiglu_act + (int64_t)(row0 + tid_e / LPR) * p.swiglu_act_ld + oc0 / 2 + 16 * (seg >> 2) + 8 * (seg & 3);
#pragma unroll
                        for (int i = 0; i < 256 / RPP; ++i) {
                            const uint4 ur = *(const uint4*)(lsrc + i * RPP * RS + 32);
                            const uint16_t* gh = (const uint16_t*)&v[i]; const uint16_t* uh = (const uint16_t*)&ur;
                            uint4 o; uint16_t* po = (uint16_t*)&o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float g = from16<ODT>(gh[e]); po[e] = to16<ODT>(silu_f(g) * from16<ODT>(uh[e])); }
                            *(uint4*)(aout + (int64_t)i * RPP * p.swiglu_act_ld) = o;
                        }
                    }
                }
            } else
#pragma unroll 1
            for (int rl = tid_e / LPR; rl < PROWS; rl += RPP) {
                const int row = row0 + tile_row(rl);
                if (row >= p.M || oc >= n_out) continue;
                const uint4 v = *(const uint4*)(smem + rl * RS + seg * 16);
                bf16_t* out = (bf16_t*)p.C + (int64_t)row * p.ldc + oc;
                auto store8 = [&](bf16_t* o_, const uint4 v_) __attribute__((always_inline)) {
                    if (oc + 7 < n_out && (p.ldc & 7) == 0) *(uint4*)o_ = v_;
                    else {      // ragged columns: element stores with STATIC component indices (a dynamic index sends the vector through scratch)
                        const uint32_t w_[4] = {v_.x, v_.y, v_.z, v_.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (oc + j < n_out) o_[j] = (bf16_t)(w_[j >> 1] >> (16 * (j & 1)));
                    }
                };
                store8(out, v);
                if (SPLIT && !lo_as_tiles) store8(out + p.lo_off, *(const uint4*)(smem + LO_OFF + rl * RS + seg * 16));
                if constexpr (EPI == EPI_BF16 && !SPLIT) {
                    if (p.swiglu_act != nullptr && (seg & 3) < 2) {      // edge tiles: same rule as above (N = 2 I is a multiple of 32)
                        constexpr int ODT = out16<DT>::value;
                        const uint4 ur = *(const uint4*)(smem + rl * RS + seg * 16 + 32);
                        const uint16_t* gh = (const uint16_t*)&v; const uint16_t* uh = (const uint16_t*)&ur;
                        uint4 o; uint16_t* po = (uint16_t*)&o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float g = from16<ODT>(gh[e]); po[e] = to16<ODT>(silu_f(g) * from16<ODT>(uh[e])); }
                        *(uint4*)(p.swiglu_act + (int64_t)row * p.swiglu_act_ld + oc0 / 2 + 16 * (seg >> 2) + 8 * (seg & 3)) = o;
                    }
                }
            }
            stamp(5);
            }
        } else {  // EPI_RESID / EPI_F32: f32 tile, two passes of 128 rows
            constexpr int RS = 1024 + 16;                         // = 16 mod 256: 16 rows of a ds_write_b128 lane group on distinct banks
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                if (wm == h) {
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        char* lrow = smem + (16 * mi + rsub) * RS;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            *(float4*)(lrow + (64 * wn + 16 * ni + 4 * tq) * 4) = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
                        }
                    }
                }
                __syncthreads();
                const int col = col0 + 4 * lane;
                const bool vec = (col + 3 < p.N) && ((p.ldc & 3) == 0);
                if constexpr (EPI == EPI_RESID) {
                    const float* rsrc = p.resid_in ? p.resid_in : (const float*)p.C;      // C = resid_in + acc (in place when resid_in is null)
                    if (vec) {   // common case: 16 independent 1-KiB row loads in flight, then add + store
                        const float4 rb = p.bias ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);   // resid += acc + bias (vision tower's Linear layers)
                        float4 o[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int row = min(row0 + 128 * h + wave + 8 * i, p.M - 1);
                            o[i] = *(const float4*)(rsrc + (int64_t)row * p.ldc + col);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int rl = wave + 8 * i;
                            const int row = row0 + 128 * h + rl;
                            const float4 v = *(const float4*)(smem + rl * RS + lane * 16);
                            if (row < p.M) *(float4*)((float*)p.C + (int64_t)row * p.ldc + col) = make_float4(o[i].x + (v.x + rb.x), o[i].y + (v.y + rb.y), o[i].z + (v.z + rb.z), o[i].w + (v.w + rb.w));
#ifdef GEMM_ABLATE_NORMFOLD   // timing-only build (round 6, profiles/r06_normfold_bound.md; results are wrong by construction): what folding the FOLLOWING RMSNorm into this epilogue
                              // would add to it -- the 16-bit copy of the new residual row times the norm's weight (the next GEMM's A operand) and the row's sum of squares per
                              // column tile -- with the row scaled by its tile-local rms so that the values downstream keep the magnitudes of normalised activations
                            if (p.swiglu_act != nullptr && row < p.M) {
                                const float4 nv = make_float4(o[i].x + (v.x + rb.x), o[i].y + (v.y + rb.y), o[i].z + (v.z + rb.z), o[i].w + (v.w + rb.w));
                                float ss = nv.x * nv.x + nv.y * nv.y + nv.z * nv.z + nv.w * nv.w;
#pragma unroll
                                for (int m_ = 1; m_ < 64; m_ <<= 1) ss += __shfl_xor(ss, m_);
                                const float r = rsqrtf(ss * (1.0f / 256.0f) + 1e-6f);
                                const float4 wv = *(const float4*)(p.col_scale + col);
                                const uint16_t h0 = to16<ODT>(nv.x * wv.x * r), h1 = to16<ODT>(nv.y * wv.y * r), h2 = to16<ODT>(nv.z * wv.z * r), h3 = to16<ODT>(nv.w * wv.w * r);
                                *(uint2*)(p.swiglu_act + (int64_t)row * p.swiglu_act_ld + col) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
                                if (lane == 0) ((float*)p.lse_part)[(int64_t)row * ntn + tn] = ss;
                            }
#endif
                        }
                    } else {
#pragma unroll 1
                        for (int i = 0; i < 16; ++i) {
                            const int rl = wave + 8 * i;
                            const int row = row0 + 128 * h + rl;
                            if (row >= p.M || col >= p.N) continue;
                            const float4 v = *(const float4*)(smem + rl * RS + lane * 16);
                            const float x[4] = {v.x, v.y, v.z, v.w};
                            float* out = (float*)p.C + (int64_t)row * p.ldc + col;
                            const float* in = rsrc + (int64_t)row * p.ldc + col;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (col + j < p.N) out[j] = in[j] + x[j] + (p.bias ? p.bias[col + j] : 0.f);
                        }
                    }
                } else {
#pragma unroll 4
                    for (int i = 0; i < 16; ++i) {
                        const int rl = wave + 8 * i;
                        const int row = row0 + 128 * h + rl;
                        if (row >= p.M || col >= p.N) continue;
                        const float4 v = *(const float4*)(smem + rl * RS + lane * 16);
                        float x[4] = {v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale};
                        float* out = (float*)p.C + (int64_t)row * p.ldc + col;
                        if (vec) *(float4*)out = make_float4(x[0], x[1], x[2], x[3]);
                        else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (col + j < p.N) out[j] = x[j];
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (p.debug_stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        __syncthreads();   // the staged C tile (LDS) is reused by the next tile's ring
    }
    }  // persistent tile loop
}

#include <stdlib.h>
#include <algorithm>
static int g_gemm_persistent = getenv("BLIM_GEMM_PERSISTENT") ? atoi(getenv("BLIM_GEMM_PERSISTENT")) : 1;
static int g_gemm_tile_map = getenv("BLIM_GEMM_TILE_MAP") ? atoi(getenv("BLIM_GEMM_TILE_MAP")) : -1;   // -1: by shape
static int g_gemm_group_m = getenv("BLIM_GEMM_GROUP_M") ? atoi(getenv("BLIM_GEMM_GROUP_M")) : GROUP_M;   // M-tiles per band of the tile order
static int g_f16_saturate = getenv("BLIM_F16_SATURATE") ? atoi(getenv("BLIM_F16_SATURATE")) : 1;   // 0: fp16 stores overflow to inf again (diagnostics)
static int g_gemm_skip_epi = getenv("BLIM_GEMM_SKIP_EPI") ? atoi(getenv("BLIM_GEMM_SKIP_EPI")) : 0;
static unsigned long long* g_gemm_stamps = nullptr;
static int g_stamp_epi = getenv("BLIM_GEMM_STAMP_EPI") ? atoi(getenv("BLIM_GEMM_STAMP_EPI")) : -1;   // stamp only this epilogue / this K (tools/epi_stamps.py)
static int g_stamp_k = getenv("BLIM_GEMM_STAMP_K") ? atoi(getenv("BLIM_GEMM_STAMP_K")) : -1;
void gemm_set_debug_stamps(unsigned long long* buf) { g_gemm_stamps = buf; }

template <int EPI>
static int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu = (n_cu / 8) * 8;   // virtual block ids must keep blockIdx % 8
        if (n_cu < 8) n_cu = 8;
    }
    const int persistent = g_gemm_persistent ? n_cu : ntm * ntn;
    const dim3 grid(ntm * ntn < persistent ? ntm * ntn : persistent);
    if constexpr (EPI == EPI_QKV || EPI == EPI_SWIGLU || EPI == EPI_RESID || EPI == EPI_LSE) {
        if (p.A6) {            // 16-bit main pass + e2m3 pass over the A operand's lo part (phase 2 of the kernel).  fp16 engines: the default second pass; bf16 engines
                               // (round 6): option "precise_lo6" = 1, the fast form of their compensated mode (the lo part is 2^-9 of the value there: DESIGN.md section 4)
#define LO6_LAUNCH(DTX)                                                                                                                                  \
            if constexpr (EPI == EPI_RESID || EPI == EPI_LSE) hipLaunchKernelGGL((gemm_kernel<EPI, DTX, false, true>), grid, dim3(NTHREADS), 0, stream, p);   \
            else if (p.lo_off != 0) hipLaunchKernelGGL((gemm_kernel<EPI, DTX, true, true>), grid, dim3(NTHREADS), 0, stream, p);                            \
            else if constexpr (EPI == EPI_SWIGLU) hipLaunchKernelGGL((gemm_kernel<EPI, DTX, false, true>), grid, dim3(NTHREADS), 0, stream, p);              \
            else { blim_set_error("lo6 QKV GEMM: hi | lo outputs only"); return BLIM_ERR_ARG; }
            if (p.dtype == DT_F16) { LO6_LAUNCH(DT_F16) } else { LO6_LAUNCH(DT_BF16) }
#undef LO6_LAUNCH
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess) { blim_set_error("gemm launch failed: %s", hipGetErrorString(e4)); return BLIM_ERR_HIP; }
            return BLIM_OK;
        }
    }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_QKV || EPI == EPI_SWIGLU) {
        if (p.lo_off != 0) {   // compensated outputs: 16-bit engines (fp16: ~21 significant bits per activation, bf16: ~16)
            if (p.dtype == DT_F16) hipLaunchKernelGGL((gemm_kernel<EPI, DT_F16, true>), grid, dim3(NTHREADS), 0, stream, p);
            else if (p.dtype == DT_BF16) hipLaunchKernelGGL((gemm_kernel<EPI, DT_BF16, true>), grid, dim3(NTHREADS), 0, stream, p);
            else { blim_set_error("split (hi|lo) GEMM outputs need a 16-bit engine"); return BLIM_ERR_ARG; }
            hipError_t e2 = hipGetLastError();
            if (e2 != hipSuccess) { blim_set_error("gemm launch failed: %s", hipGetErrorString(e2)); return BLIM_ERR_HIP; }
            return BLIM_OK;
        }
    }
    if constexpr (EPI == EPI_RESID) {
        if (p.dtype == DT_F8 && p.a_mx) {
            hipLaunchKernelGGL((gemm_kernel<EPI, DT_F8, false, true>), grid, dim3(NTHREADS), 0, stream, p);
            hipError_t e3 = hipGetLastError();
            if (e3 != hipSuccess) { blim_set_error("gemm launch failed: %s", hipGetErrorString(e3)); return BLIM_ERR_HIP; }
            return BLIM_OK;
        }
    }
    if (p.dtype == DT_F8) hipLaunchKernelGGL((gemm_kernel<EPI, DT_F8>), grid, dim3(NTHREADS), 0, stream, p);
    else if (p.dtype == DT_F16) hipLaunchKernelGGL((gemm_kernel<EPI, DT_F16>), grid, dim3(NTHREADS), 0, stream, p);
    else hipLaunchKernelGGL((gemm_kernel<EPI, DT_BF16>), grid, dim3(NTHREADS), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        blim_set_error("gemm launch failed: %s", hipGetErrorString(e));
        return BLIM_ERR_HIP;
    }
    return BLIM_OK;
}

static int launch_one(GemmEpi epi, const GemmParams& p_in, hipStream_t stream);

// The kernel addresses its operands with 32-bit byte offsets from the (scalar) base pointers.  An A operand of 4 GiB or more
// (e.g. the [T, I] SwiGLU output beyond 113 k tokens at 7B) is processed as consecutive row chunks, each a whole number of
// 256-row tiles: rows are independent in every epilogue, so the chunks are ordinary launches on shifted bases.
int launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
    ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && p.lda > 0);
    ARG_CHECK(p.dtype == DT_BF16 || p.dtype == DT_F16 || p.dtype == DT_F8);
    const int64_t es = p.dtype == DT_F8 ? 1 : 2;
    const int64_t row_bytes = p.lda * es;
    if ((int64_t)p.M * row_bytes < (1ll << 32)) return launch_one(epi, p, stream);
    const int64_t chunk = ((1ll << 32) - 1) / row_bytes / BM * BM;
    ARG_CHECK(chunk >= BM);
    const int64_t c_es = (epi == EPI_F32 || epi == EPI_RESID) ? 4 : 2;
    const int ntn = (p.N + BN - 1) / BN;
    for (int64_t r0 = 0; r0 < p.M; r0 += chunk) {
        GemmParams q = p;
        q.M = (int)std::min<int64_t>(chunk, p.M - r0);
        q.A = (const bf16_t*)((const char*)p.A + r0 * row_bytes);
        if (p.C) q.C = (char*)p.C + r0 * p.ldc * c_es;
        if (p.row_scale) q.row_scale = p.row_scale + r0;
        if (p.rope_rows) q.rope_rows = p.rope_rows + r0 * 16;        // chunk-major table: the row offset inside every chunk (rope_stride unchanged)
        if (p.a_mx) q.a_mx = p.a_mx + r0;                         // r0 is a whole number of 256-row tiles: the table is tile-major inside a K-step
        if (p.out_mx) q.out_mx = p.out_mx + r0;
        if (p.A6) q.A6 = p.A6 + (r0 / BM) * (int64_t)(p.K6 / 128) * F6_TILE_BYTES;     // tile-major: a chunk is a whole number of 256-row tiles
        if (p.out6) q.out6 = p.out6 + (r0 / BM) * (int64_t)(p.N / BN) * F6_TILE_BYTES;
        if (p.labels) q.labels = p.labels + r0;
        if (p.lse_part) q.lse_part = p.lse_part + r0 * ntn;
        if (p.label_logit) q.label_logit = p.label_logit + r0;
        if (p.resid_in) q.resid_in = p.resid_in + r0 * p.ldc;
        if (p.swiglu_gu) q.swiglu_gu = p.swiglu_gu + r0 * p.swiglu_ld;
        if (p.swiglu_act) q.swiglu_act = p.swiglu_act + r0 * p.swiglu_act_ld;
        const int rc = launch_one(epi, q, stream);
        if (rc != BLIM_OK) return rc;
    }
    return BLIM_OK;
}

static int launch_one(GemmEpi epi, const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.debug_skip_epilogue = g_gemm_skip_epi;
    static const int dbg_k6 = getenv("BLIM_GEMM_LO6_K6") ? atoi(getenv("BLIM_GEMM_LO6_K6")) : 0;      // timing aid: shorten the e2m3 pass (wrong results)
    if (p.A6 && dbg_k6 > 0 && dbg_k6 < p.K6) p.K6 = dbg_k6;     // (wrong tiles too: the images are [tile][K6 / 128 steps])
    if (!g_f16_saturate) p.f16_saturate = 0;
    p.group_m = g_gemm_group_m > 0 ? g_gemm_group_m : GROUP_M;
    // measured (one MI355X, A/B in one process): narrow outputs (N = 3584 / 4608: o_proj, down_proj, qkv) gain 3-5 % from the
    // round-robin map, the wide ones (gate|up 37888, lm_head) lose 4 % -- there the XCDs already walk the same W columns in step
    // ... but only with a chip's worth of tiles: the round-robin map hands each XCD 32 consecutive tiles, so a small call (M of 1-2 k rows x N = 3584:
    // 56 tiles) ran on TWO of the eight XCDs and their share of the fabric -- down_proj 2.9 ms instead of 1.0 (round 4, rocprofv3 of the calibration calls)
    const int64_t n_tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    p.tile_map = g_gemm_tile_map >= 0 ? g_gemm_tile_map : (((p.N + BN - 1) / BN <= 32 && n_tiles >= 256) ? 1 : 0);
    p.debug_stamps = ((g_stamp_epi < 0 || g_stamp_epi == (int)epi) && (g_stamp_k < 0 || g_stamp_k == p.K)) ? g_gemm_stamps : nullptr;
    ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0);
    ARG_CHECK(p.A && p.W);
    ARG_CHECK(p.dtype == DT_BF16 || p.dtype == DT_F16 || p.dtype == DT_F8);
    const int es = p.dtype == DT_F8 ? 1 : 2;
    ARG_CHECK((int64_t)p.K * es % 128 == 0);                  // whole 128-byte K-steps
    ARG_CHECK(p.lda * es % 16 == 0);
    ARG_CHECK(p.dtype != DT_F8 || ((p.row_scale || p.a_mx) && p.col_scale));
    ARG_CHECK((!p.a_mx && !p.out_mx) || (p.dtype == DT_F8 && p.mx_stride >= (int64_t)((p.M + BM - 1) / BM) * 256));
    ARG_CHECK(!p.a_mx || epi == EPI_RESID);                      // MX-scaled A operand: instantiated for the down projection (fp8)
    // lo6: the A operand's lo part and W as e2m3 tile images (gemm.hpp)
    ARG_CHECK(!p.A6 || ((p.dtype == DT_F16 || p.dtype == DT_BF16) && p.W6 && p.w_wrap_k == 0 && p.K6 > 0 && p.K6 % 128 == 0 && (epi == EPI_RESID || epi == EPI_QKV || epi == EPI_SWIGLU || epi == EPI_LSE)));
    ARG_CHECK(!p.out_mx || (epi == EPI_SWIGLU && p.N % 256 == 0 && p.ldc % 16 == 0));
    ARG_CHECK(!p.out6 || (epi == EPI_SWIGLU && p.A6 && p.lo_off > 0 && p.N % 256 == 0));
    ARG_CHECK(p.w_wrap_k == 0 || (p.K == 2 * p.w_wrap_k && (int64_t)p.w_wrap_k * es % 128 == 0));   // A = [hi | lo]: W is walked twice
    ARG_CHECK(p.lo_off == 0 || epi == EPI_BF16 || epi == EPI_QKV || epi == EPI_SWIGLU);
    ARG_CHECK((int64_t)p.M * p.lda * es < (1ll << 32) && (int64_t)p.N * (p.w_wrap_k > 0 ? p.w_wrap_k : p.K) * es < (1ll << 32));  // 32-bit operand offsets
    switch (epi) {
        case EPI_BF16:
            ARG_CHECK(p.C && p.ldc % 4 == 0);
            ARG_CHECK(!p.swiglu_gu || (p.N % 16 == 0 && p.swiglu_ld % 8 == 0 && p.swiglu_ld >= 2 * (int64_t)p.N));
            ARG_CHECK(!p.swiglu_act || (p.N % 32 == 0 && p.swiglu_act_ld % 8 == 0 && p.swiglu_act_ld >= p.N / 2 && !p.swiglu_gu));
            return launch_t<EPI_BF16>(p, stream);
        case EPI_F32: ARG_CHECK(p.C && p.bias == nullptr); return launch_t<EPI_F32>(p, stream);
        case EPI_RESID: ARG_CHECK(p.C && p.ldc % 4 == 0); return launch_t<EPI_RESID>(p, stream);
        case EPI_QKV:
            ARG_CHECK(p.C && p.bias && p.rope_rows && p.rope_stride >= p.M && p.N % 128 == 0 && p.rope_cols % 128 == 0 && p.ldc % 4 == 0);
            return launch_t<EPI_QKV>(p, stream);
        case EPI_SWIGLU: ARG_CHECK(p.C && p.N % 32 == 0 && p.ldc % 4 == 0); return launch_t<EPI_SWIGLU>(p, stream);
        case EPI_LSE: ARG_CHECK(p.labels && p.lse_part && p.label_logit); return launch_t<EPI_LSE>(p, stream);
    }
    blim_set_error("unknown epilogue %d", (int)epi);
    return BLIM_ERR_ARG;
}
