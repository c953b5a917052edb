// fp6 (e2m3) operands on v_mfma_scale_f32_16x16x128_f8f6f4, probed with exact data before the GEMM's second pass relies on them:
//  (1) bit packing: element j of a lane = bits [6j, 6j + 6) of its 192-bit operand (v[0:5] of the 8-register tuple)?
//  (2) which (lane group g, element j) of the first operand meets which (g', j') of the second (same k)?
//  (3) which lane group's E8M0 scale byte multiplies element (g, j) -- on either side?
//  (4) MFMA-only rate: fp6 x fp6 against fp8 x fp8 and bf16 (random and zero operands; the power limit shows here).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f6_probe tools/mfma_f6_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define FMT_F6 2   // cbsz / blgp: 0 e4m3, 1 e5m2, 2 e2m3, 3 e3m2, 4 e2m1

__global__ void probe(const uint32_t* a_lane, const uint32_t* b_lane, float* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = a_lane[l * 8 + i]; b[i] = b_lane[l * 8 + i]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT_F6, FMT_F6, 0, sa[l], 0, sb[l]);
    for (int j = 0; j < 4; ++j) c[l * 4 + j] = acc[j];
}
// all (g, j) x (g', j') pairs in one launch: operand A = 1.0 at (row 0, g, j), B = 1.0 at (col 0, g', j'); out[gj][g'j'] = C[0][0]
__global__ void pairs(float* out) {
    const int l = threadIdx.x;
    for (int gj = 0; gj < 128; ++gj)
        for (int hk = 0; hk < 128; ++hk) {
            v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
            const int g = gj >> 5, j = gj & 31, h = hk >> 5, k = hk & 31;
            if (l == 16 * g) { const int bit = 6 * j; a[bit >> 5] |= 0x08 << (bit & 31); if ((bit & 31) > 26) a[(bit >> 5) + 1] |= 0x08 >> (32 - (bit & 31)); }
            if (l == 16 * h) { const int bit = 6 * k; b[bit >> 5] |= 0x08 << (bit & 31); if ((bit & 31) > 26) b[(bit >> 5) + 1] |= 0x08 >> (32 - (bit & 31)); }
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT_F6, FMT_F6, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (l == 0) out[gj * 128 + hk] = acc[0];
        }
}

static void set6(uint32_t* lane_words, int j, uint32_t code) {
    const int bit = 6 * j;
    uint64_t v = (uint64_t)(code & 63) << (bit & 31);
    lane_words[bit >> 5] |= (uint32_t)v;
    if ((bit & 31) > 26) lane_words[(bit >> 5) + 1] |= (uint32_t)(v >> 32);
}

template <int FMT>
__global__ __launch_bounds__(512) void rate(const int* src, float* out, int iters) {
    const int l = threadIdx.x & 63;
    v8i a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(l * 8 + i) * 8 + j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[4096 + (l * 4 + i) * 8 + j];
    v4f acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    uint32_t *da, *db; float* dc; int *dsa, *dsb;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    int one[64]; for (int l = 0; l < 64; ++l) one[l] = 0x7f7f7f7f;
    uint32_t ha[512], hb[512]; float hc[256];
    // (1) bit packing + value codes: A row 0 = code at element j of lane 0, B = 1.0 everywhere; C[0][0] = value(code)
    {
        const uint32_t codes[6] = {0x08, 0x1f, 0x01, 0x28, 0x10, 0x07};
        const float want[6] = {1.0f, 7.5f, 0.125f, -1.0f, 2.0f, 0.875f};
        int bad = 0;
        memset(hb, 0, sizeof hb);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set6(hb + 8 * l, j, 0x08);
        hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice); hipMemcpy(dsa, one, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, one, 256, hipMemcpyHostToDevice);
        for (int ci = 0; ci < 6; ++ci)
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 32; ++j) {
                    memset(ha, 0, sizeof ha);
                    set6(ha + 8 * (16 * g), j, codes[ci]);
                    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
                    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
                    if (hc[0] != want[ci]) { if (bad < 8) printf("  packing: code %02x at (g %d, j %d) read %g, want %g\n", codes[ci], g, j, hc[0], want[ci]); ++bad; }
                }
        printf("(1) e2m3 element j = bits [6j, 6j+6) of the lane's 192 bits, codes {1, 7.5, 0.125, -1, 2, 0.875}: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    }
    // (2) pairing
    {
        float* dout; hipMalloc(&dout, 128 * 128 * 4);
        hipLaunchKernelGGL(pairs, dim3(1), dim3(64), 0, 0, dout);
        static float ho[128 * 128]; hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        int diag = 0, off = 0;
        for (int i = 0; i < 128; ++i) for (int k = 0; k < 128; ++k) { if (ho[i * 128 + k] != 0.f) { if (i == k) ++diag; else ++off; } }
        printf("(2) first-operand element (g, j) meets second-operand element (g', j') iff equal: diagonal hits %d / 128, off-diagonal hits %d\n", diag, off);
    }
    // (3) scale blocks: only (g, j) of row 0 live on one side, ones on the other; lane group G supplies scale 2^(G+1) on the probed side
    for (int which = 0; which < 2; ++which) {
        printf("(3) %s operand: scale block (= lane group whose byte applies) of element (g, j):\n", which ? "second" : "first");
        int s[64]; for (int l = 0; l < 64; ++l) { const int e = 127 + 1 + (l >> 4); s[l] = e | e << 8 | e << 16 | e << 24; }
        hipMemcpy(which ? dsb : dsa, s, 256, hipMemcpyHostToDevice); hipMemcpy(which ? dsa : dsb, one, 256, hipMemcpyHostToDevice);
        uint32_t ones[512]; memset(ones, 0, sizeof ones);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set6(ones + 8 * l, j, 0x08);
        for (int g = 0; g < 4; ++g) {
            printf("   g=%d: ", g);
            for (int j = 0; j < 32; ++j) {
                memset(ha, 0, sizeof ha);
                set6(ha + 8 * (16 * g), j, 0x08);
                hipMemcpy(which ? db : da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(which ? da : db, ones, 2048, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
                hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
                const float v = hc[0];
                printf("%c", v == 2.f ? '0' : v == 4.f ? '1' : v == 8.f ? '2' : v == 16.f ? '3' : '?');
            }
            printf("\n");
        }
    }
    // (4) rate: two waves per SIMD (512 threads) and ONE wave per SIMD (256 threads: what a ping-pong of wave groups gives the matrix pipe)
    for (int threads : {512, 256}) {
        int* src; float* out; hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
        static int h[8192]; uint32_t x = 12345;
        for (int zero = 0; zero < 2; ++zero) {
            for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0 : (int)((x & 0x87878787u) | 0x38383838u); }
            hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
            for (int fmt = 0; fmt < 2; ++fmt) {
                const int iters = 20000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                if (fmt) hipLaunchKernelGGL(rate<FMT_F6>, dim3(256), dim3(threads), 0, 0, src, out, 100); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, src, out, 100);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (fmt) hipLaunchKernelGGL(rate<FMT_F6>, dim3(256), dim3(threads), 0, 0, src, out, iters); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, src, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = 256.0 * (threads / 64) * iters * 32 * 2.0 * 16 * 16 * 128;
                printf("(4) %d waves per SIMD, %s operands, %s 16x16x128: %.2f ms  %.0f TFLOP/s\n", threads / 256, zero ? "zero" : "random", fmt ? "mx-fp6 (e2m3)" : "mx-fp8 (e4m3)", ms, flops / ms / 1e9);
            }
        }
    }
    return 0;
}
