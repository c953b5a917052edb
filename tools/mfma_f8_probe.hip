// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3): which (row, k) does byte j of lane l hold, and what do the
// E8M0 scale operands do?  Exact small-integer data, compared against host references for several layout hypotheses.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_f8_probe tools/mfma_f8_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* a_lane, const uint8_t* b_lane, float* c, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)a_lane)[l * 8 + i]; b[i] = ((const int*)b_lane)[l * 8 + i]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
    for (int j = 0; j < 4; ++j) c[l * 4 + j] = acc[j];
}

static uint8_t e4m3(float v) {   // exact for the small integers / powers of two used here
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);          // v = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                       // biased exponent of 1.xxx form
    int M = (int)lrintf((m * 2.f - 1.f) * 8.f);
    if (E <= 0) { M = (int)lrintf(v / powf(2.f, -9.f)); return s | (uint8_t)M; }   // subnormal: units of 2^-9
    return s | (uint8_t)((E << 3) | (M & 7));
}

int main() {
    float A[16][128], B[16][128];   // B given as [col][k]
    srand(1);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { A[i][k] = (float)(rand() % 9 - 4); B[i][k] = (float)(rand() % 7 - 3); }
    // hypotheses: k index of byte j (0..31) in lane group g = l >> 4
    auto kmap = [](int h, int g, int j) { return h == 0 ? 32 * g + j : (h == 1 ? 16 * g + (j & 15) + 64 * (j >> 4) : 8 * g + (j & 7) + 32 * (j >> 3)); };
    uint8_t *da, *db; float* dc; hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 4 * 4);
    for (int h = 0; h < 3; ++h) {
        uint8_t ha[64 * 32], hb[64 * 32];
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { const int k = kmap(h, l >> 4, j); ha[l * 32 + j] = e4m3(A[l & 15][k]); hb[l * 32 + j] = e4m3(B[l & 15][k]); }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            const int sa = sc == 0 ? 0x7f7f7f7f : (sc == 1 ? 0x80808080 : 0x7f7f7f7f), sb = sc == 2 ? 0x7e7e7e7e : 0x7f7f7f7f;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb);
            float hc[256]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
            // C layout: col = l & 15, row = (l >> 4) * 4 + j
            double err = 0, ref_mag = 0; float ratio = 0;
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
                const int row = (l >> 4) * 4 + j, col = l & 15; double r = 0;
                for (int k = 0; k < 128; ++k) r += (double)A[row][k] * B[col][k];
                err += fabs(hc[l * 4 + j] - r); ref_mag += fabs(r);
                if (fabs(r) > 20 && ratio == 0) ratio = hc[l * 4 + j] / (float)r;
            }
            printf("hypothesis %d scales a=%08x b=%08x: sum|err| = %.1f of %.1f, sample ratio got/ref = %.4f\n", h, sa, sb, err, ref_mag, ratio);
        }
    }
    return 0;
}
