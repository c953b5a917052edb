// Probe 3: which outputs move when ONE lane's scale VGPR (second operand) is doubled?  Data layout h0 (k = 32 * (l >> 4) + j).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int WHICH>
__global__ void probe(const uint8_t* a_lane, const uint8_t* b_lane, float* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)a_lane)[l * 8 + i]; b[i] = ((const int*)b_lane)[l * 8 + i]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa[l], 0, sb[l]);
    for (int j = 0; j < 4; ++j) c[l * 4 + j] = acc[j];
}
static uint8_t e4m3(float v) {
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);
    int E = e - 1 + 7; int M = (int)lrintf((m * 2.f - 1.f) * 8.f);
    return s | (uint8_t)((E << 3) | (M & 7));
}
int main() {
    float A[16][128], B[16][128];
    srand(1);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { A[i][k] = (float)(rand() % 7 - 3); B[i][k] = (float)(rand() % 5 - 2); }
    uint8_t ha[64 * 32], hb[64 * 32];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { const int k = 32 * (l >> 4) + j; ha[l * 32 + j] = e4m3(A[l & 15][k]); hb[l * 32 + j] = e4m3(B[l & 15][k]); }
    uint8_t *da, *db; float* dc; int *dsa, *dsb; hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    int one[64]; for (int l = 0; l < 64; ++l) one[l] = 0x7f7f7f7f;
    float base[256];
    hipMemcpy(dsa, one, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, one, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb); hipMemcpy(base, dc, sizeof base, hipMemcpyDeviceToHost);
    double err = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { const int row = (l >> 4) * 4 + j, col = l & 15; double r = 0; for (int k = 0; k < 128; ++k) r += (double)A[row][k] * B[col][k]; err += fabs(base[l * 4 + j] - r); }
    printf("uniform scales: sum|err| = %.2f\n", err);
    for (int which = 0; which < 2; ++which)
        for (int L : {0, 5, 16, 21, 37, 63}) {
            int s[64]; for (int l = 0; l < 64; ++l) s[l] = 0x7f7f7f7f; s[L] = 0x80808080;
            hipMemcpy(which ? dsb : dsa, s, 256, hipMemcpyHostToDevice); hipMemcpy(which ? dsa : dsb, one, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
            float hc[256]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
            // classify the changed outputs: set of rows, set of cols, and which k block explains the delta
            int rows = 0, cols = 0, n = 0; int blk_ok[4] = {1, 1, 1, 1};
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
                const float d = hc[l * 4 + j] - base[l * 4 + j];
                const int row = (l >> 4) * 4 + j, col = l & 15;
                if (d != 0.f) { rows |= 1 << row; cols |= 1 << col; ++n; }
                for (int g = 0; g < 4; ++g) { double part = 0; for (int k = 32 * g; k < 32 * g + 32; ++k) part += (double)A[row][k] * B[col][k];
                    const bool in_sel = which ? (col == (L & 15)) : (row == (L & 15));
                    if (fabs((in_sel ? part : 0.0) - d) > 1e-3) blk_ok[g] = 0; }
            }
            printf("doubling the %s scale of lane %2d: %3d outputs change; rows mask %04x cols mask %04x; delta == k-block partial sum for blocks:%s%s%s%s\n",
                   which ? "second-operand" : "first-operand", L, n, rows, cols, blk_ok[0] ? " 0" : "", blk_ok[1] ? " 1" : "", blk_ok[2] ? " 2" : "", blk_ok[3] ? " 3" : "");
        }
    return 0;
}
