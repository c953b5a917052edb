// Probe 4: WHICH of a lane group's bytes does the scale of lane group G (second operand) multiply?  Only the bytes [8q, 8q+8) of
// lane group g' are non-zero in both operands; the output then changes under a doubled scale of group G iff those bytes are in G's block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const uint8_t* a_lane, const uint8_t* b_lane, float* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)a_lane)[l * 8 + i]; b[i] = ((const int*)b_lane)[l * 8 + i]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa[l], 0, sb[l]);
    for (int j = 0; j < 4; ++j) c[l * 4 + j] = acc[j];
}
int main() {
    uint8_t *da, *db; float* dc; int *dsa, *dsb; hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    int one[64]; for (int l = 0; l < 64; ++l) one[l] = 0x7f7f7f7f;
    for (int which = 0; which < 2; ++which) {
        printf("%s operand: rows = lane group g' whose bytes [8q, 8q+8) are live (q = 0..3 left to right), columns = lane group G whose scale is doubled; X = affected\n", which ? "second" : "first");
        for (int gp = 0; gp < 4; ++gp) {
            printf("  g'=%d:", gp);
            for (int q = 0; q < 4; ++q) {
                uint8_t h[2048]; memset(h, 0, sizeof h);
                for (int l = 16 * gp; l < 16 * gp + 16; ++l) for (int j = 8 * q; j < 8 * q + 8; ++j) h[l * 32 + j] = 0x38;   // 1.0 in e4m3
                hipMemcpy(da, h, 2048, hipMemcpyHostToDevice); hipMemcpy(db, h, 2048, hipMemcpyHostToDevice);
                printf("  q%d[", q);
                for (int G = 0; G < 4; ++G) {
                    int s[64]; for (int l = 0; l < 64; ++l) s[l] = (l >> 4) == G ? 0x80808080 : 0x7f7f7f7f;
                    hipMemcpy(which ? dsb : dsa, s, 256, hipMemcpyHostToDevice); hipMemcpy(which ? dsa : dsb, one, 256, hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
                    float hc[256]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
                    printf("%s", hc[0] == 16.f ? "X" : (hc[0] == 8.f ? "." : "?"));
                }
                printf("]");
            }
            printf("\n");
        }
    }
    return 0;
}
