// Probe: does MODE.FP16_OVFL (hwreg MODE bit 23) make f32 -> f16 conversions saturate to +-65504 instead of +-inf on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void probe(const float* in, uint16_t* out, int n, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1) = 1
    const int i = threadIdx.x;
    if (i < n) {
        _Float16 h = (_Float16)in[i];
        out[i] = __builtin_bit_cast(uint16_t, h);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {in[i], -in[i]};
        h2 r = __builtin_convertvector(v, h2);
        uint32_t u = __builtin_bit_cast(uint32_t, r);
        out[n + 2 * i] = (uint16_t)(u & 0xffff); out[n + 2 * i + 1] = (uint16_t)(u >> 16);
    }
}
int main() {
    const float h_in[8] = {1.0f, 65504.0f, 65520.0f, 1.0e6f, -1.0e6f, INFINITY, -INFINITY, NAN};
    float* d_in; uint16_t* d_out; uint16_t h_out[24];
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, 8, ovfl);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d scalar cvt:", ovfl);
        for (int i = 0; i < 8; ++i) printf(" %04x", h_out[i]);
        printf("\n            packed cvt (x, -x):");
        for (int i = 0; i < 8; ++i) printf(" %04x/%04x", h_out[8 + 2 * i], h_out[8 + 2 * i + 1]);
        printf("\n");
    }
    return 0;
}
