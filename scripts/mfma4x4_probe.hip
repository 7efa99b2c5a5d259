// layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950: D[r] at lane l  ==  a[4*(l/4) + r] * b[l] ?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
            if (hd[l * 4 + r] != ha[4 * (l / 4) + r] * hb[l]) ++bad;
    printf("mfma 4x4x1 layout: %d mismatches; lane5: %g %g %g %g (a[4..7]*b[5] = %g %g %g %g)\n", bad, hd[20], hd[21], hd[22],
           hd[23], ha[4] * hb[5], ha[5] * hb[5], ha[6] * hb[5], ha[7] * hb[5]);
    return bad != 0;
}
