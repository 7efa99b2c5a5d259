// coissue_probe.hip - can a VALU wave (v_fmac_f32 with SGPR weights, s_load-fed) run beside an MFMA wave
// (v_mfma_f32_32x32x2_f32) on the same SIMD at (nearly) both waves' full rates?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit0: waves 0-3 run the MFMA loop; bit1: waves 4-7 run the VALU loop.  512 threads: waves w and w+4 share a SIMD.
__global__ __launch_bounds__(512, 2) void probe(const float *__restrict__ wts, float *out, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[6];
        for (int a = 0; a < 6; ++a) acc[a] = f32x16{0};
        float av = (float)lane * 1e-3f, bv = (float)(lane + 1) * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        }
        float s = 0;
        for (int a = 0; a < 6; ++a) s += acc[a][0] + acc[a][7];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        float acc[32];
#pragma unroll
        for (int f = 0; f < 32; ++f) acc[f] = 0.0f;
        float xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = (float)(lane + k) * 1e-3f;
        const float *w = wts + __builtin_amdgcn_readfirstlane(wave - 4) * 4096;
        for (int it = 0; it < iters; ++it) {
            const float *wr = w + (it & 15) * 256;      // 32 features x 8 k of scalar weights per "octet"
#pragma unroll
            for (int f = 0; f < 32; ++f)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[f] = __builtin_fmaf(wr[f * 8 + k], xv[k], acc[f]);
        }
        float s = 0;
#pragma unroll
        for (int f = 0; f < 32; ++f) s += acc[f];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    float *w, *o;
    hipMalloc(&w, 16 * 4096 * 4);
    hipMalloc(&o, 256 * 512 * 4);
    std::vector<float> hw(16 * 4096, 1e-3f);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, w, o, iters, mode);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, w, o, iters, mode);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // per iteration: MFMA wave 24 MFMAs (= one octet of 3x2 tiles); VALU wave 256 FMAs (32 features x 8 k)
        printf("mode %d (%s): %.3f ms, %.1f cycles/iter at 2.4 GHz\n", mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both",
               ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
