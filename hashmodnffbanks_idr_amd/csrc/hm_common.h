// hm_common.h - shared host/device definitions for libhashmod (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/hashmod.h"

// Level table handed to kernels BY VALUE (lands in kernarg / SGPRs; no device allocation).
struct HmLevels {
    int32_t L;
    int32_t F;
    int32_t E;                       // 3 + 2L + L*F
    int32_t pad_;
    int32_t res[HM_MAX_LEVELS];
    uint32_t rows[HM_MAX_LEVELS];
    uint32_t magic[HM_MAX_LEVELS];   // floor(2^32 / rows) for non power-of-two rows, 0 => use mask
    uint32_t row_off[HM_MAX_LEVELS]; // first row of the level inside the fused table
};

struct hm_grid_desc {
    HmLevels lv;
    uint64_t total_rows;
};

void hm_set_error(const std::string &msg);
int hm_fail(int code, const std::string &msg);

#define HM_CHECK_ARG(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) return hm_fail(HM_ERR_INVALID, (msg));       \
    } while (0)

#define HM_CHECK_LAUNCH(what)                                                                     \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) return hm_fail(HM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e__)); \
    } while (0)

#ifdef __HIPCC__
// Small device-side fills / copies as ordinary KERNELS.  hipMemsetAsync / hipMemcpyAsync become MEMSET / MEMCPY
// nodes when the call is recorded into a HIP graph; on ROCm 7.0 those nodes were seen to lose their place
// relative to the neighbouring kernel nodes when a graph is re-launched back to back (zeroed cursors in the
// middle of a ray search -> wild indices -> GPU memory fault).  Kernel nodes keep their order.
static __global__ __launch_bounds__(256) void hm_zero_u32_kernel(uint32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static __global__ __launch_bounds__(64) void hm_copy_u32_kernel(uint32_t *dst, const uint32_t *src, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
static inline void hm_zero_u32_async(void *p, int64_t n_words, hipStream_t st) {
    if (n_words > 0)
        hipLaunchKernelGGL(hm_zero_u32_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st,
                           static_cast<uint32_t *>(p), n_words);
}

// nn.Softplus(beta, threshold) and its derivatives (shared by hm_elem.hip and the GEMM epilogues), evaluated with the
// native exp2 / log2 / rcp units (v_exp_f32, v_log_f32, v_rcp_f32: 1 ulp each) instead of libm's expf / log1pf and IEEE
// divisions - those were ~100 VALU instructions per activation on the epilogue of ~64 GEMMs per training step (+ 6 us
// on a 27 us GEMM), and VALU work is serial with the MFMAs on gfx950.
//   softplus(z) = log1p(exp(bz))/beta = max(z, 0) + ln(1 + exp(-|bz|))/beta     (bz = beta z <= threshold)
// the logarithm term is <= ln 2 / beta, so its 1-2 ulp error is < 2e-9 absolute for beta = 100; s1, s2 carry ~3e-7
// relative error (tests/test_gemm_ep_gpu.py compares all three with torch at 2e-5).
struct SpDeriv {
    float s1, s2;
};
__device__ __forceinline__ float hm_softplus_fwd(float z, float beta, float thr) {
    const float bz = z * beta;
    const float t = __builtin_amdgcn_exp2f(-fabsf(bz) * 1.4426950408889634f);
    const float l = __builtin_amdgcn_logf(1.0f + t) * (0.6931471805599453f * __builtin_amdgcn_rcpf(beta));
    return bz > thr ? z : fmaxf(z, 0.0f) + l;
}
// s1 = d softplus/dz = e/(e+1), s2 = d^2 softplus/dz^2 = beta*e/(e+1)^2 (no cancellation in 1 - s1), e = exp(beta z)
__device__ __forceinline__ SpDeriv hm_sp_deriv(float z, float beta, float thr) {
    SpDeriv d;
    const float bz = z * beta;
    if (bz > thr) {
        d.s1 = 1.0f;
        d.s2 = 0.0f;
    } else {
        const float e = __builtin_amdgcn_exp2f(bz * 1.4426950408889634f);
        const float r = __builtin_amdgcn_rcpf(e + 1.0f);
        d.s1 = e * r;
        d.s2 = beta * d.s1 * r;
    }
    return d;
}

// hash of one voxel corner: reference hashGridEmbedding.py:32-40 restated in uint32
// (primes 1, 3, 2654435761; xor fold; unsigned modulo by the level's row count).
__device__ __forceinline__ uint32_t hm_mod_rows(uint32_t h, uint32_t rows, uint32_t magic) {
    if (magic == 0u) return h & (rows - 1u);  // power of two (also rows == 1)
    uint32_t q = __umulhi(h, magic);          // q in {floor(h/rows) - 1, floor(h/rows)}
    uint32_t r = h - q * rows;
    return r >= rows ? r - rows : r;
}
__device__ __forceinline__ uint32_t hm_hash3(uint32_t ux, uint32_t uy, uint32_t uz) {
    return ux ^ (uy * 3u) ^ (uz * 2654435761u);
}
// xi = trunc(x*res): fp32 product, then truncation toward zero (hashGridEmbedding.py:84-85)
__device__ __forceinline__ int32_t hm_trunc_voxel(float x, int32_t res) {
    return (int32_t)__fmul_rn(x, (float)res);
}
#endif
