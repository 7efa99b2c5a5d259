// hm_loss.hip - IDRLoss value AND gradients in one launch.
//
// Reference: code/model/loss.py:4-70
//     rgb_loss     = sum_{surface rays} |rgb - rgb_gt| / N                      (:14-20, L1 reduction='sum')
//     eikonal_loss = mean_i (||grad_theta_i||_2 - 1)^2                          (:43-48)
//     mask_loss    = (1/alpha) * sum_{other rays} BCEwithLogits(-alpha*sdf, object_mask) / N   (:22-41)
//     loss         = rgb_loss + w_eik * eikonal_loss + w_mask * mask_loss       (:62-64)
// with surface = network_object_mask & object_mask, others = ~surface.  torch runs this as ~45 elementwise /
// reduction kernels forward and ~30 backward over a few thousand elements; here one workgroup computes the four
// scalars and d loss / d (rgb_values, sdf_output, grad_theta), so backward is a multiplication by the upstream
// scalar.  Sums are fp32 trees (1024 lanes, then shuffles): the same precision class as torch's reductions.
#include "hm_common.h"

#include <math.h>

namespace {

constexpr int kLT = 1024;

__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < kLT / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(kLT) void idr_loss_kernel(const float *__restrict__ rgb, const float *__restrict__ rgb_gt,
                                                       const float *__restrict__ sdf, const uint8_t *__restrict__ hit,
                                                       const uint8_t *__restrict__ inside, int64_t n,
                                                       const float *__restrict__ grad_theta, int64_t m, float w_eik,
                                                       float w_mask, float alpha, float *__restrict__ terms,
                                                       float *__restrict__ d_rgb, float *__restrict__ d_sdf,
                                                       float *__restrict__ d_grad) {
    __shared__ float red[kLT / 64];
    const float inv_n = 1.0f / (float)n;
    float s_rgb = 0.0f, s_mask = 0.0f, s_eik = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += kLT) {
        const bool surface = hit[i] && inside[i];
        float e = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = rgb[i * 3 + c] - rgb_gt[i * 3 + c];
            e += fabsf(d);
            const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            d_rgb[i * 3 + c] = surface ? sg * inv_n : 0.0f;
        }
        if (surface) s_rgb += e;
        // binary_cross_entropy_with_logits(l, t) = max(l, 0) - l t + log1p(exp(-|l|)),  l = -alpha * sdf
        const float l = -alpha * sdf[i];
        const float t = inside[i] ? 1.0f : 0.0f;
        const float bce = fmaxf(l, 0.0f) - l * t + log1pf(expf(-fabsf(l)));
        const float sig = 1.0f / (1.0f + expf(-l));
        if (!surface) s_mask += bce;
        // d/d sdf of w_mask * (1/alpha) * bce / N  =  w_mask * (1/alpha) * (sigmoid(l) - t) * (-alpha) / N
        d_sdf[i] = surface ? 0.0f : -w_mask * (sig - t) * inv_n;
    }
    const float inv_m = m > 0 ? 1.0f / (float)m : 0.0f;
    for (int64_t i = threadIdx.x; i < m; i += kLT) {
        const float gx = grad_theta[i * 3], gy = grad_theta[i * 3 + 1], gz = grad_theta[i * 3 + 2];
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        const float r = nrm - 1.0f;
        s_eik += r * r;
        const float k = nrm > 0.0f ? w_eik * 2.0f * r * inv_m / nrm : 0.0f;
        d_grad[i * 3] = k * gx; d_grad[i * 3 + 1] = k * gy; d_grad[i * 3 + 2] = k * gz;
    }
    const float t_rgb = block_sum(s_rgb, red) * inv_n;
    const float t_mask = (1.0f / alpha) * block_sum(s_mask, red) * inv_n;
    const float t_eik = block_sum(s_eik, red) * inv_m;
    if (threadIdx.x == 0) {
        terms[0] = t_rgb + w_eik * t_eik + w_mask * t_mask;
        terms[1] = t_rgb;
        terms[2] = t_eik;
        terms[3] = t_mask;
    }
}

}  // namespace

extern "C" {

int hm_idr_loss(const float *rgb, const float *rgb_gt, const float *sdf, const uint8_t *hit, const uint8_t *inside,
                int64_t n_rays, const float *grad_theta, int64_t n_grad, float eikonal_weight, float mask_weight,
                float alpha, float *terms, float *d_rgb, float *d_sdf, float *d_grad, void *stream) {
    HM_CHECK_ARG(n_rays >= 1 && n_grad >= 0, "hm_idr_loss: bad row counts");
    HM_CHECK_ARG(rgb && rgb_gt && sdf && hit && inside && terms && d_rgb && d_sdf, "hm_idr_loss: NULL pointer");
    HM_CHECK_ARG(n_grad == 0 || (grad_theta && d_grad), "hm_idr_loss: NULL grad_theta pointer");
    HM_CHECK_ARG(alpha > 0.0f, "hm_idr_loss: alpha must be positive");
    hipLaunchKernelGGL(idr_loss_kernel, dim3(1), dim3(kLT), 0, reinterpret_cast<hipStream_t>(stream), rgb, rgb_gt, sdf,
                       hit, inside, n_rays, grad_theta, n_grad, eikonal_weight, mask_weight, alpha, terms, d_rgb, d_sdf,
                       d_grad);
    HM_CHECK_LAUNCH("hm_idr_loss");
    return HM_OK;
}

}  // extern "C"
