// hm_optim.hip - clip_grad_norm_(max_norm) + dense Adam over ALL parameters in three launches.
//
// Replaces the tail of the reference iteration (training/idr_train.py:306-309):
//     torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0);  optimizer.step()   # torch.optim.Adam (:128)
// which torch runs as ~110 small foreach / elementwise kernels (one group per dtype/shape bucket, separate
// passes for the norms, the clip, exp_avg, exp_avg_sq, the bias corrections and the update).  The work is
// one streaming pass over (p, g, m, v) - 16 B read + 16 B written per parameter, HBM-bound and dominated by the
// 10.4 M-entry hash table - so it is written as one pass:
//   1. adam_begin_kernel   : zero the norm accumulator, ++step of every tensor that has a gradient
//   2. grad_sqnorm_kernel  : sum g^2 over every tensor: per-workgroup tree -> partials[chunk], then ONE workgroup adds
//                            the partials in a fixed order (norm_finalize_kernel).  No float atomics: the norm - and
//                            with it the clip coefficient and every updated parameter - is bitwise reproducible, so
//                            data-parallel replicas fed the same all-reduced gradients stay bitwise identical
//   3. adam_update_kernel  : g *= clip;  m, v, p updated with torch.optim.Adam's formulas
// The tensor table travels BY VALUE in the kernel arguments (<= 4 KB, like torch's multi_tensor_apply), so the
// call can be captured into a HIP graph without any host-written device buffer.
#include "hm_common.h"

#include <math.h>

namespace {

constexpr int kMaxT = HM_ADAM_MAX_TENSORS;  // tensors per launch
constexpr int kChunk = 8192;                // elements per workgroup
constexpr int kOT = 256;

struct AdamTable {
    hm_adam_tensor t[kMaxT];
    int32_t chunk_start[kMaxT + 1];  // cumulative chunk count
    int32_t n;
};

__global__ __launch_bounds__(128) void adam_begin_kernel(AdamTable tb, float *norm_sq, int zero_norm) {
    if (threadIdx.x == 0 && zero_norm) *norm_sq = 0.0f;
    if ((int)threadIdx.x < tb.n) *tb.t[threadIdx.x].step += 1;   // torch keeps one step count per parameter
}

__device__ __forceinline__ int find_tensor(const AdamTable &tb, int chunk) {
    int lo = 0, hi = tb.n - 1;  // largest i with chunk_start[i] <= chunk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tb.chunk_start[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(kOT) void grad_sqnorm_kernel(AdamTable tb, float *partials) {
    __shared__ float red[kOT / 64];
    const int ti = find_tensor(tb, blockIdx.x);
    const hm_adam_tensor T = tb.t[ti];
    const int64_t beg = (int64_t)(blockIdx.x - tb.chunk_start[ti]) * kChunk;
    const int64_t end = min(beg + kChunk, T.numel);
    float acc = 0.0f;
    if ((reinterpret_cast<uintptr_t>(T.grad) & 15u) == 0) {
        const float4 *g4 = reinterpret_cast<const float4 *>(T.grad);
        const int64_t e4 = beg + ((end - beg) & ~(int64_t)3);
        for (int64_t i = beg + 4 * threadIdx.x; i < e4; i += 4 * kOT) {
            const float4 v = g4[i >> 2];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = e4 + threadIdx.x; i < end; i += kOT) acc += T.grad[i] * T.grad[i];
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += kOT) acc += T.grad[i] * T.grad[i];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < kOT / 64; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

// fixed-order sum of the per-chunk partial sums (double accumulators: thread t adds partials t, t+1024, ...; then
// a fixed tree) -> *norm_sq.  One workgroup; a few thousand partials at most (29 M-row table: 7153 chunks).
__global__ __launch_bounds__(1024) void norm_finalize_kernel(const float *partials, int n, float *norm_sq) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *norm_sq = (float)red[0];
}

struct AdamHyper {
    float lr, beta1, beta2, eps, max_norm;
};

__global__ __launch_bounds__(kOT) void adam_update_kernel(AdamTable tb, AdamHyper hp, const float *norm_sq,
                                                          float *norm_out) {
    __shared__ float sh[3];  // clip coefficient, step size, 1/sqrt(bias_correction2)
    const int ti = find_tensor(tb, blockIdx.x);
    const hm_adam_tensor T = tb.t[ti];
    if (threadIdx.x == 0) {
        const float total = sqrtf(*norm_sq);
        float coef = 1.0f;
        if (hp.max_norm > 0.0f) coef = fminf(hp.max_norm / (total + 1e-6f), 1.0f);   // clip_grad_norm_
        const double t = (double)*T.step;
        const double bc1 = 1.0 - pow((double)hp.beta1, t), bc2 = 1.0 - pow((double)hp.beta2, t);
        sh[0] = coef;
        sh[1] = (float)((double)hp.lr / bc1);
        sh[2] = (float)(1.0 / sqrt(bc2));
        if (blockIdx.x == 0 && norm_out) *norm_out = total;
    }
    __syncthreads();
    const float coef = sh[0], step_size = sh[1], rsq_bc2 = sh[2];
    const float b1 = hp.beta1, b2 = hp.beta2, eps = hp.eps;
    const int64_t beg = (int64_t)(blockIdx.x - tb.chunk_start[ti]) * kChunk;
    const int64_t end = min(beg + kChunk, T.numel);
    auto upd = [&](float &p, float &g, float &m, float &v) {
        g = g * coef;
        m = m + (g - m) * (1.0f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
        v = v * b2 + ((1.0f - b2) * g) * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        const float denom = sqrtf(v) * rsq_bc2 + eps;  // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
        p = p - step_size * (m / denom);               // param.addcdiv_(exp_avg, denom, value=-step_size)
    };
    const bool al = ((reinterpret_cast<uintptr_t>(T.param) | reinterpret_cast<uintptr_t>(T.grad) |
                      reinterpret_cast<uintptr_t>(T.exp_avg) | reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15u) == 0;
    int64_t e4 = beg;
    if (al) {
        e4 = beg + ((end - beg) & ~(int64_t)3);
        float4 *p4 = reinterpret_cast<float4 *>(T.param), *g4 = reinterpret_cast<float4 *>(T.grad);
        float4 *m4 = reinterpret_cast<float4 *>(T.exp_avg), *v4 = reinterpret_cast<float4 *>(T.exp_avg_sq);
        for (int64_t i = beg + 4 * threadIdx.x; i < e4; i += 4 * kOT) {
            float4 p = p4[i >> 2], g = g4[i >> 2], m = m4[i >> 2], v = v4[i >> 2];
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            p4[i >> 2] = p; g4[i >> 2] = g; m4[i >> 2] = m; v4[i >> 2] = v;
        }
    }
    for (int64_t i = e4 + threadIdx.x; i < end; i += kOT) {
        float p = T.param[i], g = T.grad[i], m = T.exp_avg[i], v = T.exp_avg_sq[i];
        upd(p, g, m, v);
        T.param[i] = p; T.grad[i] = g; T.exp_avg[i] = m; T.exp_avg_sq[i] = v;
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int64_t hm_adam_scratch_floats(const hm_adam_tensor *tensors, int n_tensors) {
    if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return hm_fail(HM_ERR_INVALID, "hm_adam_scratch_floats: bad argument");
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (tensors[i].numel < 0) return hm_fail(HM_ERR_INVALID, "hm_adam_scratch_floats: negative numel");
        chunks += (tensors[i].numel + kChunk - 1) / kChunk;
    }
    return 2 + chunks;
}

int hm_adam_step(const hm_adam_tensor *tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                 float max_norm, float *scratch_dev, void *stream) {
    HM_CHECK_ARG(n_tensors >= 0, "hm_adam_step: negative tensor count");
    HM_CHECK_ARG(scratch_dev, "hm_adam_step: NULL scratch pointer");
    HM_CHECK_ARG(n_tensors == 0 || tensors, "hm_adam_step: NULL tensor table");
    HM_CHECK_ARG(lr >= 0.0f && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f,
                 "hm_adam_step: invalid hyper-parameter");
    for (int i = 0; i < n_tensors; ++i) {
        HM_CHECK_ARG(tensors[i].numel >= 0, "hm_adam_step: negative numel");
        HM_CHECK_ARG(tensors[i].numel == 0 || (tensors[i].param && tensors[i].grad && tensors[i].exp_avg &&
                                               tensors[i].exp_avg_sq && tensors[i].step),
                     "hm_adam_step: NULL tensor pointer");
        HM_CHECK_ARG(tensors[i].numel < ((int64_t)1 << 40), "hm_adam_step: tensor too large");
    }
    hipStream_t st = as_stream(stream);
    const AdamHyper hp{lr, beta1, beta2, eps, max_norm};
    auto fill = [&](int first, AdamTable &tb) {  // table of up to kMaxT non-empty tensors starting at `first`
        tb.n = 0;
        tb.chunk_start[0] = 0;
        int i = first;
        for (; i < n_tensors && tb.n < kMaxT; ++i) {
            if (tensors[i].numel == 0) continue;
            const int64_t chunks = (tensors[i].numel + kChunk - 1) / kChunk;
            if ((int64_t)tb.chunk_start[tb.n] + chunks > (int64_t)0x7fffffff) break;
            tb.t[tb.n] = tensors[i];
            tb.chunk_start[tb.n + 1] = tb.chunk_start[tb.n] + (int32_t)chunks;
            ++tb.n;
        }
        return i;
    };
    AdamTable tb;
    if (n_tensors == 0) return HM_OK;
    for (int first = 0, k = 0; first < n_tensors; ++k) {
        const int next = fill(first, tb);
        HM_CHECK_ARG(next > first, "hm_adam_step: tensor too large for one launch");
        hipLaunchKernelGGL(adam_begin_kernel, dim3(1), dim3(128), 0, st, tb, scratch_dev, k == 0 ? 1 : 0);
        first = next;
    }
    if (max_norm > 0.0f) {
        int64_t n_part = 0;
        for (int first = 0; first < n_tensors;) {
            const int next = fill(first, tb);
            HM_CHECK_ARG(next > first, "hm_adam_step: tensor too large for one launch");
            if (tb.n > 0)
                hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)tb.chunk_start[tb.n]), dim3(kOT), 0, st, tb,
                                   scratch_dev + 2 + n_part);
            n_part += tb.chunk_start[tb.n];
            first = next;
        }
        HM_CHECK_ARG(n_part < 0x7fffffff, "hm_adam_step: too many chunks");
        hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(1024), 0, st, scratch_dev + 2, (int)n_part, scratch_dev);
    }
    for (int first = 0; first < n_tensors;) {
        const int next = fill(first, tb);
        HM_CHECK_ARG(next > first, "hm_adam_step: tensor too large for one launch");
        if (tb.n > 0)
            hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)tb.chunk_start[tb.n]), dim3(kOT), 0, st, tb, hp,
                               scratch_dev, scratch_dev + 1);
        first = next;
    }
    HM_CHECK_LAUNCH("hm_adam_step");
    return HM_OK;
}

}  // extern "C"
