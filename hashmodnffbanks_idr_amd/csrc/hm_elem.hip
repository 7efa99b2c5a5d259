// hm_elem.hip - fused elementwise / reduction kernels of the grad-enabled MLP path.
//
// nn.Softplus(beta=100, threshold=20) forward, backward and DOUBLE backward
// (reference: model/implicit_differentiable_renderer.py:84,104-105; the double backward is what the
// eikonal term's create_graph=True gradient needs, :116-128): autograd's generic formula for the
// softplus double backward launches ~9 elementwise kernels per layer over [points, 512] tensors;
// here each direction is one bandwidth-bound pass.  Also the bias-gradient column sum.
#include "hm_common.h"

namespace {

constexpr int kET = 256;

__global__ __launch_bounds__(kET) void softplus_fwd_kernel(const float *__restrict__ z, float *__restrict__ y,
                                                           int64_t n, float beta, float thr) {
    const int64_t i = ((int64_t)blockIdx.x * kET + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(z + i);
        float4 o;
        o.x = hm_softplus_fwd(v.x, beta, thr);
        o.y = hm_softplus_fwd(v.y, beta, thr);
        o.z = hm_softplus_fwd(v.z, beta, thr);
        o.w = hm_softplus_fwd(v.w, beta, thr);
        *reinterpret_cast<float4 *>(y + i) = o;
    } else {
        for (int64_t k = i; k < n; ++k) y[k] = hm_softplus_fwd(z[k], beta, thr);
    }
}

// gz = gy * s1(z)
__global__ __launch_bounds__(kET) void softplus_bwd_kernel(const float *__restrict__ z, const float *__restrict__ gy,
                                                           float *__restrict__ gz, int64_t n, float beta, float thr) {
    const int64_t i = ((int64_t)blockIdx.x * kET + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(z + i);
        const float4 g = *reinterpret_cast<const float4 *>(gy + i);
        float4 o;
        o.x = g.x * hm_sp_deriv(v.x, beta, thr).s1;
        o.y = g.y * hm_sp_deriv(v.y, beta, thr).s1;
        o.z = g.z * hm_sp_deriv(v.z, beta, thr).s1;
        o.w = g.w * hm_sp_deriv(v.w, beta, thr).s1;
        *reinterpret_cast<float4 *>(gz + i) = o;
    } else {
        for (int64_t k = i; k < n; ++k) gz[k] = gy[k] * hm_sp_deriv(z[k], beta, thr).s1;
    }
}

// backward of gz = gy*s1(z) for an incoming gg:  d_gy = gg*s1(z),  d_z = gg*gy*s2(z)
__global__ __launch_bounds__(kET) void softplus_bwd_bwd_kernel(const float *__restrict__ z,
                                                               const float *__restrict__ gy,
                                                               const float *__restrict__ gg,
                                                               float *__restrict__ d_gy, float *__restrict__ d_z,
                                                               int64_t n, float beta, float thr) {
    const int64_t i = ((int64_t)blockIdx.x * kET + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(z + i);
        const float4 g = *reinterpret_cast<const float4 *>(gy + i);
        const float4 q = *reinterpret_cast<const float4 *>(gg + i);
        const SpDeriv a = hm_sp_deriv(v.x, beta, thr), b = hm_sp_deriv(v.y, beta, thr), c = hm_sp_deriv(v.z, beta, thr),
                      d = hm_sp_deriv(v.w, beta, thr);
        *reinterpret_cast<float4 *>(d_gy + i) = make_float4(q.x * a.s1, q.y * b.s1, q.z * c.s1, q.w * d.s1);
        *reinterpret_cast<float4 *>(d_z + i) =
            make_float4(q.x * g.x * a.s2, q.y * g.y * b.s2, q.z * g.z * c.s2, q.w * g.w * d.s2);
    } else {
        for (int64_t k = i; k < n; ++k) {
            const SpDeriv a = hm_sp_deriv(z[k], beta, thr);
            d_gy[k] = gg[k] * a.s1;
            d_z[k] = gg[k] * gy[k] * a.s2;
        }
    }
}

// out[n] += sum_m x[m, n]   (bias gradient); rows are split over blockIdx.y, fp32 atomics combine slabs
__global__ __launch_bounds__(kET) void colsum_kernel(const float *__restrict__ x, int64_t M, int64_t N, int64_t ld,
                                                     float *__restrict__ out, int rows_per_block) {
    const int64_t n = (int64_t)blockIdx.x * kET + threadIdx.x;
    if (n >= N) return;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t m1 = min(M, m0 + rows_per_block);
    // eight independent row streams per thread keep the loads of a slab in flight together
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    const float *p = x + m0 * ld + n;
    int64_t m = m0;
    for (; m + 8 <= m1; m += 8, p += 8 * ld) {
        a0 += p[0]; a1 += p[ld]; a2 += p[2 * ld]; a3 += p[3 * ld];
        a4 += p[4 * ld]; a5 += p[5 * ld]; a6 += p[6 * ld]; a7 += p[7 * ld];
    }
    for (; m < m1; ++m, p += ld) a0 += *p;
    atomicAdd(out + n, ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
}


// the same for a list of matrices in ONE launch (the bias gradients of one backward pass): blockIdx.z = item
struct ColsumTable {
    hm_colsum_item it[HM_COLSUM_MAX_ITEMS];
};
__global__ __launch_bounds__(kET) void colsum_multi_kernel(ColsumTable t, int rows_per_block) {
    const hm_colsum_item I = t.it[blockIdx.z];
    const int64_t n = (int64_t)blockIdx.x * kET + threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
    if (n >= I.N || m0 >= I.M) return;
    const int64_t m1 = min(I.M, m0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    const int64_t ld = I.ld;
    const float *p = I.x + m0 * ld + n;
    int64_t m = m0;
    for (; m + 8 <= m1; m += 8, p += 8 * ld) {
        a0 += p[0]; a1 += p[ld]; a2 += p[2 * ld]; a3 += p[3 * ld];
        a4 += p[4 * ld]; a5 += p[5 * ld]; a6 += p[6 * ld]; a7 += p[7 * ld];
    }
    for (; m < m1; ++m, p += ld) a0 += *p;
    atomicAdd(I.out + n, ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
}

// ---- soft clamp of the SDF column (implicit_differentiable_renderer.py:112, density_net.py:20-30) ----------------
// forward: out = zL with column 0 replaced by sdf = tanh(s / (2 + rho(s))), rho under no_grad;
//          c = d sdf / d s = (1 - sdf^2) / (2 + rho), denom = 2 + rho  (what the gradient sweeps reuse)
__global__ __launch_bounds__(kET) void sdf_head_fwd_kernel(const float *__restrict__ zl, int64_t n, int64_t cols,
                                                           float beta_rho, float *__restrict__ out,
                                                           float *__restrict__ sdf, float *__restrict__ c,
                                                           float *__restrict__ denom) {
    const int64_t i = (int64_t)blockIdx.x * kET + threadIdx.x;
    if (i >= n * cols) return;
    const int64_t row = i / cols, col = i - row * cols;
    float v = zl[i];
    if (col == 0) {
        const float s = v;
        const float sg = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
        const float rho = (1.0f / beta_rho) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta_rho));
        const float d = 2.0f + rho;
        const float t = tanhf(s / d);
        sdf[row] = t;
        denom[row] = d;
        c[row] = (1.0f - t * t) / d;
        v = t;
    }
    out[i] = v;
}

// backward: zb = d_out with column 0 replaced by d_out[:,0] * c (+ cb * (-2 sdf c / denom), the share of the
// gradient sweep's adjoint, cb = c-bar; NULL when no gradient flows through d sdf / d x)
__global__ __launch_bounds__(kET) void sdf_head_bwd_kernel(const float *__restrict__ d_out, int64_t n, int64_t cols,
                                                           const float *__restrict__ sdf, const float *__restrict__ c,
                                                           const float *__restrict__ denom, const float *__restrict__ cb,
                                                           float *__restrict__ zb) {
    const int64_t i = (int64_t)blockIdx.x * kET + threadIdx.x;
    if (i >= n * cols) return;
    const int64_t row = i / cols, col = i - row * cols;
    float v = d_out[i];
    if (col == 0) {
        v = v * c[row];
        if (cb) v += cb[row] * (-2.0f * sdf[row] * c[row] / denom[row]);
    }
    zb[i] = v;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

// ---- SIREN activation sin(w0 x) (embeddings/Sine.py:10-12) and NeRF positional encoding (frequency_enc.py:6-51) with
// their first / second order passes - the filter-bank embedders' grad path spends a third of its launches on the torch
// expressions of these two (per layer: mul, sin | cos, mul, mul | ~8 more; per chunk: 12 x (mul, sin / cos), cat, ...)
__global__ __launch_bounds__(kET) void sine_kernel(int order, const float *__restrict__ x, const float *__restrict__ gy,
                                                   const float *__restrict__ gg, float *__restrict__ out0,
                                                   float *__restrict__ out1, int64_t n, float w0) {
    const int64_t i = (int64_t)blockIdx.x * kET + threadIdx.x;
    if (i >= n) return;
    const float u = __fmul_rn(x[i], w0);     // torch.sin(input * w0): fp32 product first
    if (order == 0) {
        out0[i] = sinf(u);
    } else if (order == 1) {
        out0[i] = __fmul_rn(__fmul_rn(gy[i], cosf(u)), w0);          // (gy * cos(u)) * w0, autograd's order
    } else {
        float sn, cs;
        sincosf(u, &sn, &cs);
        out0[i] = __fmul_rn(__fmul_rn(gg[i], w0), cs);               // d/d gy
        out1[i] = -__fmul_rn(__fmul_rn(__fmul_rn(gg[i], w0), gy[i]), __fmul_rn(sn, w0));   // d/d x
    }
}

struct PosEncArgs {
    float freq[16];
    int32_t n_freq, dim;
};

// rows [c | c | sin(f0 c) | cos(f0 c) | sin(f1 c) | ...], width 2 dim + 2 n_freq dim (the reference's embed() lists the
// input twice when include_input is set); one thread per (row, component)
__global__ __launch_bounds__(kET) void posenc_kernel(int order, PosEncArgs a, const float *__restrict__ c, int64_t ldc,
                                                     const float *__restrict__ g, int64_t ldg,
                                                     const float *__restrict__ gg, float *__restrict__ out0,
                                                     int64_t ld0, float *__restrict__ out1, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * kET + threadIdx.x;
    const int D = a.dim;
    const int64_t i = t / D;
    const int d = (int)(t - i * D);
    if (i >= n) return;
    const float v = c[i * ldc + d];
    if (order == 0) {
        float *o = out0 + i * ld0;
        o[d] = v;
        o[D + d] = v;
        for (int k = 0; k < a.n_freq; ++k) {
            float sn, cs;
            sincosf(__fmul_rn(v, a.freq[k]), &sn, &cs);
            o[2 * D + 2 * k * D + d] = sn;
            o[2 * D + (2 * k + 1) * D + d] = cs;
        }
    } else if (order == 1) {
        const float *gr = g + i * ldg;
        float acc = gr[d] + gr[D + d];
        for (int k = 0; k < a.n_freq; ++k) {
            float sn, cs;
            sincosf(__fmul_rn(v, a.freq[k]), &sn, &cs);
            acc += a.freq[k] * (cs * gr[2 * D + 2 * k * D + d] - sn * gr[2 * D + (2 * k + 1) * D + d]);
        }
        out0[i * ld0 + d] = acc;
    } else {
        // backward of order 1 along gg [n, D]: out0 = d/d g rows [n, W], out1 = d/d c [n, D]
        const float *gr = g + i * ldg;
        const float q = gg[i * D + d];
        float *o = out0 + i * ld0;
        o[d] = q;
        o[D + d] = q;
        float acc = 0.0f;
        for (int k = 0; k < a.n_freq; ++k) {
            float sn, cs;
            sincosf(__fmul_rn(v, a.freq[k]), &sn, &cs);
            const float f = a.freq[k];
            o[2 * D + 2 * k * D + d] = q * f * cs;
            o[2 * D + (2 * k + 1) * D + d] = -(q * f * sn);
            acc += f * f * (sn * gr[2 * D + 2 * k * D + d] + cs * gr[2 * D + (2 * k + 1) * D + d]);
        }
        out1[i * D + d] = -(q * acc);
    }
}

// ---- per-row normalisation (y - mean) / sqrt(var + eps), biased variance: what nn.InstanceNorm1d does to the 2-D tensor
// StyleAttention hands it (style_Attention/styleMod.py:41-43), with its first / second order passes.
//   F(y, g)   = (g - mean(g) - yh mean(g yh)) / sigma          (backward; yh = normalised row; linear and symmetric in g)
//   order 2   : d/dg (gg . F) = F(y, gg);   d/dy_j (gg . F) = -[ yh_j (a - b c / n) + c (gg_j - mean(gg) - yh_j b / n)
//                                                                + b (g_j - mean(g) - yh_j c / n) ] / (n sigma^2)
//               with a = sum(gg g) - sum(gg) sum(g) / n,  b = sum(gg yh),  c = sum(g yh)
// Eight lanes per row (each keeps <= 16 of the row's values in registers; rows are 56 / 72 floats wide), sums by shuffles.
constexpr int kRowLanes = 8, kRowMaxPerLane = 16;

__device__ __forceinline__ float row_sum8(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

__global__ __launch_bounds__(kET) void rownorm_kernel(int order, const float *__restrict__ y, const float *__restrict__ g,
                                                      const float *__restrict__ gg, float *__restrict__ out0,
                                                      float *__restrict__ out1, int64_t rows, int W, float eps) {
    const int64_t t = (int64_t)blockIdx.x * kET + threadIdx.x;
    int64_t r = t / kRowLanes;
    const int sub = (int)(t % kRowLanes);
    const bool live = r < rows;
    if (!live) r = rows - 1;                     // keep the whole wave in the shuffles
    const int cnt = (W - sub + kRowLanes - 1) / kRowLanes;      // this lane's elements: sub, sub + 8, ...
    const float inv_n = 1.0f / (float)W;
    float yv[kRowMaxPerLane], gv[kRowMaxPerLane], qv[kRowMaxPerLane];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowMaxPerLane; ++i) {
        yv[i] = i < cnt ? y[r * W + sub + kRowLanes * i] : 0.0f;
        s += yv[i];
    }
    const float mean = row_sum8(s) * inv_n;
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowMaxPerLane; ++i) {
        const float d = i < cnt ? yv[i] - mean : 0.0f;
        var += d * d;
    }
    var = row_sum8(var) * inv_n;
    const float sigma = sqrtf(var + eps);
    const float rs = 1.0f / sigma;
    if (order == 0) {
#pragma unroll
        for (int i = 0; i < kRowMaxPerLane; ++i)
            if (live && i < cnt) out0[r * W + sub + kRowLanes * i] = (yv[i] - mean) / sigma;
        return;
    }
    float sg = 0.0f, c = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowMaxPerLane; ++i) {
        gv[i] = i < cnt ? g[r * W + sub + kRowLanes * i] : 0.0f;
        yv[i] = i < cnt ? (yv[i] - mean) * rs : 0.0f;          // yh from here on
        sg += gv[i];
        c += gv[i] * yv[i];
    }
    sg = row_sum8(sg);
    c = row_sum8(c);
    if (order == 1) {
#pragma unroll
        for (int i = 0; i < kRowMaxPerLane; ++i)
            if (live && i < cnt) out0[r * W + sub + kRowLanes * i] = (gv[i] - sg * inv_n - yv[i] * c * inv_n) * rs;
        return;
    }
    float sq = 0.0f, b = 0.0f, a = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowMaxPerLane; ++i) {
        qv[i] = i < cnt ? gg[r * W + sub + kRowLanes * i] : 0.0f;
        sq += qv[i];
        b += qv[i] * yv[i];
        a += qv[i] * gv[i];
    }
    sq = row_sum8(sq);
    b = row_sum8(b);
    a = row_sum8(a) - sq * sg * inv_n;
    const float tt = a - b * c * inv_n;
    const float k2 = rs * rs * inv_n;
#pragma unroll
    for (int i = 0; i < kRowMaxPerLane; ++i) {
        if (!(live && i < cnt)) continue;
        const float fq = qv[i] - sq * inv_n - yv[i] * b * inv_n, fg = gv[i] - sg * inv_n - yv[i] * c * inv_n;
        out0[r * W + sub + kRowLanes * i] = fq * rs;
        out1[r * W + sub + kRowLanes * i] = -k2 * (yv[i] * tt + c * fq + b * fg);
    }
}

// ---- weight-norm fold of SEVERAL layers in one launch: W = g * v / ||v||_row (nn.utils.weight_norm, dim = 0) and its
// backward - torch launches one _weight_norm kernel per layer and pass (14 layers: 23 + 14 launches per step).
// Same arithmetic as ATen's weight_norm_fwd/bwd_first_dim kernels: w = (g v) (1/norm);
// grad_g = <grad_w, v> / norm, grad_v = g (grad_w / norm - v <grad_w, v> / norm^3).  One workgroup per row.
struct WnLayer {
    const float *v, *g, *gw;    // [rows, cols], [rows], grad_w [rows, cols] (backward)
    float *w, *norm, *gv, *gg;  // forward: w, norm [rows]; backward: grad_v, grad_g
    int32_t rows, cols, row0, pad_;
};
struct WnTable {
    WnLayer layer[HM_MAX_LAYERS * 2];
    int32_t n;
};

__device__ __forceinline__ float wn_block_sum(float v, float *red) {   // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kET) void weight_norm_multi_kernel(WnTable t, int backward) {
    __shared__ float red[4];
    int li = 0;
    while (li + 1 < t.n && (int)blockIdx.x >= t.layer[li + 1].row0) ++li;
    const WnLayer &L = t.layer[li];
    const int row = (int)blockIdx.x - L.row0;
    const float *v = L.v + (int64_t)row * L.cols;
    if (!backward) {
        float s = 0.0f;
        for (int k = threadIdx.x; k < L.cols; k += kET) s += v[k] * v[k];
        const float norm = sqrtf(wn_block_sum(s, red));
        const float rnorm = 1.0f / norm, g = L.g[row];
        if (threadIdx.x == 0) L.norm[row] = norm;
        float *w = L.w + (int64_t)row * L.cols;
        for (int k = threadIdx.x; k < L.cols; k += kET) w[k] = g * v[k] * rnorm;
    } else {
        const float *gw = L.gw + (int64_t)row * L.cols;
        float s = 0.0f;
        for (int k = threadIdx.x; k < L.cols; k += kET) s += gw[k] * v[k];
        const float dot = wn_block_sum(s, red);
        const float rnorm = 1.0f / L.norm[row], rnorm3 = rnorm * rnorm * rnorm, g = L.g[row];
        if (threadIdx.x == 0) L.gg[row] = dot * rnorm;
        float *gv = L.gv + (int64_t)row * L.cols;
        for (int k = threadIdx.x; k < L.cols; k += kET) gv[k] = g * (rnorm * gw[k] - rnorm3 * v[k] * dot);
    }
}

extern "C" {

int hm_weight_norm_multi(int backward, int n_layers, const hm_wn_layer *layers, void *stream) {
    HM_CHECK_ARG(n_layers >= 0 && n_layers <= HM_MAX_LAYERS * 2, "hm_weight_norm_multi: too many layers");
    if (n_layers == 0) return HM_OK;
    HM_CHECK_ARG(layers != nullptr, "hm_weight_norm_multi: NULL table");
    WnTable t;
    int row0 = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hm_wn_layer &s = layers[i];
        HM_CHECK_ARG(s.rows >= 1 && s.cols >= 1 && s.v && s.g && s.norm, "hm_weight_norm_multi: bad layer");
        HM_CHECK_ARG(backward ? (s.grad_w && s.grad_v && s.grad_g) : (s.w != nullptr), "hm_weight_norm_multi: NULL pointer");
        WnLayer &d = t.layer[i];
        d.v = s.v; d.g = s.g; d.gw = s.grad_w; d.w = s.w; d.norm = s.norm; d.gv = s.grad_v; d.gg = s.grad_g;
        d.rows = s.rows; d.cols = s.cols; d.row0 = row0; d.pad_ = 0;
        row0 += s.rows;
    }
    t.n = n_layers;
    hipLaunchKernelGGL(weight_norm_multi_kernel, dim3((unsigned)row0), dim3(kET), 0, as_stream(stream), t, backward);
    HM_CHECK_LAUNCH("hm_weight_norm_multi");
    return HM_OK;
}

int hm_rownorm(int order, const float *y, const float *g, const float *gg, float *out0, float *out1, int64_t rows,
               int width, float eps, void *stream) {
    HM_CHECK_ARG(order >= 0 && order <= 2, "hm_rownorm: order must be 0 (forward), 1 (backward) or 2 (double backward)");
    HM_CHECK_ARG(rows >= 0 && width >= 1 && width <= kRowLanes * kRowMaxPerLane, "hm_rownorm: width must be 1 .. 128");
    if (rows == 0) return HM_OK;
    HM_CHECK_ARG(y && out0 && (order == 0 || g) && (order < 2 || (gg && out1)), "hm_rownorm: NULL pointer");
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)((rows * kRowLanes + kET - 1) / kET)), dim3(kET), 0,
                       as_stream(stream), order, y, g, gg, out0, out1, rows, width, eps);
    HM_CHECK_LAUNCH("hm_rownorm");
    return HM_OK;
}

int hm_sine(int order, const float *x, const float *gy, const float *gg, float *out0, float *out1, int64_t n, float w0,
            void *stream) {
    HM_CHECK_ARG(order >= 0 && order <= 2, "hm_sine: order must be 0 (forward), 1 (backward) or 2 (double backward)");
    HM_CHECK_ARG(n >= 0, "hm_sine: n < 0");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && out0 && (order == 0 || gy) && (order < 2 || (gg && out1)), "hm_sine: NULL pointer");
    hipLaunchKernelGGL(sine_kernel, dim3((unsigned)((n + kET - 1) / kET)), dim3(kET), 0, as_stream(stream), order, x, gy,
                       gg, out0, out1, n, w0);
    HM_CHECK_LAUNCH("hm_sine");
    return HM_OK;
}

int hm_posenc(int order, const float *freqs, int n_freq, int dim, const float *c, int64_t c_stride, const float *g,
              int64_t g_stride, const float *gg, float *out0, int64_t out0_stride, float *out1, int64_t n, void *stream) {
    HM_CHECK_ARG(order >= 0 && order <= 2, "hm_posenc: order must be 0 (forward), 1 (backward) or 2 (double backward)");
    HM_CHECK_ARG(n >= 0 && n_freq >= 1 && n_freq <= 16 && dim >= 1 && dim <= 64, "hm_posenc: bad shape");
    const int width = 2 * dim + 2 * n_freq * dim;
    HM_CHECK_ARG(c_stride >= dim, "hm_posenc: c_stride < dim");
    HM_CHECK_ARG(order == 0 || g_stride >= width, "hm_posenc: g_stride < row width");
    HM_CHECK_ARG(out0_stride >= (order == 1 ? dim : width), "hm_posenc: out0_stride too small");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(freqs && c && out0 && (order == 0 || g) && (order < 2 || (gg && out1)), "hm_posenc: NULL pointer");
    PosEncArgs a;
    for (int k = 0; k < 16; ++k) a.freq[k] = k < n_freq ? freqs[k] : 0.0f;   // (host array)
    a.n_freq = n_freq; a.dim = dim;
    const int64_t threads = n * dim;
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((threads + kET - 1) / kET)), dim3(kET), 0, as_stream(stream), order,
                       a, c, c_stride, g, g_stride, gg, out0, out0_stride, out1, n);
    HM_CHECK_LAUNCH("hm_posenc");
    return HM_OK;
}

int hm_softplus(int order, const float *z, const float *gy, const float *gg, float *out0, float *out1, int64_t n,
                float beta, float threshold, void *stream) {
    HM_CHECK_ARG(order >= 0 && order <= 2, "hm_softplus: order must be 0 (forward), 1 (backward) or 2 (double backward)");
    HM_CHECK_ARG(n >= 0, "hm_softplus: n < 0");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(z && out0, "hm_softplus: NULL pointer");
    HM_CHECK_ARG(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(gy) |
                   reinterpret_cast<uintptr_t>(gg) | reinterpret_cast<uintptr_t>(out1)) & 15u) == 0,
                 "hm_softplus: pointers must be 16-byte aligned");
    const unsigned grid = (unsigned)((n + kET * 4 - 1) / (kET * 4));
    hipStream_t st = as_stream(stream);
    if (order == 0) {
        hipLaunchKernelGGL(softplus_fwd_kernel, dim3(grid), dim3(kET), 0, st, z, out0, n, beta, threshold);
    } else if (order == 1) {
        HM_CHECK_ARG(gy != nullptr, "hm_softplus: gy is NULL");
        hipLaunchKernelGGL(softplus_bwd_kernel, dim3(grid), dim3(kET), 0, st, z, gy, out0, n, beta, threshold);
    } else {
        HM_CHECK_ARG(gy && gg && out1, "hm_softplus: NULL pointer");
        hipLaunchKernelGGL(softplus_bwd_bwd_kernel, dim3(grid), dim3(kET), 0, st, z, gy, gg, out0, out1, n, beta,
                           threshold);
    }
    HM_CHECK_LAUNCH("hm_softplus");
    return HM_OK;
}

int hm_sdf_head(int backward, const float *in, int64_t n, int64_t cols, float beta_rho, float *out, float *sdf,
                float *c, float *denom, const float *cb, void *stream) {
    HM_CHECK_ARG(n >= 0 && cols >= 1, "hm_sdf_head: bad shape");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(in && out && sdf && c && denom, "hm_sdf_head: NULL pointer");
    HM_CHECK_ARG(backward || beta_rho > 0.0f, "hm_sdf_head: beta must be positive");
    const unsigned grid = (unsigned)((n * cols + kET - 1) / kET);
    if (!backward)
        hipLaunchKernelGGL(sdf_head_fwd_kernel, dim3(grid), dim3(kET), 0, as_stream(stream), in, n, cols, beta_rho, out,
                           sdf, c, denom);
    else
        hipLaunchKernelGGL(sdf_head_bwd_kernel, dim3(grid), dim3(kET), 0, as_stream(stream), in, n, cols, sdf, c, denom,
                           cb, out);
    HM_CHECK_LAUNCH("hm_sdf_head");
    return HM_OK;
}

static int colsum_impl(const float *x, int64_t M, int64_t N, int64_t ld, float *out, bool zero_first, void *stream) {
    HM_CHECK_ARG(M >= 0 && N >= 0 && ld >= N, "hm_colsum: bad shape");
    if (N == 0) return HM_OK;
    HM_CHECK_ARG(out != nullptr, "hm_colsum: out is NULL");
    hipStream_t st = as_stream(stream);
    if (zero_first) hm_zero_u32_async(out, N, st);
    if (M == 0) return HM_OK;
    HM_CHECK_ARG(x != nullptr, "hm_colsum: x is NULL");
    int rows = 32;
    int64_t slabs = (M + rows - 1) / rows;
    if (slabs > 65535) { rows = (int)((M + 65534) / 65535); slabs = (M + rows - 1) / rows; }
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + kET - 1) / kET), (unsigned)slabs), dim3(kET), 0, st, x, M, N,
                       ld, out, rows);
    HM_CHECK_LAUNCH("hm_colsum");
    return HM_OK;
}

// 2-D copy as a KERNEL (hipMemcpy2DAsync / hipMemcpyAsync would become MEMCPY nodes of a captured graph)
__global__ __launch_bounds__(256) void copy2d_kernel(float *__restrict__ dst, int64_t ldd,
                                                     const float *__restrict__ src, int64_t lds_, int64_t rows,
                                                     int64_t cols, int vec4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec4) {
        const int64_t c4 = cols >> 2;
        if (i >= rows * c4) return;
        const int64_t r = i / c4, c = (i - r * c4) << 2;
        *reinterpret_cast<float4 *>(dst + r * ldd + c) = *reinterpret_cast<const float4 *>(src + r * lds_ + c);
    } else {
        if (i >= rows * cols) return;
        const int64_t r = i / cols, c = i - r * cols;
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}

int hm_copy2d_f32(float *dst, int64_t ld_dst, const float *src, int64_t ld_src, int64_t rows, int64_t cols,
                  void *stream) {
    HM_CHECK_ARG(rows >= 0 && cols >= 0 && ld_dst >= cols && ld_src >= cols, "hm_copy2d_f32: bad shape");
    if (rows == 0 || cols == 0) return HM_OK;
    HM_CHECK_ARG(dst && src, "hm_copy2d_f32: NULL pointer");
    const bool v4 = (cols % 4 == 0) && (ld_dst % 4 == 0) && (ld_src % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0);
    const int64_t work = v4 ? rows * (cols / 4) : rows * cols;
    HM_CHECK_ARG((work + 255) / 256 < (1ll << 31), "hm_copy2d_f32: too large");
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, as_stream(stream), dst, ld_dst,
                       src, ld_src, rows, cols, v4 ? 1 : 0);
    HM_CHECK_LAUNCH("hm_copy2d_f32");
    return HM_OK;
}

int hm_colsum(const float *x, int64_t M, int64_t N, int64_t ld, float *out, void *stream) {
    return colsum_impl(x, M, N, ld, out, true, stream);
}

int hm_colsum_acc(const float *x, int64_t M, int64_t N, int64_t ld, float *out, void *stream) {
    return colsum_impl(x, M, N, ld, out, false, stream);
}

int hm_colsum_acc_multi(const hm_colsum_item *items, int n_items, void *stream) {
    HM_CHECK_ARG(n_items >= 0 && (n_items == 0 || items), "hm_colsum_acc_multi: bad argument");
    for (int first = 0; first < n_items; first += HM_COLSUM_MAX_ITEMS) {
        const int cnt = n_items - first < HM_COLSUM_MAX_ITEMS ? n_items - first : HM_COLSUM_MAX_ITEMS;
        ColsumTable t;
        int64_t max_n = 0, max_m = 0;
        for (int i = 0; i < cnt; ++i) {
            const hm_colsum_item &I = items[first + i];
            HM_CHECK_ARG(I.M >= 0 && I.N >= 0 && I.ld >= I.N && (I.M == 0 || I.N == 0 || (I.x && I.out)),
                         "hm_colsum_acc_multi: bad item");
            t.it[i] = I;
            if (I.M > 0 && I.N > 0) {
                max_n = I.N > max_n ? I.N : max_n;
                max_m = I.M > max_m ? I.M : max_m;
            }
        }
        for (int i = cnt; i < HM_COLSUM_MAX_ITEMS; ++i) t.it[i] = hm_colsum_item{nullptr, nullptr, 0, 0, 0};
        if (max_n == 0) continue;
        int rows = 32;
        int64_t slabs = (max_m + rows - 1) / rows;
        if (slabs > 65535) { rows = (int)((max_m + 65534) / 65535); slabs = (max_m + rows - 1) / rows; }
        hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)((max_n + kET - 1) / kET), (unsigned)slabs, (unsigned)cnt),
                           dim3(kET), 0, as_stream(stream), t, rows);
    }
    HM_CHECK_LAUNCH("hm_colsum_acc_multi");
    return HM_OK;
}


}  // extern "C"

namespace {
// Camera rays and their bounding-sphere intersections in ONE launch (fixed cameras of the static training step):
// rend_util.get_camera_params (rend_util.py:48-75: lift -> pose x pixel -> normalise) and rend_util.get_sphere_intersection
// (:141-162) are ~38 elementwise / reduction / 4x4-bmm launches over a few thousand rays in torch.  The arithmetic follows
// the reference expressions term by term (separate roundings; the K = 4 / K = 3 products of the two bmm calls and the
// squared norms as k-ordered fma chains).
__global__ __launch_bounds__(256) void camera_rays_kernel(const float *__restrict__ uv, const float *__restrict__ pose,
                                                          const float *__restrict__ intr, int64_t n_img, int64_t n_pix,
                                                          float r2, float *__restrict__ dirs, float *__restrict__ cam,
                                                          float *__restrict__ t_sphere, uint8_t *__restrict__ hit) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_img * n_pix) return;
    const int64_t b = i / n_pix;
    const float *P = pose + b * 16, *K = intr + b * 16;
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
    const float x = uv[i * 2], y = uv[i * 2 + 1];
    // lift (z = 1): (x - cx + cy * sk / fy - sk * y / fy) / fx * z ;  (y - cy) / fy * z
    float xl = __fsub_rn(x, cx);
    xl = __fadd_rn(xl, __fdiv_rn(__fmul_rn(cy, sk), fy));
    xl = __fsub_rn(xl, __fdiv_rn(__fmul_rn(sk, y), fy));
    xl = __fmul_rn(__fdiv_rn(xl, fx), 1.0f);
    const float yl = __fmul_rn(__fdiv_rn(__fsub_rn(y, cy), fy), 1.0f);
    float d[3], c[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        float w = __fmul_rn(P[rr * 4], xl);
        w = __fmaf_rn(P[rr * 4 + 1], yl, w);
        w = __fmaf_rn(P[rr * 4 + 2], 1.0f, w);
        w = __fmaf_rn(P[rr * 4 + 3], 1.0f, w);
        c[rr] = P[rr * 4 + 3];
        d[rr] = __fsub_rn(w, c[rr]);
    }
    const float nrm = __fsqrt_rn(__fmaf_rn(d[2], d[2], __fmaf_rn(d[1], d[1], __fmul_rn(d[0], d[0]))));
    const float den = fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        d[rr] = __fdiv_rn(d[rr], den);
        dirs[i * 3 + rr] = d[rr];
    }
    if (i - b * n_pix == 0) { cam[b * 3] = c[0]; cam[b * 3 + 1] = c[1]; cam[b * 3 + 2] = c[2]; }
    // sphere: dot = <d, c>;  under = dot^2 - (|c|^2 - r^2)
    const float dot = __fmaf_rn(d[2], c[2], __fmaf_rn(d[1], c[1], __fmul_rn(d[0], c[0])));
    const float cn = __fsqrt_rn(__fmaf_rn(c[2], c[2], __fmaf_rn(c[1], c[1], __fmul_rn(c[0], c[0]))));
    const float under = __fsub_rn(__fmul_rn(dot, dot), __fsub_rn(__fmul_rn(cn, cn), r2));
    const bool m = under > 0.0f;
    const float root = __fsqrt_rn(m ? under : 1.0f);
    float t0 = __fsub_rn(__fmul_rn(root, -1.0f), dot), t1 = __fsub_rn(root, dot);
    t0 = m ? t0 : 0.0f; t1 = m ? t1 : 0.0f;
    t_sphere[i * 2] = fmaxf(t0, 0.0f);
    t_sphere[i * 2 + 1] = fmaxf(t1, 0.0f);
    hit[i] = m ? 1 : 0;
}
}  // namespace

extern "C" {

int hm_camera_rays(const float *uv, const float *pose, const float *intrinsics, int64_t n_images, int64_t n_pixels,
                   float radius, float *ray_dirs, float *cam_loc, float *t_sphere, uint8_t *hit, void *stream) {
    HM_CHECK_ARG(n_images >= 0 && n_pixels >= 0, "hm_camera_rays: bad size");
    if (n_images * n_pixels == 0) return HM_OK;
    HM_CHECK_ARG(uv && pose && intrinsics && ray_dirs && cam_loc && t_sphere && hit, "hm_camera_rays: NULL pointer");
    const int64_t n = n_images * n_pixels;
    const double r2 = (double)radius * (double)radius;          // (python: r ** 2 in double, then a float32 scalar)
    hipLaunchKernelGGL(camera_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), uv, pose, intrinsics, n_images, n_pixels, (float)r2, ray_dirs,
                       cam_loc, t_sphere, hit);
    HM_CHECK_LAUNCH("hm_camera_rays");
    return HM_OK;
}

}  // extern "C"
