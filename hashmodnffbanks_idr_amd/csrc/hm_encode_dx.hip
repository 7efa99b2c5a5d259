// hm_encode_dx.hip - input-gradient kernels of the hash-grid encoder for frac_mode = trilinear (opt-in, non-parity
// mode: the reference's interpolation weights are degenerate, xf = x - x.float() == 0 at hashGridEmbedding.py:86, so
// there d(features)/dx is identically zero and none of this runs).
//
// With xs_d = x_d * res, t_d = xs_d - floor(xs_d) and corner bits b_d in {0,1}:
//   feat[l,f]          = sum_c  w_x w_y w_z T[id_c][f],                w_d = b_d ? t_d : 1 - t_d
//   d feat / d x_d     = res   * sum_c s_d      prod_{e != d} w_e  T,  s_d = b_d ? +1 : -1
//   d2 feat / dx_d dx_e = res^2 * sum_c s_d s_e  w_k               T   (d != e, k the third axis; 0 for d == e)
// The three products IDR's eikonal / normal terms need (ImplicitNetwork.gradient differentiates the embedding with
// create_graph=True, implicit_differentiable_renderer.py:116-127) are
//   hm_encode_bwd_input     gx[i,:]   = J_i^T d_feat[i,:]                   (and, given a, its derivative along a)
//   hm_encode_jvp           out[i,:]  = J_i a[i,:]                          (backward of gx w.r.t. d_feat)
//   hm_encode_bwd_table_jvp d_table  += scatter of (dJ_i/dT . a) d_feat     (backward of gx w.r.t. the table)
// where J_i = d feat[i,:] / d x[i,:]  [L*F, 3].  Inside a voxel the formulas are exact; across voxel faces the
// features are continuous and piecewise trilinear, their gradient jumps - like any trilinear grid.
#include "hm_common.h"

namespace {

constexpr int kThreads = 256;

struct CornerJet {
    float w[3];    // per-axis interpolation weight of this corner
    float s[3];    // its derivative sign
    uint32_t id;   // row inside the level
    float r;       // level resolution
};

__device__ __forceinline__ CornerJet corner_jet(const HmLevels &lv, int l, int c, float x0, float x1, float x2) {
    CornerJet j;
    const int32_t res = lv.res[l];
    const float xin[3] = {x0, x1, x2};
    uint32_t u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float xs = __fmul_rn(xin[d], (float)res);
        const float fl = floorf(xs);
        const float t = __fsub_rn(xs, fl);
        const int bit = (c >> d) & 1;
        u[d] = (uint32_t)(int32_t)fl + (uint32_t)bit;
        j.w[d] = bit ? t : __fsub_rn(1.0f, t);
        j.s[d] = bit ? 1.0f : -1.0f;
    }
    j.id = hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]);
    j.r = (float)res;
    return j;
}

// q = sum_d a_d * d(w_x w_y w_z)/dx_d
__device__ __forceinline__ float jet_dot(const CornerJet &j, float a0, float a1, float a2) {
    return j.r * (a0 * j.s[0] * j.w[1] * j.w[2] + a1 * j.s[1] * j.w[0] * j.w[2] + a2 * j.s[2] * j.w[0] * j.w[1]);
}

__device__ __forceinline__ float sum8(float v) {   // over the 8 corner lanes of one (point, level)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// out[i, l*F + f] = sum_c q_c T[id_c][f]; one lane per (point, level, corner)
__global__ __launch_bounds__(kThreads) void encode_jvp_kernel(HmLevels lv, const float *__restrict__ x, int64_t n,
                                                              const float *__restrict__ table,
                                                              const float *__restrict__ a, float *__restrict__ out,
                                                              int64_t out_stride) {
    const int L = lv.L, F = lv.F;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(gid & 7);
    const int64_t pl = gid >> 3;
    int64_t i = pl / L;
    const int l = (int)(pl - i * L);
    const bool live = i < n;
    if (!live) i = n - 1;               // keep the whole wave in the shuffles
    const CornerJet j = corner_jet(lv, l, c, x[i * 3], x[i * 3 + 1], x[i * 3 + 2]);
    const float q = jet_dot(j, a[i * 3], a[i * 3 + 1], a[i * 3 + 2]);
    const float *row = table + ((uint64_t)lv.row_off[l] + j.id) * F;
    for (int f = 0; f < F; ++f) {
        const float v = sum8(q * row[f]);
        if (live && c == 0) out[i * out_stride + l * F + f] = v;
    }
}

// d_table[id_c][f] += q_c * d_feat[i, l*F + f]
__global__ __launch_bounds__(kThreads) void encode_bwd_table_jvp_kernel(HmLevels lv, const float *__restrict__ x,
                                                                        int64_t n, const float *__restrict__ a,
                                                                        const float *__restrict__ d_feat,
                                                                        int64_t d_feat_stride,
                                                                        float *__restrict__ d_table) {
    const int L = lv.L, F = lv.F;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(gid & 7);
    const int64_t pl = gid >> 3;
    const int64_t i = pl / L;
    const int l = (int)(pl - i * L);
    if (i >= n) return;
    const CornerJet j = corner_jet(lv, l, c, x[i * 3], x[i * 3 + 1], x[i * 3 + 2]);
    const float q = jet_dot(j, a[i * 3], a[i * 3 + 1], a[i * 3 + 2]);
    if (q == 0.0f) return;
    const float *g = d_feat + i * d_feat_stride + l * F;
    float *row = d_table + ((uint64_t)lv.row_off[l] + j.id) * F;
    for (int f = 0; f < F; ++f) atomicAdd(row + f, q * g[f]);
}

// gx[i,d] = sum_{l,c} (d_feat[i,l,:] . T[id_c]) * coefficient_d; 8 lanes per point (one per corner) walk the levels,
// so the sum order is fixed (no atomics).  SECOND: coefficient_e = sum_{d != e} a_d * d2(w_x w_y w_z)/dx_d dx_e.
template <bool SECOND>
__global__ __launch_bounds__(kThreads) void encode_bwd_input_kernel(HmLevels lv, const float *__restrict__ x,
                                                                    int64_t n, const float *__restrict__ table,
                                                                    const float *__restrict__ d_feat,
                                                                    int64_t d_feat_stride,
                                                                    const float *__restrict__ a,
                                                                    float *__restrict__ gx) {
    const int L = lv.L, F = lv.F;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(gid & 7);
    int64_t i = gid >> 3;
    const bool live = i < n;
    if (!live) i = n - 1;
    const float x0 = x[i * 3], x1 = x[i * 3 + 1], x2 = x[i * 3 + 2];
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    if (SECOND) {
        a0 = a[i * 3]; a1 = a[i * 3 + 1]; a2 = a[i * 3 + 2];
    }
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    for (int l = 0; l < L; ++l) {
        const CornerJet j = corner_jet(lv, l, c, x0, x1, x2);
        const float *row = table + ((uint64_t)lv.row_off[l] + j.id) * F;
        const float *g = d_feat + i * d_feat_stride + l * F;
        float val = 0.0f;
        for (int f = 0; f < F; ++f) val += g[f] * row[f];
        if (!SECOND) {
            val *= j.r;
            g0 += val * j.s[0] * j.w[1] * j.w[2];
            g1 += val * j.s[1] * j.w[0] * j.w[2];
            g2 += val * j.s[2] * j.w[0] * j.w[1];
        } else {
            val *= j.r * j.r;
            const float s01 = j.s[0] * j.s[1] * j.w[2], s02 = j.s[0] * j.s[2] * j.w[1], s12 = j.s[1] * j.s[2] * j.w[0];
            g0 += val * (a1 * s01 + a2 * s02);
            g1 += val * (a0 * s01 + a2 * s12);
            g2 += val * (a0 * s02 + a1 * s12);
        }
    }
    g0 = sum8(g0); g1 = sum8(g1); g2 = sum8(g2);
    if (live && c == 0) {
        gx[i * 3] = g0; gx[i * 3 + 1] = g1; gx[i * 3 + 2] = g2;
    }
}

// ---- Fourier-feature columns of the embedding row [x | sin(a) | cos(a) | hash features], a_c = 2 pi x . B[:, c] -------
// (frequency_enc.py:63-67; same fp32 expression as the forward kernel: s = 2 pi x, k-ordered fma chain)
__device__ __forceinline__ float fourier_arg(const float *__restrict__ Bf, int L, int c, float s0, float s1, float s2) {
    float a = __fmul_rn(s0, Bf[c]);
    a = __fmaf_rn(s1, Bf[L + c], a);
    return __fmaf_rn(s2, Bf[2 * L + c], a);
}

// Both Fourier kernels run SIXTEEN lanes per point (lane c takes channels c, c + 16): one thread per point was a serial
// chain of L sincosf and 2 L strided loads on 12 workgroups for a 3072-point batch - 11 / 20 us per call, six calls per
// step.  The three sums are reduced over the 16 lanes with xor shuffles.
constexpr int kFLanes = 16;
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// gx = d_row[:, 0:3] + 2 pi sum_c B[:, c] (cos_c d_sin_c - sin_c d_cos_c): backward of the row w.r.t. x (the hash
// features of the reference's frac mode do not depend on x)
__global__ __launch_bounds__(kThreads) void fourier_bwd_input_kernel(const float *__restrict__ x, int64_t n,
                                                                     const float *__restrict__ Bf, int L,
                                                                     const float *__restrict__ d_row, int64_t ld,
                                                                     float *__restrict__ gx) {
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c0 = (int)(gid % kFLanes);
    const bool live = gid / kFLanes < n;
    const int64_t i = live ? gid / kFLanes : n - 1;      // (every lane of a group joins the shuffles)
    const float two_pi = 6.283185307179586f;
    const float s0 = __fmul_rn(two_pi, x[i * 3]), s1 = __fmul_rn(two_pi, x[i * 3 + 1]), s2 = __fmul_rn(two_pi, x[i * 3 + 2]);
    const float *d = d_row + i * ld;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    for (int c = c0; c < L; c += kFLanes) {
        float sn, cs;
        sincosf(fourier_arg(Bf, L, c, s0, s1, s2), &sn, &cs);
        const float q = cs * d[3 + c] - sn * d[3 + L + c];
        g0 += q * Bf[c]; g1 += q * Bf[L + c]; g2 += q * Bf[2 * L + c];
    }
    g0 = sum16(g0); g1 = sum16(g1); g2 = sum16(g2);
    if (live && c0 == 0) {
        gx[i * 3] = d[0] + two_pi * g0;
        gx[i * 3 + 1] = d[1] + two_pi * g1;
        gx[i * 3 + 2] = d[2] + two_pi * g2;
    }
}

// backward of gx(x, d_row) along gg [n,3]:  with p_c = B[:, c] . gg
//   dd_row = [gg | 2 pi p_c cos_c | -2 pi p_c sin_c | 0 ...],   d_x = -(2 pi)^2 sum_c B[:, c] p_c (sin_c d_sin_c + cos_c d_cos_c)
__global__ __launch_bounds__(kThreads) void fourier_bwd_input_bwd_kernel(const float *__restrict__ x, int64_t n,
                                                                         const float *__restrict__ Bf, int L,
                                                                         const float *__restrict__ d_row, int64_t ld,
                                                                         const float *__restrict__ gg,
                                                                         float *__restrict__ d_x,
                                                                         float *__restrict__ dd_row, int64_t ld2,
                                                                         int width) {
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c0 = (int)(gid % kFLanes);
    const bool live = gid / kFLanes < n;
    const int64_t i = live ? gid / kFLanes : n - 1;
    const float two_pi = 6.283185307179586f;
    const float s0 = __fmul_rn(two_pi, x[i * 3]), s1 = __fmul_rn(two_pi, x[i * 3 + 1]), s2 = __fmul_rn(two_pi, x[i * 3 + 2]);
    const float *d = d_row + i * ld;
    const float q0 = gg[i * 3], q1 = gg[i * 3 + 1], q2 = gg[i * 3 + 2];
    float *o = (dd_row && live) ? dd_row + i * ld2 : nullptr;
    if (o) {
        if (c0 == 0) { o[0] = q0; o[1] = q1; o[2] = q2; }
        for (int k = 3 + 2 * L + c0; k < width; k += kFLanes) o[k] = 0.0f;
    }
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    for (int c = c0; c < L; c += kFLanes) {
        float sn, cs;
        sincosf(fourier_arg(Bf, L, c, s0, s1, s2), &sn, &cs);
        const float b0 = Bf[c], b1 = Bf[L + c], b2 = Bf[2 * L + c];
        const float pc = b0 * q0 + b1 * q1 + b2 * q2;
        if (o) {
            o[3 + c] = two_pi * pc * cs;
            o[3 + L + c] = -(two_pi * pc * sn);
        }
        const float r = pc * (sn * d[3 + c] + cs * d[3 + L + c]);
        g0 += r * b0; g1 += r * b1; g2 += r * b2;
    }
    g0 = sum16(g0); g1 = sum16(g1); g2 = sum16(g2);
    if (d_x && live && c0 == 0) {
        const float k2 = -(two_pi * two_pi);
        d_x[i * 3] = k2 * g0; d_x[i * 3 + 1] = k2 * g1; d_x[i * 3 + 2] = k2 * g2;
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

bool grid_ok(int64_t threads, int64_t &grid) {
    grid = (threads + kThreads - 1) / kThreads;
    return grid <= 0x7fffffffLL;
}

}  // namespace

extern "C" {

int hm_encode_bwd_input(const hm_grid_desc *desc, const float *x, int64_t n, const float *table, const float *d_feat,
                        int64_t d_feat_stride, const float *a, float *gx, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_bwd_input: desc is NULL");
    HM_CHECK_ARG(n >= 0, "hm_encode_bwd_input: n < 0");
    HM_CHECK_ARG(d_feat_stride >= desc->lv.L * desc->lv.F, "hm_encode_bwd_input: d_feat_stride < L*F");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && table && d_feat && gx, "hm_encode_bwd_input: NULL pointer");
    int64_t grid;
    HM_CHECK_ARG(grid_ok(n * 8, grid), "hm_encode_bwd_input: n too large for one launch");
    if (a)
        hipLaunchKernelGGL(encode_bwd_input_kernel<true>, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream),
                           desc->lv, x, n, table, d_feat, d_feat_stride, a, gx);
    else
        hipLaunchKernelGGL(encode_bwd_input_kernel<false>, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream),
                           desc->lv, x, n, table, d_feat, d_feat_stride, a, gx);
    HM_CHECK_LAUNCH("hm_encode_bwd_input");
    return HM_OK;
}

int hm_encode_jvp(const hm_grid_desc *desc, const float *x, int64_t n, const float *table, const float *a, float *out,
                  int64_t out_stride, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_jvp: desc is NULL");
    HM_CHECK_ARG(n >= 0, "hm_encode_jvp: n < 0");
    HM_CHECK_ARG(out_stride >= desc->lv.L * desc->lv.F, "hm_encode_jvp: out_stride < L*F");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && table && a && out, "hm_encode_jvp: NULL pointer");
    int64_t grid;
    HM_CHECK_ARG(grid_ok(n * desc->lv.L * 8, grid), "hm_encode_jvp: n too large for one launch");
    hipLaunchKernelGGL(encode_jvp_kernel, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream), desc->lv, x, n,
                       table, a, out, out_stride);
    HM_CHECK_LAUNCH("hm_encode_jvp");
    return HM_OK;
}

int hm_encode_bwd_table_jvp(const hm_grid_desc *desc, const float *x, int64_t n, const float *a, const float *d_feat,
                            int64_t d_feat_stride, float *d_table, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_bwd_table_jvp: desc is NULL");
    HM_CHECK_ARG(n >= 0, "hm_encode_bwd_table_jvp: n < 0");
    HM_CHECK_ARG(d_feat_stride >= desc->lv.L * desc->lv.F, "hm_encode_bwd_table_jvp: d_feat_stride < L*F");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && a && d_feat && d_table, "hm_encode_bwd_table_jvp: NULL pointer");
    int64_t grid;
    HM_CHECK_ARG(grid_ok(n * desc->lv.L * 8, grid), "hm_encode_bwd_table_jvp: n too large for one launch");
    hipLaunchKernelGGL(encode_bwd_table_jvp_kernel, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream),
                       desc->lv, x, n, a, d_feat, d_feat_stride, d_table);
    HM_CHECK_LAUNCH("hm_encode_bwd_table_jvp");
    return HM_OK;
}

int hm_fourier_bwd_input(const float *x, int64_t n, const float *B_fourier, int n_channels, const float *d_row,
                         int64_t d_row_stride, float *gx, void *stream) {
    HM_CHECK_ARG(n >= 0 && n_channels >= 1 && n_channels <= HM_MAX_LEVELS, "hm_fourier_bwd_input: bad size");
    HM_CHECK_ARG(d_row_stride >= 3 + 2 * n_channels, "hm_fourier_bwd_input: d_row_stride < 3 + 2 n_channels");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && B_fourier && d_row && gx, "hm_fourier_bwd_input: NULL pointer");
    int64_t grid;
    HM_CHECK_ARG(grid_ok(n * kFLanes, grid), "hm_fourier_bwd_input: n too large for one launch");
    hipLaunchKernelGGL(fourier_bwd_input_kernel, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream), x, n,
                       B_fourier, n_channels, d_row, d_row_stride, gx);
    HM_CHECK_LAUNCH("hm_fourier_bwd_input");
    return HM_OK;
}

int hm_fourier_bwd_input_bwd(const float *x, int64_t n, const float *B_fourier, int n_channels, const float *d_row,
                             int64_t d_row_stride, const float *gg, float *d_x, float *dd_row, int64_t dd_row_stride,
                             int width, void *stream) {
    HM_CHECK_ARG(n >= 0 && n_channels >= 1 && n_channels <= HM_MAX_LEVELS, "hm_fourier_bwd_input_bwd: bad size");
    HM_CHECK_ARG(d_row_stride >= 3 + 2 * n_channels, "hm_fourier_bwd_input_bwd: d_row_stride < 3 + 2 n_channels");
    HM_CHECK_ARG(!dd_row || (width >= 3 + 2 * n_channels && dd_row_stride >= width), "hm_fourier_bwd_input_bwd: dd_row shape");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && B_fourier && d_row && gg && (d_x || dd_row), "hm_fourier_bwd_input_bwd: NULL pointer");
    int64_t grid;
    HM_CHECK_ARG(grid_ok(n * kFLanes, grid), "hm_fourier_bwd_input_bwd: n too large for one launch");
    hipLaunchKernelGGL(fourier_bwd_input_bwd_kernel, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream), x, n,
                       B_fourier, n_channels, d_row, d_row_stride, gg, d_x, dd_row, dd_row_stride, width);
    HM_CHECK_LAUNCH("hm_fourier_bwd_input_bwd");
    return HM_OK;
}

}  // extern "C"
