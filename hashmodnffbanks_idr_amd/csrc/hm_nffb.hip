// hm_nffb.hip - fused forward of the Fourier-filter-bank embedders ('FFB', 'StyleModNFFB'; BASELINE configs 3 / 5).
//
// Replaces FourierFilterBanks.forward evaluated without gradient
// (reference: model/embeddings/nffb3d.py:122-194 with frequency_enc.py:6-51, Sine.py:5-25 and
// style_Attention/styleMod.py:16-43): hash-grid encode of u = (x + bound) / (2 bound), the grid row cut into chunks
// of 2F = 4 values, NeRF positional encoding of chunk l-1 (identity kept twice, L octave frequencies), the SIREN trunk
// ff_lin0 .. ff_lin{L-2} of width W = 8 + 8L on x / bound, per layer l >= 1
//     e = posenc(chunk_{l-1}) [-> StyleAttention: per-row instance norm of linear_transform(.)] + trunk_l,
//     features += out_layer(e),
// output [u | features / L].  In torch this is ~12 GEMMs of width 56 / 72 plus ~40 elementwise kernels per call.
//
// Mapping (gfx950): ONE THREAD PER POINT.  The layers are far too narrow for a matrix tile to pay (56 / 72 columns,
// a sin or a per-row normalisation after every product), and on gfx950 the fp32 MFMA rate EQUALS the fp32 VALU rate
// (157 TFLOP/s both), so the exact-fp32 product is written for the VALU with the weights as SCALAR operands: the loop
// over output features is wave-uniform, each weight row arrives through the scalar cache (s_load_dwordx8/16) and
// every v_fmac_f32 takes its weight from an SGPR - no LDS operand traffic, no weight registers.  A thread keeps its
// W activations in registers (static indices); results whose index is the loop counter go through a private LDS
// column [W][threads] (conflict-free: lane = bank).  k-ordered fma chains, -ffp-contract=off file: the arithmetic is
// the same dot-product order torch's sgemm uses up to its blocking (parity tolerances in the tests).
#include "hm_common.h"

#include <math.h>

namespace {

constexpr int kNT = 128;   // threads (= points) per workgroup: 2 x [72][128] floats of LDS = 73.7 KB -> two per CU

struct NffbArgs {   // by value
    const float *trunk_w[HM_MAX_LEVELS];   // ff_lin0 [W,3], ff_lin1.. [W,W]
    const float *trunk_b[HM_MAX_LEVELS];
    const float *out_w, *out_b;             // out_layer [W,W], [W]
    const float *style_w, *style_b;         // StyleAttention.linear_transform [W,W], [W] (NULL: plain FFB)
    float bound, w0, style_eps;
};

template <int FRAC>
__device__ __forceinline__ void nffb_corner(float x, int32_t res, int bit, uint32_t &u, float &w) {
    const float xs = __fmul_rn(x, (float)res);
    if (FRAC == HM_FRAC_REFERENCE) {
        u = (uint32_t)((int32_t)xs) + (uint32_t)bit;
        w = bit ? 0.0f : 1.0f;
    } else {
        const float fl = floorf(xs);
        const float xf = __fsub_rn(xs, fl);
        u = (uint32_t)((int32_t)fl) + (uint32_t)bit;
        w = bit ? xf : __fsub_rn(1.0f, xf);
    }
}

// y[j] = b[j] + sum_k Wm[j][k] * v[k], j = 0..W-1 (wave-uniform loop: weights are scalar operands), written to this
// thread's LDS column col[j * kNT]
template <int W, int K>
__device__ __forceinline__ void matvec_to_lds(const float *__restrict__ Wm, const float *__restrict__ b,
                                              const float (&v)[K], float *col) {
    for (int j = 0; j < W; ++j) {
        const float *row = Wm + j * K;
        float acc = __fmul_rn(row[0], v[0]);
#pragma unroll
        for (int k = 1; k < K; ++k) acc = __fmaf_rn(row[k], v[k], acc);
        col[j * kNT] = __fadd_rn(acc, b[j]);
    }
}

template <int FRAC, int LV, bool STYLE>
__global__ __launch_bounds__(kNT) void nffb_fwd_kernel(HmLevels lv, NffbArgs a, const float *__restrict__ x, int64_t n,
                                                       const float *__restrict__ table,
                                                       const float *__restrict__ Bf, float *__restrict__ out,
                                                       int64_t out_stride, const int32_t *__restrict__ n_dev) {
    constexpr int W = 8 + 8 * LV;
    constexpr int NG = 4 * (LV - 2);   // grid values that are ever consumed: chunks 0 .. LV-3
    extern __shared__ __align__(16) float nffb_lds[];
    float *T = nffb_lds;               // [W][kNT] dynamic-index results of the current product
    float *FE = nffb_lds + W * kNT;    // [W][kNT] feature accumulator
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    const int tid = threadIdx.x;
    float *tcol = T + tid, *fcol = FE + tid;
    for (int64_t base = (int64_t)blockIdx.x * kNT; base < n; base += (int64_t)gridDim.x * kNT) {
        const int64_t i = base + tid;
        const bool live = i < n;
        const float p0 = live ? x[i * 3] : 0.0f, p1 = live ? x[i * 3 + 1] : 0.0f, p2 = live ? x[i * 3 + 2] : 0.0f;
        // trunk input and grid input (nffb3d.py:131-132)
        float xn[3] = {__fdiv_rn(p0, a.bound), __fdiv_rn(p1, a.bound), __fdiv_rn(p2, a.bound)};
        const float two_b = __fmul_rn(2.0f, a.bound);
        const float u0 = __fdiv_rn(__fadd_rn(p0, a.bound), two_b), u1 = __fdiv_rn(__fadd_rn(p1, a.bound), two_b),
                    u2 = __fdiv_rn(__fadd_rn(p2, a.bound), two_b);
        // ---- grid row without its 3 pass-through columns: [sin(L) | cos(L) | level features], first NG values ----
        float g[NG];
        {
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, u0), s1 = __fmul_rn(two_pi, u1), s2 = __fmul_rn(two_pi, u2);
#pragma unroll
            for (int c = 0; c < LV; ++c) {
                float ang = __fmul_rn(s0, Bf[c]);
                ang = __fmaf_rn(s1, Bf[LV + c], ang);
                ang = __fmaf_rn(s2, Bf[2 * LV + c], ang);
                float sn, cs;
                sincosf(ang, &sn, &cs);
                if (c < NG) g[c] = sn;
                if (LV + c < NG) g[LV + c] = cs;
            }
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                if (2 * LV + 2 * l >= NG) continue;      // levels beyond L-5 never reach the output
                float acc0 = 0.0f, acc1 = 0.0f;
                const float2 *tl = reinterpret_cast<const float2 *>(table) + lv.row_off[l];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    uint32_t ux, uy, uz;
                    float wx, wy, wz;
                    nffb_corner<FRAC>(u0, lv.res[l], c & 1, ux, wx);
                    nffb_corner<FRAC>(u1, lv.res[l], (c >> 1) & 1, uy, wy);
                    nffb_corner<FRAC>(u2, lv.res[l], (c >> 2) & 1, uz, wz);
                    const float w = __fmul_rn(__fmul_rn(wx, wy), wz);
                    if (w != 0.0f) {
                        const float2 r = tl[hm_mod_rows(hm_hash3(ux, uy, uz), lv.rows[l], lv.magic[l])];
                        acc0 = __fadd_rn(acc0, __fmul_rn(r.x, w));
                        acc1 = __fadd_rn(acc1, __fmul_rn(r.y, w));
                    }
                }
                g[2 * LV + 2 * l] = acc0;
                if (2 * LV + 2 * l + 1 < NG) g[2 * LV + 2 * l + 1] = acc1;
            }
        }
        // ---- trunk layer 0: 3 -> W, sin(w0 .) ------------------------------------------------------------------
        float xv[W];
        matvec_to_lds<W, 3>(a.trunk_w[0], a.trunk_b[0], xn, tcol);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            xv[k] = sinf(__fmul_rn(tcol[k * kNT], a.w0));
            fcol[k * kNT] = 0.0f;
        }
        // ---- layers 1 .. LV-2 --------------------------------------------------------------------------------------
#pragma unroll 1
        for (int layer = 1; layer < LV - 1; ++layer) {
            matvec_to_lds<W, W>(a.trunk_w[layer], a.trunk_b[layer], xv, tcol);
#pragma unroll
            for (int k = 0; k < W; ++k) xv[k] = sinf(__fmul_rn(tcol[k * kNT], a.w0));
            // positional encoding of chunk layer-1: [c, c, sin(c f0), cos(c f0), sin(c f1), ...], f_m = 2^m
            float e[W];
            {
                float c4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // g[] is register-resident: select with a static unrolled scan
                    float v = 0.0f;
#pragma unroll
                    for (int m = 0; m < NG; ++m) v = (m == 4 * (layer - 1) + r) ? g[m] : v;
                    c4[r] = v;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = c4[r];
                    e[4 + r] = c4[r];
                }
#pragma unroll
                for (int m = 0; m < LV; ++m) {
                    const float f = (float)(1 << m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sn, cs;
                        sincosf(__fmul_rn(c4[r], f), &sn, &cs);
                        e[8 + 8 * m + r] = sn;
                        e[8 + 8 * m + 4 + r] = cs;
                    }
                }
            }
            if (STYLE) {
                // StyleAttention: linear_transform(e) * softmax over a size-1 dim (== 1), then the per-row
                // InstanceNorm over the W features (biased variance), styleMod.py:30-43
                matvec_to_lds<W, W>(a.style_w, a.style_b, e, tcol);
                float mean = 0.0f;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    e[k] = tcol[k * kNT];
                    mean += e[k];
                }
                mean = mean / (float)W;
                float var = 0.0f;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const float d = e[k] - mean;
                    var += d * d;
                }
                var = var / (float)W;
                const float den = sqrtf(var + a.style_eps);
#pragma unroll
                for (int k = 0; k < W; ++k) e[k] = (e[k] - mean) / den;
            }
#pragma unroll
            for (int k = 0; k < W; ++k) e[k] = __fadd_rn(e[k], xv[k]);
            matvec_to_lds<W, W>(a.out_w, a.out_b, e, tcol);
#pragma unroll
            for (int k = 0; k < W; ++k) fcol[k * kNT] = __fadd_rn(fcol[k * kNT], tcol[k * kNT]);
        }
        if (live) {
            float *o = out + i * out_stride;
            o[0] = u0; o[1] = u1; o[2] = u2;
#pragma unroll
            for (int k = 0; k < W; ++k) o[3 + k] = __fdiv_rn(fcol[k * kNT], (float)LV);
        }
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <int FRAC, int LV, bool STYLE>
int launch_nffb1(unsigned grid, hipStream_t st, const HmLevels &lv, const NffbArgs &a, const float *x, int64_t n,
                 const float *table, const float *Bf, float *out, int64_t out_stride, const int32_t *n_dev) {
    const size_t lds = sizeof(float) * 2 * (8 + 8 * LV) * kNT;
    static thread_local bool attr_done = false;   // (one flag per template instance)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(nffb_fwd_kernel<FRAC, LV, STYLE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((nffb_fwd_kernel<FRAC, LV, STYLE>), dim3(grid), dim3(kNT), lds, st, lv, a, x, n, table, Bf, out,
                       out_stride, n_dev);
    return HM_OK;
}

template <int FRAC, int LV>
int launch_nffb(bool style, unsigned grid, hipStream_t st, const HmLevels &lv, const NffbArgs &a, const float *x,
                int64_t n, const float *table, const float *Bf, float *out, int64_t out_stride, const int32_t *n_dev) {
    return style ? launch_nffb1<FRAC, LV, true>(grid, st, lv, a, x, n, table, Bf, out, out_stride, n_dev)
                 : launch_nffb1<FRAC, LV, false>(grid, st, lv, a, x, n, table, Bf, out, out_stride, n_dev);
}

}  // namespace

extern "C" {

int hm_nffb_fwd(const hm_grid_desc *desc, const hm_nffb_desc *nf, const float *x, int64_t n, const float *table,
                const float *B_fourier, float *out, int64_t out_stride, int frac_mode, const int32_t *n_dev,
                void *stream) {
    HM_CHECK_ARG(desc && nf, "hm_nffb_fwd: NULL descriptor");
    const HmLevels &lv = desc->lv;
    HM_CHECK_ARG(lv.F == 2, "hm_nffb_fwd: the filter-bank embedders use F = 2 features per level");
    HM_CHECK_ARG(lv.L == 6 || lv.L == 8, "hm_nffb_fwd: built for L = 6 and L = 8 levels (the shipped configurations)");
    HM_CHECK_ARG(nf->n_levels == lv.L, "hm_nffb_fwd: descriptor / grid level count mismatch");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_nffb_fwd: bad frac_mode");
    HM_CHECK_ARG(n >= 0 && out_stride >= 3 + 8 + 8 * lv.L, "hm_nffb_fwd: bad n / out_stride");
    HM_CHECK_ARG(nf->bound > 0.0f, "hm_nffb_fwd: bound must be positive");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && table && B_fourier && out && nf->out_w && nf->out_b, "hm_nffb_fwd: NULL pointer");
    NffbArgs a;
    for (int l = 0; l < lv.L - 1; ++l) {
        HM_CHECK_ARG(nf->trunk_w[l] && nf->trunk_b[l], "hm_nffb_fwd: NULL trunk layer");
        a.trunk_w[l] = nf->trunk_w[l];
        a.trunk_b[l] = nf->trunk_b[l];
    }
    for (int l = lv.L - 1; l < HM_MAX_LEVELS; ++l) a.trunk_w[l] = a.trunk_b[l] = nullptr;
    a.out_w = nf->out_w; a.out_b = nf->out_b;
    a.style_w = nf->style_w; a.style_b = nf->style_b;
    HM_CHECK_ARG((nf->style_w == nullptr) == (nf->style_b == nullptr), "hm_nffb_fwd: style weight / bias must come together");
    a.bound = nf->bound; a.w0 = nf->w0; a.style_eps = nf->style_eps;
    const bool style = nf->style_w != nullptr;
    const int64_t blocks = (n + kNT - 1) / kNT;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    hipStream_t st = as_stream(stream);
    int rc;
#define HM_NFFB(FR)                                                                                                    \
    rc = (lv.L == 6) ? launch_nffb<FR, 6>(style, grid, st, lv, a, x, n, table, B_fourier, out, out_stride, n_dev)      \
                     : launch_nffb<FR, 8>(style, grid, st, lv, a, x, n, table, B_fourier, out, out_stride, n_dev)
    if (frac_mode == HM_FRAC_REFERENCE) { HM_NFFB(HM_FRAC_REFERENCE); } else { HM_NFFB(HM_FRAC_TRILINEAR); }
#undef HM_NFFB
    if (rc != HM_OK) return rc;
    HM_CHECK_LAUNCH("hm_nffb_fwd");
    return HM_OK;
}

}  // extern "C"
