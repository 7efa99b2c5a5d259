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
// Mapping (gfx950): EIGHT LANES PER POINT on the VALU.  The layers are far too narrow for a matrix tile to pay
// (56 / 72 columns, a sin or a per-row normalisation after every product), and on gfx950 the fp32 MFMA rate EQUALS
// the fp32 VALU rate (157 TFLOP/s both), so the exact-fp32 products run as k-ordered fma chains on the VALU.  Lane s
// of a point owns rows s, s+8, ... (7 or 9 of them) of every product; its weight rows arrive as 16-byte vector loads
// from the CU's L1 (the matrices are 12 - 21 KB), the input vector sits replicated in the point's lanes' registers
// (static indices), and the outputs are exchanged through a per-point LDS row.  Splitting a point over 8 (or 32) lanes is
// what makes the ~50 small calls of a ray search cheap: the first version (one thread per point, weights as scalar
// operands) had a 120 us serial chain per call however few points it held; here the chain is 8x shorter and a
// 4096-point round fills every CU.  -ffp-contract=off file: same dot-product order as torch's sgemm up to blocking.
#include "hm_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kNT = 256;   // threads per workgroup
// LP = lanes per point (template value): each lane owns ceil(W / LP) of every layer's outputs (W = 56 / 72).
//   LP = 8  (32 points per workgroup): big launches, throughput
//   LP = 32 (8 points per workgroup): the tracer's march / secant rounds (<= 8192 live points).  A call is one serial
//           chain per point - ~14 matrix-vector products whose weight rows are loaded through the L1 - and took ~92 us
//           whatever its size; four times fewer rows per lane cut that chain, and every row is still one lane's k-ordered
//           fma chain, so the values do not depend on LP.
constexpr int kSmallCount = 8192;

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

// Lane `sub` of a point computes rows sub, sub + 8, ... of y = Wm v + b (k-ordered fma chain per row, the weights
// read as 16-byte vector loads: the matrices are 12 - 21 KB and stay in the CU's L1) and leaves them in the point's LDS
// row; the barrier-separated read-back gives every lane of the point the whole vector again.
// SINE: the Sine activation sin(w0 .) is applied by the lane that owns the row, BEFORE the exchange (applied after the
// read-back every one of the point's lanes would evaluate all W sines again).
template <int W, int K, bool SINE, int LP>
__device__ __forceinline__ void matvec_rows(const float *__restrict__ Wm, const float *__restrict__ b,
                                            const float (&v)[K], int sub, float *prow, float w0 = 0.0f) {
    constexpr int R = (W + LP - 1) / LP;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int jrow = sub + LP * r;
        if (W % LP != 0 && jrow >= W) break;      // (LP = 32: the last slice of rows is partial)
        const float *row = Wm + jrow * K;
        float acc;
        if (K % 4 == 0) {
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            float4 w4 = row4[0];
            acc = __fmul_rn(w4.x, v[0]);
            acc = __fmaf_rn(w4.y, v[1], acc);
            acc = __fmaf_rn(w4.z, v[2], acc);
            acc = __fmaf_rn(w4.w, v[3], acc);
#pragma unroll
            for (int k4 = 1; k4 < K / 4; ++k4) {
                w4 = row4[k4];
                acc = __fmaf_rn(w4.x, v[4 * k4], acc);
                acc = __fmaf_rn(w4.y, v[4 * k4 + 1], acc);
                acc = __fmaf_rn(w4.z, v[4 * k4 + 2], acc);
                acc = __fmaf_rn(w4.w, v[4 * k4 + 3], acc);
            }
        } else {
            acc = __fmul_rn(row[0], v[0]);
#pragma unroll
            for (int k = 1; k < K; ++k) acc = __fmaf_rn(row[k], v[k], acc);
        }
        const float y = __fadd_rn(acc, b[jrow]);
        prow[jrow] = SINE ? sinf(__fmul_rn(y, w0)) : y;
    }
}

template <int FRAC, int LV, bool STYLE, int LP>
__global__ __launch_bounds__(kNT) void nffb_fwd_kernel(HmLevels lv, NffbArgs a, const float *__restrict__ x, int64_t n,
                                                       const float *__restrict__ table,
                                                       const float *__restrict__ Bf, float *__restrict__ out,
                                                       int64_t out_stride, const int32_t *__restrict__ n_dev,
                                                       int64_t run_min, int64_t run_max) {
    constexpr int W = 8 + 8 * LV;
    constexpr int kLP = LP, kPP = kNT / LP;
    constexpr int R = (W + kLP - 1) / kLP;
    constexpr int NG = 4 * (LV - 2);   // grid values that are ever consumed: chunks 0 .. LV-3
    constexpr int WP = W + 4;          // padded LDS row (keeps 16-byte alignment, staggers the banks of the 32 rows)
    __shared__ __align__(16) float T[kPP * WP];    // per point: the vector being exchanged between its lanes
    __shared__ float G[kPP * NG];                  // per point: the consumed part of the grid row
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;      // the other lanes-per-point variant owns this batch size
    const int tid = threadIdx.x;
    const int pt = tid / kLP, sub = tid % kLP;
    float *prow = T + pt * WP;
    float *grow = G + pt * NG;
    for (int64_t base = (int64_t)blockIdx.x * kPP; base < n; base += (int64_t)gridDim.x * kPP) {
        const int64_t i = base + pt;
        const bool live = i < n;
        const float p0 = live ? x[i * 3] : 0.0f, p1 = live ? x[i * 3 + 1] : 0.0f, p2 = live ? x[i * 3 + 2] : 0.0f;
        // trunk input and grid input (nffb3d.py:131-132)
        float xn[3] = {__fdiv_rn(p0, a.bound), __fdiv_rn(p1, a.bound), __fdiv_rn(p2, a.bound)};
        const float two_b = __fmul_rn(2.0f, a.bound);
        const float u0 = __fdiv_rn(__fadd_rn(p0, a.bound), two_b), u1 = __fdiv_rn(__fadd_rn(p1, a.bound), two_b),
                    u2 = __fdiv_rn(__fadd_rn(p2, a.bound), two_b);
        __syncthreads();   // the previous tile has left T / G
        // ---- grid row without its 3 pass-through columns: [sin(L) | cos(L) | level features]; the lanes of a
        //      point share the 2L sin/cos channels and the (<= L-4) consumed levels ------------------------------
        {
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, u0), s1 = __fmul_rn(two_pi, u1), s2 = __fmul_rn(two_pi, u2);
            for (int c = sub; c < LV; c += kLP) {
                float ang = __fmul_rn(s0, Bf[c]);
                ang = __fmaf_rn(s1, Bf[LV + c], ang);
                ang = __fmaf_rn(s2, Bf[2 * LV + c], ang);
                float sn, cs;
                sincosf(ang, &sn, &cs);
                if (c < NG) grow[c] = sn;
                if (LV + c < NG) grow[LV + c] = cs;
            }
            for (int l = sub; l < LV; l += kLP) {
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
                grow[2 * LV + 2 * l] = acc0;
                if (2 * LV + 2 * l + 1 < NG) grow[2 * LV + 2 * l + 1] = acc1;
            }
        }
        // ---- trunk layer 0: 3 -> W, sin(w0 .) ------------------------------------------------------------------
        float xv[W];
        float feat[R];     // this lane's rows of the feature accumulator
#pragma unroll
        for (int r = 0; r < R; ++r) feat[r] = 0.0f;
        matvec_rows<W, 3, true, LP>(a.trunk_w[0], a.trunk_b[0], xn, sub, prow, a.w0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < W; ++k) xv[k] = prow[k];
        // ---- layers 1 .. LV-2 --------------------------------------------------------------------------------------
#pragma unroll 1
        for (int layer = 1; layer < LV - 1; ++layer) {
            __syncthreads();
            matvec_rows<W, W, true, LP>(a.trunk_w[layer], a.trunk_b[layer], xv, sub, prow, a.w0);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < W; ++k) xv[k] = prow[k];
            // positional encoding of chunk layer-1: [c, c, sin(c f0), cos(c f0), sin(c f1), ...], f_m = 2^m;
            // lane `sub` computes entries sub, sub + 8, ...: entry k >= 8 is sin (k % 8 < 4) or cos of c[k % 4] * 2^((k-8)/8)
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = sub + kLP * r;
                if (W % kLP != 0 && k >= W) break;
                const float c = grow[4 * (layer - 1) + (k & 3)];
                float v = c;
                if (k >= 8) {
                    const float arg = __fmul_rn(c, (float)(1 << ((k - 8) >> 3)));
                    v = (k & 4) ? cosf(arg) : sinf(arg);
                }
                prow[k] = v;
            }
            __syncthreads();
            float e[W];
#pragma unroll
            for (int k = 0; k < W; ++k) e[k] = prow[k];
            if (STYLE) {
                // StyleAttention: linear_transform(e) * softmax over a size-1 dim (== 1), then the per-row
                // InstanceNorm over the W features (biased variance), styleMod.py:30-43
                __syncthreads();
                matvec_rows<W, W, false, LP>(a.style_w, a.style_b, e, sub, prow);
                __syncthreads();
                float mean = 0.0f;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    e[k] = prow[k];
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
            __syncthreads();
            matvec_rows<W, W, false, LP>(a.out_w, a.out_b, e, sub, prow);
            // (each lane reads back only the rows it wrote: no barrier needed)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (W % kLP == 0 || sub + kLP * r < W) feat[r] = __fadd_rn(feat[r], prow[sub + kLP * r]);
        }
        if (live) {
            float *o = out + i * out_stride;
            if (sub < 3) o[sub] = sub == 0 ? u0 : (sub == 1 ? u1 : u2);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (W % kLP == 0 || sub + kLP * r < W) o[3 + sub + kLP * r] = __fdiv_rn(feat[r], (float)LV);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Big launches (the tracer's coarse scans: 200 - 400 k points): the SAME embedder tiled on the matrix cores.
// In the lanes-per-point kernel above every POINT re-reads every weight matrix through the L1 (12.5 - 21 KB per
// product, ~10 products: 41 GB of L1 traffic for the 410 k coarse points of config 3 -> 2.9 ms, bound by the L1, at
// 4 % of the VALU peak).  Here a WAVE owns 16 points for the whole embedder and every product is
//     Y[rows, 16 points] = Wm[rows, k] * V[k, 16 points]      on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains),
// so a weight matrix is read once per 16 points: A fragments are 16-byte loads of Wm[16t + (lane & 15)][16kb + 4(lane >> 4)
// .. + 3] straight from the L1 (rows / k beyond W read as zero), B fragments are ds_read_b128 of the wave's activation
// image V[k/4][point][4].  The accumulator of lane (point j, quarter q) holds rows 16t + 4q .. + 3 of tile t - exactly
// one 16-byte k-group of the next product's V image (the identity the 16-point SDF body uses), so Sine, the positional
// encoding, StyleAttention's per-row normalisation (two lane exchanges over the point's four lanes) and the feature
// accumulation all run on the accumulator layout.  Waves share nothing: no workgroup barrier anywhere.
typedef float nf_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMfmaWaves = 4;      // waves per workgroup (16 points each) of the big-batch launch; 1 for the tracer's rounds

template <int W, int NT>
__device__ __forceinline__ void nffb_matvec_mfma(const float *__restrict__ Wm, const float *src, nf_f32x4 (&acc)[NT],
                                                 int j, int q) {
    constexpr int KB = NT;                        // 16-wide k blocks of the padded K = 16 NT
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = nf_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int k0 = 16 * kb + 4 * q;
        const float4 b = *reinterpret_cast<const float4 *>(src + ((4 * kb + q) * 16 + j) * 4);
        float4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = 16 * t + j;
            const bool ok = row < W && k0 < W;    // (W % 4 == 0: a k-group is wholly inside or wholly outside)
            const float4 v = *reinterpret_cast<const float4 *>(Wm + (ok ? row * W + k0 : 0));
            w[t] = ok ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].w, b.w, acc[t], 0, 0, 0);
    }
}

template <int FRAC, int LV, bool STYLE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void nffb_fwd_mfma_kernel(HmLevels lv, NffbArgs a, const float *__restrict__ x,
                                                                        int64_t n, const float *__restrict__ table,
                                                                        const float *__restrict__ Bf,
                                                                        float *__restrict__ out, int64_t out_stride,
                                                                        const int32_t *__restrict__ n_dev, int64_t run_min,
                                                                        int64_t run_max) {
    constexpr int W = 8 + 8 * LV;
    constexpr int NT = (W + 15) / 16, WP = 16 * NT;     // 16-row tiles / padded width (64 at W = 56, 80 at W = 72)
    constexpr int NG = 4 * (LV - 2);                    // grid values that are ever consumed: chunks 0 .. LV-3
    constexpr int kImg = WP * 16;                       // floats of one activation image [WP/4][16][4]
    __shared__ __align__(16) float lds[WAVES * (2 * kImg + 16 * NG)];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    float *XV = lds + wave * (2 * kImg + 16 * NG);      // trunk state
    float *EV = XV + kImg;                              // positional encoding / style / sum image
    float *G = EV + kImg;                               // [16][NG] consumed grid values of the wave's points
    // the padded k-groups (rows W .. WP-1) are multiplied by zero weights: they must hold finite numbers
    for (int i = lane; i < 2 * kImg; i += 64) XV[i] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    const int64_t n_tiles = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t i = tile * 16 + j;
        const bool live = i < n;
        const float p0 = live ? x[i * 3] : 0.0f, p1 = live ? x[i * 3 + 1] : 0.0f, p2 = live ? x[i * 3 + 2] : 0.0f;
        const float xn[3] = {__fdiv_rn(p0, a.bound), __fdiv_rn(p1, a.bound), __fdiv_rn(p2, a.bound)};
        const float two_b = __fmul_rn(2.0f, a.bound);
        const float u0 = __fdiv_rn(__fadd_rn(p0, a.bound), two_b), u1 = __fdiv_rn(__fadd_rn(p1, a.bound), two_b),
                    u2 = __fdiv_rn(__fadd_rn(p2, a.bound), two_b);
        // ---- consumed part of the grid row [sin(L) | cos(L) | level features]: the point's four lanes share it ----
        {
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, u0), s1 = __fmul_rn(two_pi, u1), s2 = __fmul_rn(two_pi, u2);
            for (int c = q; c < NG; c += 4) {
                float v;
                if (c < 2 * LV) {
                    const int ch = c < LV ? c : c - LV;
                    float ang = __fmul_rn(s0, Bf[ch]);
                    ang = __fmaf_rn(s1, Bf[LV + ch], ang);
                    ang = __fmaf_rn(s2, Bf[2 * LV + ch], ang);
                    float sn, cs;
                    sincosf(ang, &sn, &cs);
                    v = c < LV ? sn : cs;
                } else {
                    const int l = (c - 2 * LV) >> 1, f = (c - 2 * LV) & 1;
                    float acc = 0.0f;
                    const float2 *tl = reinterpret_cast<const float2 *>(table) + lv.row_off[l];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) {
                        uint32_t ux, uy, uz;
                        float wx, wy, wz;
                        nffb_corner<FRAC>(u0, lv.res[l], cc & 1, ux, wx);
                        nffb_corner<FRAC>(u1, lv.res[l], (cc >> 1) & 1, uy, wy);
                        nffb_corner<FRAC>(u2, lv.res[l], (cc >> 2) & 1, uz, wz);
                        const float w = __fmul_rn(__fmul_rn(wx, wy), wz);
                        if (w != 0.0f) {
                            const float2 r = tl[hm_mod_rows(hm_hash3(ux, uy, uz), lv.rows[l], lv.magic[l])];
                            acc = __fadd_rn(acc, __fmul_rn(f ? r.y : r.x, w));
                        }
                    }
                    v = acc;
                }
                G[j * NG + c] = v;
            }
        }
        // ---- trunk layer 0: 3 -> W, sin(w0 .): the lane's rows 4g .. 4g+3 for the groups g = q, q+4, ... ---------------
        for (int g = q; g < WP / 4; g += 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * g + e;
                float y = 0.0f;
                if (row < W) {
                    const float *wr = a.trunk_w[0] + row * 3;
                    float acc = __fmul_rn(wr[0], xn[0]);
                    acc = __fmaf_rn(wr[1], xn[1], acc);
                    acc = __fmaf_rn(wr[2], xn[2], acc);
                    y = sinf(__fmul_rn(__fadd_rn(acc, a.trunk_b[0][row]), a.w0));
                }
                v[e] = y;
            }
            *reinterpret_cast<float4 *>(XV + (g * 16 + j) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_wave_barrier();
        nf_f32x4 feat[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) feat[t] = nf_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int layer = 1; layer < LV - 1; ++layer) {
            nf_f32x4 acc[NT];
            // trunk: xv <- sin(w0 (W_l xv + b_l)), in place (every read of the old image precedes the writes)
            nffb_matvec_mfma<W, NT>(a.trunk_w[layer], XV, acc, j, q);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * t + 4 * q + e;
                    v[e] = row < W ? sinf(__fmul_rn(__fadd_rn(acc[t][e], a.trunk_b[layer][row]), a.w0)) : 0.0f;
                }
                *reinterpret_cast<float4 *>(XV + ((4 * t + q) * 16 + j) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
            // positional encoding of chunk layer-1 into EV: group g holds entries 4g .. 4g+3 = f(c[0..3] * 2^m),
            // f = identity (g < 2), sin (g even) or cos (g odd), m = (g - 2) / 2     (frequency_enc.py:6-51)
            for (int g = q; g < WP / 4; g += 4) {
                float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (4 * g < W) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = G[j * NG + 4 * (layer - 1) + e];
                        float r = c;
                        if (g >= 2) {
                            const float arg = __fmul_rn(c, (float)(1 << ((g - 2) >> 1)));
                            r = (g & 1) ? cosf(arg) : sinf(arg);
                        }
                        v[e] = r;
                    }
                }
                *reinterpret_cast<float4 *>(EV + (g * 16 + j) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
            __builtin_amdgcn_wave_barrier();
            if (STYLE) {
                // StyleAttention: linear_transform(e) (its softmax over a size-1 dim is 1), then the per-row InstanceNorm
                // over the W features (biased variance), styleMod.py:30-43; the point's rows sit in its four lanes
                nffb_matvec_mfma<W, NT>(a.style_w, EV, acc, j, q);
                float sum = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 16 * t + 4 * q + e;
                        const float y = row < W ? __fadd_rn(acc[t][e], a.style_b[row]) : 0.0f;
                        acc[t][e] = y;
                        sum += y;
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float mean = sum / (float)W;
                float var = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 16 * t + 4 * q + e;
                        const float d = row < W ? acc[t][e] - mean : 0.0f;
                        var += d * d;
                    }
                var += __shfl_xor(var, 16);
                var += __shfl_xor(var, 32);
                const float den = sqrtf(var / (float)W + a.style_eps);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 xv = *reinterpret_cast<const float4 *>(XV + ((4 * t + q) * 16 + j) * 4);
                    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 16 * t + 4 * q + e;
                        v[e] = row < W ? __fadd_rn((acc[t][e] - mean) / den, xe[e]) : 0.0f;
                    }
                    *reinterpret_cast<float4 *>(EV + ((4 * t + q) * 16 + j) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float4 *ep = reinterpret_cast<float4 *>(EV + ((4 * t + q) * 16 + j) * 4);
                    const float4 ev = *ep, xv = *reinterpret_cast<const float4 *>(XV + ((4 * t + q) * 16 + j) * 4);
                    *ep = make_float4(__fadd_rn(ev.x, xv.x), __fadd_rn(ev.y, xv.y), __fadd_rn(ev.z, xv.z), __fadd_rn(ev.w, xv.w));
                }
            }
            __builtin_amdgcn_wave_barrier();
            // shared output layer on e = posenc (+ style) + trunk; features accumulate in the accumulator layout
            nffb_matvec_mfma<W, NT>(a.out_w, EV, acc, j, q);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * t + 4 * q + e;
                    if (row < W) feat[t][e] = __fadd_rn(feat[t][e], __fadd_rn(acc[t][e], a.out_b[row]));
                }
            __builtin_amdgcn_wave_barrier();
        }
        if (live) {
            float *o = out + i * out_stride;
            if (q == 0) {
                o[0] = u0; o[1] = u1; o[2] = u2;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * t + 4 * q + e;
                    if (row < W) o[3 + row] = __fdiv_rn(feat[t][e], (float)LV);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <int FRAC, int LV, bool STYLE>
int launch_nffb1(hipStream_t st, const HmLevels &lv, const NffbArgs &a, const float *x, int64_t n, const float *table,
                 const float *Bf, float *out, int64_t out_stride, const int32_t *n_dev) {
    // with a device-side count the host cannot know the batch size: both variants are enqueued and each returns at
    // once unless the live count falls in its range (like the fused SDF kernels)
    const bool small = n <= kSmallCount, big = n_dev ? n > kSmallCount : !small;
    const int64_t kBig = (int64_t)1 << 62;
    static const int mfma_cfg = [] { const char *e = getenv("HM_NFFB_MFMA"); return e ? atoi(e) : 1; }();
    if (small || n_dev) {
        const int64_t cap = n < kSmallCount ? n : kSmallCount;
        if (mfma_cfg == 2) {   // (experiment: the matrix-core kernel for the tracer's rounds too, one wave per workgroup)
            const int64_t blocks = (cap + 15) / 16;
            hipLaunchKernelGGL((nffb_fwd_mfma_kernel<FRAC, LV, STYLE, 1>), dim3((unsigned)blocks), dim3(64), 0, st, lv, a, x, n,
                               table, Bf, out, out_stride, n_dev, (int64_t)0, (int64_t)kSmallCount);
        } else {
            const int64_t blocks = (cap + kNT / 32 - 1) / (kNT / 32);
            hipLaunchKernelGGL((nffb_fwd_kernel<FRAC, LV, STYLE, 32>), dim3((unsigned)blocks), dim3(kNT), 0, st, lv, a, x, n,
                               table, Bf, out, out_stride, n_dev, (int64_t)0, (int64_t)kSmallCount);
        }
    }
    if (big) {
        if (mfma_cfg) {     // matrix-core tiles, a wave per 16 points
            const int64_t blocks = (n + 16 * kMfmaWaves - 1) / (16 * kMfmaWaves);
            hipLaunchKernelGGL((nffb_fwd_mfma_kernel<FRAC, LV, STYLE, kMfmaWaves>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)),
                               dim3(64 * kMfmaWaves), 0, st, lv, a, x, n, table, Bf, out, out_stride, n_dev,
                               (int64_t)kSmallCount + 1, kBig);
        } else {            // (HM_NFFB_MFMA=0: the 8-lanes-per-point VALU kernel, for A/B measurements)
            const int64_t blocks = (n + kNT / 8 - 1) / (kNT / 8);
            hipLaunchKernelGGL((nffb_fwd_kernel<FRAC, LV, STYLE, 8>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kNT),
                               0, st, lv, a, x, n, table, Bf, out, out_stride, n_dev, (int64_t)kSmallCount + 1, kBig);
        }
    }
    return HM_OK;
}

template <int FRAC, int LV>
int launch_nffb(bool style, hipStream_t st, const HmLevels &lv, const NffbArgs &a, const float *x, int64_t n,
                const float *table, const float *Bf, float *out, int64_t out_stride, const int32_t *n_dev) {
    return style ? launch_nffb1<FRAC, LV, true>(st, lv, a, x, n, table, Bf, out, out_stride, n_dev)
                 : launch_nffb1<FRAC, LV, false>(st, lv, a, x, n, table, Bf, out, out_stride, n_dev);
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
    hipStream_t st = as_stream(stream);
    int rc;
#define HM_NFFB(FR)                                                                                                    \
    rc = (lv.L == 6) ? launch_nffb<FR, 6>(style, st, lv, a, x, n, table, B_fourier, out, out_stride, n_dev)            \
                     : launch_nffb<FR, 8>(style, st, lv, a, x, n, table, B_fourier, out, out_stride, n_dev)
    if (frac_mode == HM_FRAC_REFERENCE) { HM_NFFB(HM_FRAC_REFERENCE); } else { HM_NFFB(HM_FRAC_TRILINEAR); }
#undef HM_NFFB
    if (rc != HM_OK) return rc;
    HM_CHECK_LAUNCH("hm_nffb_fwd");
    return HM_OK;
}

}  // extern "C"
