// hm_sdf_bf16.hip - bf16 variant of the fused no-grad SDF forward (BASELINE configs[4]: "bf16, fused MLP path").
//
// Same operator as hm_sdf.hip's sdf_fwd_kernel (reference: model/implicit_differentiable_renderer.py:89-113 under
// no_grad), sdf-only output, for the COARSE searches of the ray tracer: the 100-sample sign-change scan and the
// closest-approach scan (model/ray_tracing.py:189-249, 270-298) - 80 % of the SDF evaluations of an iteration.  The
// sphere-tracing rounds and the secant refinement that produce the final hit stay on the exact-fp32 kernels, so the
// bf16 error (~1e-4 abs in the SDF) can only move the choice of a bracketing sample / closest sample, never the
// refined intersection.  The reference has no reduced-precision behaviour; the criterion is the
// measured error against the fp32 kernel and loss-curve agreement, both in tests/test_bf16_gpu.py.
//
// Mapping: v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, so the tile is bound by its weight stream, not by
// the matrix pipe: a workgroup (8 waves, one per CU) owns 96 points - X as bf16 [k/8][point][8] = 96 KB, the embedding
// in fp32 [e/4][point][4] = 27.6 KB - and every wave computes 2 feature tiles x 3 point tiles per 1-KB weight block
// (6 MFMAs = 192 cycles per 2 KB streamed: ~100 GB/s per CU at the matrix rate, i.e. the L2 stream and the pipe are
// about balanced).  What stays fp32: the embedding and every product that consumes it (layer 0 and the skip layer's
// embedding segment run on v_mfma_f32_32x32x2_f32 from the fp32 operand image - the raw coordinates never see 8-bit
// mantissas), all accumulators, bias / Softplus, the last layer's dot product and the clamp.  Hidden activations and
// hidden-layer weights are bf16 (round-to-nearest-even).  The C/D register layout of the two MFMA shapes is the same
// (row = 8(reg/4) + 4(lane/32) + reg%4, col = lane%32), so both accumulate into the same tiles.
#include "hm_common.h"

#include <math.h>

namespace {

#include "hm_sdf_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kPB = 96;            // points per workgroup tile
constexpr int kTB16 = 512;         // threads
constexpr int kWB = 8;             // waves
constexpr int kPT = kPB / 32;      // point tiles per wave (3)
constexpr int kEmbGF = kPB * 4;    // floats per k-group row of the fp32 embedding image
constexpr int kXOct = kPB * 8;     // bf16 elements per k-octet row of X

union Frag16 {   // 16 bytes = 8 bf16 = one MFMA operand fragment
    float4 f;
    bf16x8 h;
};

template <int FRAC>
__global__ __launch_bounds__(kTB16, 2) void sdf_fwd_bf16_kernel(HmLevels lv, SdfNet net, const float *__restrict__ x,
                                                                 int64_t n, const float *__restrict__ table,
                                                                 const float *__restrict__ Bf,
                                                                 float *__restrict__ out, int64_t out_stride,
                                                                 const int32_t *__restrict__ n_dev, int64_t run_min,
                                                                 int64_t run_max) {
    extern __shared__ __align__(16) float lds[];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;
    __bf16 *X = reinterpret_cast<__bf16 *>(lds);                                   // [x_oct][kPB][8] bf16
    const int x_oct = net.x_groups / 2;                                           // k-octets of the widest layer
    float *EMB = lds + (size_t)x_oct * kXOct / 2;                                 // [emb_groups][kPB][4] fp32
    float *SX = EMB + (size_t)net.emb_groups * kEmbGF;                            // [kPB][3] raw points (+ pad)
    float *RED = SX + kPB * 4;                                                    // [5][kPB] last-layer partial sums

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: descriptors / scalar offsets of the weight stream)
    const int lane = tid & 63;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int L = lv.L, F = lv.F, E = lv.E;
    const int64_t n_tiles = (n + kPB - 1) / kPB;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kPB;
        const int cnt = (int)min((int64_t)kPB, n - base);
        __syncthreads();
        if (net.emb_stride == 0 && tid < kPB * 3) SX[tid] = (tid < cnt * 3) ? x[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---------------- embedding -> EMB[(e/4)][p][e%4] (fp32) ------------------------------------------------
        if (net.emb_stride > 0) {
            load_emb_tile(EMB, x, net.emb_stride, base, cnt, E, net.emb_groups, kPB, kEmbGF, tid, kTB16);
        } else {
            auto put = [&](int p, int e, float v) { EMB[(e >> 2) * kEmbGF + p * 4 + (e & 3)] = v; };
            const int n_slot = 2 * L + 1;      // slot 0: pass-through + padding, 1..L: Fourier channel, L+1..2L: level
            for (int idx = tid; idx < kPB * n_slot; idx += kTB16) {
                const int p = idx % kPB, slot = idx / kPB;
                const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
                if (slot == 0) {
                    put(p, 0, x0); put(p, 1, x1); put(p, 2, x2);
                    for (int e = E; e < net.emb_groups * 4; ++e) put(p, e, 0.0f);
                } else if (slot <= L) {
                    const int c = slot - 1;
                    const float two_pi = 6.283185307179586f;
                    float a = __fmul_rn(__fmul_rn(two_pi, x0), Bf[c]);
                    a = __fmaf_rn(__fmul_rn(two_pi, x1), Bf[L + c], a);
                    a = __fmaf_rn(__fmul_rn(two_pi, x2), Bf[2 * L + c], a);
                    float sn, cs;
                    sincosf(a, &sn, &cs);
                    put(p, 3 + c, sn);
                    put(p, 3 + L + c, cs);
                } else {
                    const int l = slot - L - 1;
                    float acc[8];
                    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
                    const float *tl = table + (size_t)lv.row_off[l] * F;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        uint32_t ux, uy, uz;
                        float wx, wy, wz;
                        corner<FRAC>(x0, lv.res[l], c & 1, ux, wx);
                        corner<FRAC>(x1, lv.res[l], (c >> 1) & 1, uy, wy);
                        corner<FRAC>(x2, lv.res[l], (c >> 2) & 1, uz, wz);
                        const float w = __fmul_rn(__fmul_rn(wx, wy), wz);
                        if (w != 0.0f) {
                            const uint32_t id = hm_mod_rows(hm_hash3(ux, uy, uz), lv.rows[l], lv.magic[l]);
                            for (int f = 0; f < F; ++f) acc[f] = __fadd_rn(acc[f], __fmul_rn(tl[(size_t)id * F + f], w));
                        }
                    }
                    for (int f = 0; f < F; ++f) put(p, 3 + 2 * L + l * F + f, acc[f]);
                }
            }
        }
        __syncthreads();

        // ---------------- layers ------------------------------------------------------------------------------
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            if (li == net.n_layers - 1) {
                // sdf-only last layer (one segment, previous layer's output): fp32 VALU dot of row 0 of the fp32
                // image with the bf16 activations; thread = (point, k slice of 5)
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed);
                const int p = tid % kPB, sl = tid / kPB;       // 480 threads busy
                float part = 0.0f;
                if (sl < 5) {
                    const int n_o = Ly.seg_octets[0];
                    for (int kb = sl; kb < n_o; kb += 5) {
                        const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(X + (size_t)kb * kXOct + p * 8);
                        const float4 w0 = W0[(size_t)kb * 64], w1 = W0[(size_t)kb * 64 + 32];   // k = 8kb+0..3 / +4..7
                        part = __fmaf_rn((float)xv[0], w0.x, part);
                        part = __fmaf_rn((float)xv[1], w0.y, part);
                        part = __fmaf_rn((float)xv[2], w0.z, part);
                        part = __fmaf_rn((float)xv[3], w0.w, part);
                        part = __fmaf_rn((float)xv[4], w1.x, part);
                        part = __fmaf_rn((float)xv[5], w1.y, part);
                        part = __fmaf_rn((float)xv[6], w1.z, part);
                        part = __fmaf_rn((float)xv[7], w1.w, part);
                    }
                    RED[sl * kPB + p] = part;
                }
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int s5 = 0; s5 < 5; ++s5) sacc += RED[s5 * kPB + tid];
                    out[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt = Ly.n_tiles;
            const int t0 = 2 * wave;
            const int ntw = max(0, min(2, nt - t0));
            f32x16 acc[2][kPT];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < kPT; ++q) acc[a][q] = f32x16{0};
            if (ntw > 0) {
                const int n_oct = Ly.seg_octets[0] + Ly.seg_octets[1];
                const int nb = Ly.seg_blocks16[0] + Ly.seg_blocks16[1];
                int oct0 = 0, blk0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    if (Ly.seg_octets[seg] == 0) continue;
                    if (Ly.seg_src[seg] == 1) {
                        // ---- embedding segment: exact fp32 (v_mfma_f32_32x32x2_f32, fp32 operand image) -----------
                        const float4 *A0 = reinterpret_cast<const float4 *>(Ly.w_packed) +
                                           ((size_t)t0 * n_oct + oct0) * 64 + lane;
                        const float4 *A1 = A0 + (ntw > 1 ? (size_t)n_oct * 64 : 0);
                        for (int g = 0; g < Ly.seg_octets[seg]; ++g) {
                            const float4 a0 = A0[(size_t)g * 64], a1 = A1[(size_t)g * 64];
                            const float *src = EMB + (2 * g + h) * kEmbGF;
#pragma unroll
                            for (int q = 0; q < kPT; ++q) {
                                const float4 b = *reinterpret_cast<const float4 *>(src + (32 * q + j) * 4);
                                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc[0][q], 0, 0, 0);
                                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc[1][q], 0, 0, 0);
                                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc[0][q], 0, 0, 0);
                                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc[1][q], 0, 0, 0);
                                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc[0][q], 0, 0, 0);
                                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc[1][q], 0, 0, 0);
                                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc[0][q], 0, 0, 0);
                                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc[1][q], 0, 0, 0);
                            }
                        }
                    } else {
                        // ---- hidden segment: bf16 (v_mfma_f32_32x32x16_bf16), 1-KB weight blocks through a 4-slot ring
                        const int nbs = Ly.seg_blocks16[seg];
                        // (buffer loads: descriptor on the wave's first feature tile, lane * 16 the one VGPR offset, block /
                        //  tile offsets scalar - no address VALU between the MFMAs, hm_sdf_common.h: ld_w16)
                        const __amdgpu_buffer_rsrc_t rA =
                            w_rsrc(reinterpret_cast<const float *>(Ly.w_packed_bf16) + ((size_t)t0 * nb + blk0) * 256);
                        const int a1 = ntw > 1 ? nb * 1024 : 0, lane16 = lane * 16;
                        Frag16 r0[4], r1[4];
#pragma unroll
                        for (int st = 0; st < 3; ++st) {
                            const int off = min(st, nbs - 1) * 1024;
                            r0[st].f = ld_w16(rA, lane16, off);
                            r1[st].f = ld_w16(rA, lane16, a1 + off);
                        }
                        auto block = [&](int t, const Frag16 &a0, const Frag16 &a1) {
                            const __bf16 *src = X + (size_t)(2 * t + h) * kXOct;
#pragma unroll
                            for (int q = 0; q < kPT; ++q) {
                                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(src + (32 * q + j) * 8);
                                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b, acc[0][q], 0, 0, 0);
                                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b, acc[1][q], 0, 0, 0);
                            }
                        };
                        const int n_full = nbs & ~3;
                        for (int tt = 0; tt < n_full; tt += 4) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int t = tt + u;
                                {
                                    const int off = min(t + 3, nbs - 1) * 1024;
                                    r0[(u + 3) & 3].f = ld_w16(rA, lane16, off);
                                    r1[(u + 3) & 3].f = ld_w16(rA, lane16, a1 + off);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                block(t, r0[u], r1[u]);
                            }
                        }
                        if (n_full + 0 < nbs) block(n_full + 0, r0[0], r1[0]);
                        if (n_full + 1 < nbs) block(n_full + 1, r0[1], r1[1]);
                        if (n_full + 2 < nbs) block(n_full + 2, r0[2], r1[2]);
                    }
                    oct0 += Ly.seg_octets[seg];
                    blk0 += Ly.seg_blocks16[seg];
                }
            }
            __syncthreads();  // every wave has finished reading X / EMB for this layer

            // epilogue: registers 4q..4q+3 of a tile = features 8q + 4h + {0..3} -> 4 bf16 of one k-octet of the next layer
            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a >= ntw) continue;
                const int fbase = 32 * (t0 + a);
#pragma unroll
                for (int pt = 0; pt < kPT; ++pt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = fbase + 8 * q + 4 * h;
                        const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                        float v0 = acc[a][pt][4 * q + 0] + bb.x, v1 = acc[a][pt][4 * q + 1] + bb.y,
                              v2 = acc[a][pt][4 * q + 2] + bb.z, v3 = acc[a][pt][4 * q + 3] + bb.w;
                        if (act) {
                            softplus100_4(v0, v1, v2, v3);
                        }
                        if (div) {
                            v0 = __fdiv_rn(v0, sqrt2); v1 = __fdiv_rn(v1, sqrt2); v2 = __fdiv_rn(v2, sqrt2);
                            v3 = __fdiv_rn(v3, sqrt2);
                        }
                        bf16x4 o;
                        o[0] = (__bf16)v0; o[1] = (__bf16)v1; o[2] = (__bf16)v2; o[3] = (__bf16)v3;
                        *reinterpret_cast<bf16x4 *>(X + (size_t)(f >> 3) * kXOct + (32 * pt + j) * 8 + 4 * h) = o;
                    }
                }
            }
            if (li == 0 && net.emb_groups > 0) {
                // the skip layer consumes cat[x, emb]/sqrt(2): rescale the kept embedding once, in place
                for (int i = tid; i < net.emb_groups * kEmbGF; i += kTB16) EMB[i] = __fdiv_rn(EMB[i], sqrt2);
            }
            __syncthreads();
        }
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static int sdf_bf16_impl(const HmLevels &lv, const hm_mlp_desc *mlp, const float *x, int64_t emb_stride, int64_t n,
                         const float *table, const float *B_fourier, float *out, int64_t out_stride, int frac_mode,
                         const int32_t *n_dev, int64_t run_min, void *stream) {
    HM_CHECK_ARG(mlp, "hm_sdf_fwd_bf16: NULL descriptor");
    HM_CHECK_ARG(n >= 0, "hm_sdf_fwd_bf16: n < 0");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_sdf_fwd_bf16: bad frac_mode");
    HM_CHECK_ARG(mlp->n_layers >= 2 && mlp->n_layers <= HM_MAX_LAYERS, "hm_sdf_fwd_bf16: n_layers out of range");
    SdfNet net;
    net.n_layers = mlp->n_layers;
    net.beta = mlp->beta;
    net.emb_stride = emb_stride;
    const int emb_oct = (lv.E + 7) / 8, emb_b16 = (lv.E + 15) / 16;
    net.emb_groups = emb_oct * 2;
    int x_groups = 0;
    for (int l = 0; l < mlp->n_layers; ++l) {
        const hm_mlp_layer &Ly = mlp->layer[l];
        HM_CHECK_ARG(Ly.w_packed && Ly.bias && Ly.w_packed_bf16, "hm_sdf_fwd_bf16: layer lacks the fp32 or the bf16 image");
        HM_CHECK_ARG(Ly.n_tiles >= 1 && Ly.n_tiles <= 2 * kWB, "hm_sdf_fwd_bf16: layer wider than 512 features");
        HM_CHECK_ARG(Ly.seg_octets[0] >= 1 && Ly.seg_octets[1] >= 0, "hm_sdf_fwd_bf16: bad segment length");
        for (int s = 0; s < 2; ++s) {
            if (Ly.seg_octets[s] == 0) continue;
            if (Ly.seg_src[s] == 1) {
                HM_CHECK_ARG(Ly.seg_octets[s] == emb_oct && Ly.seg_blocks16[s] == emb_b16,
                             "hm_sdf_fwd_bf16: embedding segment must span ceil(E/8) octets / ceil(E/16) blocks");
            } else {
                HM_CHECK_ARG(l > 0 && Ly.seg_blocks16[s] * 16 <= mlp->layer[l - 1].n_tiles * 32 &&
                                 Ly.seg_blocks16[s] * 16 >= mlp->layer[l - 1].out_dim,
                             "hm_sdf_fwd_bf16: hidden segment does not match the previous layer");
            }
        }
        x_groups = max(x_groups, Ly.n_tiles * 8);
        net.layer[l] = Ly;
    }
    const hm_mlp_layer &last = mlp->layer[mlp->n_layers - 1];
    HM_CHECK_ARG(last.seg_octets[1] == 0 && last.seg_src[0] == 0, "hm_sdf_fwd_bf16: the last layer must read the previous layer only");
    net.x_groups = x_groups;
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && out && (emb_stride > 0 || (table && B_fourier)), "hm_sdf_fwd_bf16: NULL pointer");
    const size_t lds = (size_t)(x_groups / 2) * kXOct * 2 + sizeof(float) * ((size_t)net.emb_groups * kEmbGF + kPB * 4 + 5 * kPB);
    HM_CHECK_ARG(lds <= 160 * 1024, "hm_sdf_fwd_bf16: network does not fit the 160 KB LDS tile");
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_bf16_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_bf16_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    const int64_t tiles = (n + kPB - 1) / kPB;
    const int64_t grid = tiles < 256 ? tiles : 256;
    const int64_t big = (int64_t)1 << 62;
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(sdf_fwd_bf16_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kTB16), lds,
                           as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, big);
    else
        hipLaunchKernelGGL(sdf_fwd_bf16_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kTB16), lds,
                           as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, big);
    HM_CHECK_LAUNCH("hm_sdf_fwd_bf16");
    return HM_OK;
}

}  // namespace

extern "C" {

int hm_sdf_fwd_bf16(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n, const float *table,
                    const float *B_fourier, float *out, int64_t out_stride, int frac_mode, const int32_t *n_dev,
                    int64_t run_min, void *stream) {
    HM_CHECK_ARG(desc, "hm_sdf_fwd_bf16: NULL descriptor");
    return sdf_bf16_impl(desc->lv, mlp, x, 0, n, table, B_fourier, out, out_stride, frac_mode, n_dev, run_min, stream);
}

int hm_sdf_fwd_emb_bf16(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n,
                        float *out, int64_t out_stride, const int32_t *n_dev, int64_t run_min, void *stream) {
    HM_CHECK_ARG(emb_width >= 1 && emb_width <= 512 && emb_stride >= emb_width, "hm_sdf_fwd_emb_bf16: bad embedding width / stride");
    HmLevels lv = {};
    lv.L = 0; lv.F = 2; lv.E = emb_width;
    return sdf_bf16_impl(lv, mlp, emb, emb_stride, n, nullptr, nullptr, out, out_stride, HM_FRAC_REFERENCE, n_dev,
                         run_min, stream);
}

}  // extern "C"
