// hm_sdf_split.hip - the fused no-grad SDF forward on the 16-bit matrix cores with SPLIT operands.
//
// Same operator as hm_sdf.hip's sdf_fwd_kernel (reference: model/implicit_differentiable_renderer.py:89-113 under
// no_grad; density_net.py:20-30), sdf-only output, for the coarse scans of the ray tracer (model/ray_tracing.py:189-249,
// 270-298).  Every operand of every matrix product - weights, hidden activations AND the embedding - is carried as a
// pair (hi, lo) of 16-bit floats with  c v ~= hi + lo  (c = a power-of-two operand scale) and the product is evaluated as
//         W x  ~=  (Wh xh + Wh xl + Wl xh) / (c_w c_x)            (three MFMAs, fp32 accumulate; Wl xl is dropped)
// on v_mfma_f32_32x32x16_{bf16,f16}, which run at 16x the rate of the fp32 MFMA:
//   HM_SPLIT_BF16X2  hi, lo bf16 (s = 0): 16 significant bits per operand, relative error 2^-16 per product - the
//                    "bf16" configuration of BASELINE configs[4] with 250x the accuracy of plain bf16 operands;
//   HM_SPLIT_F16X2   hi, lo fp16; activations are stored scaled by 2^4 and weights by 2^8 (so that the lo parts of
//                    ordinary magnitudes are normal fp16 numbers; smaller ones are subnormals, which the MFMA keeps):
//                    22 significant bits per operand, relative error <= 3 * 2^-22 per product - below the rounding noise
//                    an fp32 accumulation over K = 512 adds anyway (~ sqrt(K) 2^-24); activations beyond |x| = 4094
//                    become non-finite and are counted by the tracer's non-finite counter.
// The reference has no such mode; tests/test_split_gpu.py measures both kinds against the exact-fp32 kernel and against
// an fp64 evaluation, and applies SURVEY.md 8(d)'s loss-curve criterion.
//
// Mapping: a workgroup (8 waves, one per CU) owns 64 points.  Activations live in LDS as two planes [k/8][point][8] of
// 2-byte elements (hi, lo) = 128 KB at 512 features, the embedding likewise (2 x 10 KB at E = 67) - the same 4 bytes
// per element as the fp32 kernel.  Wave w owns feature tiles 2w, 2w+1 for both 32-point tiles: per 16-wide k block it
// streams 4 KB of the split weight image ([tile][block][hi | lo][lane][8]) through a 4-slot register ring, reads the
// four B fragments with ds_read_b128 and issues 12 MFMAs (384 cycles), i.e. ~100 GB/s of weight stream per CU at the
// matrix rate: stream and pipe are about balanced, as in the bf16 kernel.  fp32: accumulators, bias, Softplus, the
// sqrt(2) of the layer before the skip, the last layer's dot product and the clamp.  The 1/sqrt(2) of the skip
// layer's EMBEDDING segment is folded into that segment's weights at pack time (hm_pack_mlp_layer_split).
#include "hm_common.h"

#include <math.h>

namespace {

#include "hm_sdf_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int KIND> struct Split;
template <> struct Split<HM_SPLIT_BF16X2> {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x4 V4;
    static constexpr float xs = 1.0f, ws = 1.0f, inv = 1.0f;   // operand scales (bf16 has fp32's exponent range: none)
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Split<HM_SPLIT_F16X2> {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x4 V4;
    // fp16 lo parts are ~2^-11 of their value: activations are stored scaled by 2^4 and weights by 2^8 so that the lo
    // parts of ordinary magnitudes (|x| >= 0.008, |w| >= 5e-4) are NORMAL fp16 numbers; smaller ones become subnormal,
    // which the MFMA keeps (scripts/f16_denorm_probe.hip) - their absolute error stays below 2^-25 / scale.  The
    // accumulators hold 2^12 x the sums; `inv` is applied together with the bias.  Range: |x| <= 4094, |w| <= 255.
    static constexpr float xs = 16.0f, ws = 256.0f, inv = 1.0f / 4096.0f;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// v (already multiplied by the operand's scale) -> hi + lo
template <int KIND>
__device__ __forceinline__ void split_val(float v, typename Split<KIND>::T &hi, typename Split<KIND>::T &lo) {
    typedef typename Split<KIND>::T T;
    hi = (T)v;                                              // round to nearest even
    lo = (T)__fsub_rn(v, (float)hi);
}

constexpr int kPS = 64;            // points per workgroup tile
constexpr int kTS = 512;           // threads
constexpr int kWS = 8;             // waves
constexpr int kOctE = kPS * 8;     // 2-byte elements per k-octet row of a plane

template <int KIND>
union FragS {   // 16 bytes = 8 two-byte elements = one MFMA operand fragment
    float4 f;
    typename Split<KIND>::V8 h;
};

template <int KIND, int FRAC>
__global__ __launch_bounds__(kTS, 2) void sdf_fwd_split_kernel(HmLevels lv, SdfNet net, const float *__restrict__ x,
                                                                int64_t n, const float *__restrict__ table,
                                                                const float *__restrict__ Bf,
                                                                float *__restrict__ out, int64_t out_stride,
                                                                const int32_t *__restrict__ n_dev, int64_t run_min,
                                                                int64_t run_max) {
    typedef Split<KIND> S;
    typedef typename S::T T;
    typedef typename S::V8 V8;
    typedef typename S::V4 V4;
    extern __shared__ __align__(16) float lds[];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;
    const int x_oct = net.x_groups / 2, e_oct = net.emb_groups / 2;
    T *XH = reinterpret_cast<T *>(lds);                       // [x_oct][kPS][8]
    T *XL = XH + (size_t)x_oct * kOctE;
    T *EH = XL + (size_t)x_oct * kOctE;                       // [e_oct][kPS][8]
    T *EL = EH + (size_t)e_oct * kOctE;
    float *SX = reinterpret_cast<float *>(EL + (size_t)e_oct * kOctE);   // [kPS][3] raw points (+ pad)
    float *RED = SX + kPS * 4;                                            // [8][kPS] last-layer partial sums

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: descriptors / scalar offsets of the weight stream)
    const int lane = tid & 63;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int L = lv.L, F = lv.F, E = lv.E;
    const int64_t n_tiles = (n + kPS - 1) / kPS;
    const int e_pad = e_oct * 8;

    auto put = [&](int p, int e, float v) {
        T hi, lo;
        split_val<KIND>(v * S::xs, hi, lo);
        const size_t o = (size_t)(e >> 3) * kOctE + p * 8 + (e & 7);
        EH[o] = hi;
        EL[o] = lo;
    };

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kPS;
        const int cnt = (int)min((int64_t)kPS, n - base);
        __syncthreads();
        if (net.emb_stride == 0 && tid < kPS * 3) SX[tid] = (tid < cnt * 3) ? x[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---------------- embedding (fp32 arithmetic) -> split planes EH / EL ---------------------------------
        if (net.emb_stride > 0) {
            for (int i = tid; i < kPS * e_pad; i += kTS) {
                const int p = i / e_pad, e = i - p * e_pad;
                put(p, e, (p < cnt && e < E) ? x[(base + p) * net.emb_stride + e] : 0.0f);
            }
        } else {
            const int n_slot = 2 * L + 1;      // slot 0: pass-through + padding, 1..L: Fourier channel, L+1..2L: level
            for (int idx = tid; idx < kPS * n_slot; idx += kTS) {
                const int p = idx % kPS, slot = idx / kPS;
                const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
                if (slot == 0) {
                    put(p, 0, x0); put(p, 1, x1); put(p, 2, x2);
                    for (int e = E; e < e_pad; ++e) put(p, e, 0.0f);
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
        // weight ring: D slots per image part, D-1 blocks in flight (fp16 kind: two accumulator sets leave room for
        // three slots only - a fourth spills).  It lives across the layers: the first D-1 blocks of layer l+1 are
        // requested before layer l's epilogue (weights do not depend on the activations), so the stream does not
        // restart from an empty pipe behind the two barriers of every layer.
        constexpr int D = 4;
        FragS<KIND> r0h[D], r0l[D], r1h[D], r1l[D];
        bool ring_ready = false;
        // (weight stream as buffer loads: descriptor on the wave's first feature tile, lane * 16 the one VGPR offset, block /
        //  tile offsets scalar - no address arithmetic on the VALU between the MFMAs, hm_sdf_common.h: ld_w16)
        const int lane16 = lane * 16;
        auto ring_fill = [&](const __amdgpu_buffer_rsrc_t &rs, int a1, int nbs) {
#pragma unroll
            for (int st = 0; st < D - 1; ++st) {
                const int off = min(st, nbs - 1) * 2048;
                r0h[st].f = ld_w16(rs, lane16, off); r0l[st].f = ld_w16(rs, lane16, off + 1024);
                r1h[st].f = ld_w16(rs, lane16, a1 + off); r1l[st].f = ld_w16(rs, lane16, a1 + off + 1024);
            }
        };
        auto prefetch_layer = [&](int l) {      // segment 0 of layer l, this wave's feature tiles
            const hm_mlp_layer &Lp = net.layer[l];
            const int ntp = max(0, min(2, Lp.n_tiles - 2 * wave));
            if (ntp <= 0) return false;
            const int nbp = Lp.seg_blocks16[0] + Lp.seg_blocks16[1];
            const __amdgpu_buffer_rsrc_t rp =
                w_rsrc(reinterpret_cast<const float *>(Lp.w_packed_split) + ((size_t)(2 * wave) * nbp) * 512);
            ring_fill(rp, ntp > 1 ? nbp * 2048 : 0, Lp.seg_blocks16[0]);
            return true;
        };
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            if (li == net.n_layers - 1) {
                // sdf-only last layer (one segment, previous layer's output): fp32 VALU dot of row 0 of the fp32 image
                // with the reassembled activations; thread = (point, k slice of 8)
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed);
                const int p = tid & (kPS - 1), sl = tid >> 6;
                float part = 0.0f;
                const int n_o = Ly.seg_octets[0];
                for (int kb = sl; kb < n_o; kb += kWS) {
                    const V8 xh = *reinterpret_cast<const V8 *>(XH + (size_t)kb * kOctE + p * 8);
                    const V8 xl = *reinterpret_cast<const V8 *>(XL + (size_t)kb * kOctE + p * 8);
                    const float4 w0 = W0[(size_t)kb * 64], w1 = W0[(size_t)kb * 64 + 32];   // k = 8kb+0..3 / +4..7
                    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        part = __fmaf_rn(((float)xl[e] + (float)xh[e]) * (1.0f / S::xs), wv[e], part);
                }
                RED[sl * kPS + p] = part;
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int s8 = 0; s8 < kWS; ++s8) sacc += RED[s8 * kPS + tid];
                    out[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt = Ly.n_tiles;
            const int t0 = 2 * wave;
            const int ntw = max(0, min(2, nt - t0));
            f32x16 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[a][q] = f32x16{0};
            if (ntw > 0) {
                const int nb = Ly.seg_blocks16[0] + Ly.seg_blocks16[1];
                int blk0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const int nbs = Ly.seg_blocks16[seg];
                    if (nbs == 0) continue;
                    const T *srcH = Ly.seg_src[seg] == 1 ? EH : XH;
                    const T *srcL = Ly.seg_src[seg] == 1 ? EL : XL;
                    // image: [tile][block][hi | lo][lane] float4
                    const __amdgpu_buffer_rsrc_t rA =
                        w_rsrc(reinterpret_cast<const float *>(Ly.w_packed_split) + ((size_t)t0 * nb + blk0) * 512);
                    const int a1 = ntw > 1 ? nb * 2048 : 0;
                    if (!(seg == 0 && ring_ready)) ring_fill(rA, a1, nbs);
                    ring_ready = false;
                    auto block = [&](int t, const FragS<KIND> &a0h, const FragS<KIND> &a0l, const FragS<KIND> &a1h,
                                     const FragS<KIND> &a1l) {
                        const size_t o = (size_t)(2 * t + h) * kOctE + j * 8;
                        const V8 bh0 = *reinterpret_cast<const V8 *>(srcH + o);
                        const V8 bh1 = *reinterpret_cast<const V8 *>(srcH + o + 32 * 8);
                        const V8 bl0 = *reinterpret_cast<const V8 *>(srcL + o);
                        const V8 bl1 = *reinterpret_cast<const V8 *>(srcL + o + 32 * 8);
                        // hi x hi, then hi x lo, then lo x hi: accumulators that are used twice are four MFMAs apart
                        acc[0][0] = S::mfma(a0h.h, bh0, acc[0][0]);
                        acc[0][1] = S::mfma(a0h.h, bh1, acc[0][1]);
                        acc[1][0] = S::mfma(a1h.h, bh0, acc[1][0]);
                        acc[1][1] = S::mfma(a1h.h, bh1, acc[1][1]);
                        acc[0][0] = S::mfma(a0h.h, bl0, acc[0][0]);
                        acc[0][1] = S::mfma(a0h.h, bl1, acc[0][1]);
                        acc[1][0] = S::mfma(a1h.h, bl0, acc[1][0]);
                        acc[1][1] = S::mfma(a1h.h, bl1, acc[1][1]);
                        acc[0][0] = S::mfma(a0l.h, bh0, acc[0][0]);
                        acc[0][1] = S::mfma(a0l.h, bh1, acc[0][1]);
                        acc[1][0] = S::mfma(a1l.h, bh0, acc[1][0]);
                        acc[1][1] = S::mfma(a1l.h, bh1, acc[1][1]);
                    };
                    const int n_full = nbs - nbs % D;
                    for (int tt = 0; tt < n_full; tt += D) {
#pragma unroll
                        for (int u = 0; u < D; ++u) {
                            const int t = tt + u;
                            {
                                const int off = min(t + D - 1, nbs - 1) * 2048;
                                r0h[(u + D - 1) % D].f = ld_w16(rA, lane16, off);
                                r0l[(u + D - 1) % D].f = ld_w16(rA, lane16, off + 1024);
                                r1h[(u + D - 1) % D].f = ld_w16(rA, lane16, a1 + off);
                                r1l[(u + D - 1) % D].f = ld_w16(rA, lane16, a1 + off + 1024);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            block(t, r0h[u], r0l[u], r1h[u], r1l[u]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < D - 1; ++u)      // the left-over blocks are already in ring slots 0 .. D-2
                        if (n_full + u < nbs) block(n_full + u, r0h[u], r0l[u], r1h[u], r1l[u]);
                    blk0 += nbs;
                }
            }
            if (li + 1 < net.n_layers - 1) ring_ready = prefetch_layer(li + 1);
            __syncthreads();  // every wave has finished reading the planes for this layer

            // epilogue: registers 4q..4q+3 of a tile = features 8q + 4h + {0..3} -> 4 elements of one k-octet of the next layer
            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a >= ntw) continue;
                const int fbase = 32 * (t0 + a);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = fbase + 8 * q + 4 * h;
                        const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[a][pt][4 * q + e];
                        v[0] = __fmaf_rn(v[0], S::inv, bb.x); v[1] = __fmaf_rn(v[1], S::inv, bb.y);
                        v[2] = __fmaf_rn(v[2], S::inv, bb.z); v[3] = __fmaf_rn(v[3], S::inv, bb.w);
                        V4 oh, ol;
                        if (act) softplus100_4(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float u = v[e];
                            if (div) u = __fdiv_rn(u, sqrt2);
                            T hi, lo;
                            split_val<KIND>(u * S::xs, hi, lo);
                            oh[e] = hi;
                            ol[e] = lo;
                        }
                        const size_t o = (size_t)(f >> 3) * kOctE + (32 * pt + j) * 8 + 4 * h;
                        *reinterpret_cast<V4 *>(XH + o) = oh;
                        *reinterpret_cast<V4 *>(XL + o) = ol;
                    }
                }
            }
            __syncthreads();
        }
    }
}

struct PackSplitArgs {
    const float *W;
    int64_t ldw;
    int32_t out_dim, n_tiles, w0, w1, p16_0, nb;
    float scale0, scale1;
};

template <int KIND>
__global__ __launch_bounds__(256) void pack_layer_split_kernel(PackSplitArgs a, typename Split<KIND>::T *img) {
    // img[(((u*nb + t)*2 + part)*64 + l)*8 + jj] = part(scale * W[32u + (l&31)][16t + 8(l>>5) + jj])
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.n_tiles * a.nb * 512;
    if (d >= total) return;
    const int jj = d & 7, l = (d >> 3) & 63;
    const int64_t blk = d >> 9;
    const int t = (int)(blk % a.nb), u = (int)(blk / a.nb);
    const int row = 32 * u + (l & 31), kpos = 16 * t + 8 * (l >> 5) + jj;
    int col;
    float sc;
    if (kpos < a.p16_0) {
        col = kpos < a.w0 ? kpos : -1;
        sc = a.scale0;
    } else {
        const int kk = kpos - a.p16_0;
        col = kk < a.w1 ? a.w0 + kk : -1;
        sc = a.scale1;
    }
    const float w = (row < a.out_dim && col >= 0) ? __fmul_rn(a.W[(int64_t)row * a.ldw + col], sc) : 0.0f;
    typename Split<KIND>::T hi, lo;
    split_val<KIND>(w * Split<KIND>::ws, hi, lo);
    const int64_t o = (blk * 2) * 512 + l * 8 + jj;
    img[o] = hi;
    img[o + 512] = lo;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <int KIND, int FRAC>
static int launch_split(const HmLevels &lv, const SdfNet &net, const float *x, int64_t n, const float *table,
                        const float *B_fourier, float *out, int64_t out_stride, const int32_t *n_dev, int64_t run_min,
                        size_t lds, void *stream) {
    static thread_local bool attr_done = false;
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_split_kernel<KIND, FRAC>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    const int64_t tiles = (n + kPS - 1) / kPS;
    const int64_t grid = tiles < 256 ? tiles : 256;
    const int64_t big = (int64_t)1 << 62;
    hipLaunchKernelGGL((sdf_fwd_split_kernel<KIND, FRAC>), dim3((unsigned)grid), dim3(kTS), lds, as_stream(stream), lv,
                       net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, big);
    HM_CHECK_LAUNCH("hm_sdf_fwd_split");
    return HM_OK;
}

static int sdf_split_impl(const HmLevels &lv, const hm_mlp_desc *mlp, const float *x, int64_t emb_stride, int64_t n,
                          const float *table, const float *B_fourier, float *out, int64_t out_stride, int frac_mode,
                          const int32_t *n_dev, int64_t run_min, void *stream) {
    HM_CHECK_ARG(mlp, "hm_sdf_fwd_split: NULL descriptor");
    HM_CHECK_ARG(n >= 0, "hm_sdf_fwd_split: n < 0");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_sdf_fwd_split: bad frac_mode");
    HM_CHECK_ARG(mlp->n_layers >= 2 && mlp->n_layers <= HM_MAX_LAYERS, "hm_sdf_fwd_split: n_layers out of range");
    HM_CHECK_ARG(mlp->split_kind == HM_SPLIT_BF16X2 || mlp->split_kind == HM_SPLIT_F16X2,
                 "hm_sdf_fwd_split: the descriptor carries no split image (split_kind)");
    SdfNet net;
    net.n_layers = mlp->n_layers;
    net.beta = mlp->beta;
    net.emb_stride = emb_stride;
    const int emb_b16 = (lv.E + 15) / 16;
    net.emb_groups = emb_b16 * 4;      // k-groups of 4 = 2 octets per 16-block
    int x_groups = 0;
    for (int l = 0; l < mlp->n_layers; ++l) {
        const hm_mlp_layer &Ly = mlp->layer[l];
        HM_CHECK_ARG(Ly.w_packed && Ly.bias && Ly.w_packed_split, "hm_sdf_fwd_split: layer lacks the fp32 or the split image");
        HM_CHECK_ARG(Ly.n_tiles >= 1 && Ly.n_tiles <= 2 * kWS, "hm_sdf_fwd_split: layer wider than 512 features");
        HM_CHECK_ARG(Ly.seg_blocks16[0] >= 1 && Ly.seg_blocks16[1] >= 0, "hm_sdf_fwd_split: bad segment length");
        for (int s = 0; s < 2; ++s) {
            if (Ly.seg_blocks16[s] == 0) continue;
            if (Ly.seg_src[s] == 1) {
                HM_CHECK_ARG(Ly.seg_blocks16[s] == emb_b16, "hm_sdf_fwd_split: embedding segment must span ceil(E/16) blocks");
            } else {
                HM_CHECK_ARG(l > 0 && Ly.seg_blocks16[s] * 16 <= mlp->layer[l - 1].n_tiles * 32 &&
                                 Ly.seg_blocks16[s] * 16 >= mlp->layer[l - 1].out_dim,
                             "hm_sdf_fwd_split: hidden segment does not match the previous layer");
            }
        }
        x_groups = max(x_groups, Ly.n_tiles * 8);
        net.layer[l] = Ly;
    }
    const hm_mlp_layer &last = mlp->layer[mlp->n_layers - 1];
    HM_CHECK_ARG(last.seg_octets[1] == 0 && last.seg_src[0] == 0, "hm_sdf_fwd_split: the last layer must read the previous layer only");
    net.x_groups = x_groups;
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && out && (emb_stride > 0 || (table && B_fourier)), "hm_sdf_fwd_split: NULL pointer");
    const size_t lds = (size_t)(x_groups / 2 + net.emb_groups / 2) * kOctE * 2 * 2 + sizeof(float) * (kPS * 4 + kWS * kPS);
    HM_CHECK_ARG(lds <= 160 * 1024, "hm_sdf_fwd_split: network does not fit the 160 KB LDS tile");
    if (mlp->split_kind == HM_SPLIT_BF16X2) {
        if (frac_mode == HM_FRAC_REFERENCE)
            return launch_split<HM_SPLIT_BF16X2, HM_FRAC_REFERENCE>(lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, lds, stream);
        return launch_split<HM_SPLIT_BF16X2, HM_FRAC_TRILINEAR>(lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, lds, stream);
    }
    if (frac_mode == HM_FRAC_REFERENCE)
        return launch_split<HM_SPLIT_F16X2, HM_FRAC_REFERENCE>(lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, lds, stream);
    return launch_split<HM_SPLIT_F16X2, HM_FRAC_TRILINEAR>(lv, net, x, n, table, B_fourier, out, out_stride, n_dev, run_min, lds, stream);
}

}  // namespace

extern "C" {

int hm_pack_mlp_layer_split(const float *W, int64_t ldw, int out_dim, int seg_width0, int seg_width1, float seg_scale0,
                            float seg_scale1, int split_kind, void *w_packed_split, void *stream) {
    HM_CHECK_ARG(W && w_packed_split, "hm_pack_mlp_layer_split: NULL pointer");
    HM_CHECK_ARG(out_dim >= 1 && seg_width0 >= 1 && seg_width1 >= 0 && ldw >= seg_width0 + seg_width1,
                 "hm_pack_mlp_layer_split: bad shape");
    HM_CHECK_ARG(split_kind == HM_SPLIT_BF16X2 || split_kind == HM_SPLIT_F16X2, "hm_pack_mlp_layer_split: bad split_kind");
    PackSplitArgs a = {};
    a.W = W; a.ldw = ldw; a.out_dim = out_dim;
    a.n_tiles = (out_dim + 31) / 32;
    a.w0 = seg_width0; a.w1 = seg_width1;
    a.p16_0 = (seg_width0 + 15) / 16 * 16;
    a.nb = a.p16_0 / 16 + (seg_width1 + 15) / 16;
    a.scale0 = seg_scale0; a.scale1 = seg_scale1;
    const int64_t total = (int64_t)a.n_tiles * a.nb * 512;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (split_kind == HM_SPLIT_BF16X2)
        hipLaunchKernelGGL(pack_layer_split_kernel<HM_SPLIT_BF16X2>, dim3(grid), dim3(256), 0, as_stream(stream), a,
                           static_cast<__bf16 *>(w_packed_split));
    else
        hipLaunchKernelGGL(pack_layer_split_kernel<HM_SPLIT_F16X2>, dim3(grid), dim3(256), 0, as_stream(stream), a,
                           static_cast<_Float16 *>(w_packed_split));
    HM_CHECK_LAUNCH("hm_pack_mlp_layer_split");
    return HM_OK;
}

int hm_sdf_fwd_split(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n, const float *table,
                     const float *B_fourier, float *out, int64_t out_stride, int frac_mode, const int32_t *n_dev,
                     int64_t run_min, void *stream) {
    HM_CHECK_ARG(desc, "hm_sdf_fwd_split: NULL descriptor");
    return sdf_split_impl(desc->lv, mlp, x, 0, n, table, B_fourier, out, out_stride, frac_mode, n_dev, run_min, stream);
}

int hm_sdf_fwd_emb_split(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n,
                         float *out, int64_t out_stride, const int32_t *n_dev, int64_t run_min, void *stream) {
    HM_CHECK_ARG(emb_width >= 1 && emb_width <= 512 && emb_stride >= emb_width, "hm_sdf_fwd_emb_split: bad embedding width / stride");
    HmLevels lv = {};
    lv.L = 0; lv.F = 2; lv.E = emb_width;
    return sdf_split_impl(lv, mlp, emb, emb_stride, n, nullptr, nullptr, out, out_stride, HM_FRAC_REFERENCE, n_dev,
                          run_min, stream);
}

}  // extern "C"
