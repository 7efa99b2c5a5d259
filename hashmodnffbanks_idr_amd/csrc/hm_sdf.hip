// hm_sdf.hip - fused no-grad SDF network forward for gfx950: hash-grid encode + all MLP layers
// + tanh/Laplace clamp in ONE kernel; activations never leave the CU.
//
// Replaces ImplicitNetwork.forward under torch.no_grad()
// (reference: model/implicit_differentiable_renderer.py:89-113, density_net.py:20-30) - the
// callable RayTracing evaluates ~123 times per ray.
//
// Mapping (MI355X-first, not a GEMM-library call chain):
//   * a workgroup (8 waves, 512 threads, 1 per CU) owns a tile of 64 points for ALL layers;
//   * activations live in LDS as X[k/4][point][4] (fp32, 16-B k-groups), 128 KB + the embedding
//     (kept for the skip connection) 18 KB  -> 146 KB of the CU's 160 KB;
//   * each layer is D^T[feature, point] = W[feature, k] * X[k, point] on v_mfma_f32_32x32x2_f32
//     (exact fp32, k-ordered fma chain): W is the A operand streamed straight from L2 into
//     registers as coalesced 1-KB dwordx4 wave loads of a pre-packed image (each wave owns 64
//     output features, so weights are never shared between waves and LDS staging would be pure
//     overhead); X is the B operand read with conflict-free ds_read_b128;
//   * the accumulator tile has the point on the lane and 4 consecutive features per register
//     quad, i.e. exactly one 16-B k-group of the NEXT layer: the epilogue (bias, Softplus(100),
//     optional /sqrt(2)) writes it back with ds_write_b128 - no transpose, no shuffles.
// Small batches (the ray search issues ~30 dependent calls of 1...4096 points per iteration) use 16-, 8- and
// 4-point tiles so that every call is ONE tile per CU; the tile size is chosen on the device from the live
// point count (sdf_fwd_small_kernel -> sdf_m16_body / sdf_m8_body<8|4>).
#include "hm_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

#include "hm_sdf_common.h"
#include "hm_trace_dev.h"   // the ray search's state machine, run by the persistent march kernel below

// third value of the kernels' FRAC template parameter: `x` holds PRECOMPUTED embedding rows (hm_sdf_fwd_emb).  A template
// value rather than a run-time branch on net.emb_stride: with both input paths in one kernel body the register
// allocation of the hash-grid variant changed (178 -> 205 VGPRs, 33 -> 42 spilled SGPRs) and it lost 4 %.
constexpr int kFracEmb = 2;

#ifdef HM_SDF_PHASE_PROBE
// scripts/sdf_phase_probe.py: cycle stamps of one tile's phases (workgroup 0, wave 0, second tile), never in the product build
__device__ unsigned long long hm_probe_ts[128];
#define HM_PROBE(i_)                                                                                  \
    do {                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0 && it == 1 && (i_) < 120) hm_probe_ts[(i_)] = wall_clock64(); \
        if (blockIdx.x == 0 && threadIdx.x == 0 && it == 1 && ((i_) == 0 || (i_) == 100))                 \
            hm_probe_ts[120 + ((i_) != 0)] = clock64();   /* shader-clock cycles over the same tile */      \
    } while (0)
// every wave's own stamp (lane 0) at the end of layer 2's k-loop -> ts[64 + wave], after its first barrier -> ts[72 + wave]
#define HM_PROBE_WAVES(base_)                                                                                  \
    do {                                                                                                       \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && it == 1 && li == 2) hm_probe_ts[(base_) + (threadIdx.x >> 6)] = wall_clock64(); \
    } while (0)
// small-tile bodies: workgroup 0 / thread 0 stamps of its first tile -> ts[i]
#define HM_PROBE_S(i_)                                                                                \
    do {                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (i_) < 120) hm_probe_ts[(i_)] = wall_clock64();   \
    } while (0)
#else
#define HM_PROBE(i_) do { } while (0)
#define HM_PROBE_WAVES(base_) do { } while (0)
#define HM_PROBE_S(i_) do { } while (0)
#endif

constexpr int kPts = 64;        // points per workgroup tile
constexpr int kThreadsSdf = 512;
constexpr int kWaves = 8;
constexpr int kGroupFloats = kPts * 4;  // floats per k-group row of X: [point][4]

// Filler tiles.  A launch whose own points end in a partly filled last round can take the 64-point tiles that complete
// the round from a SECOND point set that has to be evaluated anyway (the ray search: the closest-approach scan, whose
// values nothing in the sampler's launches waits for).  Which of that set's tiles a launch takes follows from the
// device-side counts alone, the same way in every launch of the chain: a launch with TA own tiles on G workgroups takes
// pad = (G - TA mod G) mod G filler tiles if that many are still available and it runs on the 64-point kernel at all,
// none otherwise (its remainder then follows the half-tile rule).
struct SdfFill {               // resolved, per launch
    const float *x;
    float *out;
    int64_t n;                 // points of the filler set
    int64_t tile0;             // its first tile that is still free
};
struct SdfFillArgs {           // kernel argument (all zero: no filler)
    const float *x;
    float *out;
    const int32_t *n_dev;      // device-side point count of the filler set
    const int32_t *prev_dev;   // own-point count of the launch that took filler tiles before this one (or NULL)
    int32_t prev_grid;
    int32_t pad_;
};

__device__ __forceinline__ int64_t fill_quota(int64_t n_own, int64_t grid, int64_t avail, int64_t run_min) {
    if (n_own < run_min || grid <= 0) return 0;
    const int64_t ta = (n_own + 63) / 64;
    const int64_t pad = (grid - ta % grid) % grid;
    return (pad > 0 && avail >= pad) ? pad : 0;
}

// The 64-point tile loop of sdf_fwd_kernel / sdf_scan_secant_kernel.  DYN = false: the static schedule below
// (tile = round * grid + workgroup); DYN = true: the SAME tiles in the same enumeration, handed out by an atomic cursor
// (zero at launch) - workgroups that start late (they carried secant rays first) simply take fewer of them.
template <int FRAC, bool DYN>
__device__ __forceinline__ void sdf64_run(const HmLevels &lv, const SdfNet &net, const float *__restrict__ x, int64_t n,
                                          const float *__restrict__ table, const float *__restrict__ Bf,
                                          float *__restrict__ out, int64_t out_stride, int out_cols, float *lds,
                                          unsigned *cursor, const SdfFill &fb) {
    __shared__ unsigned s_next_tile;
    float *X = lds;
    float *EMB = lds + (size_t)net.x_groups * kGroupFloats;
    float *SX = EMB + (size_t)net.emb_groups * kGroupFloats;  // [64][3] raw points
    float *RED = SX + kPts * 4;                                // [8][64] cross-wave partial sums

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the weight pointers below stay in SGPRs)
    const int lane = tid & 63;
    const int j = lane & 31;  // point within a 32-point tile / feature row within a 32-feature tile
    const int h = lane >> 5;
    const int L = lv.L, F = lv.F;
    const int E = lv.E;
    // Tile schedule: `rounds` full rounds of 64-point tiles (tile = round * grid + workgroup), then ONE remainder tile
    // per workgroup.  When the remainder fits 32 points per workgroup it is cut into 32-point HALF tiles (the MFMAs of
    // the second point block are skipped), so the last round costs about half a round instead of a full one on part of
    // the chip: 204 800 points on 256 CUs are 12 rounds + 256 half tiles instead of 13 rounds on 128 CUs.  A point's
    // value does not depend on the tile it sits in (tests: permutation equivariance bit for bit).
    const int64_t G = gridDim.x;
    // filler tiles (static schedule only): TA own tiles, then tiles fb.tile0 .. of the second set up to the end of the round
    const int64_t TA = (n + kPts - 1) / kPts;
    const int64_t n_fill = DYN ? 0 : fill_quota(n, G, max((fb.n + kPts - 1) / kPts - fb.tile0, (int64_t)0), 0);
    const int64_t rounds = n_fill > 0 ? (TA + n_fill) / G : (n / kPts) / G;
    const int64_t rem_base = rounds * G * kPts;
    const int64_t rem_step = (n - rem_base <= G * 32) ? 32 : kPts;

    for (int64_t it = 0;; ++it) {
        int64_t itr = it, wg = blockIdx.x;
        if (DYN) {
            __syncthreads();   // every thread has read the previous hand-out
            if (tid == 0) s_next_tile = atomicAdd(cursor, 1u);
            __syncthreads();
            const int64_t u = s_next_tile;
            itr = u / G;
            wg = u - itr * G;
        }
        if (itr > rounds) break;
        int64_t base;
        int cnt;
        const float *__restrict__ xs = x;       // this tile's point set
        float *__restrict__ os = out;
        if (n_fill > 0) {                        // whole rounds: own tiles (the last one may be ragged), then filler tiles
            if (itr == rounds) break;
            const int64_t v = itr * G + wg;
            if (v < TA) {
                base = v * kPts;
                cnt = (int)min((int64_t)kPts, n - base);
            } else {
                base = (fb.tile0 + (v - TA)) * kPts;
                cnt = (int)min((int64_t)kPts, fb.n - base);
                xs = fb.x;
                os = fb.out;
            }
        } else if (itr < rounds) {
            base = (itr * G + wg) * kPts;
            cnt = kPts;
        } else {
            base = rem_base + wg * rem_step;
            if (base >= n) break;
            cnt = (int)min(rem_step, n - base);
        }
        const bool half = cnt <= 32;   // (also a ragged last tile of <= 32 points)
        HM_PROBE(0);
        __syncthreads();  // previous tile's output stage is done with X
        if (FRAC != kFracEmb && tid < kPts * 3) SX[tid] = (tid < cnt * 3) ? xs[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---------------- encode -> EMB[(e/4)][p][e%4] ------------------------------------
        if constexpr (FRAC == kFracEmb) {
            load_emb_tile(EMB, xs, net.emb_stride, base, cnt, E, net.emb_groups, kPts, kGroupFloats, tid, kThreadsSdf);
        } else {
            const int p = tid & (kPts - 1);
            const int grp = tid >> 6;  // 0..7
            const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
            auto put = [&](int e, float v) { EMB[(e >> 2) * kGroupFloats + p * 4 + (e & 3)] = v; };
            if (grp == 0) {
                put(0, x0); put(1, x1); put(2, x2);
                for (int e = E; e < net.emb_groups * 4; ++e) put(e, 0.0f);  // zero the k padding
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = grp; c < L; c += kWaves) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                put(3 + c, sn);
                put(3 + L + c, cs);
            }
            for (int l = grp; l < L; l += kWaves) {
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
                    if (w != 0.0f) {  // zero-weight corners add exactly 0 (reference mode: only corner 0 survives)
                        const uint32_t id = hm_mod_rows(hm_hash3(ux, uy, uz), lv.rows[l], lv.magic[l]);
                        for (int f = 0; f < F; ++f) acc[f] = __fadd_rn(acc[f], __fmul_rn(tl[(size_t)id * F + f], w));
                    }
                }
                for (int f = 0; f < F; ++f) put(3 + 2 * L + l * F + f, acc[f]);
            }
        }
        __syncthreads();
        HM_PROBE(1);

        // ---------------- layers ------------------------------------------------------------
        // weight ring (4 slots per feature tile) lives across layers: the first three octets of layer l+1 are
        // requested before layer l's epilogue (weights do not depend on the activations), so the stream does not
        // restart from an empty pipe behind the two barriers of every layer
        float4 r0[4], r1[4];
        bool ring_ready = false;
        const int lane16 = lane * 16;
        auto ldw = [&](const __amdgpu_buffer_rsrc_t &rs, int soff) -> float4 { return ld_w16(rs, lane16, soff); };
        auto prefetch64 = [&](int l) {
            const hm_mlp_layer &Lp = net.layer[l];
            const int nop = Lp.seg_octets[0] + Lp.seg_octets[1];
            const int ntp = max(0, min(2, Lp.n_tiles - 2 * wave));
            // buffer loads: descriptor (SGPRs) on the wave's first feature tile, lane * 16 as the one loop-invariant VGPR
            // offset, the octet / tile offset as the scalar offset - no per-load 64-bit address arithmetic on the VALU,
            // whose instructions are serial with the MFMAs of both waves on the SIMD
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(Lp.w_packed) + ((size_t)(2 * wave) * nop) * 256, 0, 0x7fffffff, 0x00020000);
            const int t1 = ntp > 1 ? nop * 1024 : 0;
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int off = min(st, nop - 1) * 1024;
                r0[st] = ldw(rs, off);
                r1[st] = ldw(rs, t1 + off);
            }
        };
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            const int n_oct = Ly.seg_octets[0] + Ly.seg_octets[1];
            if (li == net.n_layers - 1 && out_cols == 1) {
                // sdf-only: one output feature.  A 32-row MFMA tile would idle 7 of 8 waves for a full
                // layer time; instead every wave takes a k-slice of the dot product on the VALU
                // (row 0 of the packed image: tile 0, lanes 0 and 32).
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed);
                float part = 0.0f;
                int kg0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const float *src = (Ly.seg_src[seg] == 0) ? X : EMB;
                    const int ng = 2 * Ly.seg_octets[seg];
                    for (int kg = wave; kg < ng; kg += kWaves) {
                        const float4 xv = *reinterpret_cast<const float4 *>(src + kg * kGroupFloats + lane * 4);
                        const int kk = kg0 + kg;
                        const float4 wv = W0[(size_t)(kk >> 1) * 64 + 32 * (kk & 1)];
                        part = __fmaf_rn(xv.x, wv.x, part);
                        part = __fmaf_rn(xv.y, wv.y, part);
                        part = __fmaf_rn(xv.z, wv.z, part);
                        part = __fmaf_rn(xv.w, wv.w, part);
                    }
                    kg0 += ng;
                }
                RED[wave * kPts + lane] = part;
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int w8 = 0; w8 < kWaves; ++w8) sacc += RED[w8 * kPts + tid];
                    os[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt = Ly.n_tiles;
            const int t0 = 2 * wave;
            const int ntw = max(0, min(2, nt - t0));
            f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
            if (ntw > 0) {
                // weight stream: 4-deep register ring (3 octets = 6 KB per wave in flight), unconditional
                // clamped loads so that hipcc emits counted vmcnt waits instead of draining per octet
                const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(Ly.w_packed) + ((size_t)t0 * n_oct) * 256, 0, 0x7fffffff, 0x00020000);
                const int tA1 = ntw > 1 ? n_oct * 1024 : 0;     // byte offset of the wave's second feature tile
                const int no0 = Ly.seg_octets[0];
                const float *src0 = (Ly.seg_src[0] == 0) ? X : EMB;
                const float *src1 = (Ly.seg_src[1] == 0) ? X : EMB;
                if (!ring_ready) prefetch64(li);
                ring_ready = false;
                auto loadB = [&](int gg, float4 &b0, float4 &b1) {
                    const int gc = min(gg, n_oct - 1);
                    const float *src = (gc < no0) ? src0 + (2 * gc + h) * kGroupFloats
                                                  : src1 + (2 * (gc - no0) + h) * kGroupFloats;
                    b0 = *reinterpret_cast<const float4 *>(src + j * 4);
                    b1 = *reinterpret_cast<const float4 *>(src + (32 + j) * 4);
                };
                auto mfma16 = [&](const float4 &a0, const float4 &a1, const float4 &b0, const float4 &b1) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc11, 0, 0, 0);
                };
                // (full tiles request the B fragments of octet gg + 1 before octet gg's MFMAs - two register sets, loadB / mfma16:
                //  with the ds_reads behind the scheduling barrier of their own octet every octet starts with an exposed
                //  LDS round trip.  Alone this changed nothing - one wave's k-loop 62.4 -> 54.2 us per layer, but then 11 us
                //  at the barrier, scripts/sdf_phase_probe.py; on top of the buffer-load weight stream: 132.0 -> 134.2
                //  TFLOP/s)
                // whole groups of four octets run without an exit test: with a `break` inside the unrolled group
                // hipcc cannot count the loads in flight across the back edge and drains them (vmcnt(0)) at every
                // loop head; the 1-3 left-over octets are already in ring slots 0..2
                // half tile: the same stream, point block 0 only (8 MFMAs per octet)
                auto octet_h = [&](int gg, const float4 &a0, const float4 &a1) {
                    const float *src = (gg < no0) ? src0 + (2 * gg + h) * kGroupFloats
                                                  : src1 + (2 * (gg - no0) + h) * kGroupFloats;
                    const float4 b0 = *reinterpret_cast<const float4 *>(src + j * 4);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc00, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc10, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc00, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc10, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc00, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc10, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc00, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc10, 0, 0, 0);
                };
                const int n_full = n_oct & ~3;
                if (!half) {
                    float4 bA0, bA1, bB0, bB1;      // B fragments: even octets in set A, odd octets in set B
                    loadB(0, bA0, bA1);
                    for (int gg0 = 0; gg0 < n_full; gg0 += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int gg = gg0 + u;
                            {
                                const int off = min(gg + 3, n_oct - 1) * 1024;
                                r0[(u + 3) & 3] = ldw(rsA, off);
                                r1[(u + 3) & 3] = ldw(rsA, tA1 + off);
                            }
                            if (u & 1) loadB(gg + 1, bA0, bA1); else loadB(gg + 1, bB0, bB1);
                            __builtin_amdgcn_sched_barrier(0);   // (loads stay ahead of this octet's MFMAs; see the 16-point body)
                            if (u & 1) mfma16(r0[u], r1[u], bB0, bB1); else mfma16(r0[u], r1[u], bA0, bA1);
                        }
                    }
                    if (n_full + 0 < n_oct) { loadB(n_full + 1, bB0, bB1); mfma16(r0[0], r1[0], bA0, bA1); }
                    if (n_full + 1 < n_oct) { loadB(n_full + 2, bA0, bA1); mfma16(r0[1], r1[1], bB0, bB1); }
                    if (n_full + 2 < n_oct) mfma16(r0[2], r1[2], bA0, bA1);
                } else {
                    for (int gg0 = 0; gg0 < n_full; gg0 += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int gg = gg0 + u;
                            {
                                const int off = min(gg + 3, n_oct - 1) * 1024;
                                r0[(u + 3) & 3] = ldw(rsA, off);
                                r1[(u + 3) & 3] = ldw(rsA, tA1 + off);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            octet_h(gg, r0[u], r1[u]);
                        }
                    }
                    if (n_full + 0 < n_oct) octet_h(n_full + 0, r0[0], r1[0]);
                    if (n_full + 1 < n_oct) octet_h(n_full + 1, r0[1], r1[1]);
                    if (n_full + 2 < n_oct) octet_h(n_full + 2, r0[2], r1[2]);
                }
            }
            if (li + 1 < net.n_layers && !(li + 1 == net.n_layers - 1 && out_cols == 1) &&
                net.layer[li + 1].n_tiles - 2 * wave > 0) {
                prefetch64(li + 1);
                ring_ready = true;
            }
            HM_PROBE(2 + 4 * li);
            HM_PROBE_WAVES(64);
            __syncthreads();  // every wave has finished reading X / EMB for this layer
            HM_PROBE(3 + 4 * li);
            HM_PROBE_WAVES(72);

            // epilogue: registers 4q..4q+3 of a tile = features 8q+4h+{0..3} = one k-group of the next layer
            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
            auto store_tile = [&](const f32x16 &acc, int ft, int pt) {
                const int fbase = 32 * (t0 + ft);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = fbase + 8 * q + 4 * h;
                    const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                    float v0 = acc[4 * q + 0] + bb.x, v1 = acc[4 * q + 1] + bb.y, v2 = acc[4 * q + 2] + bb.z,
                          v3 = acc[4 * q + 3] + bb.w;
                    if (act) {
                        softplus100_4(v0, v1, v2, v3);
                    }
                    if (div) {
                        v0 = __fdiv_rn(v0, sqrt2); v1 = __fdiv_rn(v1, sqrt2); v2 = __fdiv_rn(v2, sqrt2);
                        v3 = __fdiv_rn(v3, sqrt2);
                    }
                    *reinterpret_cast<float4 *>(X + (f >> 2) * kGroupFloats + (32 * pt + j) * 4) =
                        make_float4(v0, v1, v2, v3);
                }
            };
            if (ntw > 0) {
                store_tile(acc00, 0, 0);
                if (!half) store_tile(acc01, 0, 1);
            }
            if (ntw > 1) {
                store_tile(acc10, 1, 0);
                if (!half) store_tile(acc11, 1, 1);
            }
            if (li == 0 && net.emb_groups > 0) {
                // the skip layer consumes cat[x, emb]/sqrt(2): rescale the kept embedding once, in place
                for (int i = tid; i < net.emb_groups * kGroupFloats; i += kThreadsSdf)
                    EMB[i] = __fdiv_rn(EMB[i], sqrt2);
            }
            HM_PROBE(4 + 4 * li);
            __syncthreads();
            HM_PROBE(5 + 4 * li);
        }
        HM_PROBE(100);

        // ---------------- output: X[(f/4)][p][f%4] -> out[p][f] -----------------------------
        const hm_mlp_layer &last = net.layer[net.n_layers - 1];
        if (out_cols != 1) {
            const int od = last.out_dim;
            for (int i = tid; i < cnt * od; i += kThreadsSdf) {
                const int p = i / od, f = i - p * od;
                float v = X[(f >> 2) * kGroupFloats + p * 4 + (f & 3)];
                if (f == 0) v = sdf_clamp(v, net.beta);
                os[(base + p) * out_stride + f] = v;
            }
        }
    }
}

template <int FRAC>
__global__ __launch_bounds__(kThreadsSdf, 2) void sdf_fwd_kernel(HmLevels lv, SdfNet net,
                                                                  const float *__restrict__ x, int64_t n,
                                                                  const float *__restrict__ table,
                                                                  const float *__restrict__ Bf,
                                                                  float *__restrict__ out, int64_t out_stride,
                                                                  int out_cols, const int32_t *__restrict__ n_dev, int64_t run_min,
                                                                  int64_t run_max, SdfFillArgs fa) {
    extern __shared__ __align__(16) float lds[];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));  // device-side point count (sync-free callers)
    if (n < run_min || n > run_max) return;          // the other tile-size kernel owns this batch size
    SdfFill fb = {nullptr, nullptr, 0, 0};
    if (fa.n_dev) {
        fb.x = fa.x;
        fb.out = fa.out;
        fb.n = max(*fa.n_dev, 0);
        if (fa.prev_dev)
            fb.tile0 = fill_quota(max(*fa.prev_dev, 0), fa.prev_grid, (fb.n + kPts - 1) / kPts, run_min);
    }
    sdf64_run<FRAC, false>(lv, net, x, n, table, Bf, out, out_stride, out_cols, lds, nullptr, fb);
}


// ------------------------------------------------------------------------------------------------
// 32-point tiles, TWO workgroups per CU (4 waves each, 74 KB of LDS each) - the throughput variant for big batches.
// One 64-point workgroup per CU leaves the matrix pipes idle whenever all of its waves sit in a layer's epilogue,
// at one of the two barriers per layer, in the encode phase or in the last layer's VALU dot (23 % of the tile time,
// 121 of 157 TFLOP/s).  Two independent workgroups drift out of phase and fill each other's bubbles; the price is the
// weight stream per CU (two tiles x 7.9 MB per ~410 us = 39 GB/s, well under the ~85 GB/s a CU draws from L2).
// Wave w owns feature tiles 4w .. 4w+3 (128 features) for all 32 points: per k octet 4 coalesced 1-KB A loads, one
// ds_read_b128 of B and 16 MFMAs - the same arithmetic intensity per LDS byte as the 64-point kernel.
constexpr int kPts32 = 32;
constexpr int kThreads32 = 256;
constexpr int kWaves32 = 4;
constexpr int kGroupFloats32 = kPts32 * 4;

template <int FRAC>
__global__ __launch_bounds__(kThreads32, 2) void sdf_fwd_p32_kernel(HmLevels lv, SdfNet net,
                                                                     const float *__restrict__ x, int64_t n,
                                                                     const float *__restrict__ table,
                                                                     const float *__restrict__ Bf,
                                                                     float *__restrict__ out, int64_t out_stride,
                                                                     int out_cols, const int32_t *__restrict__ n_dev,
                                                                     int64_t run_min, int64_t run_max) {
    extern __shared__ __align__(16) float lds[];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;
    float *X = lds;
    float *EMB = lds + (size_t)net.x_groups * kGroupFloats32;
    float *SX = EMB + (size_t)net.emb_groups * kGroupFloats32;  // [32][3] raw points (+ pad)
    float *RED = SX + kPts32 * 4;                                // [8][32] partial sums of the sdf-only last layer

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int j = lane & 31;  // point / feature row within a 32-feature tile
    const int h = lane >> 5;
    const int L = lv.L, F = lv.F;
    const int E = lv.E;
    const int64_t n_tiles = (n + kPts32 - 1) / kPts32;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kPts32;
        const int cnt = (int)min((int64_t)kPts32, n - base);
        __syncthreads();
        if (FRAC != kFracEmb && tid < kPts32 * 3) SX[tid] = (tid < cnt * 3) ? x[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---------------- encode -> EMB[(e/4)][p][e%4]: thread -> (point p, slot grp of 8) ----------------
        if constexpr (FRAC == kFracEmb) {
            load_emb_tile(EMB, x, net.emb_stride, base, cnt, E, net.emb_groups, kPts32, kGroupFloats32, tid, kThreads32);
        } else {
            const int p = tid & (kPts32 - 1);
            const int grp = tid >> 5;  // 0..7
            const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
            auto put = [&](int e, float v) { EMB[(e >> 2) * kGroupFloats32 + p * 4 + (e & 3)] = v; };
            if (grp == 0) {
                put(0, x0); put(1, x1); put(2, x2);
                for (int e = E; e < net.emb_groups * 4; ++e) put(e, 0.0f);
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = grp; c < L; c += 8) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                put(3 + c, sn);
                put(3 + L + c, cs);
            }
            for (int l = grp; l < L; l += 8) {
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
                for (int f = 0; f < F; ++f) put(3 + 2 * L + l * F + f, acc[f]);
            }
        }
        __syncthreads();

        // ---------------- layers ------------------------------------------------------------
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            const int n_oct = Ly.seg_octets[0] + Ly.seg_octets[1];
            if (li == net.n_layers - 1 && out_cols == 1) {
                // sdf-only last layer: VALU dot product; lane (point j, half h) of wave w walks k-groups 2w + h (mod 8)
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed);
                float part = 0.0f;
                int kg0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const float *src = (Ly.seg_src[seg] == 0) ? X : EMB;
                    const int ng = 2 * Ly.seg_octets[seg];
                    for (int kg = 2 * wave + h; kg < ng; kg += 2 * kWaves32) {
                        const float4 xv = *reinterpret_cast<const float4 *>(src + kg * kGroupFloats32 + j * 4);
                        const int kk = kg0 + kg;
                        const float4 wv = W0[(size_t)(kk >> 1) * 64 + 32 * (kk & 1)];
                        part = __fmaf_rn(xv.x, wv.x, part);
                        part = __fmaf_rn(xv.y, wv.y, part);
                        part = __fmaf_rn(xv.z, wv.z, part);
                        part = __fmaf_rn(xv.w, wv.w, part);
                    }
                    kg0 += ng;
                }
                RED[(2 * wave + h) * kPts32 + j] = part;
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int w8 = 0; w8 < 2 * kWaves32; ++w8) sacc += RED[w8 * kPts32 + tid];
                    out[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt = Ly.n_tiles;
            const int t0 = 4 * wave;
            const int ntw = max(0, min(4, nt - t0));
            f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
            if (ntw > 0) {
                const float4 *A0 = reinterpret_cast<const float4 *>(Ly.w_packed) + ((size_t)t0 * n_oct) * 64 + lane;
                const size_t ts = (size_t)n_oct * 64;
                const float4 *A1 = A0 + (ntw > 1 ? ts : 0);
                const float4 *A2 = A0 + (ntw > 2 ? 2 * ts : 0);
                const float4 *A3 = A0 + (ntw > 3 ? 3 * ts : 0);
                const int no0 = Ly.seg_octets[0];
                const float *src0 = (Ly.seg_src[0] == 0) ? X : EMB;
                const float *src1 = (Ly.seg_src[1] == 0) ? X : EMB;
                float4 r0[4], r1[4], r2[4], r3[4];
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    const size_t off = (size_t)min(st, n_oct - 1) * 64;
                    r0[st] = A0[off]; r1[st] = A1[off]; r2[st] = A2[off]; r3[st] = A3[off];
                }
                auto octet = [&](int gg, const float4 &a0, const float4 &a1, const float4 &a2, const float4 &a3) {
                    const float *src = (gg < no0) ? src0 + (2 * gg + h) * kGroupFloats32
                                                  : src1 + (2 * (gg - no0) + h) * kGroupFloats32;
                    const float4 b = *reinterpret_cast<const float4 *>(src + j * 4);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b.x, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, b.x, acc3, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b.y, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, b.y, acc3, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, b.z, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, b.z, acc3, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, b.w, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, b.w, acc3, 0, 0, 0);
                };
                // (whole groups of four octets without an exit test: see the 64-point kernel)
                const int n_full = n_oct & ~3;
                for (int gg0 = 0; gg0 < n_full; gg0 += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int gg = gg0 + u;
                        {
                            const size_t off = (size_t)min(gg + 3, n_oct - 1) * 64;
                            r0[(u + 3) & 3] = A0[off]; r1[(u + 3) & 3] = A1[off];
                            r2[(u + 3) & 3] = A2[off]; r3[(u + 3) & 3] = A3[off];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        octet(gg, r0[u], r1[u], r2[u], r3[u]);
                    }
                }
                if (n_full + 0 < n_oct) octet(n_full + 0, r0[0], r1[0], r2[0], r3[0]);
                if (n_full + 1 < n_oct) octet(n_full + 1, r0[1], r1[1], r2[1], r3[1]);
                if (n_full + 2 < n_oct) octet(n_full + 2, r0[2], r1[2], r2[2], r3[2]);
            }
            __syncthreads();  // every wave has finished reading X / EMB for this layer

            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
            auto store_tile = [&](const f32x16 &acc, int ft) {
                const int fbase = 32 * (t0 + ft);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = fbase + 8 * q + 4 * h;
                    const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                    float v0 = acc[4 * q + 0] + bb.x, v1 = acc[4 * q + 1] + bb.y, v2 = acc[4 * q + 2] + bb.z,
                          v3 = acc[4 * q + 3] + bb.w;
                    if (act) {
                        softplus100_4(v0, v1, v2, v3);
                    }
                    if (div) {
                        v0 = __fdiv_rn(v0, sqrt2); v1 = __fdiv_rn(v1, sqrt2); v2 = __fdiv_rn(v2, sqrt2);
                        v3 = __fdiv_rn(v3, sqrt2);
                    }
                    *reinterpret_cast<float4 *>(X + (f >> 2) * kGroupFloats32 + j * 4) = make_float4(v0, v1, v2, v3);
                }
            };
            if (ntw > 0) store_tile(acc0, 0);
            if (ntw > 1) store_tile(acc1, 1);
            if (ntw > 2) store_tile(acc2, 2);
            if (ntw > 3) store_tile(acc3, 3);
            if (li == 0 && net.emb_groups > 0) {
                for (int i = tid; i < net.emb_groups * kGroupFloats32; i += kThreads32)
                    EMB[i] = __fdiv_rn(EMB[i], sqrt2);
            }
            __syncthreads();
        }

        const hm_mlp_layer &last = net.layer[net.n_layers - 1];
        if (out_cols != 1) {
            const int od = last.out_dim;
            for (int i = tid; i < cnt * od; i += kThreads32) {
                const int p = i / od, f = i - p * od;
                float v = X[(f >> 2) * kGroupFloats32 + p * 4 + (f & 3)];
                if (f == 0) v = sdf_clamp(v, net.beta);
                out[(base + p) * out_stride + f] = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 16-point tile variant for SMALL batches (sphere-tracing rounds evaluate only 2 points per ray):
// with 64-point tiles a 4096-point call occupies 64 of the 256 CUs for a full 9-layer latency.
// Here a workgroup owns 16 points (v_mfma_f32_16x16x4_f32, 4 feature tiles of 16 per wave), so
// the same call spreads over 256 workgroups; the price is 4x the weight traffic per point, which
// L2 absorbs at this size.  Same LDS image X[k/4][point][4] (64 floats per k-group), same epilogue
// identity: lane (point j, quarter q) holds features 4q..4q+3 of a tile = one k-group.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPts16 = 16;
constexpr int kRing16 = 4;   // weight ring depth of the 16-point kernel (k-blocks)
constexpr int kGroupFloats16 = kPts16 * 4;

template <int FRAC>
__device__ __forceinline__ void sdf_m16_body(const HmLevels &lv, const SdfNet &net, const float *__restrict__ x,
                                             int64_t n, const float *__restrict__ table,
                                             const float *__restrict__ Bf, float *__restrict__ out,
                                             int64_t out_stride, int out_cols, float *lds, int64_t tile_first,
                                             int64_t tile_step) {
    const int emb_groups16 = ((lv.E + 15) / 16) * 4;
    // two activation images, used alternately (layer l reads one, its epilogue writes the other): no barrier between a
    // layer's k-loop and its epilogue - a wave that finishes early does not wait for the others before its Softplus
    float *X0 = lds;
    float *X1 = lds + (size_t)net.x_groups * kGroupFloats16;
    float *EMB = X1 + (size_t)net.x_groups * kGroupFloats16;
    float *SX = EMB + (size_t)emb_groups16 * kGroupFloats16;  // [16][3]
    float *RED = SX + kPts16 * 4;                              // [8][16]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: descriptors / scalar offsets of the weight stream)
    const int lane = tid & 63;
    const int lane16 = lane * 16;
    const int j = lane & 15;  // point
    const int q = lane >> 4;  // k quarter / feature quarter
    const int L = lv.L, F = lv.F, E = lv.E;
    const int64_t n_tiles = (n + kPts16 - 1) / kPts16;

    for (int64_t tile = tile_first; tile < n_tiles; tile += tile_step) {
        const int64_t base = tile * kPts16;
        const int cnt = (int)min((int64_t)kPts16, n - base);
        float *X = X0, *Xn = X1;     // X: the image the current layer reads; Xn: the one its epilogue fills
        __syncthreads();
        if (FRAC != kFracEmb && tid < kPts16 * 3) SX[tid] = (tid < cnt * 3) ? x[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---- encode: thread -> (point p, slot c); 32 slots cover channels / levels
        if constexpr (FRAC == kFracEmb) {
            load_emb_tile(EMB, x, net.emb_stride, base, cnt, E, emb_groups16, kPts16, kGroupFloats16, tid, kThreadsSdf);
        } else {
            const int p = tid & (kPts16 - 1);
            const int c0 = tid >> 4;  // 0..31
            const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
            auto put = [&](int e, float v) { EMB[(e >> 2) * kGroupFloats16 + p * 4 + (e & 3)] = v; };
            if (c0 == 0) {
                put(0, x0); put(1, x1); put(2, x2);
                for (int e = E; e < emb_groups16 * 4; ++e) put(e, 0.0f);
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = c0; c < L; c += 32) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                put(3 + c, sn);
                put(3 + L + c, cs);
            }
            for (int l = c0; l < L; l += 32) {
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
                for (int f = 0; f < F; ++f) put(3 + 2 * L + l * F + f, acc[f]);
            }
        }
        __syncthreads();

        float4 ring[kRing16][4];
        bool ring_ready = false;
        auto prefetch16 = [&](int l) {
            const hm_mlp_layer &Lp = net.layer[l];
            const int nbp = Lp.seg_blocks16[0] + Lp.seg_blocks16[1];
            const int ntp = max(0, min(4, Lp.n_tiles * 2 - 4 * wave));
            const __amdgpu_buffer_rsrc_t rp = w_rsrc(Lp.w_packed_m16 + ((size_t)(4 * wave) * nbp) * 256);
            const int ts = nbp * 1024;     // bytes to the next feature tile
            const int p1 = (1 < ntp ? 1 : 0) * ts, p2 = (2 < ntp ? 2 : 0) * ts, p3 = (3 < ntp ? 3 : 0) * ts;
#pragma unroll
            for (int st = 0; st < kRing16 - 1; ++st) {
                const int off = min(st, nbp - 1) * 1024;
                ring[st][0] = ld_w16(rp, lane16, off); ring[st][1] = ld_w16(rp, lane16, p1 + off);
                ring[st][2] = ld_w16(rp, lane16, p2 + off); ring[st][3] = ld_w16(rp, lane16, p3 + off);
            }
        };
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            const int nb = Ly.seg_blocks16[0] + Ly.seg_blocks16[1];  // 16-wide k blocks
            if (li == net.n_layers - 1 && out_cols == 1) {
                // sdf-only last layer: VALU dot product, lane (point j, quarter q) walks k-groups == q (mod 4)
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed_m16);
                float part = 0.0f;
                int t0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const float *src = (Ly.seg_src[seg] == 0) ? X : EMB;
                    const int nbs = Ly.seg_blocks16[seg];
                    for (int t = wave; t < nbs; t += kWaves) {
                        const float4 xv = *reinterpret_cast<const float4 *>(src + (4 * t + q) * kGroupFloats16 + j * 4);
                        const float4 wv = W0[(size_t)(t0 + t) * 64 + q * 16];
                        part = __fmaf_rn(xv.x, wv.x, part);
                        part = __fmaf_rn(xv.y, wv.y, part);
                        part = __fmaf_rn(xv.z, wv.z, part);
                        part = __fmaf_rn(xv.w, wv.w, part);
                    }
                    t0 += nbs;
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                if (q == 0) RED[wave * kPts16 + j] = part;
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int w8 = 0; w8 < kWaves; ++w8) sacc += RED[w8 * kPts16 + tid];
                    out[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt16 = Ly.n_tiles * 2;
            const int u0 = 4 * wave;
            const int ntw = max(0, min(4, nt16 - u0));
            f32x4 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (ntw > 0) {
                // Weight stream: a kRing16-deep register ring (kRing16-1 k-blocks of 4 KB per wave in flight).
                // A small batch is bound by the round trip of the packed weights (7.9 MB per workgroup, served
                // from the Infinity Cache at ~2 us), not by the MFMAs: bytes in flight per CU set the rate.
                // The first kRing16-1 blocks of a layer were requested before the previous layer's epilogue
                // (`prefetch16` below), so the pipeline does not drain at layer boundaries.
                // All loads are unconditional (clamped index) so hipcc emits counted vmcnt waits.
                const __amdgpu_buffer_rsrc_t rA = w_rsrc(Ly.w_packed_m16 + ((size_t)u0 * nb) * 256);
                const int tstride = nb * 1024;  // bytes to the next feature tile
                const int o1 = (1 < ntw ? 1 : 0) * tstride, o2 = (2 < ntw ? 2 : 0) * tstride,
                          o3 = (3 < ntw ? 3 : 0) * tstride;
                const int nb0 = Ly.seg_blocks16[0];
                const float *src0 = (Ly.seg_src[0] == 0) ? X : EMB;
                const float *src1 = (Ly.seg_src[1] == 0) ? X : EMB;
                if (!ring_ready) prefetch16(li);
                ring_ready = false;
                // B fragment of k-block t (LDS), requested one block ahead of its MFMAs (two registers sets, as in the
                // 64-point kernel)
                auto loadB16 = [&](int t) -> float4 {
                    const int tc = min(t, nb - 1);
                    const float *src = (tc < nb0) ? src0 + (4 * tc + q) * kGroupFloats16
                                                  : src1 + (4 * (tc - nb0) + q) * kGroupFloats16;
                    return *reinterpret_cast<const float4 *>(src + j * 4);
                };
                auto block16 = [&](const float4 &b, const float4 (&w)[4]) {
                    // component-major order: four INDEPENDENT accumulators between two uses of the same one
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].x, b.x, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].y, b.y, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].z, b.z, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a].w, b.w, acc[a], 0, 0, 0);
                };
                // whole groups of kRing16 blocks run without an exit test (with a `break` inside the unrolled group
                // hipcc drains all loads in flight - vmcnt(0) - at every loop head); the left-over blocks are
                // already in ring slots 0 .. kRing16-2
                const int nb_full = (nb / kRing16) * kRing16;
                static_assert(kRing16 % 2 == 0, "the B double buffer alternates with the parity of the block in the group");
                float4 bE = loadB16(0), bO;      // even / odd blocks
                for (int tt = 0; tt < nb_full; tt += kRing16) {
#pragma unroll
                    for (int u = 0; u < kRing16; ++u) {
                        const int t = tt + u;
                        {
                            const int off = min(t + kRing16 - 1, nb - 1) * 1024;
                            ring[(u + kRing16 - 1) % kRing16][0] = ld_w16(rA, lane16, off);
                            ring[(u + kRing16 - 1) % kRing16][1] = ld_w16(rA, lane16, o1 + off);
                            ring[(u + kRing16 - 1) % kRing16][2] = ld_w16(rA, lane16, o2 + off);
                            ring[(u + kRing16 - 1) % kRing16][3] = ld_w16(rA, lane16, o3 + off);
                        }
                        // keep the four loads HERE: left to itself hipcc sinks them below this block's MFMAs (their
                        // destination registers double as MFMA temporaries), which halves the bytes in flight
                        if (u & 1) bE = loadB16(t + 1); else bO = loadB16(t + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (u & 1) block16(bO, ring[u]); else block16(bE, ring[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < kRing16 - 1; ++u)      // (nb_full is even: block nb_full + u has the parity of u)
                    if (nb_full + u < nb) {
                        if (u & 1) { bE = loadB16(nb_full + u + 1); block16(bO, ring[u]); }
                        else { bO = loadB16(nb_full + u + 1); block16(bE, ring[u]); }
                    }
            }
            // request the next layer's first blocks now: they travel while this layer's epilogue and the two
            // barriers run (weights do not depend on the activations)
            if (li + 1 < net.n_layers && !(li + 1 == net.n_layers - 1 && out_cols == 1)) {
                if (2 * net.layer[li + 1].n_tiles - 4 * wave > 0) {
                    prefetch16(li + 1);
                    ring_ready = true;
                }
            }
            if (li == 0) __syncthreads();   // (layer 0 only: its epilogue rescales EMB in place, which every wave has read)
            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a >= ntw) continue;
                const int f = 16 * (u0 + a) + 4 * q;
                const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                float v0 = acc[a][0] + bb.x, v1 = acc[a][1] + bb.y, v2 = acc[a][2] + bb.z, v3 = acc[a][3] + bb.w;
                if (act) {
                    softplus100_4(v0, v1, v2, v3);
                }
                if (div) {
                    v0 = __fdiv_rn(v0, sqrt2); v1 = __fdiv_rn(v1, sqrt2); v2 = __fdiv_rn(v2, sqrt2);
                    v3 = __fdiv_rn(v3, sqrt2);
                }
                *reinterpret_cast<float4 *>(Xn + (f >> 2) * kGroupFloats16 + j * 4) = make_float4(v0, v1, v2, v3);
            }
            if (li == 0) {
                for (int i = tid; i < emb_groups16 * kGroupFloats16; i += kThreadsSdf) EMB[i] = __fdiv_rn(EMB[i], sqrt2);
            }
            __syncthreads();
            { float *t_ = X; X = Xn; Xn = t_; }
        }

        const hm_mlp_layer &last = net.layer[net.n_layers - 1];
        if (out_cols != 1) {
            const int od = last.out_dim;
            for (int i = tid; i < cnt * od; i += kThreadsSdf) {
                const int p = i / od, f = i - p * od;
                float v = X[(f >> 2) * kGroupFloats16 + p * 4 + (f & 3)];
                if (f == 0) v = sdf_clamp(v, net.beta);
                out[(base + p) * out_stride + f] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// 8-point tiles for the smallest batches (late sphere-tracing rounds, secant steps: <= 2048 live points).
// A 16-point tile costs the same MFMA time however few of its points are live (8 x 13.6 us per network on
// one CU); with 8 points per workgroup the matrix work halves and the call is bound by the weight stream
// alone.  v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 outer products: lane l = 16q + j supplies
//   A = W[16u + j][k]  (k in quarter q of the 16-wide k-block: the SAME packed image as the 16-point kernel)
//   B = X[point l & 3][k]
// and block (q, j / 4) accumulates, for its four features and four points, the partial sum over the k of
// quarter q; the four quarters are added with two lane exchanges at the end of the layer, which also hand
// each lane exactly one (feature quad, point) of the next layer's X image.
constexpr int kPts8 = 8;
constexpr int kRing8 = 5;    // weight ring depth of the 8-point body (k-blocks)
constexpr int kGroupFloats8 = kPts8 * 4;

// PTS = 8: two point groups per tile; PTS = 4: one (<= 1024 live points: the second group's MFMAs are skipped,
// the LDS image keeps its 8-point stride)
template <int FRAC, int PTS>
__device__ __forceinline__ void sdf_m8_body(const HmLevels &lv, const SdfNet &net, const float *__restrict__ x,
                                            int64_t n, const float *__restrict__ table,
                                            const float *__restrict__ Bf, float *__restrict__ out,
                                            int64_t out_stride, int out_cols, float *lds, int64_t tile_first,
                                            int64_t tile_step) {
    const int emb_groups16 = ((lv.E + 15) / 16) * 4;
    // two activation images, used alternately (layer l reads one, its epilogue writes the other): no barrier between a
    // layer's k-loop and its epilogue - a wave that finishes early does not wait for the others before its Softplus
    float *X0 = lds;
    float *X1 = lds + (size_t)net.x_groups * kGroupFloats8;
    float *EMB = X1 + (size_t)net.x_groups * kGroupFloats8;
    float *SX = EMB + (size_t)emb_groups16 * kGroupFloats8;  // [8][3] (+ pad)
    float *RED = SX + kPts8 * 4;                              // [8 waves][8 points]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: descriptors / scalar offsets of the weight stream)
    const int lane = tid & 63;
    const int lane16 = lane * 16;
    const int q = lane >> 4;         // k quarter of the A operand
    const int jj = (lane & 15) >> 2; // feature quad within the 16-feature tile
    const int p4 = lane & 3;         // point within a group of four
    const int L = lv.L, F = lv.F, E = lv.E;
    constexpr bool TWO = PTS == 8;
    constexpr int RD8 = kRing8;   // (one block deeper for the 4-point variant measured the same: 94 vs 93 us)
    const int64_t n_tiles = (n + PTS - 1) / PTS;

    for (int64_t tile = tile_first; tile < n_tiles; tile += tile_step) {
        const int64_t base = tile * PTS;
        const int cnt = (int)min((int64_t)PTS, n - base);
        HM_PROBE_S(0);
        float *X = X0, *Xn = X1;     // X: the image the current layer reads; Xn: the one its epilogue fills
        __syncthreads();
        if (FRAC != kFracEmb && tid < kPts8 * 3) SX[tid] = (tid < cnt * 3) ? x[base * 3 + tid] : 0.0f;
        __syncthreads();

        // ---- encode: thread -> (point p, slot c0); 64 slots cover channels / levels
        if constexpr (FRAC == kFracEmb) {
            load_emb_tile(EMB, x, net.emb_stride, base, cnt, E, emb_groups16, PTS, kGroupFloats8, tid, kThreadsSdf);
        } else {
            const int p = tid & (kPts8 - 1);
            const int c0 = tid >> 3;  // 0..63
            const float x0 = SX[p * 3], x1 = SX[p * 3 + 1], x2 = SX[p * 3 + 2];
            auto put = [&](int e, float v) { EMB[(e >> 2) * kGroupFloats8 + p * 4 + (e & 3)] = v; };
            if (c0 == 0) {
                put(0, x0); put(1, x1); put(2, x2);
                for (int e = E; e < emb_groups16 * 4; ++e) put(e, 0.0f);
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = c0; c < L; c += 64) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                put(3 + c, sn);
                put(3 + L + c, cs);
            }
            for (int l = c0; l < L; l += 64) {
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
                for (int f = 0; f < F; ++f) put(3 + 2 * L + l * F + f, acc[f]);
            }
        }
        __syncthreads();
        HM_PROBE_S(1);

        float4 ring[RD8][4];
        bool ring_ready = false;
        auto prefetch8 = [&](int l) {
            const hm_mlp_layer &Lp = net.layer[l];
            const int nbp = Lp.seg_blocks16[0] + Lp.seg_blocks16[1];
            const int ntp = max(0, min(4, Lp.n_tiles * 2 - 4 * wave));
            const __amdgpu_buffer_rsrc_t rp = w_rsrc(Lp.w_packed_m16 + ((size_t)(4 * wave) * nbp) * 256);
            const int ts = nbp * 1024;     // bytes to the next feature tile
            const int p1 = (1 < ntp ? 1 : 0) * ts, p2 = (2 < ntp ? 2 : 0) * ts, p3 = (3 < ntp ? 3 : 0) * ts;
#pragma unroll
            for (int st = 0; st < RD8 - 1; ++st) {
                const int off = min(st, nbp - 1) * 1024;
                ring[st][0] = ld_w16(rp, lane16, off); ring[st][1] = ld_w16(rp, lane16, p1 + off);
                ring[st][2] = ld_w16(rp, lane16, p2 + off); ring[st][3] = ld_w16(rp, lane16, p3 + off);
            }
        };
        for (int li = 0; li < net.n_layers; ++li) {
            const hm_mlp_layer &Ly = net.layer[li];
            const int nb = Ly.seg_blocks16[0] + Ly.seg_blocks16[1];  // 16-wide k blocks
            if (li == net.n_layers - 1 && out_cols == 1) {
                // sdf-only last layer: VALU dot product.  lane -> (point, k quarter, block parity)
                const float4 *W0 = reinterpret_cast<const float4 *>(Ly.w_packed_m16);
                const int pp = lane & 7, qq = (lane >> 3) & 3, th = lane >> 5;
                float part = 0.0f;
                int t0 = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const float *src = (Ly.seg_src[seg] == 0) ? X : EMB;
                    const int nbs = Ly.seg_blocks16[seg];
                    for (int t = 2 * wave + th; t < nbs; t += 2 * kWaves) {
                        const float4 xv = *reinterpret_cast<const float4 *>(src + (4 * t + qq) * kGroupFloats8 + pp * 4);
                        const float4 wv = W0[(size_t)(t0 + t) * 64 + qq * 16];
                        part = __fmaf_rn(xv.x, wv.x, part);
                        part = __fmaf_rn(xv.y, wv.y, part);
                        part = __fmaf_rn(xv.z, wv.z, part);
                        part = __fmaf_rn(xv.w, wv.w, part);
                    }
                    t0 += nbs;
                }
                part += __shfl_xor(part, 8);
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                if (lane < kPts8) RED[wave * kPts8 + lane] = part;
                __syncthreads();
                if (tid < cnt) {
                    float sacc = Ly.bias[0];
                    for (int w8 = 0; w8 < kWaves; ++w8) sacc += RED[w8 * kPts8 + tid];
                    out[(base + tid) * out_stride] = sdf_clamp(sacc, net.beta);
                }
                break;
            }
            const int nt16 = Ly.n_tiles * 2;
            const int u0 = 4 * wave;
            const int ntw = max(0, min(4, nt16 - u0));
            f32x4 acc0[4], acc1[4];   // points 0-3 / 4-7
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc0[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                acc1[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (ntw > 0) {
                const __amdgpu_buffer_rsrc_t rA = w_rsrc(Ly.w_packed_m16 + ((size_t)u0 * nb) * 256);
                const int tstride = nb * 1024;
                const int o1 = (1 < ntw ? 1 : 0) * tstride, o2 = (2 < ntw ? 2 : 0) * tstride,
                          o3 = (3 < ntw ? 3 : 0) * tstride;
                const int nb0 = Ly.seg_blocks16[0];
                const float *src0 = (Ly.seg_src[0] == 0) ? X : EMB;
                const float *src1 = (Ly.seg_src[1] == 0) ? X : EMB;
                if (!ring_ready) prefetch8(li);
                ring_ready = false;
                auto block8 = [&](int t, const float4 (&w)[4]) {
                    const float *src = (t < nb0) ? src0 + (4 * t + q) * kGroupFloats8
                                                 : src1 + (4 * (t - nb0) + q) * kGroupFloats8;
                    const float4 b0 = *reinterpret_cast<const float4 *>(src + p4 * 4);
                    const float4 b1 = TWO ? *reinterpret_cast<const float4 *>(src + (p4 + 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float4 av = w[a];
                        acc0[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, b0.x, acc0[a], 0, 0, 0);
                        if (TWO) acc1[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, b1.x, acc1[a], 0, 0, 0);
                        acc0[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, b0.y, acc0[a], 0, 0, 0);
                        if (TWO) acc1[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, b1.y, acc1[a], 0, 0, 0);
                        acc0[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, b0.z, acc0[a], 0, 0, 0);
                        if (TWO) acc1[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, b1.z, acc1[a], 0, 0, 0);
                        acc0[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, b0.w, acc0[a], 0, 0, 0);
                        if (TWO) acc1[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, b1.w, acc1[a], 0, 0, 0);
                    }
                };
                const int nb_full = (nb / RD8) * RD8;   // (no exit test inside the unrolled group: see the 16-point body)
                for (int tt = 0; tt < nb_full; tt += RD8) {
#pragma unroll
                    for (int u = 0; u < RD8; ++u) {
                        const int t = tt + u;
                        {
                            const int off = min(t + RD8 - 1, nb - 1) * 1024;
                            ring[(u + RD8 - 1) % RD8][0] = ld_w16(rA, lane16, off);
                            ring[(u + RD8 - 1) % RD8][1] = ld_w16(rA, lane16, o1 + off);
                            ring[(u + RD8 - 1) % RD8][2] = ld_w16(rA, lane16, o2 + off);
                            ring[(u + RD8 - 1) % RD8][3] = ld_w16(rA, lane16, o3 + off);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // loads stay ahead of this block's MFMAs
                        block8(t, ring[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < RD8 - 1; ++u)
                    if (nb_full + u < nb) block8(nb_full + u, ring[u]);
            }
            if (li + 1 < net.n_layers && !(li + 1 == net.n_layers - 1 && out_cols == 1)) {
                if (2 * net.layer[li + 1].n_tiles - 4 * wave > 0) {
                    prefetch8(li + 1);
                    ring_ready = true;
                }
            }
            HM_PROBE_S(2 + 4 * li);
            // add the four k quarters; lane q ends up with the complete sums of feature tile a == q
            // (exchange across lane bit 5 keeps tiles {0,1} or {2,3}, across bit 4 keeps one of the pair)
            f32x4 r0, r1;
            {
                const bool hi2 = (q & 2) != 0, hi1 = (q & 1) != 0;
                f32x4 k0[2], k1[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float keep0 = hi2 ? acc0[a + 2][r] : acc0[a][r], send0 = hi2 ? acc0[a][r] : acc0[a + 2][r];
                        const float keep1 = hi2 ? acc1[a + 2][r] : acc1[a][r], send1 = hi2 ? acc1[a][r] : acc1[a + 2][r];
                        k0[a][r] = keep0 + __shfl_xor(send0, 32);
                        k1[a][r] = TWO ? keep1 + __shfl_xor(send1, 32) : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float keep0 = hi1 ? k0[1][r] : k0[0][r], send0 = hi1 ? k0[0][r] : k0[1][r];
                    const float keep1 = hi1 ? k1[1][r] : k1[0][r], send1 = hi1 ? k1[0][r] : k1[1][r];
                    r0[r] = keep0 + __shfl_xor(send0, 16);
                    r1[r] = TWO ? keep1 + __shfl_xor(send1, 16) : 0.0f;
                }
            }
            HM_PROBE_S(3 + 4 * li);
            if (li == 0) __syncthreads();   // (layer 0 only: its epilogue rescales EMB in place, which every wave has read)
            HM_PROBE_S(4 + 4 * li);
            const bool act = Ly.activation != 0;
            const bool div = Ly.post_div_sqrt2 != 0;
            const float sqrt2 = 1.41421356237309515f;
            if (q < ntw) {
                const int f = 16 * (u0 + q) + 4 * jj;
                const float4 bb = *reinterpret_cast<const float4 *>(Ly.bias + f);
                float v[8] = {r0[0] + bb.x, r0[1] + bb.y, r0[2] + bb.z, r0[3] + bb.w,
                              r1[0] + bb.x, r1[1] + bb.y, r1[2] + bb.z, r1[3] + bb.w};
                if (act) {
                    softplus100_4(v[0], v[1], v[2], v[3]);
                    softplus100_4(v[4], v[5], v[6], v[7]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (div) v[i] = __fdiv_rn(v[i], sqrt2);
                float *dst = Xn + (f >> 2) * kGroupFloats8;
                *reinterpret_cast<float4 *>(dst + p4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
                if (TWO) *reinterpret_cast<float4 *>(dst + (p4 + 4) * 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (li == 0) {
                for (int i = tid; i < emb_groups16 * kGroupFloats8; i += kThreadsSdf) EMB[i] = __fdiv_rn(EMB[i], sqrt2);
            }
            __syncthreads();
            { float *t_ = X; X = Xn; Xn = t_; }
            HM_PROBE_S(5 + 4 * li);
        }
        HM_PROBE_S(100);

        const hm_mlp_layer &last = net.layer[net.n_layers - 1];
        if (out_cols != 1) {
            const int od = last.out_dim;
            for (int i = tid; i < cnt * od; i += kThreadsSdf) {
                const int p = i / od, f = i - p * od;
                float v = X[(f >> 2) * kGroupFloats8 + p * 4 + (f & 3)];
                if (f == 0) v = sdf_clamp(v, net.beta);
                out[(base + p) * out_stride + f] = v;
            }
        }
    }
}

// small batches: one launch, the tile size is chosen on the device from the live point count
template <int FRAC>
__global__ __launch_bounds__(kThreadsSdf, 1) void sdf_fwd_small_kernel(HmLevels lv, SdfNet net,
                                                                        const float *__restrict__ x, int64_t n,
                                                                        const float *__restrict__ table,
                                                                        const float *__restrict__ Bf,
                                                                        float *__restrict__ out, int64_t out_stride,
                                                                        int out_cols, const int32_t *__restrict__ n_dev,
                                                                        int64_t run_min, int64_t run_max, int64_t m8_max,
                                                                        int64_t m4_max) {
    extern __shared__ __align__(16) float lds[];
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (n < run_min || n > run_max) return;
    if (n <= m4_max)
        sdf_m8_body<FRAC, 4>(lv, net, x, n, table, Bf, out, out_stride, out_cols, lds, blockIdx.x, gridDim.x);
    else if (n <= m8_max)
        sdf_m8_body<FRAC, 8>(lv, net, x, n, table, Bf, out, out_stride, out_cols, lds, blockIdx.x, gridDim.x);
    else
        sdf_m16_body<FRAC>(lv, net, x, n, table, Bf, out, out_stride, out_cols, lds, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// Persistent TAIL of the sphere-tracing march (reference: model/ray_tracing.py:98-187).  The search needs
// 1 + sphere_tracing_iters rounds when no ray runs a line search and up to 1 + iters * (1 + line_step_iters) = 41 when
// one does in every iteration; the launch-per-round form must enqueue all 41 (SDF launch, update launch) pairs, and
// the 30 surplus pairs - empty launches, ~4.7 us of dispatch each - cost more than a live round.  Rays are independent,
// so the surplus rounds need no grid at all: after the guaranteed rounds this kernel gives every workgroup EIGHT rays
// (at most 16 pending points, in 16 slots of its own behind the compact list; cursor in LDS) and lets it carry them
// through all remaining rounds - evaluate the pending points on the 16-point body, run trace_advance_ray for its own
// rays, repeat until they are done.  Nothing is exchanged between workgroups, nothing synchronises the grid; a
// workgroup without stragglers (the usual case: every one) leaves at once.  The guaranteed rounds stay launches of
// their own: there the compact list balances the live points over all CUs (a march of the whole search inside this
// kernel was measured: -0.21 ms per step with every ray live to the end, +0.20 ms with a third of them live - the
// slowest workgroup sets the time).
// (the per-ray functions are out of line: inlined, their registers are live across the tile body)
__device__ __attribute__((noinline)) void march_adopt_ray(const TraceArgs &a, int64_t i, int32_t *cursor, int32_t slot_base) {
    // a ray that is not done has pending points in the compact list of round `first`: move them to this workgroup's slots
    const TraceWs &w = a.w;
    if (w.stage[i] == ST_DONE) return;
    const int32_t ss = w.slot_s[i], se = w.slot_e[i];
    w.slot_s[i] = ss >= 0 ? append_point(a, cursor, slot_base, i, w.t_s[i]) : -1;
    w.slot_e[i] = se >= 0 ? append_point(a, cursor, slot_base, i, w.t_e[i]) : -1;
}
__device__ __attribute__((noinline)) void march_advance_ray(const TraceArgs &a, int64_t ray, int32_t *cursor, int32_t slot_base) {
    trace_advance_ray(a, ray, cursor, slot_base);
}

template <int FRAC>
__global__ __launch_bounds__(kThreadsSdf, 1) void trace_march_tail_kernel(HmLevels lv, SdfNet net,
                                                                           const float *__restrict__ table,
                                                                           const float *__restrict__ Bf, TraceArgs a,
                                                                           int first, int rounds, int lds_floats, int body16) {
    extern __shared__ __align__(16) float lds[];
    if (a.w.cnt[C_ROUND0 + first] == 0) return;     // no ray has a pending point: every state machine has finished
    int32_t *ctl = reinterpret_cast<int32_t *>(lds + lds_floats);   // [0] cursor of the round being filled
    const int tid = threadIdx.x;
    const int32_t slot_base = (int32_t)a.w.cap + (int32_t)blockIdx.x * 16;   // (second region of pts / vals)
    const int64_t ray = (int64_t)blockIdx.x * 8 + tid;
    const bool mine = tid < 8 && ray < a.n;
    const float *xp = a.w.pts + (int64_t)slot_base * 3;
    float *vp = a.w.vals + slot_base;
    if (tid == 0) ctl[0] = 0;
    __syncthreads();
    if (mine) march_adopt_ray(a, ray, ctl, slot_base);
    __syncthreads();
    for (int r = first; r < rounds; ++r) {
        const int n_loc = ctl[0];
        if (n_loc == 0) break;      // (uniform) every ray of this workgroup is done
        // the search's evaluation count (statistics; round `first` was counted when its points were appended)
        if (tid == 0 && r > first) atomicAdd(a.w.cnt + C_ROUND0 + r, n_loc);
        // the small-tile body that fits the workgroup's OWN pending points: a straggler's rounds cost 83 instead of 130 us
        // (600 training steps, r3aq: 7.00 ms per step with every tail round on the 16-point body against 6.43 ms with the
        // launch-per-round form, whose compact list runs such rounds on 4-point tiles).  body16 (tile_points = 16): the
        // 16-point body always - bit-identical to hm_sdf_fwd with 16-point tiles, how the tests compare this kernel
        // with the generic tracer.  (With the three bodies inlined the kernel spills ~60 VGPRs, none inside an MFMA loop.)
        if (n_loc <= 4 && !body16)
            sdf_m8_body<FRAC, 4>(lv, net, xp, n_loc, table, Bf, vp, 1, 1, lds, 0, 1 << 30);
        else if (n_loc <= 8 && !body16)
            sdf_m8_body<FRAC, 8>(lv, net, xp, n_loc, table, Bf, vp, 1, 1, lds, 0, 1 << 30);
        else
            sdf_m16_body<FRAC>(lv, net, xp, n_loc, table, Bf, vp, 1, 1, lds, 0, 1 << 30);
        __syncthreads();            // the values (global stores of this workgroup) are visible to its threads
        if (tid == 0) ctl[0] = 0;
        __syncthreads();
        if (mine) march_advance_ray(a, ray, ctl, slot_base);
        __syncthreads();
    }
}

// diagnostic (hm_diag_mfma_f32_stream): the MFMA stream of the 64-point kernel's k-loop without its memory traffic
__global__ __launch_bounds__(kThreadsSdf, 2) void mfma_f32_stream_kernel(int iters, float *out) {
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float a0 = (float)(threadIdx.x & 63) * 1e-3f, a1 = a0 + 0.5f, b0 = 1.0f - a0, b1 = 0.25f + a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
    }
    out[(size_t)blockIdx.x * kThreadsSdf + threadIdx.x] = acc00[0] + acc01[5] + acc10[9] + acc11[15];
}

// ---------------------------------------------------------------------------------------------------------
// Persistent secant refinement (reference: model/ray_tracing.py:251-268): the secant rays are a compact list, a tile
// of them (4 / 8 / 16 by the list's length - the rule of sdf_fwd_small_kernel, so the values are those of one
// hm_sdf_fwd launch per iteration) stays with its workgroup through all iterations: evaluate the tile's midpoints, run
// secant_advance_ray for its rays, repeat.  Nothing is exchanged between workgroups: 8 (SDF launch, update launch) pairs
// become one launch.
__device__ __attribute__((noinline)) void secant_step_ray(const TraceArgs &a, int64_t q, int last) {
    secant_advance_ray(a, q, last);
}

template <int FRAC>
__device__ __forceinline__ void secant_role(const HmLevels &lv, const SdfNet &net, const float *__restrict__ table,
                                            const float *__restrict__ Bf, const TraceArgs &a, int n_iters, int64_t n,
                                            int pts, float *lds) {
    const int64_t n_tiles = (n + pts - 1) / pts;
    for (int it = 0; it < n_iters; ++it) {
        if (pts == 4)
            sdf_m8_body<FRAC, 4>(lv, net, a.w.pts, n, table, Bf, a.w.vals, 1, 1, lds, blockIdx.x, gridDim.x);
        else if (pts == 8)
            sdf_m8_body<FRAC, 8>(lv, net, a.w.pts, n, table, Bf, a.w.vals, 1, 1, lds, blockIdx.x, gridDim.x);
        else
            sdf_m16_body<FRAC>(lv, net, a.w.pts, n, table, Bf, a.w.vals, 1, 1, lds, blockIdx.x, gridDim.x);
        __syncthreads();       // the tile's values (global stores of this workgroup) are visible to its threads
        for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const int64_t q = tile * pts + threadIdx.x;
            if ((int)threadIdx.x < pts && q < n) secant_step_ray(a, q, it == n_iters - 1 ? 1 : 0);
        }
        __syncthreads();       // the next iteration's points are written
    }
}

template <int FRAC>
__global__ __launch_bounds__(kThreadsSdf, 1) void trace_secant_kernel(HmLevels lv, SdfNet net,
                                                                       const float *__restrict__ table,
                                                                       const float *__restrict__ Bf, TraceArgs a,
                                                                       int n_iters, int64_t m8_max, int64_t m4_max) {
    extern __shared__ __align__(16) float lds[];
    const int64_t n = a.w.cnt[C_NSEC];
    if (n <= 0) return;
    const int pts = n <= m4_max ? 4 : (n <= m8_max ? 8 : 16);
    secant_role<FRAC>(lv, net, table, Bf, a, n_iters, n, pts, lds);
}

// ---------------------------------------------------------------------------------------------------------
// Closest-approach scan + secant refinement in ONE launch (training, tile size left to the library).  The secant
// iterations are a chain of eight dependent small-tile evaluations that keeps a few dozen workgroups busy for most of a
// millisecond while the rest of the chip idles; the closest-approach scan of the mask-loss rays (ray_tracing.py:71-92,
// ~77 k points at the bench workload) depends on neither the sampler nor the secant.  Here the workgroups that own secant
// tiles run that chain first (trace_secant_kernel's code), every other workgroup starts on the scan's 64-point tiles at
// once, and the tiles are handed out by an atomic cursor, so the late workgroups simply take fewer.  Values: the scan's
// are those of sdf_fwd_kernel (same tiles); the secant uses 16-point tiles when the scan is long enough to hide their
// latency (a 16-point tile does four times the rays of a 4-point one for 1.6x the time: least chip time), and the list
// length rule of trace_secant_kernel otherwise.
template <int FRAC>
__global__ __launch_bounds__(kThreadsSdf, 2) void sdf_scan_secant_kernel(HmLevels lv, SdfNet net,
                                                                          const float *__restrict__ table,
                                                                          const float *__restrict__ Bf, TraceArgs a,
                                                                          int n_iters, int64_t m8_max, int64_t m4_max,
                                                                          int64_t scan_off, int64_t scan_min,
                                                                          int64_t scan_hide, int c_prev1, int grid1,
                                                                          int c_prev2, int grid2) {
    extern __shared__ __align__(16) float lds[];
    const int64_t n_scan = max(a.w.cnt[C_NSEL_PTS], 0);
    const int64_t n_sec = a.w.cnt[C_NSEC];
    if (n_sec > 0 && n_iters > 0) {
        const int pts = n_scan >= scan_hide ? 16 : (n_sec <= m4_max ? 4 : (n_sec <= m8_max ? 8 : 16));
        secant_role<FRAC>(lv, net, table, Bf, a, n_iters, n_sec, pts, lds);
    }
    if (n_scan < scan_min) return;        // (a short scan is the small-tile launch's job)
    // the scan's first tiles completed the last rounds of the sampler's launches (filler tiles, see fill_quota)
    const int64_t tb = (n_scan + kPts - 1) / kPts;
    int64_t done = c_prev1 >= 0 ? fill_quota(max(a.w.cnt[c_prev1], 0), grid1, tb, scan_min) : 0;
    if (c_prev2 >= 0) done += fill_quota(max(a.w.cnt[c_prev2], 0), grid2, tb - done, scan_min);
    const int64_t skip = done * kPts;
    if (skip >= n_scan) return;
    const SdfFill none = {nullptr, nullptr, 0, 0};
    sdf64_run<FRAC, true>(lv, net, a.w.pts + (scan_off + skip) * 3, n_scan - skip, table, Bf, a.w.vals + scan_off + skip, 1, 1,
                          lds, reinterpret_cast<unsigned *>(a.w.cnt + C_TILE_CURSOR), none);
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int hm_diag_mfma_f32_stream(int workgroups, int iters, float *out, void *stream) {
    HM_CHECK_ARG(workgroups >= 1 && workgroups <= 65535 && iters >= 1 && out, "hm_diag_mfma_f32_stream: bad argument");
    hipLaunchKernelGGL(mfma_f32_stream_kernel, dim3((unsigned)workgroups), dim3(kThreadsSdf), 0, as_stream(stream), iters, out);
    HM_CHECK_LAUNCH("hm_diag_mfma_f32_stream");
    return HM_OK;
}

#ifdef HM_SDF_PHASE_PROBE
HM_API int hm_probe_read(unsigned long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_probe_ts), sizeof(unsigned long long) * (size_t)(n < 128 ? n : 128)) == hipSuccess ? 0 : -1;
}
#endif

static int sdf_fwd_impl(const HmLevels &lv, const hm_mlp_desc *mlp, const float *x, int64_t emb_stride, int64_t n,
                        const float *table, const float *B_fourier, float *out, int64_t out_stride, int out_cols,
                        int frac_mode, int tile_points, const int32_t *n_dev, int max_workgroups, void *stream,
                        const SdfFillArgs *fill = nullptr, int *grid64_out = nullptr);

// hm_sdf_fwd for the ray search's sampler launches (not exported): the 64-point launch completes its last round with
// tiles of a second point set (fill_quota; x_fill / out_fill / n_fill_dev: the closest-approach scan's points, values and
// device-side count; prev_dev / prev_grid: own-point count and grid of the launch that took filler tiles before this one).
// *grid64_out = the 64-point launch's grid (0: there was none), for the next launch of the chain.
int hm_sdf_fwd_fill(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n, const float *table,
                    const float *B_fourier, float *out, int frac_mode, const int32_t *n_dev, const float *x_fill,
                    float *out_fill, const int32_t *n_fill_dev, const int32_t *prev_dev, int prev_grid, int *grid64_out,
                    void *stream) {
    HM_CHECK_ARG(desc && mlp, "hm_sdf_fwd_fill: NULL descriptor");
    HM_CHECK_ARG(n == 0 || (table && B_fourier && n_dev && x_fill && out_fill && n_fill_dev), "hm_sdf_fwd_fill: NULL pointer");
    SdfFillArgs fa = {x_fill, out_fill, n_fill_dev, prev_grid > 0 ? prev_dev : nullptr, prev_grid, 0};
    if (grid64_out) *grid64_out = 0;
    return sdf_fwd_impl(desc->lv, mlp, x, 0, n, table, B_fourier, out, 1, 1, frac_mode, 0, n_dev, 0, stream, &fa, grid64_out);
}

int hm_sdf_fwd(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n, const float *table,
               const float *B_fourier, float *out, int64_t out_stride, int out_cols, int frac_mode, int tile_points,
               const int32_t *n_dev, int max_workgroups, void *stream) {
    HM_CHECK_ARG(desc && mlp, "hm_sdf_fwd: NULL descriptor");
    HM_CHECK_ARG(n == 0 || (table && B_fourier), "hm_sdf_fwd: NULL pointer");
    return sdf_fwd_impl(desc->lv, mlp, x, 0, n, table, B_fourier, out, out_stride, out_cols, frac_mode, tile_points,
                        n_dev, max_workgroups, stream);
}

int hm_sdf_fwd_emb(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n, float *out,
                   int64_t out_stride, int out_cols, int tile_points, const int32_t *n_dev, int max_workgroups,
                   void *stream) {
    HM_CHECK_ARG(mlp, "hm_sdf_fwd_emb: NULL descriptor");
    HM_CHECK_ARG(emb_width >= 1 && emb_width <= 512 && emb_stride >= emb_width, "hm_sdf_fwd_emb: bad embedding width / stride");
    HmLevels lv = {};
    lv.L = 0; lv.F = 2; lv.E = emb_width;
    return sdf_fwd_impl(lv, mlp, emb, emb_stride, n, nullptr, nullptr, out, out_stride, out_cols, HM_FRAC_REFERENCE,
                        tile_points, n_dev, max_workgroups, stream);
}

// kernel-side image of the network descriptor (+ the checks every fused SDF launch makes on it)
static int sdf_net_from_desc(const HmLevels &lv, const hm_mlp_desc *mlp, int64_t emb_stride, SdfNet &net, bool &have16) {
    HM_CHECK_ARG(mlp && mlp->n_layers >= 1 && mlp->n_layers <= HM_MAX_LAYERS, "hm_sdf_fwd: n_layers out of range");
    net.n_layers = mlp->n_layers;
    net.beta = mlp->beta;
    net.emb_stride = emb_stride;
    const int emb_oct = (lv.E + 7) / 8;
    const int emb_b16 = (lv.E + 15) / 16;
    net.emb_groups = emb_oct * 2;
    int x_groups = 0;
    have16 = true;
    for (int l = 0; l < mlp->n_layers; ++l) {
        const hm_mlp_layer &Ly = mlp->layer[l];
        HM_CHECK_ARG(Ly.w_packed && Ly.bias, "hm_sdf_fwd: layer has NULL weights/bias");
        HM_CHECK_ARG(Ly.n_tiles >= 1 && Ly.n_tiles <= 2 * kWaves, "hm_sdf_fwd: layer wider than 512 features");
        HM_CHECK_ARG(Ly.out_dim >= 1 && Ly.out_dim <= Ly.n_tiles * 32, "hm_sdf_fwd: out_dim / n_tiles mismatch");
        HM_CHECK_ARG(Ly.seg_octets[0] >= 1 && Ly.seg_octets[1] >= 0, "hm_sdf_fwd: bad segment length");
        if (!Ly.w_packed_m16) have16 = false;
        for (int s = 0; s < 2; ++s) {
            if (Ly.seg_octets[s] == 0) continue;
            if (Ly.seg_src[s] == 1) {
                HM_CHECK_ARG(Ly.seg_octets[s] == emb_oct, "hm_sdf_fwd: embedding segment must span ceil(E/8) octets");
                HM_CHECK_ARG(!Ly.w_packed_m16 || Ly.seg_blocks16[s] == emb_b16,
                             "hm_sdf_fwd: embedding segment must span ceil(E/16) 16-blocks");
            } else {
                HM_CHECK_ARG(Ly.seg_src[s] == 0 && l > 0, "hm_sdf_fwd: layer 0 must read the embedding");
                HM_CHECK_ARG(Ly.seg_octets[s] * 8 <= mlp->layer[l - 1].n_tiles * 32,
                             "hm_sdf_fwd: layer reads more inputs than the previous layer produces");
                HM_CHECK_ARG(Ly.seg_octets[s] * 8 >= mlp->layer[l - 1].out_dim,
                             "hm_sdf_fwd: layer reads fewer inputs than the previous layer produces");
                HM_CHECK_ARG(!Ly.w_packed_m16 || (Ly.seg_blocks16[s] * 16 <= mlp->layer[l - 1].n_tiles * 32 &&
                                                  Ly.seg_blocks16[s] * 16 >= mlp->layer[l - 1].out_dim),
                             "hm_sdf_fwd: 16-block segment length does not match the previous layer");
            }
        }
        x_groups = max(x_groups, Ly.n_tiles * 8);
        net.layer[l] = Ly;
    }
    net.x_groups = x_groups;
    return HM_OK;
}

// internal entry of the ray search (hm_trace.hip, not exported): rounds [first, rounds) of the sphere-tracing march as
// ONE launch (trace_march_tail_kernel).  `trace_args` = a TraceArgs of hm_trace_dev.h.
int hm_trace_march_tail(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table, const float *B_fourier,
                        int frac_mode, int body16, const void *trace_args, int first, int rounds, void *stream) {
    HM_CHECK_ARG(desc && mlp && table && B_fourier && trace_args, "hm_trace_march_tail: NULL argument");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_trace_march_tail: bad frac_mode");
    HM_CHECK_ARG(first >= 1 && first < 64 && rounds <= 64, "hm_trace_march_tail: bad round range");
    const TraceArgs &a = *static_cast<const TraceArgs *>(trace_args);
    SdfNet net;
    bool have16 = false;
    const int rc = sdf_net_from_desc(desc->lv, mlp, 0, net, have16);
    if (rc != HM_OK) return rc;
    HM_CHECK_ARG(have16, "hm_trace_march_tail: needs w_packed_m16 in every layer");
    if (a.n == 0 || first >= rounds) return HM_OK;
    HM_CHECK_ARG(a.w.cap >= ((a.n + 7) / 8) * 16, "hm_trace_march_tail: point buffer too small");
    const int emb_b16 = (desc->lv.E + 15) / 16;
    const int lds_floats = (2 * net.x_groups + emb_b16 * 4) * kGroupFloats16 + kPts16 * 4 + kWaves * kPts16;   // two activation images
    const size_t lds = sizeof(float) * (size_t)lds_floats + 64;
    HM_CHECK_ARG(lds <= 96 * 1024, "hm_trace_march_tail: network does not fit the 16-point LDS tile");
    static thread_local bool attr_tail_done = false;
    if (!attr_tail_done) {  // opt in to >64 KB dynamic LDS once (not a stream operation)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_march_tail_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_march_tail_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_tail_done = true;
    }
    const unsigned grid = (unsigned)((a.n + 7) / 8);
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(trace_march_tail_kernel<HM_FRAC_REFERENCE>, dim3(grid), dim3(kThreadsSdf), lds,
                           as_stream(stream), desc->lv, net, table, B_fourier, a, first, rounds, lds_floats, body16);
    else
        hipLaunchKernelGGL(trace_march_tail_kernel<HM_FRAC_TRILINEAR>, dim3(grid), dim3(kThreadsSdf), lds,
                           as_stream(stream), desc->lv, net, table, B_fourier, a, first, rounds, lds_floats, body16);
    HM_CHECK_LAUNCH("hm_trace_march_tail");
    return HM_OK;
}

// internal entry of the ray search (hm_trace.hip, not exported): all secant iterations as ONE launch
// (trace_secant_kernel).  tile_points 0 = the tile size follows the list's length as in hm_sdf_fwd; 4 / 8 / 16 = fixed.
int hm_trace_secant_persistent(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table, const float *B_fourier,
                               int frac_mode, int tile_points, const void *trace_args, int n_iters, void *stream) {
    HM_CHECK_ARG(desc && mlp && table && B_fourier && trace_args, "hm_trace_secant_persistent: NULL argument");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_trace_secant_persistent: bad frac_mode");
    HM_CHECK_ARG(tile_points == 0 || tile_points == 4 || tile_points == 8 || tile_points == 16,
                 "hm_trace_secant_persistent: tile_points must be 0, 4, 8 or 16");
    const TraceArgs &a = *static_cast<const TraceArgs *>(trace_args);
    SdfNet net;
    bool have16 = false;
    const int rc = sdf_net_from_desc(desc->lv, mlp, 0, net, have16);
    if (rc != HM_OK) return rc;
    HM_CHECK_ARG(have16, "hm_trace_secant_persistent: needs w_packed_m16 in every layer");
    if (a.n == 0 || n_iters <= 0) return HM_OK;
    const int64_t kBig = (int64_t)1 << 62;
    int64_t m8_max = 0, m4_max = 0;      // (the thresholds of hm_sdf_fwd's small-tile launch)
    if (tile_points == 0) { m8_max = 2048; m4_max = 1024; }
    else if (tile_points == 8) m8_max = kBig;
    else if (tile_points == 4) { m8_max = kBig; m4_max = kBig; }
    const int emb_b16 = (desc->lv.E + 15) / 16;
    const size_t lds = sizeof(float) * ((size_t)(2 * net.x_groups + emb_b16 * 4) * kGroupFloats16 + kPts16 * 4 +
                                        kWaves * kPts16);
    HM_CHECK_ARG(lds <= 96 * 1024, "hm_trace_secant_persistent: network does not fit the 16-point LDS tile");
    static thread_local bool attr_sec_done = false;
    if (!attr_sec_done) {  // opt in to >64 KB dynamic LDS once (not a stream operation)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_secant_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_secant_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_sec_done = true;
    }
    // workgroups for the longest list the call can see (every ray a secant ray), one tile each up to one per CU
    const int64_t nmax = a.n;
    int64_t tiles = m8_max >= nmax ? 0 : (nmax + kPts16 - 1) / kPts16;
    if (m8_max > 0) tiles = max(tiles, ((nmax < m8_max ? nmax : m8_max) + kPts8 - 1) / kPts8);
    if (m4_max > 0) tiles = max(tiles, ((nmax < m4_max ? nmax : m4_max) + 3) / 4);
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(trace_secant_kernel<HM_FRAC_REFERENCE>, dim3(grid), dim3(kThreadsSdf), lds, as_stream(stream),
                           desc->lv, net, table, B_fourier, a, n_iters, m8_max, m4_max);
    else
        hipLaunchKernelGGL(trace_secant_kernel<HM_FRAC_TRILINEAR>, dim3(grid), dim3(kThreadsSdf), lds, as_stream(stream),
                           desc->lv, net, table, B_fourier, a, n_iters, m8_max, m4_max);
    HM_CHECK_LAUNCH("hm_trace_secant_persistent");
    return HM_OK;
}

// Closest-approach scan (points at pts + scan_off, count cnt[C_NSEL_PTS]) and the secant refinement in one launch
// (sdf_scan_secant_kernel); scans of <= 8192 points run on the small-tile launch in front of it.  tile_points = 0 only.
int hm_trace_scan_secant(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table, const float *B_fourier,
                         int frac_mode, const void *trace_args, int n_iters, int64_t scan_off, int64_t scan_capacity,
                         int c_prev1, int grid1, int c_prev2, int grid2, void *stream) {
    HM_CHECK_ARG(desc && mlp && table && B_fourier && trace_args, "hm_trace_scan_secant: NULL argument");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_trace_scan_secant: bad frac_mode");
    const TraceArgs &a = *static_cast<const TraceArgs *>(trace_args);
    SdfNet net;
    bool have16 = false;
    int rc = sdf_net_from_desc(desc->lv, mlp, 0, net, have16);
    if (rc != HM_OK) return rc;
    HM_CHECK_ARG(have16, "hm_trace_scan_secant: needs w_packed_m16 in every layer");
    if (a.n == 0) return HM_OK;
    constexpr int64_t kSmall = 8192;
    // the scan's small-count form (tile_points -1: returns at once above kSmall live points)
    rc = hm_sdf_fwd(desc, mlp, a.w.pts + scan_off * 3, scan_capacity, table, B_fourier, a.w.vals + scan_off, 1, 1, frac_mode,
                    -1, a.w.cnt + C_NSEL_PTS, 0, stream);
    if (rc != HM_OK) return rc;
    const int emb_b16 = (desc->lv.E + 15) / 16;
    const size_t lds16 = sizeof(float) * ((size_t)(2 * net.x_groups + emb_b16 * 4) * kGroupFloats16 + kPts16 * 4 +
                                          kWaves * kPts16);
    const size_t lds64 = sizeof(float) * ((size_t)(net.x_groups + net.emb_groups) * kGroupFloats + kPts * 4 + kWaves * kPts);
    HM_CHECK_ARG(lds16 <= 96 * 1024 && lds64 <= 160 * 1024 - 64, "hm_trace_scan_secant: network does not fit the LDS tiles");
    const size_t lds = lds16 > lds64 ? lds16 : lds64;
    static thread_local bool attr_done = false;
    if (!attr_done) {  // opt in to >64 KB dynamic LDS once (not a stream operation)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_scan_secant_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_scan_secant_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    // 16-point secant tiles once the scan keeps the chip busy for longer than their chain takes (8 x 131 us ~ 2.3 rounds
    // of 64-point tiles: 3 rounds = 49 152 points)
    const int64_t scan_hide = 3 * 256 * kPts;
    // c_prev*: counters holding the own-point counts of the sampler launches that took filler tiles of this scan (-1: none)
    const int p1 = grid1 > 0 ? c_prev1 : -1, p2 = grid2 > 0 ? c_prev2 : -1;
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(sdf_scan_secant_kernel<HM_FRAC_REFERENCE>, dim3(256), dim3(kThreadsSdf), lds, as_stream(stream),
                           desc->lv, net, table, B_fourier, a, n_iters, (int64_t)2048, (int64_t)1024, scan_off, kSmall + 1,
                           scan_hide, p1, grid1, p2, grid2);
    else
        hipLaunchKernelGGL(sdf_scan_secant_kernel<HM_FRAC_TRILINEAR>, dim3(256), dim3(kThreadsSdf), lds, as_stream(stream),
                           desc->lv, net, table, B_fourier, a, n_iters, (int64_t)2048, (int64_t)1024, scan_off, kSmall + 1,
                           scan_hide, p1, grid1, p2, grid2);
    HM_CHECK_LAUNCH("hm_trace_scan_secant");
    return HM_OK;
}

static int sdf_fwd_impl(const HmLevels &lv, const hm_mlp_desc *mlp, const float *x, int64_t emb_stride, int64_t n,
                        const float *table, const float *B_fourier, float *out, int64_t out_stride, int out_cols,
                        int frac_mode, int tile_points, const int32_t *n_dev, int max_workgroups, void *stream,
                        const SdfFillArgs *fill, int *grid64_out) {
    HM_CHECK_ARG(mlp, "hm_sdf_fwd: NULL descriptor");
    HM_CHECK_ARG(n >= 0, "hm_sdf_fwd: n < 0");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_sdf_fwd: bad frac_mode");
    HM_CHECK_ARG(mlp->n_layers >= 1 && mlp->n_layers <= HM_MAX_LAYERS, "hm_sdf_fwd: n_layers out of range");
    HM_CHECK_ARG(tile_points == 0 || tile_points == 4 || tile_points == 8 || tile_points == 16 || tile_points == 32 ||
                     tile_points == 64 || tile_points == -1,
                 "hm_sdf_fwd: tile_points must be -1, 0, 4, 8, 16, 32 or 64");
    SdfNet net;
    bool have16 = true;
    {
        const int rc = sdf_net_from_desc(lv, mlp, emb_stride, net, have16);
        if (rc != HM_OK) return rc;
    }
    const int mode = emb_stride > 0 ? kFracEmb : frac_mode;    // kernel template value
    const int emb_b16 = (lv.E + 15) / 16;
    const hm_mlp_layer &last = mlp->layer[mlp->n_layers - 1];
    HM_CHECK_ARG(out_cols == 1 || out_cols == last.out_dim, "hm_sdf_fwd: out_cols must be 1 or the last layer's out_dim");
    HM_CHECK_ARG(out_stride >= out_cols, "hm_sdf_fwd: out_stride < out_cols");
    HM_CHECK_ARG((tile_points != 16 && tile_points != 8 && tile_points != 4) || have16,
                 "hm_sdf_fwd: tile_points 4 / 8 / 16 need w_packed_m16 in every layer");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && out && (emb_stride > 0 || (table && B_fourier)), "hm_sdf_fwd: NULL pointer");
    // small batches: 16-point tiles spread the call over the whole chip (see sdf_m16_body / sdf_m8_body).
    // With a device-side count the host cannot know the batch size: both kernels are enqueued and
    // each returns at once unless the live count falls in its range.
    constexpr int64_t kSmall = 8192;
    const int64_t kBig = (int64_t)1 << 62;
    bool run16 = false, run64 = false;
    int64_t lo16 = 0, hi16 = kBig, lo64 = 0, hi64 = kBig;
    // 8-point tiles below kTiny live points (one tile per CU), 16-point tiles up to kSmall, 64 above
    // (4-point tiles below kMini: same weight stream, a quarter of the 16-point tile's matrix work)
    constexpr int64_t kTiny = 2048, kMini = 1024;
    int64_t m8_max = 0, m4_max = 0;
    if (tile_points == -1) {   // small counts only: n > kSmall is another launch's job (bf16 coarse kernel)
        HM_CHECK_ARG(have16, "hm_sdf_fwd: tile_points -1 needs w_packed_m16 in every layer");
        run16 = true; hi16 = kSmall; m8_max = kTiny; m4_max = kMini;
    }
    else if (tile_points == 8) { run16 = true; m8_max = kBig; }
    else if (tile_points == 4) { run16 = true; m8_max = kBig; m4_max = kBig; }
    else if (tile_points == 16) run16 = true;
    else if (tile_points == 64 || tile_points == 32 || !have16) run64 = true;
    else if (!n_dev) { run16 = n <= kSmall; run64 = !run16; m8_max = kTiny; m4_max = kMini; }
    else if (n <= kSmall) { run16 = true; m8_max = kTiny; m4_max = kMini; }
    else { run16 = run64 = true; hi16 = kSmall; lo64 = kSmall + 1; m8_max = kTiny; m4_max = kMini; }
    static thread_local bool attr_done = false;
    if (!attr_done) {  // opt in to >64 KB dynamic LDS once (not a stream operation)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_kernel<kFracEmb>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        // (small tiles: two activation images of 32 KB + the embedding = 70 KB at 512-wide layers)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_small_kernel<HM_FRAC_REFERENCE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_small_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_small_kernel<kFracEmb>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    if (run16) {
        const size_t lds = sizeof(float) * ((size_t)(2 * net.x_groups + emb_b16 * 4) * kGroupFloats16 + kPts16 * 4 +
                                            kWaves * kPts16);   // two activation images (the 8-point layout needs half of this)
        HM_CHECK_ARG(lds <= 96 * 1024, "hm_sdf_fwd: network does not fit the 16-point LDS tile");
        const int64_t nmax = n < hi16 ? n : hi16;
        int64_t tiles = m8_max >= nmax ? 0 : (nmax + kPts16 - 1) / kPts16;
        if (m8_max > 0) {
            const int64_t n8 = nmax < m8_max ? nmax : m8_max;
            tiles = max(tiles, (n8 + kPts8 - 1) / kPts8);
        }
        if (m4_max > 0) {
            const int64_t n4 = nmax < m4_max ? nmax : m4_max;
            tiles = max(tiles, (n4 + 3) / 4);
        }
        const int64_t cap = max_workgroups > 0 ? max_workgroups : 256;  // one resident workgroup per CU
        const int64_t grid = tiles < cap ? tiles : cap;
        if (mode == kFracEmb)
            hipLaunchKernelGGL(sdf_fwd_small_kernel<kFracEmb>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo16, hi16, m8_max, m4_max);
        else if (mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(sdf_fwd_small_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo16, hi16, m8_max, m4_max);
        else
            hipLaunchKernelGGL(sdf_fwd_small_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo16, hi16, m8_max, m4_max);
    }
    // big batches: one 64-point workgroup per CU; HM_SDF_P32=1 / tile_points 32: two 32-point workgroups per CU
    // (measured SLOWER: 113 vs 121.6 TFLOP/s - twice the weight stream per MFMA costs more than the overlap gains)
    static const int use_p32 = [] { const char *e = getenv("HM_SDF_P32"); return e ? atoi(e) : 0; }();
    const size_t lds32 = sizeof(float) * ((size_t)(net.x_groups + net.emb_groups) * kGroupFloats32 + kPts32 * 4 +
                                          2 * kWaves32 * kPts32);
    if (run64 && (use_p32 || tile_points == 32) && tile_points != 64 && lds32 <= 80 * 1024) {
        static thread_local bool attr32_done = false;
        if (!attr32_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_p32_kernel<HM_FRAC_REFERENCE>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_p32_kernel<HM_FRAC_TRILINEAR>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(sdf_fwd_p32_kernel<kFracEmb>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
            attr32_done = true;
        }
        const int64_t tiles = (n + kPts32 - 1) / kPts32;
        const int64_t cap = max_workgroups > 0 ? max_workgroups : 512;   // two resident workgroups per CU
        const int64_t grid = tiles < cap ? tiles : cap;
        if (mode == kFracEmb)
            hipLaunchKernelGGL(sdf_fwd_p32_kernel<kFracEmb>, dim3((unsigned)grid), dim3(kThreads32), lds32,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64);
        else if (mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(sdf_fwd_p32_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreads32), lds32,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64);
        else
            hipLaunchKernelGGL(sdf_fwd_p32_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreads32), lds32,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64);
        run64 = false;
    }
    if (run64) {
        const size_t lds = sizeof(float) * ((size_t)(net.x_groups + net.emb_groups) * kGroupFloats + kPts * 4 +
                                            kWaves * kPts);
        HM_CHECK_ARG(lds <= 160 * 1024, "hm_sdf_fwd: network does not fit the 160 KB LDS tile");
        const int64_t tiles = (n + 31) / 32;      // (up to cap * 32 points: one 32-point half tile per workgroup)
        const int64_t cap = max_workgroups > 0 ? max_workgroups : 256;
        const int64_t grid = tiles < cap ? tiles : cap;
        // filler tiles only where every launch of the chain applies the same lower bound (fill_quota's run_min)
        const SdfFillArgs none = {nullptr, nullptr, nullptr, nullptr, 0, 0};
        const SdfFillArgs fa = (fill && lo64 == kSmall + 1 && out_cols == 1 && out_stride == 1) ? *fill : none;
        if (grid64_out && fa.n_dev) *grid64_out = (int)grid;
        if (mode == kFracEmb)
            hipLaunchKernelGGL(sdf_fwd_kernel<kFracEmb>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64, none);
        else if (mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(sdf_fwd_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64, fa);
        else
            hipLaunchKernelGGL(sdf_fwd_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreadsSdf), lds,
                               as_stream(stream), lv, net, x, n, table, B_fourier, out, out_stride, out_cols, n_dev,
                               lo64, hi64, fa);
    }
    HM_CHECK_LAUNCH("hm_sdf_fwd");
    return HM_OK;
}

}  // extern "C"
