// hm_encode.hip - multi-resolution hash-grid encoder kernels for gfx950 (MI355X).
//
// Replaces MultiResHashGridMLP.forward / _HashGridMLP.forward / hash_func / FourierFeature.forward
// (reference: model/embeddings/hashGridEmbedding.py:32-40,81-102,150-155, frequency_enc.py:63-67)
// and the nn.Embedding backward autograd runs for them.
//
// Layout: one fused table [sum(rows_l), F] fp32 in HBM (level l starts at row_off[l]); points
// x [N,3]; output rows [x | sin | cos | level features], E = 3 + 2L + L*F floats per point.
//
// Forward kernel shape (wave64): 8 lanes own one point, one lane per voxel corner, so a single
// global_load_dwordx2 wave-instruction fetches the 8 corner rows of 8 points; the x/y corner
// neighbours (hash multipliers 1 and 3) fall into the same 64-B line and coalesce inside the
// instruction.  The 8-corner weighted sum is a 3-step DPP butterfly (quad_perm, quad_perm,
// row_half_mirror) - no LDS traffic.  A workgroup (4 waves) owns a tile of 64 points, its waves
// interleave the levels, and the [64, E] output tile is staged in LDS so that HBM sees only
// full-line coalesced stores.
#include "hm_common.h"

#include <stdlib.h>

#include <math.h>

namespace {

constexpr int kTile = 64;  // points per workgroup
constexpr int kThreads = 256;

// v + (v of the partner lane) as ONE VALU instruction per step (v_add_f32 with a DPP-permuted operand).  The builtin form
// (__builtin_amdgcn_update_dpp + an add) compiled to v_mov_b32 + v_mov_b32_dpp + a packed add per step - with the wait states
// 18 issue slots of the ~40 of a (level, point group) pass in the gather kernels, which are VALU-bound (r3bj).  The
// s_nop covers the two wait states a DPP read needs behind the VALU write of its source (the assembler does not insert
// them for inline code); the sum is the same IEEE addition, bit for bit.
__device__ __forceinline__ void dpp_sum8_pair(float &a0, float &a1) {
    // sums over the 8 corner lanes of a point (lanes 8p .. 8p+7 of a row of 16): partner lane ^1, ^2, then the other quad of
    // the half row - the order of the additions the builtin form had.  Two independent chains: one filler instruction and
    // one s_nop give the second wait state between a value's steps.
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1));
}

// Output-tile store with a cache policy (flags bits 0-1): 0 plain, 1 write-through `sc1` (the line is DROPPED from the
// XCD's L2 once written: 1.1 GB of output rows per launch no longer push table lines out of the 4 MB L2s), 2 `nt`,
// 3 `sc0 sc1`.  dst / n_vec describe ONE tile (wave-uniform), i is the lane's float4 index inside it.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_tile_vec(float *dst, unsigned tile_bytes, int i, const float4 &v, int mode) {
    if (mode == 0) {
        reinterpret_cast<float4 *>(dst)[i] = v;
        return;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, tile_bytes, 0x00020000);
    const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    if (mode == 1) __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, i * 16, 0, 16);        // sc1
    else if (mode == 2) __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, i * 16, 0, 2);    // nt
    else __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, i * 16, 0, 17);                  // sc0 sc1
}

// corner weight for one axis: reference mode => (bit ? xf : 1-xf) with xf == 0
template <int FRAC>
__device__ __forceinline__ void voxel_and_weight(float x, int32_t res, int bit, uint32_t &u, float &w) {
    float xs = __fmul_rn(x, (float)res);
    if (FRAC == HM_FRAC_REFERENCE) {
        int32_t xi = (int32_t)xs;  // trunc toward zero
        u = (uint32_t)xi + (uint32_t)bit;
        w = bit ? 0.0f : 1.0f;     // where(mask, 1 - xf, xf) with xf = x - x.float() = 0
    } else {
        float fl = floorf(xs);
        int32_t xi = (int32_t)fl;
        float xf = __fsub_rn(xs, fl);
        u = (uint32_t)xi + (uint32_t)bit;
        w = bit ? xf : __fsub_rn(1.0f, xf);
    }
}

template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_fwd_f2_kernel(HmLevels lv, const float *__restrict__ x, int64_t n,
                                                                 const float2 *__restrict__ table,
                                                                 const float *__restrict__ Bf,
                                                                 float *__restrict__ out, int64_t out_stride,
                                                                 int flags) {
    extern __shared__ __align__(16) float smem[];
    const int L = lv.L;
    const bool fourier = (Bf != nullptr);
    const int hoff = fourier ? 3 + 2 * L : 0;  // first hash-feature column of an output row
    const int E = hoff + 2 * L;                // row width written by this launch (F == 2)
    float *s_out = smem;              // [kTile][E]
    float *s_x = smem + kTile * E;    // [kTile][3]
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const int cnt = (int)min((int64_t)kTile, n - base);

    if (tid < kTile * 3) {
        int64_t g = base * 3 + tid;
        s_x[tid] = (tid < cnt * 3) ? x[g] : 0.0f;
    }
    __syncthreads();

    // ---- Fourier features + passthrough: thread -> (point p, channel group cg)
    if (fourier) {
        const int p = tid & (kTile - 1);
        const int cg = tid >> 6;  // 0..3
        const float x0 = s_x[p * 3 + 0], x1 = s_x[p * 3 + 1], x2 = s_x[p * 3 + 2];
        float *o = s_out + p * E;
        if (cg == 0) {
            o[0] = x0; o[1] = x1; o[2] = x2;
        }
        const float two_pi = 6.283185307179586f;  // 2*np.pi*x is evaluated in fp32 (frequency_enc.py:65)
        const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
        for (int c = cg; c < L; c += 4) {
            // [N,3]@[3,L] as a k-ordered fma chain (matches torch's CPU sgemm bit-for-bit, see oracle)
            float a = __fmul_rn(s0, Bf[c]);
            a = __fmaf_rn(s1, Bf[L + c], a);
            a = __fmaf_rn(s2, Bf[2 * L + c], a);
            float sn, cs;
            sincosf(a, &sn, &cs);
            o[3 + c] = sn;
            o[3 + L + c] = cs;
        }
    }

    // ---- hash levels: wave w takes levels w, w+4, ...; lane = (sub-point, corner)
    {
        const int wave = tid >> 6;
        const int lane = tid & 63;
        const int corner = lane & 7;
        const int sub = lane >> 3;
        const int bx = corner & 1, by = (corner >> 1) & 1, bz = (corner >> 2) & 1;
        for (int l = wave; l < L; l += 4) {
            const int32_t res = lv.res[l];
            const uint32_t rows = lv.rows[l], magic = lv.magic[l];
            const float2 *tl = table + lv.row_off[l];
            float2 v[8];
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = j * 8 + sub;
                uint32_t ux, uy, uz;
                float wx, wy, wz;
                voxel_and_weight<FRAC>(s_x[p * 3 + 0], res, bx, ux, wx);
                voxel_and_weight<FRAC>(s_x[p * 3 + 1], res, by, uy, wy);
                voxel_and_weight<FRAC>(s_x[p * 3 + 2], res, bz, uz, wz);
                const uint32_t id = hm_mod_rows(hm_hash3(ux, uy, uz), rows, magic);
                v[j] = tl[id];
                w[j] = __fmul_rn(__fmul_rn(wx, wy), wz);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a0 = __fmul_rn(v[j].x, w[j]);
                float a1 = __fmul_rn(v[j].y, w[j]);
                dpp_sum8_pair(a0, a1);
                if (corner == 0) {
                    const int p = j * 8 + sub;
                    float *o = s_out + p * E + hoff + 2 * l;
                    o[0] = a0;
                    o[1] = a1;
                }
            }
        }
    }
    __syncthreads();

    // ---- coalesced tile store
    if (out_stride == E && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        float *dst = out + base * E;  // 16-B aligned: kTile*E*4 is a multiple of 16
        const int total = cnt * E;
        const int nvec = total >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(s_out);
        for (int i = tid; i < nvec; i += kThreads) store_tile_vec(dst, (unsigned)total * 4u, i, s4[i], flags & 3);
        for (int i = (nvec << 2) + tid; i < total; i += kThreads) dst[i] = s_out[i];
    } else {
        for (int i = tid; i < cnt * E; i += kThreads) {
            int p = i / E, c = i - p * E;
            out[(base + p) * out_stride + c] = s_out[i];
        }
    }
}

// Generic-F fallback (F != 2): one thread per (point, level); correctness path for unusual configs.
// ---------------------------------------------------------------------------------------------------------
// Level-synchronous variant for big launches over tables larger than an XCD's L2.  In the tile kernel above tens
// of thousands of short-lived workgroups are in flight at random phases, so every XCD's 4 MB L2 sees all 16 levels
// at once (40 MB of tables at C2): a third of the fine-level corner fetches miss L2 and the kernel runs at the
// Infinity-Cache rate.  Here TWO persistent workgroups per CU (8 waves each) own 256 points at a time and ALL of them sweep
// the levels in the same order, one (coarse, fine) pair at a time - identical work per tile keeps them roughly in
// phase - so an XCD's L2 mostly holds the one fine level (4 MB at T = 2^19) everybody is reading: fabric reads per
// launch drop from 7.84 to 5.18 GB at C2 (PMC, profiles/).  Output rows are staged in LDS (256 x 67 floats = 67 KB per workgroup)
// and leave as full-line stores, as before.
constexpr int kTileS = 256, kThreadsS = 512;   // two workgroups per CU: one gathers while the other stores
                                               // (one of 512 points: 4.53 TB/s, two of 256: 4.65, four of 128: 4.45)   // two workgroups per CU: one gathers while the other stores

template <int FRAC, int KL>   // KL = levels gathered per sweep step (2: one (coarse, fine) pair; 4: more loads in flight)
__global__ __launch_bounds__(kThreadsS) void encode_fwd_f2_sweep_kernel(HmLevels lv, const float *__restrict__ x,
                                                                        int64_t n, const float2 *__restrict__ table,
                                                                        const float *__restrict__ Bf,
                                                                        float *__restrict__ out, int64_t out_stride,
                                                                        int flags) {
    extern __shared__ __align__(16) float smem[];
    const int L = lv.L;
    const bool fourier = (Bf != nullptr);
    const int hoff = fourier ? 3 + 2 * L : 0;
    const int E = hoff + 2 * L;
    float *s_out = smem;               // [kTileS][E]
    float *s_x = smem + kTileS * E;    // [kTileS][3]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int corner = lane & 7, sub = lane >> 3;
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = (corner >> 2) & 1;
    const int64_t n_tiles = (n + kTileS - 1) / kTileS;
    const int half = (L + KL - 1) / KL;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kTileS;
        const int cnt = (int)min((int64_t)kTileS, n - base);
        __syncthreads();   // the previous tile's rows have left s_out
        if (flags & 4)
            for (int i = tid; i < kTileS * 3; i += kThreadsS)
                s_x[i] = (i < cnt * 3) ? __builtin_nontemporal_load(x + base * 3 + i) : 0.0f;
        else
            for (int i = tid; i < kTileS * 3; i += kThreadsS) s_x[i] = (i < cnt * 3) ? x[base * 3 + i] : 0.0f;
        __syncthreads();
        if (fourier) {
            const int p = tid & (kTileS - 1);
            const int cg = tid / kTileS;  // 0..1
            const float x0 = s_x[p * 3 + 0], x1 = s_x[p * 3 + 1], x2 = s_x[p * 3 + 2];
            float *o = s_out + p * E;
            if (cg == 0) {
                o[0] = x0; o[1] = x1; o[2] = x2;
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = cg; c < L; c += 2) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                o[3 + c] = sn;
                o[3 + L + c] = cs;
            }
        }
        // wave w owns points [32w, 32w+32) of the tile: 4 passes of 8 points x 8 corners, two levels per sweep step
        // (gathering step lp+1 while step lp is reduced - two register sets - measured 2-7 % SLOWER)
        for (int lp = 0; lp < half; ++lp) {
            int lvl[KL];
#pragma unroll
            for (int k = 0; k < KL; ++k) lvl[k] = lp + k * half;
            float2 v[KL][4];
            float w[KL][4];
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int l = min(lvl[k], L - 1);
                const int32_t res = lv.res[l];
                const uint32_t rows = lv.rows[l], magic = lv.magic[l];
                const float2 *tl = table + lv.row_off[l];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = wave * 32 + j * 8 + sub;
                    uint32_t ux, uy, uz;
                    float wx, wy, wz;
                    voxel_and_weight<FRAC>(s_x[p * 3 + 0], res, bx, ux, wx);
                    voxel_and_weight<FRAC>(s_x[p * 3 + 1], res, by, uy, wy);
                    voxel_and_weight<FRAC>(s_x[p * 3 + 2], res, bz, uz, wz);
                    v[k][j] = tl[hm_mod_rows(hm_hash3(ux, uy, uz), rows, magic)];
                    w[k][j] = __fmul_rn(__fmul_rn(wx, wy), wz);
                }
            }
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                if (lvl[k] >= L) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a0 = __fmul_rn(v[k][j].x, w[k][j]);
                    float a1 = __fmul_rn(v[k][j].y, w[k][j]);
                    dpp_sum8_pair(a0, a1);
                    if (corner == 0) {
                        float *o = s_out + (wave * 32 + j * 8 + sub) * E + hoff + 2 * lvl[k];
                        o[0] = a0;
                        o[1] = a1;
                    }
                }
            }
        }
        __syncthreads();
        if (out_stride == E && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
            float *dst = out + base * E;   // 16-B aligned: kTileS*E*4 is a multiple of 16
            const int total = cnt * E;
            const int nvec = total >> 2;
            const float4 *s4 = reinterpret_cast<const float4 *>(s_out);
            for (int i = tid; i < nvec; i += kThreadsS) store_tile_vec(dst, (unsigned)total * 4u, i, s4[i], flags & 3);
            for (int i = (nvec << 2) + tid; i < total; i += kThreadsS) dst[i] = s_out[i];
        } else {
            for (int i = tid; i < cnt * E; i += kThreadsS) {
                const int p = i / E, c = i - p * E;
                out[(base + p) * out_stride + c] = s_out[i];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// z-ordered variant for big launches.  The reference hash multiplies x by 1, y by 3 and only z by the large prime
// (hashGridEmbedding.py:14,32-40): for one z-plane of a level all (x, y) corners fall into a window of 2^11 rows
// (16 KB) around (P * uz) mod rows, because ux ^ 3uy only touches the low 11 bits (res <= 512).  Points that are
// close in z therefore share their table windows on EVERY level.  The launch buckets the points into kSlabs z-slabs
// (three small passes: histogram, scan, scatter of point indices), and the gather kernel walks the points in that
// order - consecutive tiles of one XCD read a few 16-KB windows per level out of its L2 instead of scattering over
// 4 ... 33 MB per level.  x is read and the output rows are written through the permutation (rows keep their
// positions: out[i] is still encode(x[i])).  Fabric reads per launch at T = 2^22 drop from 9.4 GB to the compulsory
// traffic; see DESIGN.md / profiles for the measured numbers.
constexpr int kSlabs = 512;
constexpr int kSortChunk = 16384;   // points per workgroup of the bucketing passes

__device__ __forceinline__ int z_slab(float z) {
    const float t = (z + 1.0f) * (0.5f * kSlabs);          // [-1, 1] -> [0, kSlabs); everything outside clamps
    return (int)fminf(fmaxf(t, 0.0f), (float)(kSlabs - 1));
}

// slab_out (optional, [n] uint16): the slab of every point, so that the scatter pass reads 2 bytes per point instead of
// striding through x again
__global__ __launch_bounds__(1024) void zsort_hist_kernel(const float *__restrict__ x, int64_t n, uint32_t *hist,
                                                          uint16_t *__restrict__ slab_out) {
    __shared__ uint32_t h[kSlabs];
    for (int i = threadIdx.x; i < kSlabs; i += 1024) h[i] = 0u;
    __syncthreads();
    const int64_t beg = (int64_t)blockIdx.x * kSortChunk, end = min(beg + kSortChunk, n);
    for (int64_t i = beg + threadIdx.x; i < end; i += 1024) {
        const int sl = z_slab(x[i * 3 + 2]);
        if (slab_out) slab_out[i] = (uint16_t)sl;
        atomicAdd(&h[sl], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSlabs; i += 1024)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// hist[0..kSlabs): counts -> start offsets (the scatter pass turns them into end offsets);
// hist[kSlabs..2 kSlabs): a copy of the start offsets that survives the scatter
__global__ __launch_bounds__(kSlabs) void zsort_scan_kernel(uint32_t *hist) {
    __shared__ uint32_t s[kSlabs];
    const int t = threadIdx.x;
    s[t] = hist[t];
    __syncthreads();
    for (int o = 1; o < kSlabs; o <<= 1) {
        const uint32_t v = t >= o ? s[t - o] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    const uint32_t start = s[t] - hist[t];   // exclusive
    hist[t] = start;
    hist[kSlabs + t] = start;
}

__global__ __launch_bounds__(1024) void zsort_scatter_kernel(const float *__restrict__ x, int64_t n, uint32_t *cursor,
                                                             uint32_t *__restrict__ order,
                                                             const uint16_t *__restrict__ slab_in) {
    __shared__ uint32_t h[kSlabs], base[kSlabs];
    for (int i = threadIdx.x; i < kSlabs; i += 1024) h[i] = 0u;
    __syncthreads();
    const int64_t beg = (int64_t)blockIdx.x * kSortChunk, end = min(beg + kSortChunk, n);
    auto slab_of = [&](int64_t i) { return slab_in ? (int)slab_in[i] : z_slab(x[i * 3 + 2]); };
    for (int64_t i = beg + threadIdx.x; i < end; i += 1024) atomicAdd(&h[slab_of(i)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < kSlabs; i += 1024) {
        base[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0u;   // reserve this workgroup's range of the slab
        h[i] = 0u;
    }
    __syncthreads();
    for (int64_t i = beg + threadIdx.x; i < end; i += 1024) {
        const int sl = slab_of(i);
        order[base[sl] + atomicAdd(&h[sl], 1u)] = (uint32_t)i;
    }
}

template <int FRAC>
__global__ __launch_bounds__(kThreadsS) void encode_fwd_f2_zorder_kernel(HmLevels lv, const float *__restrict__ x,
                                                                         int64_t n, const float2 *__restrict__ table,
                                                                         const float *__restrict__ Bf,
                                                                         float *__restrict__ out, int64_t out_stride,
                                                                         const uint32_t *__restrict__ order) {
    extern __shared__ __align__(16) float smem[];
    const int L = lv.L;
    const bool fourier = (Bf != nullptr);
    const int hoff = fourier ? 3 + 2 * L : 0;
    const int E = hoff + 2 * L;
    const int ES = (E + 3) & ~3;                                        // LDS row stride: rows start on 16-byte boundaries
    float *s_out = smem;                                                // [kTileS][ES]
    float *s_x = smem + kTileS * ES;                                    // [kTileS][3]
    uint32_t *s_i = reinterpret_cast<uint32_t *>(s_x + kTileS * 3);     // [kTileS] original row of each point
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int corner = lane & 7, sub = lane >> 3;
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = (corner >> 2) & 1;
    const int64_t n_tiles = (n + kTileS - 1) / kTileS;
    const int half = (L + 1) / 2;
    // XCD-affine tile order: workgroups b and b + 8 share an XCD (round-robin dispatch; speed only, never
    // correctness).  XCD k walks the k-th eighth of the z-ordered tiles, its workgroups side by side, so that
    // an XCD's L2 sees one narrow z-range at a time.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (int)(gridDim.x >> 3);
    const int64_t span = (n_tiles + 7) / 8;

    for (int64_t k = slot; k < span; k += max(per_xcd, 1)) {
        const int64_t tile = (int64_t)xcd * span + k;
        if (tile >= n_tiles) break;
        const int64_t base = tile * kTileS;
        const int cnt = (int)min((int64_t)kTileS, n - base);
        __syncthreads();   // the previous tile's rows have left s_out
        if (tid < kTileS) s_i[tid] = tid < cnt ? order[base + tid] : 0u;
        __syncthreads();
        for (int i = tid; i < kTileS * 3; i += kThreadsS) {
            const int p = i / 3, c = i - p * 3;
            s_x[i] = (p < cnt) ? x[(int64_t)s_i[p] * 3 + c] : 0.0f;
        }
        __syncthreads();
        if (ES > E && tid < kTileS)
            for (int c = E; c < ES; ++c) s_out[tid * ES + c] = 0.0f;
#ifdef HM_ENC_DIAG_NOFOURIER
        if (false) {
#else
        if (fourier) {
#endif
            const int p = tid & (kTileS - 1);
            const int cg = tid / kTileS;  // 0..1
            const float x0 = s_x[p * 3 + 0], x1 = s_x[p * 3 + 1], x2 = s_x[p * 3 + 2];
            float *o = s_out + p * ES;
            if (cg == 0) {
                o[0] = x0; o[1] = x1; o[2] = x2;
            }
            const float two_pi = 6.283185307179586f;
            const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
            for (int c = cg; c < L; c += 2) {
                float a = __fmul_rn(s0, Bf[c]);
                a = __fmaf_rn(s1, Bf[L + c], a);
                a = __fmaf_rn(s2, Bf[2 * L + c], a);
                float sn, cs;
                sincosf(a, &sn, &cs);
                o[3 + c] = sn;
                o[3 + L + c] = cs;
            }
        }
        for (int lp = 0; lp < half; ++lp) {
            const int lvl[2] = {lp, lp + half};
            float2 v[2][4];
            float w[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int l = min(lvl[kk], L - 1);
#ifdef HM_ENC_DIAG_LMIN   // scripts/gather_level_probe.py: the kernel's time with only some of the levels gathered
                if (lvl[kk] < HM_ENC_DIAG_LMIN || lvl[kk] > HM_ENC_DIAG_LMAX) {
                    for (int j = 0; j < 4; ++j) { v[kk][j] = make_float2(0.0f, 0.0f); w[kk][j] = 0.0f; }
                    continue;
                }
#endif
                const int32_t res = lv.res[l];
                const uint32_t rows = lv.rows[l], magic = lv.magic[l];
                const float2 *tl = table + lv.row_off[l];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = wave * 32 + j * 8 + sub;
                    uint32_t ux, uy, uz;
                    float wx, wy, wz;
                    voxel_and_weight<FRAC>(s_x[p * 3 + 0], res, bx, ux, wx);
                    voxel_and_weight<FRAC>(s_x[p * 3 + 1], res, by, uy, wy);
                    voxel_and_weight<FRAC>(s_x[p * 3 + 2], res, bz, uz, wz);
                    v[kk][j] = tl[hm_mod_rows(hm_hash3(ux, uy, uz), rows, magic)];
                    w[kk][j] = __fmul_rn(__fmul_rn(wx, wy), wz);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (lvl[kk] >= L) continue;
#ifdef HM_ENC_DIAG_LMIN
                if (lvl[kk] < HM_ENC_DIAG_LMIN || lvl[kk] > HM_ENC_DIAG_LMAX) continue;
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a0 = __fmul_rn(v[kk][j].x, w[kk][j]);
                    float a1 = __fmul_rn(v[kk][j].y, w[kk][j]);
                    dpp_sum8_pair(a0, a1);
                    if (corner == 0) {
                        float *o = s_out + (wave * 32 + j * 8 + sub) * ES + hoff + 2 * lvl[kk];
                        o[0] = a0;
                        o[1] = a1;
                    }
                }
            }
        }
        __syncthreads();
        // rows go back to their ORIGINAL positions (nt: streamed).  With a padded row stride (out_stride a multiple of 4
        // floats >= ES, 16-byte aligned base: ops.encode_fwd allocates big outputs that way) ONE dwordx4 instruction of
        // ES/4 lanes writes a whole row - the 32-byte sectors of the row leave the CU together instead of as the three
        // partial-line stores of the dword path (WRITE_SIZE 1.44x the output bytes, profiles/r02_gather_pmc.json)
#ifdef HM_ENC_DIAG_NOSTORE
        if (s_out[tid] == 12345.678f)     // (never: the rows are computed but not written)
#endif
        if (((out_stride & 3) == 0) && out_stride >= ES && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
            const int nv = ES >> 2;                       // float4 per row (17 at E = 67)
            const int rows_per_wave = 64 / nv;            // 3
            const int g = lane / nv, li = lane - g * nv;
            if (g < rows_per_wave) {
                for (int p = wave * rows_per_wave + g; p < cnt; p += (kThreadsS / 64) * rows_per_wave) {
                    typedef float f32x4v __attribute__((ext_vector_type(4)));
                    const f32x4v v = *reinterpret_cast<const f32x4v *>(s_out + p * ES + 4 * li);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4v *>(out + (int64_t)s_i[p] * out_stride) + li);
                }
            }
        } else {
            const int hw = tid >> 5, hl = tid & 31;   // 16 half-waves
            for (int p = hw; p < cnt; p += kThreadsS / 32) {
                float *dst = out + (int64_t)s_i[p] * out_stride;
                const float *src = s_out + p * ES;
                for (int c = hl; c < E; c += 32) __builtin_nontemporal_store(src[c], dst + c);
            }
        }
    }
}

template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_fwd_generic_kernel(HmLevels lv, const float *__restrict__ x,
                                                                      int64_t n, const float *__restrict__ table,
                                                                      const float *__restrict__ Bf,
                                                                      float *__restrict__ out, int64_t out_stride) {
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int L = lv.L, F = lv.F;
    const int64_t i = gid / (L + 1);
    const int l = (int)(gid - i * (L + 1));
    if (i >= n) return;
    const float x0 = x[i * 3], x1 = x[i * 3 + 1], x2 = x[i * 3 + 2];
    float *o = out + i * out_stride;
    const int hoff = Bf ? 3 + 2 * L : 0;
    if (l == L) {  // Fourier part
        if (!Bf) return;
        o[0] = x0; o[1] = x1; o[2] = x2;
        const float two_pi = 6.283185307179586f;
        const float s0 = __fmul_rn(two_pi, x0), s1 = __fmul_rn(two_pi, x1), s2 = __fmul_rn(two_pi, x2);
        for (int c = 0; c < L; ++c) {
            float a = __fmul_rn(s0, Bf[c]);
            a = __fmaf_rn(s1, Bf[L + c], a);
            a = __fmaf_rn(s2, Bf[2 * L + c], a);
            float sn, cs;
            sincosf(a, &sn, &cs);
            o[3 + c] = sn;
            o[3 + L + c] = cs;
        }
        return;
    }
    const float xin[3] = {x0, x1, x2};
    float acc[8];
    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
    for (int c = 0; c < 8; ++c) {
        uint32_t u[3];
        float w = 1.0f;
        for (int d = 0; d < 3; ++d) {
            float wd;
            voxel_and_weight<FRAC>(xin[d], lv.res[l], (c >> d) & 1, u[d], wd);
            w = __fmul_rn(w, wd);
        }
        const uint32_t id = hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]);
        const float *row = table + ((uint64_t)lv.row_off[l] + id) * F;
        for (int f = 0; f < F; ++f) acc[f] = __fadd_rn(acc[f], __fmul_rn(row[f], w));
    }
    for (int f = 0; f < F; ++f) o[hoff + l * F + f] = acc[f];
}

// xi + 8 corner ids of one level (parity / debugging entry point)
__global__ __launch_bounds__(kThreads) void corner_ids_kernel(int32_t res, uint32_t rows, uint32_t magic,
                                                              const float *__restrict__ x, int64_t n,
                                                              int32_t *__restrict__ xi_out,
                                                              uint32_t *__restrict__ ids_out) {
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t i = gid >> 3;
    const int c = (int)(gid & 7);
    if (i >= n) return;
    int32_t xi[3];
    for (int d = 0; d < 3; ++d) xi[d] = hm_trunc_voxel(x[i * 3 + d], res);
    if (xi_out && c < 3) xi_out[i * 3 + c] = xi[c];
    const uint32_t ux = (uint32_t)xi[0] + (c & 1), uy = (uint32_t)xi[1] + ((c >> 1) & 1),
                   uz = (uint32_t)xi[2] + ((c >> 2) & 1);
    ids_out[i * 8 + c] = hm_mod_rows(hm_hash3(ux, uy, uz), rows, magic);
}

// Table gradient.  Reference weights: only corner 0 of a voxel carries weight (hashGridEmbedding.py:86,94 - exactly one
// row per point and level), so ONE lane per (point, level) does all there is to do (the first version
// launched eight lanes per (point, level) and left seven of them idle: 26 us per 3072-point call, now a third of it);
// trilinear weights: one lane per (point, level, corner), zero-weight corners skipped.
template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_bwd_table_kernel(HmLevels lv, const float *__restrict__ x,
                                                                    int64_t n, const float *__restrict__ d_feat,
                                                                    int64_t d_feat_stride,
                                                                    float *__restrict__ d_table) {
    constexpr int C = FRAC == HM_FRAC_REFERENCE ? 1 : 8;
    const int L = lv.L, F = lv.F;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(gid % C);
    const int64_t pl = gid / C;
    const int64_t i = pl / L;
    const int l = (int)(pl - i * L);
    if (i >= n) return;
    uint32_t u[3];
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float wd;
        voxel_and_weight<FRAC>(x[i * 3 + d], lv.res[l], (c >> d) & 1, u[d], wd);
        w = __fmul_rn(w, wd);
    }
    if (w == 0.0f) return;
    const uint32_t id = hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]);
    const float *g = d_feat + i * d_feat_stride + l * F;
    float *row = d_table + ((uint64_t)lv.row_off[l] + id) * F;
    for (int f = 0; f < F; ++f) atomicAdd(row + f, __fmul_rn(w, g[f]));
}

// The same scatter for SMALL tables (all levels together <= 4096 rows: the view-direction grid of the rendering network
// has 8 rows per level): every point of a batch lands on the same few rows, and 2048 points x 4 levels of global float
// atomics on 32 rows serialise in the L2 (45 us for that call, as much as the two 49 k-lane scatters into the 2^19-row
// table together).  Each workgroup sums its share in an LDS copy of the whole table (LDS float atomics) and adds the
// rows it touched to d_table once.
constexpr int kSmallTableRows = 4096;
template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_bwd_table_small_kernel(HmLevels lv, const float *__restrict__ x,
                                                                          int64_t n, const float *__restrict__ d_feat,
                                                                          int64_t d_feat_stride,
                                                                          float *__restrict__ d_table, int total_rows) {
    extern __shared__ __align__(16) float acc_lds[];      // [total_rows * F]
    constexpr int C = FRAC == HM_FRAC_REFERENCE ? 1 : 8;
    const int L = lv.L, F = lv.F;
    const int cells = total_rows * F;
    for (int k = threadIdx.x; k < cells; k += kThreads) acc_lds[k] = 0.0f;
    __syncthreads();
    const int64_t total = n * L * C;
    for (int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * kThreads) {
        const int c = (int)(gid % C);
        const int64_t pl = gid / C;
        const int64_t i = pl / L;
        const int l = (int)(pl - i * L);
        uint32_t u[3];
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float wd;
            voxel_and_weight<FRAC>(x[i * 3 + d], lv.res[l], (c >> d) & 1, u[d], wd);
            w = __fmul_rn(w, wd);
        }
        if (w == 0.0f) continue;
        const uint32_t id = hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]);
        const float *g = d_feat + i * d_feat_stride + l * F;
        float *row = acc_lds + ((size_t)lv.row_off[l] + id) * F;
        for (int f = 0; f < F; ++f) atomicAdd(row + f, __fmul_rn(w, g[f]));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < cells; k += kThreads) {
        const float v = acc_lds[k];
        if (v != 0.0f) atomicAdd(d_table + k, v);
    }
}

// The same scatter for data-parallel steps (parallel.TouchedRowExchange): besides adding into d_table it lists every
// table row it touches ONCE - a bit per row is claimed with atomicOr, the lanes of a wave that claimed a new row take
// consecutive slots of touched_rows with one atomicAdd per wave.  The rank's (row, value) pairs are then read off
// d_table by hm_rows_pack; nothing is sorted anywhere.
template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_bwd_table_tracked_kernel(HmLevels lv, const float *__restrict__ x,
                                                                            int64_t n, const float *__restrict__ d_feat,
                                                                            int64_t d_feat_stride,
                                                                            float *__restrict__ d_table,
                                                                            uint32_t *__restrict__ bits,
                                                                            int32_t *__restrict__ count,
                                                                            int32_t *__restrict__ rows_out, int64_t cap) {
    // reference weights: only corner 0 of a voxel carries weight (hashGridEmbedding.py:86,94), so a lane per (point,
    // level) does all there is to do; trilinear weights: a lane per (point, level, corner)
    constexpr int C = FRAC == HM_FRAC_REFERENCE ? 1 : 8;
    __shared__ int32_t s_cnt[kThreads / 64], s_base;
    const int L = lv.L, F = lv.F;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(gid % C);
    const int64_t pl = gid / C;
    const int64_t i = pl / L;
    const int l = (int)(pl - i * L);
    bool claimed = false;
    uint32_t grow = 0u;
    if (i < n) {
        uint32_t u[3];
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float wd;
            voxel_and_weight<FRAC>(x[i * 3 + d], lv.res[l], (c >> d) & 1, u[d], wd);
            w = __fmul_rn(w, wd);
        }
        if (w != 0.0f) {
            grow = lv.row_off[l] + hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]);
            const float *g = d_feat + i * d_feat_stride + l * F;
            float *row = d_table + (uint64_t)grow * F;
            for (int f = 0; f < F; ++f) atomicAdd(row + f, __fmul_rn(w, g[f]));
            const uint32_t bit = 1u << (grow & 31u);
            claimed = (atomicOr(bits + (grow >> 5), bit) & bit) == 0u;
        }
    }
    // slots for the newly claimed rows: ONE atomicAdd on the shared counter per workgroup (a returning atomic per wave
    // on one address serialises in the L2: 6144 of them cost 40 us of a 65 us launch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t m = __ballot(claimed);
    if (lane == 0) s_cnt[wave] = (int32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            const int32_t v = s_cnt[w];
            s_cnt[w] = tot;
            tot += v;
        }
        s_base = tot > 0 ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (claimed) {
        const int64_t slot = (int64_t)s_base + s_cnt[wave] + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < cap) rows_out[slot] = (int32_t)grow;     // (slot >= cap: reported by hm_rows_pack through `status`)
    }
}

// ---------------------------------------------------------------------------------------------------------
// z-ordered table gradient for big launches (F = 2).  Scattering one 4-byte atomic per lane into 64 different rows
// runs at a sixteenth of the atomic rate (the guide's 0.08 TB/s case): the atomic kernel above needs 8.2 ms for 2^22
// points.  With the points bucketed into z-slabs (same three passes as the forward) every (slab, level) pair touches
// only a handful of 2^11-row blocks of its level's table (the z-plane windows, see encode_fwd_f2_zorder_kernel), so
// one workgroup per pair accumulates its contributions in LDS copies of those blocks (ds_add_f32) and then adds each
// block to the table with DENSE, fully coalesced float atomics - 256 contiguous bytes per wave instruction, the shape
// that runs at the full 1.3 TB/s.  A pre-pass lays x and d_feat out in z order, level-major, so that the per-level
// passes read contiguous 8-byte values instead of re-gathering 128-byte rows 16 times.  Contributions that do not find
// a free LDS block (more than kBwdSlots distinct blocks in one pair: never seen in [-1,1]^3) go straight to the table.
constexpr int kBwdSlots = 8;            // LDS blocks per workgroup: 8 x 2048 rows x 2 floats = 128 KB
constexpr int kBwdBlockRows = 2048;
constexpr int kBwdThreads = 512;

__global__ __launch_bounds__(256) void zsort_pack_kernel(const float *__restrict__ x, const float *__restrict__ d_feat,
                                                         int64_t d_feat_stride, const uint32_t *__restrict__ order,
                                                         int64_t n, int LF, float *__restrict__ xs,
                                                         float *__restrict__ dfs /* [LF/2][n][2] */) {
    __shared__ float tile[64 * 33];      // 64 points x up to 32 floats (+1 pad)
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < LF; c0 += 32) {
        const int cw = min(32, LF - c0);
        __syncthreads();
        for (int i = tid; i < 64 * cw; i += 256) {      // coalesced along a point's row
            const int p = i / cw, c = i - p * cw;
            const int64_t k = k0 + p;
            tile[p * 33 + c] = k < n ? d_feat[(int64_t)order[k] * d_feat_stride + c0 + c] : 0.0f;
        }
        __syncthreads();
        for (int i = tid; i < 64 * cw; i += 256) {      // coalesced along the points of one level
            const int pr = i / 128, r = i - pr * 128;   // level pair index within the chunk, (point, f)
            const int p = r >> 1, f = r & 1;
            const int64_t k = k0 + p;
            if (2 * pr + f < cw && k < n) dfs[((int64_t)(c0 / 2 + pr) * n + k) * 2 + f] = tile[p * 33 + 2 * pr + f];
        }
    }
    for (int i = tid; i < 64 * 3; i += 256) {
        const int p = i / 3, c = i - p * 3;
        const int64_t k = k0 + p;
        if (k < n) xs[k * 3 + c] = x[(int64_t)order[k] * 3 + c];
    }
}

template <int FRAC>
__global__ __launch_bounds__(kBwdThreads) void encode_bwd_table_zorder_kernel(HmLevels lv,
                                                                             const float *__restrict__ xs,
                                                                             const float2 *__restrict__ dfs, int64_t n,
                                                                             const uint32_t *__restrict__ slab_start,
                                                                             float *__restrict__ d_table) {
    extern __shared__ __align__(16) float smem[];
    float *acc = smem;                                                  // [kBwdSlots][kBwdBlockRows][2]
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + kBwdSlots * kBwdBlockRows * 2);   // [kBwdSlots]
    const int slab = blockIdx.x, l = blockIdx.y;
    const int tid = threadIdx.x;
    const int64_t beg = slab_start[slab], end = slab + 1 < kSlabs ? (int64_t)slab_start[slab + 1] : n;
    if (beg >= end) return;
    for (int i = tid; i < kBwdSlots * kBwdBlockRows * 2; i += kBwdThreads) acc[i] = 0.0f;
    if (tid < kBwdSlots) keys[tid] = 0xffffffffu;
    __syncthreads();
    const int32_t res = lv.res[l];
    const uint32_t rows = lv.rows[l], magic = lv.magic[l];
    const float2 *g = dfs + (int64_t)l * n;
    float *tl = d_table + (size_t)lv.row_off[l] * 2;
    constexpr int C = FRAC == HM_FRAC_REFERENCE ? 1 : 8;
    for (int64_t k = beg + tid; k < end; k += kBwdThreads) {
        const float x0 = xs[k * 3], x1 = xs[k * 3 + 1], x2 = xs[k * 3 + 2];
        const float2 gv = g[k];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t ux, uy, uz;
            float wx, wy, wz;
            voxel_and_weight<FRAC>(x0, res, c & 1, ux, wx);
            voxel_and_weight<FRAC>(x1, res, (c >> 1) & 1, uy, wy);
            voxel_and_weight<FRAC>(x2, res, (c >> 2) & 1, uz, wz);
            const float w = __fmul_rn(__fmul_rn(wx, wy), wz);
            if (w == 0.0f) continue;
            const uint32_t id = hm_mod_rows(hm_hash3(ux, uy, uz), rows, magic);
            const uint32_t blk = id / kBwdBlockRows, off = id % kBwdBlockRows;
            int slot = -1;
            for (int s = 0; s < kBwdSlots; ++s) {
                uint32_t cur = keys[s];
                if (cur == 0xffffffffu) cur = atomicCAS(&keys[s], 0xffffffffu, blk);
                if (cur == blk || cur == 0xffffffffu) { slot = s; break; }
            }
            const float v0 = __fmul_rn(w, gv.x), v1 = __fmul_rn(w, gv.y);
            if (slot >= 0) {
                atomicAdd(&acc[(slot * kBwdBlockRows + off) * 2], v0);
                atomicAdd(&acc[(slot * kBwdBlockRows + off) * 2 + 1], v1);
            } else {
                atomicAdd(tl + (size_t)id * 2, v0);
                atomicAdd(tl + (size_t)id * 2 + 1, v1);
            }
        }
    }
    __syncthreads();
    for (int s = 0; s < kBwdSlots; ++s) {
        const uint32_t blk = keys[s];
        if (blk == 0xffffffffu) break;
        const uint32_t row0 = blk * kBwdBlockRows;
        const uint32_t nrow = min((uint32_t)kBwdBlockRows, rows - row0);
        for (uint32_t i = tid; i < nrow * 2; i += kBwdThreads) {      // dense, coalesced atomics
            const float v = acc[s * kBwdBlockRows * 2 + i];
            if (v != 0.0f) atomicAdd(tl + (size_t)row0 * 2 + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Deterministic table gradient (no atomics): contributions are keyed by their destination row, sorted by the caller
// (any stable sort; ops.encode_bwd_table uses the library's own radix sort, csrc/hm_sort.hip), and every run of equal keys is summed by ONE thread in
// sorted order and added to the table with a plain read-modify-write (each row has exactly one owner).  Bitwise
// reproducible for a given contribution order - what keeps data-parallel replicas identical when every rank builds
// its dense gradient from the same all-gathered contributions (round 2's point exchange; round 3's
// parallel.StaticGradExchange exchanges (row, value) lists and needs no sort).
template <int FRAC>
__global__ __launch_bounds__(kThreads) void encode_rows_kernel(HmLevels lv, const float *__restrict__ x, int64_t n,
                                                               int32_t *__restrict__ keys,
                                                               float *__restrict__ wts) {
    constexpr int C = FRAC == HM_FRAC_REFERENCE ? 1 : 8;   // reference mode: only corner 0 carries weight
    const int L = lv.L;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= n * L * C) return;
    const int c = (int)(gid % C);
    const int64_t pl = gid / C;
    const int64_t i = pl / L;
    const int l = (int)(pl - i * L);
    uint32_t u[3];
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float wd;
        voxel_and_weight<FRAC>(x[i * 3 + d], lv.res[l], (c >> d) & 1, u[d], wd);
        w = __fmul_rn(w, wd);
    }
    keys[gid] = (int32_t)(lv.row_off[l] + hm_mod_rows(hm_hash3(u[0], u[1], u[2]), lv.rows[l], lv.magic[l]));
    if (wts) wts[gid] = w;
}

__global__ __launch_bounds__(kThreads) void segment_scatter_kernel(const int32_t *__restrict__ keys,
                                                                   const int64_t *__restrict__ perm, int64_t K, int L,
                                                                   int F, int C, const float *__restrict__ d_feat,
                                                                   int64_t d_feat_stride, const float *__restrict__ wts,
                                                                   float *__restrict__ d_table) {
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= K) return;
    const int32_t key = keys[k];
    if (k > 0 && keys[k - 1] == key) return;      // not the first contribution of its row
    float acc[8];
    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
    for (int64_t m = k; m < K && keys[m] == key; ++m) {
        const int64_t jx = perm[m];
        const int64_t pl = jx / C;
        const int64_t i = pl / L;
        const int l = (int)(pl - i * L);
        const float w = wts ? wts[jx] : 1.0f;
        const float *g = d_feat + i * d_feat_stride + l * F;
        for (int f = 0; f < F; ++f) acc[f] = __fadd_rn(acc[f], __fmul_rn(w, g[f]));
    }
    float *row = d_table + (int64_t)key * F;
    for (int f = 0; f < F; ++f) row[f] = __fadd_rn(row[f], acc[f]);
}

// Counter calibration for 8-byte row gathers (diagnostic; bench.py --only gather_calib).  Every group of `group`
// consecutive lanes reads 8-byte rows of ONE pseudo-random 128-B-aligned block of a table far larger than the
// Infinity Cache, lane g of the group at byte offset g * stride_bytes inside the block: the ALGORITHMIC bytes and the
// set of 32- / 64- / 128-byte units touched are known exactly, so FETCH_SIZE / TCC_EA0_RDREQ per block tell the
// fetch granularity and the counter's unit for this access shape (the guide calibrates wide coalesced loads only).
__global__ __launch_bounds__(256) void gather_calib_kernel(const float2 *__restrict__ table, uint64_t n_blocks,
                                                           int64_t n, int group, int stride_bytes,
                                                           float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t b = (uint64_t)(i / group);
    b = (b ^ (b >> 31)) * 0x9E3779B97F4A7C15ull;       // splitmix-style scramble: distinct blocks, no locality
    b = (b ^ (b >> 29)) * 0xBF58476D1CE4E5B9ull;
    b = (b ^ (b >> 32)) % n_blocks;
    const int g = (int)(i % group);
    const float2 v = table[b * 16 + (uint64_t)(g * stride_bytes) / 8];
    out[i] = v.x + v.y;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int64_t hm_encode_bwd_workspace_bytes(const hm_grid_desc *desc, int64_t n) {
    if (!desc || n < 0) return hm_fail(HM_ERR_INVALID, "hm_encode_bwd_workspace_bytes: bad argument");
    // [slab counters 2*kSlabs u32 | order n u32 | xs 3n f32 | dfs L*F*n f32 | slab ids n u16], each part 256-byte aligned
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    return up(4 * 2 * kSlabs) + up(4 * n) + up(12 * n) + up(4 * n * desc->lv.L * desc->lv.F) + up(2 * n);
}

int hm_encode_bwd_table_ws(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                           int64_t d_feat_stride, float *d_table, int frac_mode, void *workspace,
                           int64_t workspace_bytes, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_bwd_table_ws: desc is NULL");
    const HmLevels &lv = desc->lv;
    const bool table_big = desc->total_rows * (uint64_t)lv.F * 4u > (8u << 20);
    if (!workspace || lv.F != 2 || n < (int64_t)131072 || n >= ((int64_t)1 << 31) || !table_big ||
        workspace_bytes < hm_encode_bwd_workspace_bytes(desc, n))
        return hm_encode_bwd_table(desc, x, n, d_feat, d_feat_stride, d_table, frac_mode, stream);
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_encode_bwd_table_ws: bad frac_mode");
    HM_CHECK_ARG(d_feat_stride >= lv.L * lv.F, "hm_encode_bwd_table_ws: d_feat_stride < L*F");
    HM_CHECK_ARG(x && d_feat && d_table, "hm_encode_bwd_table_ws: NULL pointer");
    hipStream_t st = as_stream(stream);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    char *wsb = static_cast<char *>(workspace);
    uint32_t *hist = reinterpret_cast<uint32_t *>(wsb);
    uint32_t *order = reinterpret_cast<uint32_t *>(wsb + up(4 * 2 * kSlabs));
    float *xs = reinterpret_cast<float *>(wsb + up(4 * 2 * kSlabs) + up(4 * (size_t)n));
    float *dfs = reinterpret_cast<float *>(wsb + up(4 * 2 * kSlabs) + up(4 * (size_t)n) + up(12 * (size_t)n));
    uint16_t *slab = reinterpret_cast<uint16_t *>(wsb + up(4 * 2 * kSlabs) + up(4 * (size_t)n) + up(12 * (size_t)n) +
                                                  up(4 * (size_t)n * lv.L * lv.F));
    hm_zero_u32_async(hist, kSlabs, st);
    const unsigned g_sort = (unsigned)((n + kSortChunk - 1) / kSortChunk);
    hipLaunchKernelGGL(zsort_hist_kernel, dim3(g_sort), dim3(1024), 0, st, x, n, hist, slab);
    hipLaunchKernelGGL(zsort_scan_kernel, dim3(1), dim3(kSlabs), 0, st, hist);
    hipLaunchKernelGGL(zsort_scatter_kernel, dim3(g_sort), dim3(1024), 0, st, x, n, hist, order,
                       static_cast<const uint16_t *>(slab));
    hipLaunchKernelGGL(zsort_pack_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, x, d_feat, d_feat_stride,
                       order, n, lv.L * lv.F, xs, dfs);
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_bwd_table_zorder_kernel<HM_FRAC_REFERENCE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_bwd_table_zorder_kernel<HM_FRAC_TRILINEAR>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_done = true;
    }
    const size_t lds = sizeof(float) * kBwdSlots * kBwdBlockRows * 2 + sizeof(uint32_t) * kBwdSlots;
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(encode_bwd_table_zorder_kernel<HM_FRAC_REFERENCE>, dim3(kSlabs, lv.L), dim3(kBwdThreads), lds, st,
                           lv, xs, reinterpret_cast<const float2 *>(dfs), n, hist + kSlabs, d_table);
    else
        hipLaunchKernelGGL(encode_bwd_table_zorder_kernel<HM_FRAC_TRILINEAR>, dim3(kSlabs, lv.L), dim3(kBwdThreads), lds, st,
                           lv, xs, reinterpret_cast<const float2 *>(dfs), n, hist + kSlabs, d_table);
    HM_CHECK_LAUNCH("hm_encode_bwd_table_ws");
    return HM_OK;
}

int hm_encode_bwd_table_tracked(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                                int64_t d_feat_stride, float *d_table, int frac_mode, uint32_t *touched_bits,
                                int32_t *touched_count, int32_t *touched_rows, int64_t cap, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_bwd_table_tracked: desc is NULL");
    HM_CHECK_ARG(n >= 0 && cap >= 0, "hm_encode_bwd_table_tracked: negative count");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_encode_bwd_table_tracked: bad frac_mode");
    HM_CHECK_ARG(d_feat_stride >= desc->lv.L * desc->lv.F, "hm_encode_bwd_table_tracked: d_feat_stride < L*F");
    HM_CHECK_ARG(desc->total_rows < ((uint64_t)1 << 31), "hm_encode_bwd_table_tracked: table too large for 32-bit row ids");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && d_feat && d_table && touched_bits && touched_count && touched_rows,
                 "hm_encode_bwd_table_tracked: NULL pointer");
    const int64_t threads = n * desc->lv.L * (frac_mode == HM_FRAC_REFERENCE ? 1 : 8);
    const int64_t grid = (threads + kThreads - 1) / kThreads;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_encode_bwd_table_tracked: n too large for one launch");
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(encode_bwd_table_tracked_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), desc->lv, x, n, d_feat, d_feat_stride, d_table, touched_bits,
                           touched_count, touched_rows, cap);
    else
        hipLaunchKernelGGL(encode_bwd_table_tracked_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), desc->lv, x, n, d_feat, d_feat_stride, d_table, touched_bits,
                           touched_count, touched_rows, cap);
    HM_CHECK_LAUNCH("hm_encode_bwd_table_tracked");
    return HM_OK;
}

int hm_encode_rows(const hm_grid_desc *desc, const float *x, int64_t n, int frac_mode, int32_t *keys_out,
                   float *weights_out, void *stream) {
    HM_CHECK_ARG(desc != nullptr && n >= 0, "hm_encode_rows: bad argument");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_encode_rows: bad frac_mode");
    HM_CHECK_ARG(desc->total_rows < ((uint64_t)1 << 31), "hm_encode_rows: table too large for 32-bit row keys");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && keys_out, "hm_encode_rows: NULL pointer");
    const int C = frac_mode == HM_FRAC_REFERENCE ? 1 : 8;
    HM_CHECK_ARG(C == 1 || weights_out, "hm_encode_rows: trilinear mode needs the weight output");
    const int64_t total = n * desc->lv.L * C;
    const int64_t grid = (total + kThreads - 1) / kThreads;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_encode_rows: n too large for one launch");
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(encode_rows_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), desc->lv, x, n, keys_out, weights_out);
    else
        hipLaunchKernelGGL(encode_rows_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), desc->lv, x, n, keys_out, weights_out);
    HM_CHECK_LAUNCH("hm_encode_rows");
    return HM_OK;
}

int hm_encode_bwd_table_sorted(const hm_grid_desc *desc, const int32_t *keys_sorted, const int64_t *perm, int64_t n_keys,
                               int corners, const float *d_feat, int64_t d_feat_stride, const float *weights,
                               float *d_table, void *stream) {
    HM_CHECK_ARG(desc != nullptr && n_keys >= 0 && (corners == 1 || corners == 8), "hm_encode_bwd_table_sorted: bad argument");
    HM_CHECK_ARG(d_feat_stride >= desc->lv.L * desc->lv.F && desc->lv.F <= 8, "hm_encode_bwd_table_sorted: bad stride / F");
    if (n_keys == 0) return HM_OK;
    HM_CHECK_ARG(keys_sorted && perm && d_feat && d_table, "hm_encode_bwd_table_sorted: NULL pointer");
    const int64_t grid = (n_keys + kThreads - 1) / kThreads;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_encode_bwd_table_sorted: too many contributions for one launch");
    hipLaunchKernelGGL(segment_scatter_kernel, dim3((unsigned)grid), dim3(kThreads), 0, as_stream(stream), keys_sorted,
                       perm, n_keys, desc->lv.L, desc->lv.F, corners, d_feat, d_feat_stride, weights, d_table);
    HM_CHECK_LAUNCH("hm_encode_bwd_table_sorted");
    return HM_OK;
}

int hm_diag_gather_calib(const float *table, int64_t table_bytes, int64_t n, int group, int stride_bytes, float *out,
                         void *stream) {
    HM_CHECK_ARG(table && out && n > 0 && table_bytes >= 128, "hm_diag_gather_calib: bad argument");
    HM_CHECK_ARG(group >= 1 && group <= 16 && stride_bytes >= 0 && stride_bytes % 8 == 0 &&
                     (group - 1) * stride_bytes + 8 <= 128,
                 "hm_diag_gather_calib: the group's rows must stay inside one 128-B block");
    const int64_t grid = (n + 255) / 256;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_diag_gather_calib: n too large");
    hipLaunchKernelGGL(gather_calib_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(table), (uint64_t)(table_bytes / 128), n, group, stride_bytes,
                       out);
    HM_CHECK_LAUNCH("hm_diag_gather_calib");
    return HM_OK;
}

int hm_corner_ids(const hm_grid_desc *desc, int level, const float *x, int64_t n, int32_t *xi_out,
                  uint32_t *ids_out, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_corner_ids: desc is NULL");
    HM_CHECK_ARG(level >= 0 && level < desc->lv.L, "hm_corner_ids: level out of range");
    HM_CHECK_ARG(n >= 0, "hm_corner_ids: n < 0");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && ids_out, "hm_corner_ids: NULL pointer");
    const int64_t threads = n * 8;
    const unsigned grid = (unsigned)((threads + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(corner_ids_kernel, dim3(grid), dim3(kThreads), 0, as_stream(stream), desc->lv.res[level],
                       desc->lv.rows[level], desc->lv.magic[level], x, n, xi_out, ids_out);
    HM_CHECK_LAUNCH("hm_corner_ids");
    return HM_OK;
}

int64_t hm_encode_workspace_bytes(const hm_grid_desc *desc, int64_t n) {
    if (!desc || n < 0) return hm_fail(HM_ERR_INVALID, "hm_encode_workspace_bytes: bad argument");
    // [slab counters 2*kSlabs u32 | order n u32 | slab ids n u16]
    return (int64_t)sizeof(uint32_t) * (n + 2 * kSlabs) + (int64_t)sizeof(uint16_t) * ((n + 1) & ~(int64_t)1);
}

static int encode_fwd_impl(const hm_grid_desc *desc, const float *x, int64_t n, const float *table,
                           const float *B_fourier, float *out, int64_t out_stride, int frac_mode, void *workspace,
                           int64_t workspace_bytes, void *stream);

int hm_encode_fwd(const hm_grid_desc *desc, const float *x, int64_t n, const float *table, const float *B_fourier,
                  float *out, int64_t out_stride, int frac_mode, void *stream) {
    return encode_fwd_impl(desc, x, n, table, B_fourier, out, out_stride, frac_mode, nullptr, 0, stream);
}

int hm_encode_fwd_ws(const hm_grid_desc *desc, const float *x, int64_t n, const float *table, const float *B_fourier,
                     float *out, int64_t out_stride, int frac_mode, void *workspace, int64_t workspace_bytes,
                     void *stream) {
    return encode_fwd_impl(desc, x, n, table, B_fourier, out, out_stride, frac_mode, workspace, workspace_bytes,
                           stream);
}

static int encode_fwd_impl(const hm_grid_desc *desc, const float *x, int64_t n, const float *table,
                           const float *B_fourier, float *out, int64_t out_stride, int frac_mode, void *workspace,
                           int64_t workspace_bytes, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_fwd: desc is NULL");
    HM_CHECK_ARG(n >= 0, "hm_encode_fwd: n < 0");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR, "hm_encode_fwd: bad frac_mode");
    const HmLevels &lv = desc->lv;
    const int width = B_fourier ? lv.E : lv.L * lv.F;
    HM_CHECK_ARG(out_stride >= width, "hm_encode_fwd: out_stride < row width");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && table && out, "hm_encode_fwd: NULL pointer");
    static const int sweep_cfg = [] { const char *e = getenv("HM_ENCODE_SWEEP"); return e ? atoi(e) : 1; }();
    // cache policy of the output-row stores / x loads (see store_tile_vec); HM_ENCODE_FLAGS overrides for experiments
    // default: nt stores of the output rows (C2 4.61 -> 4.77 TB/s, C4 3.61 -> 3.73; sc1 / nt loads / 4 levels per
    // step measured equal or slower, profiles/README.md); tables beyond 64 MiB run ONE workgroup per CU (C4: 3.89)
    static const int enc_flags = [] { const char *e = getenv("HM_ENCODE_FLAGS"); return e ? atoi(e) : 2; }();
    static const int sweep_grid_env = [] { const char *e = getenv("HM_ENCODE_GRID"); return e ? atoi(e) : 0; }();
    const int sweep_grid = sweep_grid_env > 0 ? sweep_grid_env
                                              : (desc->total_rows * (uint64_t)lv.F * 4u > (64u << 20) ? 256 : 512);
    const size_t lds_sweep = sizeof(float) * (size_t)(kTileS * width + kTileS * 3);
    const bool table_exceeds_l2 = desc->total_rows * (uint64_t)lv.F * 4u > (8u << 20);   // (C1's 0.9 MiB: tile kernel)
    static const int zorder_cfg = [] { const char *e = getenv("HM_ENCODE_ZORDER"); return e ? atoi(e) : 1; }();
    const size_t lds_z = sizeof(float) * (size_t)(kTileS * ((width + 3) & ~3) + kTileS * 3 + kTileS);
    if (lv.F == 2 && zorder_cfg != 0 && workspace && table_exceeds_l2 && n >= (int64_t)131072 && n < ((int64_t)1 << 32) &&
        lds_z <= 160 * 1024 && workspace_bytes >= (int64_t)sizeof(uint32_t) * (n + 2 * kSlabs)) {
        static thread_local bool attr_z = false;
        if (!attr_z) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_fwd_f2_zorder_kernel<HM_FRAC_REFERENCE>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_fwd_f2_zorder_kernel<HM_FRAC_TRILINEAR>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
            attr_z = true;
        }
        hipStream_t st = as_stream(stream);
        uint32_t *hist = static_cast<uint32_t *>(workspace);        // [kSlabs] counts -> start offsets -> cursors
        uint32_t *order = hist + 2 * kSlabs;                          // [n]
        // [n] uint16 slab ids behind the order array when the caller's workspace has room for them (older, smaller
        // workspaces still work: the scatter pass then recomputes the slabs from x)
        uint16_t *slab = workspace_bytes >= hm_encode_workspace_bytes(desc, n) ? reinterpret_cast<uint16_t *>(order + n)
                                                                             : nullptr;
        hm_zero_u32_async(hist, kSlabs, st);
        const unsigned g_sort = (unsigned)((n + kSortChunk - 1) / kSortChunk);
        hipLaunchKernelGGL(zsort_hist_kernel, dim3(g_sort), dim3(1024), 0, st, x, n, hist, slab);
        hipLaunchKernelGGL(zsort_scan_kernel, dim3(1), dim3(kSlabs), 0, st, hist);
        hipLaunchKernelGGL(zsort_scatter_kernel, dim3(g_sort), dim3(1024), 0, st, x, n, hist, order,
                           static_cast<const uint16_t *>(slab));
        const int64_t tiles = (n + kTileS - 1) / kTileS;
        static const int z_grid = [] { const char *e = getenv("HM_ENCODE_ZGRID"); return e ? atoi(e) : 512; }();
        const unsigned grid = (unsigned)(tiles < z_grid ? ((tiles + 7) / 8) * 8 : z_grid);
        if (frac_mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(encode_fwd_f2_zorder_kernel<HM_FRAC_REFERENCE>, dim3(grid), dim3(kThreadsS), lds_z, st,
                               lv, x, n, reinterpret_cast<const float2 *>(table), B_fourier, out, out_stride, order);
        else
            hipLaunchKernelGGL(encode_fwd_f2_zorder_kernel<HM_FRAC_TRILINEAR>, dim3(grid), dim3(kThreadsS), lds_z, st,
                               lv, x, n, reinterpret_cast<const float2 *>(table), B_fourier, out, out_stride, order);
    } else if (lv.F == 2 && sweep_cfg != 0 && table_exceeds_l2 && n >= (int64_t)131072 && lds_sweep <= 160 * 1024) {
        // big launches over big tables: level-synchronous persistent kernel (two workgroups per CU)
        static thread_local bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipSuccess;
            const void *fns[4] = {reinterpret_cast<const void *>(encode_fwd_f2_sweep_kernel<HM_FRAC_REFERENCE, 2>),
                                  reinterpret_cast<const void *>(encode_fwd_f2_sweep_kernel<HM_FRAC_TRILINEAR, 2>),
                                  reinterpret_cast<const void *>(encode_fwd_f2_sweep_kernel<HM_FRAC_REFERENCE, 4>),
                                  reinterpret_cast<const void *>(encode_fwd_f2_sweep_kernel<HM_FRAC_TRILINEAR, 4>)};
            for (int i = 0; i < 4 && e == hipSuccess; ++i)
                e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
            attr_done = true;
        }
        const int64_t tiles = (n + kTileS - 1) / kTileS;
        const unsigned grid = (unsigned)(tiles < sweep_grid ? tiles : sweep_grid);
        const bool kl4 = (enc_flags & 8) != 0;
#define HM_SWEEP_LAUNCH(FR, KLV)                                                                                    \
    hipLaunchKernelGGL((encode_fwd_f2_sweep_kernel<FR, KLV>), dim3(grid), dim3(kThreadsS), lds_sweep,               \
                       as_stream(stream), lv, x, n, reinterpret_cast<const float2 *>(table), B_fourier, out,        \
                       out_stride, enc_flags)
        if (frac_mode == HM_FRAC_REFERENCE) { if (kl4) HM_SWEEP_LAUNCH(HM_FRAC_REFERENCE, 4); else HM_SWEEP_LAUNCH(HM_FRAC_REFERENCE, 2); }
        else { if (kl4) HM_SWEEP_LAUNCH(HM_FRAC_TRILINEAR, 4); else HM_SWEEP_LAUNCH(HM_FRAC_TRILINEAR, 2); }
#undef HM_SWEEP_LAUNCH
    } else if (lv.F == 2) {
        const int64_t tiles = (n + kTile - 1) / kTile;
        HM_CHECK_ARG(tiles <= 0x7fffffffLL, "hm_encode_fwd: n too large for one launch");
        const size_t lds = sizeof(float) * (size_t)(kTile * width + kTile * 3);
        if (frac_mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(encode_fwd_f2_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)tiles), dim3(kThreads), lds,
                               as_stream(stream), lv, x, n, reinterpret_cast<const float2 *>(table), B_fourier, out,
                               out_stride, enc_flags);
        else
            hipLaunchKernelGGL(encode_fwd_f2_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)tiles), dim3(kThreads), lds,
                               as_stream(stream), lv, x, n, reinterpret_cast<const float2 *>(table), B_fourier, out,
                               out_stride, enc_flags);
    } else {
        const int64_t threads = n * (lv.L + 1);
        const int64_t grid = (threads + kThreads - 1) / kThreads;
        HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_encode_fwd: n too large for one launch");
        if (frac_mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(encode_fwd_generic_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreads), 0,
                               as_stream(stream), lv, x, n, table, B_fourier, out, out_stride);
        else
            hipLaunchKernelGGL(encode_fwd_generic_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreads), 0,
                               as_stream(stream), lv, x, n, table, B_fourier, out, out_stride);
    }
    HM_CHECK_LAUNCH("hm_encode_fwd");
    return HM_OK;
}

int hm_encode_bwd_table(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                        int64_t d_feat_stride, float *d_table, int frac_mode, void *stream) {
    HM_CHECK_ARG(desc != nullptr, "hm_encode_bwd_table: desc is NULL");
    HM_CHECK_ARG(n >= 0, "hm_encode_bwd_table: n < 0");
    HM_CHECK_ARG(frac_mode == HM_FRAC_REFERENCE || frac_mode == HM_FRAC_TRILINEAR,
                 "hm_encode_bwd_table: bad frac_mode");
    HM_CHECK_ARG(d_feat_stride >= desc->lv.L * desc->lv.F, "hm_encode_bwd_table: d_feat_stride < L*F");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(x && d_feat && d_table, "hm_encode_bwd_table: NULL pointer");
    const HmLevels &lv = desc->lv;
    const int64_t threads = n * lv.L * (frac_mode == HM_FRAC_REFERENCE ? 1 : 8);
    const int64_t grid = (threads + kThreads - 1) / kThreads;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_encode_bwd_table: n too large for one launch");
    if (desc->total_rows <= (uint64_t)kSmallTableRows && (int64_t)desc->total_rows * lv.F * 4 <= 48 * 1024 && threads >= 4096) {
        // small table, many contributions: LDS-privatised sums, at most 64 workgroups
        const unsigned g_small = (unsigned)(grid < 64 ? grid : 64);
        const size_t lds = sizeof(float) * (size_t)desc->total_rows * lv.F;
        if (frac_mode == HM_FRAC_REFERENCE)
            hipLaunchKernelGGL(encode_bwd_table_small_kernel<HM_FRAC_REFERENCE>, dim3(g_small), dim3(kThreads), lds,
                               as_stream(stream), lv, x, n, d_feat, d_feat_stride, d_table, (int)desc->total_rows);
        else
            hipLaunchKernelGGL(encode_bwd_table_small_kernel<HM_FRAC_TRILINEAR>, dim3(g_small), dim3(kThreads), lds,
                               as_stream(stream), lv, x, n, d_feat, d_feat_stride, d_table, (int)desc->total_rows);
        HM_CHECK_LAUNCH("hm_encode_bwd_table");
        return HM_OK;
    }
    if (frac_mode == HM_FRAC_REFERENCE)
        hipLaunchKernelGGL(encode_bwd_table_kernel<HM_FRAC_REFERENCE>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), lv, x, n, d_feat, d_feat_stride, d_table);
    else
        hipLaunchKernelGGL(encode_bwd_table_kernel<HM_FRAC_TRILINEAR>, dim3((unsigned)grid), dim3(kThreads), 0,
                           as_stream(stream), lv, x, n, d_feat, d_feat_stride, d_table);
    HM_CHECK_LAUNCH("hm_encode_bwd_table");
    return HM_OK;
}

}  // extern "C"
