// hm_gemm.hip - exact-fp32 MFMA GEMM for the grad-enabled path of the SDF / rendering MLPs
// (forward X*W^T+b, backward dY*W and dY^T*X, and the same three shapes again in the
// double-backward pass that ImplicitNetwork.gradient(create_graph=True) needs;
// reference: model/implicit_differentiable_renderer.py:102,116-128,211-221).
//
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]);  row-major with leading dimensions, any M/N/K
// (the MLP has K = 67, N = 445 and 257, M = number of points), guarded loads, optional split-K
// with fp32 atomics for the weight-gradient shape (small M x N, K = number of points).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); 4 waves per workgroup as 2 x 2, each owning
// TM x TN tiles of 32 x 32; both operands are staged through LDS in the k-grouped image
// [k/4][row][4] (rows XOR-swizzled by the k-group) so that both the 16-B staging stores and the
// MFMA operand fetches (ds_read_b128) are bank-conflict free; K is staged 64 deep for the 64x64
// tile so that one stage of MFMAs (~0.85 us) covers the L2 latency of the next stage's loads.
#include "hm_common.h"

#include <stdio.h>

#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int32_t M, N, K;
    int32_t transA, transB;  // op(A) = A^T if transA (A stored [K,M]); op(B) = B^T if transB (B stored [N,K])
    int64_t lda, ldb, ldc;
    int32_t k_chunk;  // K range handled by one blockIdx.z
    int32_t atomic;   // accumulate with atomics (split-K, or beta = 1)
    int32_t vecA, vecB;  // operand may be fetched with aligned 16-B loads
    int32_t nrecA, nrecB;  // pipelined kernel: bytes of the operands' buffer descriptors (reads beyond them return 0)
    hm_gemm_epilogue ep;  // fused elementwise epilogue (mode HM_EPI_NONE = plain store)
};

// LDS image of an operand tile: S[kg][row][4] (kg = k/4) with the row XOR-swizzled by kg so that both
// the 16-B staging stores (lanes = consecutive kg of one row) and the MFMA fragment loads (lanes =
// consecutive rows of one kg) are bank-conflict free.
template <int ROWS>
__device__ __forceinline__ int lds_slot(int kg, int row) {
    return (kg * ROWS + (row ^ (kg & 7))) * 4;
}

// Fetch one ROWS x BK operand tile into registers as float4 k-groups.
//   k-contiguous source (element (row,k) at P[row*ld + k]): lanes walk the k-groups of a row -> one
//   coalesced 16-B load per k-group (VEC) or four dword loads;
//   row-contiguous source (element (row,k) at P[k*ld + row]): lanes walk rows -> four coalesced dword
//   loads (k, k+1, k+2, k+3) per k-group.
// Every load is UNCONDITIONAL on a clamped in-range address and masked afterwards: a load under a
// per-element branch makes hipcc wait vmcnt(0) per element and serialises the stage's L2 round trips.
template <int ROWS, int BK, int NT, bool VEC>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int64_t ld, bool kcontig, int row0,
                                          int nrows, int k0, int kend, int tid, float4 (&r)[ROWS * BK / 4 / NT]) {
    constexpr int PER = ROWS * BK / 4 / NT;
    constexpr int KG = BK / 4;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + NT * i;
        const int row = kcontig ? e / KG : e % ROWS;
        const int kg = kcontig ? e % KG : e / ROWS;
        const int gr = row0 + row, gk = k0 + kg * 4;
        const int rc = min(gr, nrows - 1);
        const bool rv = gr < nrows;
        float4 v;
        if (VEC) {  // k-contiguous, kend % 4 == 0, 16-B aligned rows
            const int kk = min(gk, kend - 4);
            v = *reinterpret_cast<const float4 *>(P + (int64_t)rc * ld + kk);
            const bool ok = rv && gk < kend;
            v.x = ok ? v.x : 0.0f; v.y = ok ? v.y : 0.0f; v.z = ok ? v.z : 0.0f; v.w = ok ? v.w : 0.0f;
        } else {
            const int64_t sr = kcontig ? ld : 1, sk = kcontig ? 1 : ld;
            const float *base = P + (int64_t)rc * sr;
            const int c0 = min(gk, kend - 1), c1 = min(gk + 1, kend - 1), c2 = min(gk + 2, kend - 1),
                      c3 = min(gk + 3, kend - 1);
            const float x0 = base[(int64_t)c0 * sk], x1 = base[(int64_t)c1 * sk], x2 = base[(int64_t)c2 * sk],
                        x3 = base[(int64_t)c3 * sk];
            v.x = (rv && gk < kend) ? x0 : 0.0f;
            v.y = (rv && gk + 1 < kend) ? x1 : 0.0f;
            v.z = (rv && gk + 2 < kend) ? x2 : 0.0f;
            v.w = (rv && gk + 3 < kend) ? x3 : 0.0f;
        }
        r[i] = v;
    }
}

template <int ROWS, int BK, int NT>
__device__ __forceinline__ void store_tile(float *__restrict__ S, bool kcontig, int tid,
                                           const float4 (&r)[ROWS * BK / 4 / NT]) {
    constexpr int PER = ROWS * BK / 4 / NT;
    constexpr int KG = BK / 4;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + NT * i;
        int row, kg;
        if (kcontig) { row = e / KG; kg = e % KG; } else { kg = e / ROWS; row = e % ROWS; }
        *reinterpret_cast<float4 *>(S + lds_slot<ROWS>(kg, row)) = r[i];
    }
}

// Store one 32x32 accumulator tile: lane holds column n, registers hold rows mrow0 + (r&3) + 8(r>>2)
// (mrow0 already includes the lane half's +4).  EP selects the fused Softplus epilogues (include/hashmod.h).
template <bool EP>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs &g, const f32x16 &acc, int n, int mrow0, bool add_bias) {
    const float bv = add_bias ? g.bias[n] : 0.0f;
    const int mode = EP ? g.ep.mode : (int)HM_EPI_NONE;
    // epilogue operands first, all 16 (+16) loads in flight together on clamped addresses: a load under
    // a per-row guard would be waited for one at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
    float zv[8], gv[8];
    if (mode == HM_EPI_S1MUL || mode == HM_EPI_ADJOINT || mode == HM_EPI_RELUMASK) {
        const int nc = mode != HM_EPI_ADJOINT ? min(n, g.ep.nz - 1) : n;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * half + q;
            const int mc = min(mrow0 + (r & 3) + 8 * (r >> 2), g.M - 1);
            zv[q] = g.ep.z[(int64_t)mc * g.ep.ldz + nc];
        }
        if (g.ep.g) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * half + q;
                const int mc = min(mrow0 + (r & 3) + 8 * (r >> 2), g.M - 1);
                gv[q] = g.ep.g[(int64_t)mc * g.ep.ldg + nc];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) gv[q] = 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 8 * half + q;
        const int m = mrow0 + (r & 3) + 8 * (r >> 2);
        if (m >= g.M) continue;
        float v = acc[r] + bv;
        if (mode != HM_EPI_NONE) v *= g.ep.scale;
        if (g.C) {
            float *dst = g.C + (int64_t)m * g.ldc + n;
            if (g.atomic)
                atomicAdd(dst, v);
            else
                *dst = v;
        }
        if (mode == HM_EPI_SOFTPLUS) {
            g.ep.out1[(int64_t)m * g.ep.ld1 + n] = hm_softplus_fwd(v, g.ep.beta, g.ep.threshold);
        } else if (mode == HM_EPI_RELU) {
            g.ep.out1[(int64_t)m * g.ep.ld1 + n] = fmaxf(v, 0.0f);
        } else if (mode == HM_EPI_RELUMASK) {
            if (n < g.ep.nz) g.ep.out1[(int64_t)m * g.ep.ld1 + n] = (zv[q] > 0.0f ? v : 0.0f) + gv[q];
        } else if (mode == HM_EPI_S1MUL) {
            if (n < g.ep.nz)
                g.ep.out1[(int64_t)m * g.ep.ld1 + n] =
                    v * hm_sp_deriv(zv[q], g.ep.beta, g.ep.threshold).s1 + gv[q];
        } else if (mode == HM_EPI_ADJOINT) {
            const SpDeriv d = hm_sp_deriv(zv[q], g.ep.beta, g.ep.threshold);
            g.ep.out1[(int64_t)m * g.ep.ld1 + n] = v * d.s1;
            g.ep.out2[(int64_t)m * g.ep.ld2 + n] = v * gv[q] * d.s2;
            if (g.ep.out3) g.ep.out3[(int64_t)m * g.ep.ld3 + n] = gv[q] * d.s1;
        }
    }
    }
}

// KS = intra-workgroup K split: 4*KS waves, wave group g multiplies octets [g*BK/8/KS, (g+1)*BK/8/KS) of every
// stage, partial tiles are summed through LDS at the end.  Two waves per SIMD keep the matrix pipe busy
// while the next stage's loads are in flight even when the grid has only one workgroup per CU.
// EP: compiled with the fused epilogues (kept out of the plain instantiation: its extra registers would drop the
// 64x64 configuration from two resident workgroups per CU to one).  The 8-wave configuration is held to 128 VGPRs
// (4 waves per SIMD = two workgroups per CU).
// WM: wave rows of the tile (2 -> 64*TM rows; 1 -> 32*TM rows for row counts that would leave the 64-row grid
// with fewer than two workgroups per CU); the wave columns are always 2.
template <int TM, int TN, int BK, int KS, bool VA, bool VB, bool EP, int WM = 2>
__global__ __launch_bounds__(128 * WM * KS, (WM * KS == 4 ? 4 : 1)) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 32 * WM * TM, BN = 64 * TN, NT = 128 * WM * KS, WG = 2 * WM;  // WG = waves per k part
    __shared__ __align__(16) float smem[BK * (BM + BN)];
    float *As = smem, *Bs = smem + BK * BM;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int kpart = wave / WG, wsub = wave % WG;
    const int wm = wsub >> 1, wn = wsub & 1;
    const int j = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const bool a_kc = (g.transA == 0);  // A[m*lda + k]
    const bool b_kc = (g.transB != 0);  // B stored [N,K]: B[n*ldb + k]

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float4 ra[BM * BK / 4 / NT], rb[BN * BK / 4 / NT];
    load_tile<BM, BK, NT, VA>(g.A, g.lda, a_kc, m0, g.M, kbeg, kend, tid, ra);
    load_tile<BN, BK, NT, VB>(g.B, g.ldb, b_kc, n0, g.N, kbeg, kend, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();  // previous stage fully consumed
        store_tile<BM, BK, NT>(As, a_kc, tid, ra);
        store_tile<BN, BK, NT>(Bs, b_kc, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) {  // prefetch the next stage into registers while this one is multiplied
            load_tile<BM, BK, NT, VA>(g.A, g.lda, a_kc, m0, g.M, k0 + BK, kend, tid, ra);
            load_tile<BN, BK, NT, VB>(g.B, g.ldb, b_kc, n0, g.N, k0 + BK, kend, tid, rb);
        }
        constexpr int OCT = BK / 8 / KS;
#pragma unroll
        for (int oo = 0; oo < OCT; ++oo) {
            float4 a[TM], b[TN];
            const int kg = 2 * (kpart * OCT + oo) + h;
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t] = *reinterpret_cast<const float4 *>(As + lds_slot<BM>(kg, wm * 32 * TM + t * 32 + j));
#pragma unroll
            for (int t = 0; t < TN; ++t)
                b[t] = *reinterpret_cast<const float4 *>(Bs + lds_slot<BN>(kg, wn * 32 * TN + t * 32 + j));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float av = s == 0 ? a[tm].x : s == 1 ? a[tm].y : s == 2 ? a[tm].z : a[tm].w;
                        const float bv = s == 0 ? b[tn].x : s == 1 ? b[tn].y : s == 2 ? b[tn].z : b[tn].w;
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
                    }
        }
    }

    if (KS > 1) {
        // sum the wave groups' partial tiles through LDS (the staging buffers are free now)
        static_assert(KS == 1 || (TM == 1 && TN == 1), "intra-workgroup K split is built for the 64x64 tile");
        __syncthreads();
        float *red = smem;  // (KS-1) * WG waves * 16 regs * 64 lanes floats
        static_assert((KS - 1) * WG * 16 * 64 <= BK * (BM + BN), "K-split reduction does not fit the staging buffers");
        if (kpart > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((kpart - 1) * WG + wsub) * 16 + r) * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (kpart > 0) return;
#pragma unroll
        for (int p = 0; p < KS - 1; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += red[((p * WG + wsub) * 16 + r) * 64 + lane];
    }

    // epilogue: lane holds column n, registers hold rows (r&3) + 8(r>>2) + 4h
    const bool add_bias = (g.bias != nullptr) && (blockIdx.z == 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * 32 * TN + tn * 32 + j;
            if (n >= g.N) continue;
            gemm_store_tile<EP>(g, acc[tm][tn], n, m0 + wm * 32 * TM + tm * 32 + 4 * h, add_bias);
        }
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined 64x64 tile for the training shapes (M = 2048...4096 rows, N = K = 512: 256-512 tiles, i.e. ONE or two
// workgroups per CU, 6.8 us of matrix work each).  With so little work per tile the steady state of the software
// pipeline has to be tight: 32-deep K stages (16 of them at K = 512), global loads issued kPipeD-1 stages ahead into
// a register ring, two LDS buffers, and the MFMA operands of a stage held in registers (two sets) so that the LDS
// round trip of stage s+1 - store, the ONE barrier of the stage, fragment reads - is issued in the middle of stage
// s's MFMAs.  (A wave that has passed the barrier of stage s has read the fragments of stage s-1's buffer long
// before, so that buffer is free to refill.)  The loop body has NO conditionals - the host launches this kernel
// only when the K range is a whole number of kPipeD-stage groups, partial edge tiles clamp their row pointers once,
// before the loop - because with the
// generic kernel's guards in it hipcc keeps the accumulators in VGPRs across the back edge and copies all 32 of
// them to AGPRs and back every stage.
// AKC / BKC: operand is k-contiguous and 16-B aligned (one dwordx4 per k-group), else row-contiguous (four dwords).
constexpr int kPipeBK = 32, kPipeD = 4;

template <bool KC, int PER>
__device__ __forceinline__ void pipe_fetch(const float *const (&p)[PER], int64_t step, int64_t ld, int sc,
                                           float4 (&x)[PER]) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float *q = p[i] + sc * step;
        if (KC) {   // (component-wise: a whole-struct float4 copy into the array keeps it in scratch memory)
            const float4 t = *reinterpret_cast<const float4 *>(q);
            x[i] = make_float4(t.x, t.y, t.z, t.w);
        } else {
            x[i] = make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
        }
    }
}
// buffer-load form of pipe_fetch: voff = the thread's byte offset (loop invariant), soff = the stage's byte offset (scalar)
template <bool KC, int PER>
__device__ __forceinline__ void pipe_fetch_buf(const __amdgpu_buffer_rsrc_t &rs, const int (&voff)[PER], int soff, int ld4,
                                               float4 (&x)[PER]) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (KC) {
            const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0);
            x[i] = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
        } else {
            x[i] = make_float4(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], soff, 0)),
                               __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], soff + ld4, 0)),
                               __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], soff + 2 * ld4, 0)),
                               __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], soff + 3 * ld4, 0)));
        }
    }
}

template <int OCT, int AROWS = 64>
__device__ __forceinline__ void pipe_read_ops(const float *as, const float *bs, int arow, int brow, int h,
                                              float4 (&xa)[OCT], float4 (&xb)[OCT]) {
#pragma unroll
    for (int oo = 0; oo < OCT; ++oo) {
        const int kg = 2 * oo + h;
        xa[oo] = *reinterpret_cast<const float4 *>(as + lds_slot<AROWS>(kg, arow));
        xb[oo] = *reinterpret_cast<const float4 *>(bs + lds_slot<64>(kg, brow));
    }
}
__device__ __forceinline__ void pipe_mfma_oct(const float4 &a, const float4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
}


template <bool AKC, bool BKC, bool EP>
__global__ __launch_bounds__(256) void gemm_f32_pipe_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = kPipeBK, NT = 256, PER = BM * BK / 4 / NT, KG = BK / 4, OCT = BK / 8;
    __shared__ __align__(16) float As[2][BK * BM];
    __shared__ __align__(16) float Bs[2][BK * BN];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int j = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int stages = g.k_chunk / BK;   // multiple of kPipeD (host)

    const float *pa[PER], *pb[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + NT * i;
        // rows / columns past the edge of a partial tile are clamped HERE, once (they compute values nobody stores)
        const int am = min(m0 + (AKC ? e / KG : e % BM), g.M - 1), bn = min(n0 + (BKC ? e / KG : e % BN), g.N - 1);
        pa[i] = AKC ? g.A + (int64_t)am * g.lda + kbeg + (e % KG) * 4
                    : g.A + (int64_t)(kbeg + (e / BM) * 4) * g.lda + am;
        pb[i] = BKC ? g.B + (int64_t)bn * g.ldb + kbeg + (e % KG) * 4
                    : g.B + (int64_t)(kbeg + (e / BN) * 4) * g.ldb + bn;
    }
    const int64_t sa = AKC ? BK : (int64_t)BK * g.lda, sb = BKC ? BK : (int64_t)BK * g.ldb;
    const int64_t lda = g.lda, ldb = g.ldb;
    f32x16 acc, acc2;   // even / odd k-octets: two independent MFMA chains for the one wave on each SIMD
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    // the four ring slots and the two operand sets are separate named arrays (an array indexed by (u+3)%4 inside
    // the unrolled loop was demoted to scratch memory by hipcc)
    static_assert(kPipeD == 4, "the stage macro below is written out for a 4-deep ring (8-deep measured no faster)");
    float4 ra0[PER], ra1[PER], ra2[PER], ra3[PER], rb0[PER], rb1[PER], rb2[PER], rb3[PER];
    float4 opa0[OCT], opa1[OCT], opb0[OCT], opb1[OCT];
#define HM_PIPE_FETCH(S_, RA_, RB_)                                                                 \
    do {                                                                                            \
        const int sc_ = min((S_), stages - 1); /* past the end: re-read the last stage, never multiplied */ \
        pipe_fetch<AKC, PER>(pa, sa, lda, sc_, RA_);                                                \
        pipe_fetch<BKC, PER>(pb, sb, ldb, sc_, RB_);                                                \
    } while (0)
    HM_PIPE_FETCH(0, ra0, rb0);
    HM_PIPE_FETCH(1, ra1, rb1);
    HM_PIPE_FETCH(2, ra2, rb2);
    store_tile<BM, BK, NT>(As[0], AKC, tid, ra0);
    store_tile<BN, BK, NT>(Bs[0], BKC, tid, rb0);
    __syncthreads();
    pipe_read_ops<OCT>(As[0], Bs[0], wm * 32 + j, wn * 32 + j, h, opa0, opb0);
// one stage: multiply from operand set C, meanwhile fetch stage s+D-1 into ring slot F and move ring slot N (stage
// s+1) through LDS buffer NB into operand set X
#define HM_PIPE_STAGE(S_, F_, N_, C_, X_, NB_)                                                      \
    do {                                                                                            \
        HM_PIPE_FETCH((S_) + kPipeD - 1, ra##F_, rb##F_);                                                    \
        pipe_mfma_oct(opa##C_[0], opb##C_[0], acc);                                                 \
        pipe_mfma_oct(opa##C_[1], opb##C_[1], acc2);                                                \
        store_tile<BM, BK, NT>(As[NB_], AKC, tid, ra##N_);                                          \
        store_tile<BN, BK, NT>(Bs[NB_], BKC, tid, rb##N_);                                          \
        __syncthreads();                                                                            \
        pipe_read_ops<OCT>(As[NB_], Bs[NB_], wm * 32 + j, wn * 32 + j, h, opa##X_, opb##X_);        \
        pipe_mfma_oct(opa##C_[2], opb##C_[2], acc);                                                 \
        pipe_mfma_oct(opa##C_[3], opb##C_[3], acc2);                                                \
    } while (0)
    static_assert(OCT == 4, "HM_PIPE_STAGE is written for 4 octets per stage");
    for (int s0 = 0; s0 < stages; s0 += kPipeD) {
        HM_PIPE_STAGE(s0 + 0, 3, 1, 0, 1, 1);
        HM_PIPE_STAGE(s0 + 1, 0, 2, 1, 0, 0);
        HM_PIPE_STAGE(s0 + 2, 1, 3, 0, 1, 1);
        HM_PIPE_STAGE(s0 + 3, 2, 0, 1, 0, 0);
    }
#undef HM_PIPE_STAGE
#undef HM_PIPE_FETCH
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    const int n = n0 + wn * 32 + j;
    if (n < g.N) gemm_store_tile<EP>(g, acc, n, m0 + wm * 32 + 4 * h, (g.bias != nullptr) && (blockIdx.z == 0));
}

// Same pipeline with EIGHT waves: two wave groups split the octets of every stage (two waves per SIMD hide each
// other's barrier / LDS turnarounds), partial tiles are summed through LDS at the end.
// WM = 3: a 96 x 64 tile on TWELVE waves (two groups of 3 x 2), for row counts whose 64-row grid is between one and two
// rounds of the chip: M = 3072, N = 512 are 384 tiles of 64 x 64 - a CU with two of them takes twice as long as the one
// with one - but exactly 256 of 96 x 64.  The B panel is staged by the first 512 threads.
template <bool AKC, bool BKC, bool EP, int WM = 2>
__device__ __forceinline__ void gemm_pipe2_body(const GemmArgs &g, int bx, int by, int bz) {
    constexpr int BM = 32 * WM, BN = 64, BK = kPipeBK, NT = 256 * WM, NTB = 512, PER = BM * BK / 4 / NT, KG = BK / 4,
                  OCT = BK / 8 / 2;   // OCT: octets of a stage per wave group
    static_assert(PER == 1 && BN * BK / 4 / NTB == 1, "one k-group per thread and operand");
    __shared__ __align__(16) float As[2][BK * BM];
    __shared__ __align__(16) float Bs[2][BK * BN];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int kpart = wave / (2 * WM), wsub = wave % (2 * WM);   // two wave groups split the octets of every stage
    const int wm = wsub >> 1, wn = wsub & 1;
    const int j = lane & 31, h = lane >> 5;
    const int m0 = bx * BM, n0 = by * BN;
    const int kbeg = bz * g.k_chunk;
    const int stages = g.k_chunk / BK;   // multiple of kPipeD (host)

    // operand fetches are BUFFER loads: descriptor on the matrix (SGPRs), the thread's own byte offset in ONE loop-invariant
    // VGPR, the stage's offset scalar - no 64-bit address arithmetic on the VALU inside the loop (VALU instructions are
    // serial with the MFMAs of every wave on the SIMD; the fused SDF kernels gained 3 - 11 % from the same change).
    // The host takes this kernel only for operands below 2 GB.
    // The descriptors end at the operands' last element: a K range that is no multiple of the stage group (K = 445, 257)
    // runs to the next multiple, the operand whose k is the SLOW dimension returns zeros beyond its end and cancels what the
    // k-contiguous one reads from its following rows (the host admits a K tail only when there is such an operand).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, g.nrecA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B), 0, g.nrecB, 0x00020000);
    int va[PER], vb[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + NT * i;
        const int eb = (tid % NTB) + NTB * i;   // (WM = 3: threads 512.. fetch a B k-group again and do not stage it)
        // rows / columns past the edge of a partial tile are clamped HERE, once (they compute values nobody stores)
        const int am = min(m0 + (AKC ? e / KG : e % BM), g.M - 1), bn = min(n0 + (BKC ? eb / KG : eb % BN), g.N - 1);
        va[i] = (int)(4 * (AKC ? (int64_t)am * g.lda + kbeg + (e % KG) * 4
                               : (int64_t)(kbeg + (e / BM) * 4) * g.lda + am));
        vb[i] = (int)(4 * (BKC ? (int64_t)bn * g.ldb + kbeg + (eb % KG) * 4
                               : (int64_t)(kbeg + (eb / BN) * 4) * g.ldb + bn));
    }
    const bool stage_b = NT == NTB || tid < NTB;
    const int sa = 4 * (AKC ? BK : BK * (int)g.lda), sb = 4 * (BKC ? BK : BK * (int)g.ldb);   // bytes per stage
    const int lda4 = 4 * (int)g.lda, ldb4 = 4 * (int)g.ldb;
    f32x16 acc, acc2;   // even / odd k-octets: two independent MFMA chains for the one wave on each SIMD
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    // the four ring slots and the two operand sets are separate named arrays (an array indexed by (u+3)%4 inside
    // the unrolled loop was demoted to scratch memory by hipcc)
    static_assert(kPipeD == 4, "the stage macro below is written out for a 4-deep ring (8-deep measured no faster)");
    float4 ra0[PER], ra1[PER], ra2[PER], ra3[PER], rb0[PER], rb1[PER], rb2[PER], rb3[PER];
    float4 opa0[OCT], opa1[OCT], opb0[OCT], opb1[OCT];
#define HM_PIPE_FETCH(S_, RA_, RB_)                                                                 \
    do {                                                                                            \
        const int sc_ = min((S_), stages - 1); /* past the end: re-read the last stage, never multiplied */ \
        pipe_fetch_buf<AKC, PER>(rsA, va, sc_ * sa, lda4, RA_);                                     \
        pipe_fetch_buf<BKC, PER>(rsB, vb, sc_ * sb, ldb4, RB_);                                     \
    } while (0)
    HM_PIPE_FETCH(0, ra0, rb0);
    HM_PIPE_FETCH(1, ra1, rb1);
    HM_PIPE_FETCH(2, ra2, rb2);
    store_tile<BM, BK, NT>(As[0], AKC, tid, ra0);
    if (stage_b) store_tile<BN, BK, NTB>(Bs[0], BKC, tid, rb0);
    __syncthreads();
    pipe_read_ops<OCT, BM>(As[0], Bs[0], wm * 32 + j, wn * 32 + j, h + 4 * kpart, opa0, opb0);
// one stage: multiply from operand set C, meanwhile fetch stage s+D-1 into ring slot F and move ring slot N (stage
// s+1) through LDS buffer NB into operand set X
#define HM_PIPE_STAGE(S_, F_, N_, C_, X_, NB_)                                                      \
    do {                                                                                            \
        HM_PIPE_FETCH((S_) + kPipeD - 1, ra##F_, rb##F_);                                                    \
        pipe_mfma_oct(opa##C_[0], opb##C_[0], acc);                                                 \
        store_tile<BM, BK, NT>(As[NB_], AKC, tid, ra##N_);                                          \
        if (stage_b) store_tile<BN, BK, NTB>(Bs[NB_], BKC, tid, rb##N_);                            \
        __syncthreads();                                                                            \
        pipe_read_ops<OCT, BM>(As[NB_], Bs[NB_], wm * 32 + j, wn * 32 + j, h + 4 * kpart, opa##X_, opb##X_);        \
        pipe_mfma_oct(opa##C_[1], opb##C_[1], acc2);                                                \
    } while (0)
    static_assert(OCT == 2, "HM_PIPE_STAGE is written for 2 octets per stage and wave group");
    for (int s0 = 0; s0 < stages; s0 += kPipeD) {
        HM_PIPE_STAGE(s0 + 0, 3, 1, 0, 1, 1);
        HM_PIPE_STAGE(s0 + 1, 0, 2, 1, 0, 0);
        HM_PIPE_STAGE(s0 + 2, 1, 3, 0, 1, 1);
        HM_PIPE_STAGE(s0 + 3, 2, 0, 1, 0, 0);
    }
#undef HM_PIPE_STAGE
#undef HM_PIPE_FETCH
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    {   // add the second wave group's partial tile (through the staging buffers, free now)
        __syncthreads();
        float *red = &As[0][0];   // 2 WM waves * 16 registers * 64 lanes floats == 2 * BK * BM
        static_assert(2 * BK * BM >= 2 * WM * 16 * 64, "reduction buffer");
        if (kpart == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wsub * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kpart == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(wsub * 16 + r) * 64 + lane];
    }
    const int n = n0 + wn * 32 + j;
    if (n < g.N) gemm_store_tile<EP>(g, acc, n, m0 + wm * 32 + 4 * h, (g.bias != nullptr) && (bz == 0));
}

template <bool AKC, bool BKC, bool EP>
__global__ __launch_bounds__(512, 4) void gemm_f32_pipe2_kernel(GemmArgs g) {
    gemm_pipe2_body<AKC, BKC, EP>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}
template <bool AKC, bool BKC, bool EP>
__global__ __launch_bounds__(768, 3) void gemm_f32_pipe2_m96_kernel(GemmArgs g) {
    gemm_pipe2_body<AKC, BKC, EP, 3>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped form for the weight gradients of one backward pass: C_p += A_p^T B_p for up to HM_GEMM_GROUP_MAX problems in ONE
// launch (A_p [K_p, M_p], B_p [K_p, N_p] row-major: both operands row-contiguous, the shape of  dW = [u; z-bar]^T [v-bar; a]).
// Launched one by one these GEMMs are 64 tiles each and need an 8-way split-K (and a zeroing launch) to fill the chip
// at all: 33 - 45 us per layer for 14 - 20 us of matrix work.  Here every (problem, tile, k part) is a workgroup of
// the same grid: ~1000 of them, four resident per CU, so that one problem's tail overlaps the next one's head.
struct GemmGroupEntry {
    const float *A, *B;
    float *C;
    int64_t lda, ldb, ldc;
    int32_t M, N, k_chunk, tiles_m, tiles_n, split;
};
struct GemmGroupTable {
    GemmGroupEntry e[HM_GEMM_GROUP_MAX];
    int32_t start[HM_GEMM_GROUP_MAX + 1];
    int32_t n;
};
__global__ __launch_bounds__(512, 4) void gemm_f32_pipe2_group_kernel(GemmGroupTable t) {
    int p = 0;
    while (p + 1 < t.n && (int)blockIdx.x >= t.start[p + 1]) ++p;       // (uniform: <= 16 problems)
    const GemmGroupEntry &E = t.e[p];
    int local = (int)blockIdx.x - t.start[p];
    const int bz = local % E.split;
    local /= E.split;
    const int by = local % E.tiles_n, bx = local / E.tiles_n;
    GemmArgs g;
    g.A = E.A; g.B = E.B; g.bias = nullptr; g.C = E.C;
    g.M = E.M; g.N = E.N; g.K = 0;
    g.transA = 1; g.transB = 0;
    g.lda = E.lda; g.ldb = E.ldb; g.ldc = E.ldc;
    g.k_chunk = E.k_chunk;
    g.atomic = 1;
    g.vecA = g.vecB = 0;
    g.nrecA = g.nrecB = 0x7fffffff;
    g.ep.mode = HM_EPI_NONE;
    gemm_pipe2_body<false, false, false>(g, bx, by, bz);
}

// zero an M x N window of C (split-K accumulates with atomics); a plain kernel instead of
// hipMemset2DAsync so that the call can be recorded into a HIP graph on every ROCm version
__global__ __launch_bounds__(256) void zero_window_kernel(float *C, int64_t ldc, int64_t M, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N, n = i - m * N;
    C[m * ldc + n] = 0.0f;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

static int gemm_impl(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                     const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int accumulate,
                     const hm_gemm_epilogue *ep, void *stream) {
    HM_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "hm_gemm_f32: negative dimension");
    HM_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "hm_gemm_f32: dimension too large");
    if (M == 0 || N == 0) return HM_OK;
    HM_CHECK_ARG(C != nullptr || (ep && ep->mode != HM_EPI_NONE), "hm_gemm_f32: C is NULL");
    HM_CHECK_ARG(K == 0 || (A && B), "hm_gemm_f32: NULL operand");
    HM_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && (!C || ldc >= N), "hm_gemm_f32: leading dimension");
    GemmArgs g;
    g.ep = hm_gemm_epilogue{};
    if (ep && ep->mode != HM_EPI_NONE) {
        HM_CHECK_ARG(!accumulate, "hm_gemm_f32_ep: an epilogue cannot be combined with accumulate");
        HM_CHECK_ARG(ep->mode >= HM_EPI_SOFTPLUS && ep->mode <= HM_EPI_RELUMASK, "hm_gemm_f32_ep: unknown epilogue mode");
        const bool masked = ep->mode == HM_EPI_S1MUL || ep->mode == HM_EPI_RELUMASK;   // out1 has nz columns
        HM_CHECK_ARG(ep->out1 && ep->ld1 >= (masked ? ep->nz : N), "hm_gemm_f32_ep: out1");
        if (ep->mode != HM_EPI_SOFTPLUS && ep->mode != HM_EPI_RELU) {
            const int64_t nzc = masked ? ep->nz : N;
            HM_CHECK_ARG(ep->z && ep->ldz >= nzc, "hm_gemm_f32_ep: z");
            HM_CHECK_ARG(!masked || (ep->nz >= 1 && ep->nz <= N), "hm_gemm_f32_ep: nz out of range");
            HM_CHECK_ARG(!ep->g || ep->ldg >= nzc, "hm_gemm_f32_ep: g");
        }
        if (ep->mode == HM_EPI_ADJOINT) {
            HM_CHECK_ARG(ep->g && ep->out2 && ep->ld2 >= N && (!ep->out3 || ep->ld3 >= N), "hm_gemm_f32_ep: ADJOINT operands");
        }
        g.ep = *ep;
    }
    g.A = A; g.B = B; g.bias = bias; g.C = C;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.transA = transA; g.transB = transB;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    // operands may be fetched with 16-B loads when every k-group of every row is 16-B aligned
    const bool a_kc = !transA, b_kc = transB != 0;
    // (and the K range ends on a multiple of 4, so no k-group straddles the end)
    g.vecA = (a_kc && K % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0) ? 1 : 0;
    g.vecB = (b_kc && K % 4 == 0 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15u) == 0) ? 1 : 0;
    // tile choice: big tiles only when they still fill the chip
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    static const int small_cfg = [] { const char *e = getenv("HM_GEMM_CFG"); return e ? atoi(e) : 0; }();
    const bool big = t128 >= 256;
    // (128x64 / 64x128 tiles and a forced 128x128 tile were measured too: 30-45 % slower on M = 1750...4822)
    // 32-row tiles (HM_GEMM_HALF=1) for row counts whose 64-row grid gives a CU fewer than two workgroups
    // (M = 2048...3072, N = 512).  Measured: the training step is 2 % SLOWER with them (10.50 vs 10.28 ms) -
    // twice the B-panel traffic outweighs the extra overlap - so they stay off.
    const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
    static const int half_cfg = [] { const char *e = getenv("HM_GEMM_HALF"); return e ? atoi(e) : 0; }();
    const bool half_rows = !big && small_cfg == 0 && half_cfg != 0 && t64 >= 128 && t64 < 512 && M >= 256;
    static const int pipe_cfg = [] { const char *e = getenv("HM_GEMM_PIPE"); return e ? atoi(e) : 2; }();   // 2 = eight-wave variant
    // the pipelined kernel takes K ranges that are a whole number of 128-deep groups per split and operands that are
    // either k-contiguous + 16-B aligned or row-contiguous (partial edge tiles are fine: clamped rows, guarded stores)
    // pipe_cfg 2 fetches through buffer descriptors with 32-bit offsets (operands below 2 GB); its 16-byte loads need
    // dword alignment only (probed: unaligned buffer_load_dwordx4 returns the right dwords), so k-contiguous operands with
    // any leading dimension qualify, and a K TAIL is admitted when one operand has k as its slow dimension: the K range
    // runs to the next multiple of the stage group, that operand's descriptor returns zeros beyond its end (K = 445 and
    // 257 of the backward sweeps: 27 - 43 us on the generic kernel).  HM_GEMM_KTAIL=0: generic kernel for those (A/B).
    const int64_t bytesA = 4 * ((transA ? K - 1 : M - 1) * lda + (transA ? M : K)),
                  bytesB = 4 * ((transB ? N - 1 : K - 1) * ldb + (transB ? K : N));
    static const int ktail_cfg = [] { const char *e = getenv("HM_GEMM_KTAIL"); return e ? atoi(e) : 1; }();
    const bool buf_ok = pipe_cfg == 2 && bytesA < (1ll << 31) && bytesB < (1ll << 31);
    const bool k_whole = K % (kPipeBK * kPipeD) == 0;
    // (K >= 192: below that the rounded-up range costs more than the generic kernel's guards - K = 72 of the filter banks)
    // (and only where the generic kernel would not split K over workgroups: a tail runs as ONE chunk, so the small
    //  M x N weight-gradient shapes of the eager path - K = number of points - keep their split-K launch)
    const bool k_tail_ok = buf_ok && ktail_cfg != 0 && (!a_kc || !b_kc) && K >= 192 &&
                           (g.ep.mode != HM_EPI_NONE || t64 >= 256);
    const bool use_pipe = !big && small_cfg == 0 && !half_rows && pipe_cfg != 0 && K > 0 &&
                          (buf_ok ? (k_whole || k_tail_ok) : (k_whole && (!a_kc || g.vecA) && (!b_kc || g.vecB)));
    g.nrecA = (int32_t)(bytesA < 0x7fffffff ? bytesA : 0x7fffffff);
    g.nrecB = (int32_t)(bytesB < 0x7fffffff ? bytesB : 0x7fffffff);
    // 96-row tiles when they need fewer rounds of the chip per row of the tile (M = 3072, N = 512: 384 tiles of 64 rows -
    // the slowest CU runs two = 128 rows' worth - against 256 tiles of 96).  HM_GEMM_M96=0: always 64-row tiles (A/B).
    static const int m96_cfg = [] { const char *e = getenv("HM_GEMM_M96"); return e ? atoi(e) : 1; }();
    const int64_t t96 = ((M + 95) / 96) * ((N + 63) / 64);
    const bool m96 = use_pipe && pipe_cfg == 2 && m96_cfg != 0 && ((t96 + 255) / 256) * 96 < ((t64 + 255) / 256) * 64;
    const int64_t bm = big ? 128 : (half_rows ? 32 : (m96 ? 96 : 64)), bn = big ? 128 : 64;
    const int64_t kBK = big ? 32 : (use_pipe ? kPipeBK * kPipeD : (small_cfg == 1 || small_cfg == 2 ? 64 : 128));
    const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    int64_t split = 1;
    if (tiles < 256 && K >= 256 && g.ep.mode == HM_EPI_NONE && !(use_pipe && !k_whole)) {   // (a nonlinear epilogue needs the
                                                                                          // full sum; a K tail is not split)
        static const int split_target = [] { const char *e = getenv("HM_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 512; }();
        split = (split_target + tiles - 1) / tiles;
        const int64_t max_split = K / 128;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    int64_t k_chunk = (K + split - 1) / split;
    k_chunk = ((k_chunk + kBK - 1) / kBK) * kBK;
    if (k_chunk == 0) k_chunk = kBK;
    // the pipelined kernel has no K tail: grow the chunk until it divides K (K = 6144 over 10 splits: 640 -> 768)
    while (use_pipe && k_whole && K % k_chunk != 0 && k_chunk < K) k_chunk += kBK;
    split = K > 0 ? (K + k_chunk - 1) / k_chunk : 1;
    g.k_chunk = (int)k_chunk;
    // (whole K: the chunks divide it; K tail: ONE chunk of the rounded-up K, zeros beyond the slow operand's end)
    const bool pipe_ok = use_pipe && (k_whole ? K % k_chunk == 0 : split == 1);
    g.atomic = (accumulate || split > 1) ? 1 : 0;
    if (split > 1 && !accumulate) {
        // split-K accumulates with atomics into a zeroed C
        hipLaunchKernelGGL(zero_window_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, as_stream(stream),
                           C, ldc, M, N);
    }
    dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)split);
    HM_CHECK_ARG(grid.y <= 65535u && grid.z <= 65535u, "hm_gemm_f32: N or split too large for one launch");
#define HM_GEMM_LAUNCH(TM_, TN_, BK_, KS_, WM_)                                                                     \
    do {                                                                                                        \
        if (g.ep.mode != HM_EPI_NONE) {                                                                         \
            if (g.vecA && g.vecB)                                                                               \
                hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, true, true, true, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
            else if (g.vecA)                                                                                    \
                hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, true, false, true, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
            else if (g.vecB)                                                                                    \
                hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, false, true, true, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
            else                                                                                                \
                hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, false, false, true, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
        } else if (g.vecA && g.vecB)                                                                            \
            hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, true, true, false, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
        else if (g.vecA)                                                                                        \
            hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, true, false, false, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
        else if (g.vecB)                                                                                        \
            hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, false, true, false, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
        else                                                                                                    \
            hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_, BK_, KS_, false, false, false, WM_>), grid, dim3(128 * WM_ * KS_), 0, st, g); \
    } while (0)
    hipStream_t st = as_stream(stream);
    if (big)
        HM_GEMM_LAUNCH(2, 2, 32, 1, 2);
    else if (small_cfg == 1)
        HM_GEMM_LAUNCH(1, 1, 64, 1, 2);
    else if (small_cfg == 2)
        HM_GEMM_LAUNCH(1, 1, 64, 2, 2);
    else if (half_rows)
        HM_GEMM_LAUNCH(1, 1, 128, 4, 1);
    else if (pipe_ok) {
#define HM_PIPE_LAUNCH(AKC_, BKC_)                                                                                \
    do {                                                                                                          \
        if (m96) {                                                                                                \
            if (g.ep.mode != HM_EPI_NONE)                                                                         \
                hipLaunchKernelGGL((gemm_f32_pipe2_m96_kernel<AKC_, BKC_, true>), grid, dim3(768), 0, st, g);     \
            else                                                                                                  \
                hipLaunchKernelGGL((gemm_f32_pipe2_m96_kernel<AKC_, BKC_, false>), grid, dim3(768), 0, st, g);    \
        } else if (pipe_cfg == 2) {                                                                               \
            if (g.ep.mode != HM_EPI_NONE)                                                                         \
                hipLaunchKernelGGL((gemm_f32_pipe2_kernel<AKC_, BKC_, true>), grid, dim3(512), 0, st, g);         \
            else                                                                                                  \
                hipLaunchKernelGGL((gemm_f32_pipe2_kernel<AKC_, BKC_, false>), grid, dim3(512), 0, st, g);        \
        } else if (g.ep.mode != HM_EPI_NONE)                                                                      \
            hipLaunchKernelGGL((gemm_f32_pipe_kernel<AKC_, BKC_, true>), grid, dim3(256), 0, st, g);              \
        else                                                                                                      \
            hipLaunchKernelGGL((gemm_f32_pipe_kernel<AKC_, BKC_, false>), grid, dim3(256), 0, st, g);             \
    } while (0)
        if (a_kc && b_kc) HM_PIPE_LAUNCH(true, true);
        else if (a_kc) HM_PIPE_LAUNCH(true, false);
        else if (b_kc) HM_PIPE_LAUNCH(false, true);
        else HM_PIPE_LAUNCH(false, false);
#undef HM_PIPE_LAUNCH
    } else {
        static const int log_cfg = [] { const char *e = getenv("HM_GEMM_LOG"); return e ? atoi(e) : 0; }();
        if (log_cfg)   // debugging: which shapes miss the pipelined kernel
            fprintf(stderr, "hm_gemm_f32 generic: tA %d tB %d M %lld N %lld K %lld lda %lld ldb %lld ep %d vecA %d vecB %d split %lld\n",
                    transA, transB, (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb, g.ep.mode,
                    g.vecA, g.vecB, (long long)split);
        HM_GEMM_LAUNCH(1, 1, 128, 2, 2);
    }
#undef HM_GEMM_LAUNCH
    HM_CHECK_LAUNCH("hm_gemm_f32");
    return HM_OK;
}

extern "C" {

int hm_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int accumulate,
                void *stream) {
    return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, bias, C, ldc, accumulate, nullptr, stream);
}

int hm_gemm_f32_group_tn(const hm_gemm_group_item *items, int n_items, void *stream) {
    HM_CHECK_ARG(n_items >= 0 && (n_items == 0 || items), "hm_gemm_f32_group_tn: bad argument");
    GemmGroupTable t;
    t.n = 0;
    t.start[0] = 0;
    auto flush = [&]() -> int {
        if (t.n > 0) {
            hipLaunchKernelGGL(gemm_f32_pipe2_group_kernel, dim3((unsigned)t.start[t.n]), dim3(512), 0, as_stream(stream), t);
            HM_CHECK_LAUNCH("hm_gemm_f32_group_tn");
        }
        t.n = 0;
        t.start[0] = 0;
        return HM_OK;
    };
    for (int i = 0; i < n_items; ++i) {
        const hm_gemm_group_item &it = items[i];
        HM_CHECK_ARG(it.M >= 0 && it.N >= 0 && it.K >= 0 && it.M < (1ll << 31) && it.N < (1ll << 31) && it.K < (1ll << 31),
                     "hm_gemm_f32_group_tn: bad dimension");
        if (it.M == 0 || it.N == 0 || it.K == 0) continue;
        HM_CHECK_ARG(it.A && it.B && it.C && it.lda >= it.M && it.ldb >= it.N && it.ldc >= it.N,
                     "hm_gemm_f32_group_tn: NULL operand or leading dimension");
        if (it.K % (kPipeBK * kPipeD) != 0 || 4 * it.lda * it.K >= (1ll << 31) || 4 * it.ldb * it.K >= (1ll << 31)) {
            // no K tail in the pipelined kernel, 32-bit operand offsets: such a problem goes alone
            const int rc = gemm_impl(1, 0, it.M, it.N, it.K, it.A, it.lda, it.B, it.ldb, nullptr, it.C, it.ldc, 1, nullptr, stream);
            if (rc != HM_OK) return rc;
            continue;
        }
        if (t.n == HM_GEMM_GROUP_MAX) {
            const int rc = flush();
            if (rc != HM_OK) return rc;
        }
        GemmGroupEntry &E = t.e[t.n];
        E.A = it.A; E.B = it.B; E.C = it.C;
        E.lda = it.lda; E.ldb = it.ldb; E.ldc = it.ldc;
        E.M = (int32_t)it.M; E.N = (int32_t)it.N;
        E.tiles_m = (int32_t)((it.M + 63) / 64);
        E.tiles_n = (int32_t)((it.N + 63) / 64);
        // k parts of >= 1024 (eight 128-deep groups) that divide K
        int64_t split = it.K / 1024;
        if (split < 1) split = 1;
        if (split > 4) split = 4;
        while (split > 1 && (it.K % split != 0 || (it.K / split) % (kPipeBK * kPipeD) != 0)) --split;
        E.split = (int32_t)split;
        E.k_chunk = (int32_t)(it.K / split);
        t.start[t.n + 1] = t.start[t.n] + E.tiles_m * E.tiles_n * E.split;
        ++t.n;
    }
    return flush();
}

int hm_gemm_f32_ep(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                   const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                   const hm_gemm_epilogue *ep, void *stream) {
    HM_CHECK_ARG(ep != nullptr, "hm_gemm_f32_ep: epilogue is NULL");
    return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, bias, C, ldc, 0, ep, stream);
}

}  // extern "C"
