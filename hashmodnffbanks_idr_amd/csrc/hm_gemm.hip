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
// [k/4][row][4] so that every MFMA operand fetch is a conflict-free ds_read_b128.
#include "hm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;  // k per stage = 4 k-groups = 2 MFMA octets
constexpr int kGemmThreads = 256;

struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int32_t M, N, K;
    int32_t transA, transB;  // op(A) = A^T if transA (A stored [K,M]); op(B) = B^T if transB (B stored [N,K])
    int64_t lda, ldb, ldc;
    int32_t k_chunk;  // K range handled by one blockIdx.z
    int32_t atomic;   // accumulate with atomics (split-K, or beta = 1)
};

// stage one operand tile: ROWS x kBK elements of op(X) into S[kgroup][row][4]
template <int ROWS>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int64_t ld, bool row_contig_k, int row0,
                                          int nrows, int k0, int kend, int tid, float (&r)[ROWS * kBK / kGemmThreads]) {
    constexpr int PER = ROWS * kBK / kGemmThreads;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + kGemmThreads * i;
        int row, k;
        if (row_contig_k) {  // element (row, k) at P[row*ld + k]
            row = e / kBK;
            k = e % kBK;
        } else {  // element (row, k) at P[k*ld + row]
            k = e / ROWS;
            row = e % ROWS;
        }
        const int gr = row0 + row, gk = k0 + k;
        float v = 0.0f;
        if (gr < nrows && gk < kend) v = row_contig_k ? P[(int64_t)gr * ld + gk] : P[(int64_t)gk * ld + gr];
        r[i] = v;
    }
}

template <int ROWS>
__device__ __forceinline__ void store_tile(float *__restrict__ S, bool row_contig_k, int tid,
                                           const float (&r)[ROWS * kBK / kGemmThreads]) {
    constexpr int PER = ROWS * kBK / kGemmThreads;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + kGemmThreads * i;
        int row, k;
        if (row_contig_k) {
            row = e / kBK;
            k = e % kBK;
        } else {
            k = e / ROWS;
            row = e % ROWS;
        }
        S[((k >> 2) * ROWS + row) * 4 + (k & 3)] = r[i];
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(kGemmThreads) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ __align__(16) float As[(kBK / 4) * BM * 4];
    __shared__ __align__(16) float Bs[(kBK / 4) * BN * 4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
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

    float ra[BM * kBK / kGemmThreads], rb[BN * kBK / kGemmThreads];
    load_tile<BM>(g.A, g.lda, a_kc, m0, g.M, kbeg, kend, tid, ra);
    load_tile<BN>(g.B, g.ldb, b_kc, n0, g.N, kbeg, kend, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += kBK) {
        __syncthreads();  // previous stage fully consumed
        store_tile<BM>(As, a_kc, tid, ra);
        store_tile<BN>(Bs, b_kc, tid, rb);
        __syncthreads();
        if (k0 + kBK < kend) {  // prefetch the next stage into registers while this one is multiplied
            load_tile<BM>(g.A, g.lda, a_kc, m0, g.M, k0 + kBK, kend, tid, ra);
            load_tile<BN>(g.B, g.ldb, b_kc, n0, g.N, k0 + kBK, kend, tid, rb);
        }
#pragma unroll
        for (int o = 0; o < kBK / 8; ++o) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t] = *reinterpret_cast<const float4 *>(As + ((2 * o + h) * BM + wm * 32 * TM + t * 32 + j) * 4);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                b[t] = *reinterpret_cast<const float4 *>(Bs + ((2 * o + h) * BN + wn * 32 * TN + t * 32 + j) * 4);
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

    // epilogue: lane holds column n, registers hold rows (r&3) + 8(r>>2) + 4h
    const bool add_bias = (g.bias != nullptr) && (blockIdx.z == 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * 32 * TN + tn * 32 + j;
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                const float v = acc[tm][tn][r] + bv;
                float *dst = g.C + (int64_t)m * g.ldc + n;
                if (g.atomic)
                    atomicAdd(dst, v);
                else
                    *dst = v;
            }
        }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int hm_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int accumulate,
                void *stream) {
    HM_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "hm_gemm_f32: negative dimension");
    HM_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "hm_gemm_f32: dimension too large");
    if (M == 0 || N == 0) return HM_OK;
    HM_CHECK_ARG(C != nullptr, "hm_gemm_f32: C is NULL");
    HM_CHECK_ARG(K == 0 || (A && B), "hm_gemm_f32: NULL operand");
    HM_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "hm_gemm_f32: leading dimension");
    GemmArgs g;
    g.A = A; g.B = B; g.bias = bias; g.C = C;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.transA = transA; g.transB = transB;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    // tile choice: big tiles only when they still fill the chip
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    const bool big = t128 >= 192;
    const int64_t bm = big ? 128 : 64, bn = big ? 128 : 64;
    const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    int64_t split = 1;
    if (tiles < 256 && K >= 256) {
        split = (512 + tiles - 1) / tiles;
        const int64_t max_split = K / 128;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    int64_t k_chunk = (K + split - 1) / split;
    k_chunk = ((k_chunk + kBK - 1) / kBK) * kBK;
    if (k_chunk == 0) k_chunk = kBK;
    split = K > 0 ? (K + k_chunk - 1) / k_chunk : 1;
    g.k_chunk = (int)k_chunk;
    g.atomic = (accumulate || split > 1) ? 1 : 0;
    if (split > 1 && !accumulate) {
        // split-K accumulates with atomics into a zeroed C
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M,
                                        as_stream(stream));
        if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipMemset2DAsync: ") + hipGetErrorString(e));
    }
    dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)split);
    HM_CHECK_ARG(grid.y <= 65535u && grid.z <= 65535u, "hm_gemm_f32: N or split too large for one launch");
    if (big)
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<1, 1>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
    HM_CHECK_LAUNCH("hm_gemm_f32");
    return HM_OK;
}

}  // extern "C"
