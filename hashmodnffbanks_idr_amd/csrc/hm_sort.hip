// hm_sort.hip - stable LSD radix sort of (row id, contribution index) pairs for the deterministic table backward.
//
// hm_encode_bwd_table_sorted (csrc/hm_encode.hip) sums the contributions of one table row in the order of a STABLE sort
// of their destination rows - the bitwise-reproducible form of the embedding_dense_backward autograd runs for
// nn.Embedding in _HashGridMLP.forward (reference: model/embeddings/hashGridEmbedding.py:99-102; its CPU path, the one
// the oracle pins, is deterministic too).  Round 2 sorted with a vendor library call; this is the library's own sort:
// 8-bit digits, ceil(key_bits / 8) passes, each pass three launches
//   radix_hist     per 4096-key chunk: digit histogram -> hist[digit][chunk]
//   radix_scan     exclusive prefix over (digit-major, chunk-minor) = first output slot of every (digit, chunk)
//   radix_scatter  per chunk, 16 rounds of 256 keys in input order: a key's slot is
//                  base[digit][chunk] + (keys of the same digit earlier in the chunk); "earlier in the round" comes from
//                  eight wave ballots (match-any on the digit), "earlier rounds / waves" from LDS counters
// Stable by construction, no atomics on the data path, every launch a plain kernel (graph-capturable).
#include "hm_common.h"

namespace {

constexpr int kST = 256;              // threads per workgroup
constexpr int kRounds = 16;           // rounds of kST keys per chunk
constexpr int kChunk = kST * kRounds; // 4096 keys per workgroup

__global__ __launch_bounds__(kST) void radix_hist_kernel(const int32_t *__restrict__ keys, int64_t n, int shift,
                                                         int64_t n_chunks, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t beg = (int64_t)blockIdx.x * kChunk, end = min(beg + kChunk, n);
    for (int64_t i = beg + threadIdx.x; i < end; i += kST) atomicAdd(&h[((uint32_t)keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_chunks + blockIdx.x] = h[threadIdx.x];
}

// exclusive prefix sum of `total` uint32 values in place (one workgroup of 1024 threads, each owning a contiguous slice)
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *__restrict__ v, int64_t total) {
    __shared__ uint32_t part[1024];
    const int64_t per = (total + 1023) / 1024;
    const int64_t beg = min((int64_t)threadIdx.x * per, total), end = min(beg + per, total);
    uint32_t s = 0u;
    for (int64_t i = beg; i < end; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;     // exclusive prefix of this thread's slice
    for (int64_t i = beg; i < end; ++i) {
        const uint32_t x = v[i];
        v[i] = run;
        run += x;
    }
}

// vals_in == nullptr: the value of key i is i (first pass).  perm64_out != nullptr: values leave as int64 (last pass).
__global__ __launch_bounds__(kST) void radix_scatter_kernel(const int32_t *__restrict__ keys_in,
                                                            const int32_t *__restrict__ vals_in, int64_t n, int shift,
                                                            int64_t n_chunks, const uint32_t *__restrict__ base,
                                                            int32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out,
                                                            int64_t *__restrict__ perm64_out) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t cnt[kST / 64][256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    running[tid] = base[(int64_t)tid * n_chunks + blockIdx.x];
    const int64_t beg = (int64_t)blockIdx.x * kChunk;
    for (int r = 0; r < kRounds; ++r) {
        const int64_t i = beg + (int64_t)r * kST + tid;
        const bool valid = i < n;
        const int32_t key = valid ? keys_in[i] : 0;
        const uint32_t d = ((uint32_t)key >> shift) & 255u;
#pragma unroll
        for (int w = 0; w < kST / 64; ++w) cnt[w][tid] = 0u;
        __syncthreads();
        // lanes of this wave that hold the same digit (match-any by eight ballots), among the valid ones
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (valid && rank == 0u) cnt[wave][d] = (uint32_t)__popcll(m);
        __syncthreads();
        if (valid) {
            uint32_t dest = running[d] + rank;
            for (int w = 0; w < wave; ++w) dest += cnt[w][d];
            const int32_t val = vals_in ? vals_in[i] : (int32_t)i;
            keys_out[dest] = key;
            if (perm64_out) perm64_out[dest] = (int64_t)val;
            else vals_out[dest] = val;
        }
        __syncthreads();
        uint32_t add = 0u;
#pragma unroll
        for (int w = 0; w < kST / 64; ++w) add += cnt[w][tid];
        running[tid] += add;
        __syncthreads();
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t up256(int64_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" {

int64_t hm_sort_workspace_bytes(int64_t n) {
    if (n < 0) return hm_fail(HM_ERR_INVALID, "hm_sort_workspace_bytes: n < 0");
    const int64_t n_chunks = (n + kChunk - 1) / kChunk;
    // [keys tmp n i32 | vals a n i32 | vals b n i32 | hist 256 * n_chunks u32]
    return 3 * up256(4 * n) + up256(4 * 256 * (n_chunks > 0 ? n_chunks : 1));
}

int hm_sort_pairs_i32(const int32_t *keys, int64_t n, int key_bits, int32_t *keys_sorted, int64_t *perm, void *workspace,
                      int64_t workspace_bytes, void *stream) {
    HM_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31), "hm_sort_pairs_i32: n out of range");
    HM_CHECK_ARG(key_bits >= 1 && key_bits <= 31, "hm_sort_pairs_i32: key_bits must be in [1, 31] (non-negative keys)");
    if (n == 0) return HM_OK;
    HM_CHECK_ARG(keys && keys_sorted && perm && workspace, "hm_sort_pairs_i32: NULL pointer");
    HM_CHECK_ARG(workspace_bytes >= hm_sort_workspace_bytes(n), "hm_sort_pairs_i32: workspace too small");
    const int64_t n_chunks = (n + kChunk - 1) / kChunk;
    char *ws = static_cast<char *>(workspace);
    int32_t *ktmp = reinterpret_cast<int32_t *>(ws);
    int32_t *va = reinterpret_cast<int32_t *>(ws + up256(4 * n));
    int32_t *vb = reinterpret_cast<int32_t *>(ws + 2 * up256(4 * n));
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws + 3 * up256(4 * n));
    const int passes = (key_bits + 7) / 8;
    hipStream_t st = as_stream(stream);
    // ping-pong so that the LAST pass writes keys_sorted: with an even number of passes the first one writes the temporary
    const int32_t *kin = keys;
    const int32_t *vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool last = p == passes - 1;
        int32_t *kout = ((passes - 1 - p) % 2 == 0) ? keys_sorted : ktmp;
        int32_t *vout = (p % 2 == 0) ? va : vb;
        hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)n_chunks), dim3(kST), 0, st, kin, n, 8 * p, n_chunks, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, st, hist, 256 * n_chunks);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)n_chunks), dim3(kST), 0, st, kin, vin, n, 8 * p, n_chunks,
                           static_cast<const uint32_t *>(hist), kout, vout, last ? perm : nullptr);
        kin = kout;
        vin = vout;
    }
    HM_CHECK_LAUNCH("hm_sort_pairs_i32");
    return HM_OK;
}

}  // extern "C"
