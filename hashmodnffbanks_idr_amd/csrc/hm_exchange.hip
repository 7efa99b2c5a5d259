// hm_exchange.hip - device side of the data-parallel gradient exchange (parallel.StaticGradExchange).
//
// The reference is single-GPU; rays shard across ranks and the only coupling is ONE gradient exchange
// per iteration, after backward and before clip_grad_norm_ / Adam (training/idr_train.py:302-308).  Everything on the
// device side of that exchange is a kernel here, so that it can sit inside the two captured graphs of a training step
// and only the collectives themselves run between them:
//   hm_multi_copy_f32   the MLP gradients -> one flat bucket (one launch; the bucket is all-reduced in place and the
//                       optimizer reads the averaged gradients straight from it)
//   hm_rows_pack        the rank's hash-table gradient -> (row, value) pairs of the rows hm_encode_bwd_table_tracked
//                       listed (<= 0.4 % of the table), the dense gradient is zeroed again on the way
//   hm_rows_apply       all ranks' pairs (after ONE all-gather) -> dense gradient.  List r is added by launch r with
//                       plain read-modify-writes (rows are unique inside a list), so the sum of every row runs in rank
//                       order on every replica: bitwise identical replicas without sorting anything
//   hm_rows_clear       after the optimizer step: the touched rows of the dense gradient back to zero
#include "hm_common.h"

namespace {

constexpr int kMaxItems = HM_COPY_MAX_ITEMS;
constexpr int kChunk = 8192;   // floats per workgroup
constexpr int kT = 256;

struct CopyTable {
    hm_copy_item it[kMaxItems];
    int32_t chunk_start[kMaxItems + 1];
    int32_t n;
};

__global__ __launch_bounds__(kT) void multi_copy_kernel(CopyTable tb) {
    int lo = 0, hi = tb.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tb.chunk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const hm_copy_item T = tb.it[lo];
    const int64_t beg = (int64_t)((int)blockIdx.x - tb.chunk_start[lo]) * kChunk;
    const int64_t end = min(beg + kChunk, T.numel);
    if (((reinterpret_cast<uintptr_t>(T.src) | reinterpret_cast<uintptr_t>(T.dst)) & 15u) == 0) {
        const int64_t e4 = beg + ((end - beg) & ~(int64_t)3);
        const float4 *s4 = reinterpret_cast<const float4 *>(T.src);
        float4 *d4 = reinterpret_cast<float4 *>(T.dst);
        for (int64_t i = beg + 4 * threadIdx.x; i < e4; i += 4 * kT) d4[i >> 2] = s4[i >> 2];
        for (int64_t i = e4 + threadIdx.x; i < end; i += kT) T.dst[i] = T.src[i];
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += kT) T.dst[i] = T.src[i];
    }
}

// payload entry = [row (int32) | F values (fp32 bits)]
__global__ __launch_bounds__(kT) void rows_pack_kernel(float *__restrict__ d_table, int F,
                                                       const int32_t *__restrict__ rows, const int32_t *__restrict__ count,
                                                       int64_t cap, uint32_t *__restrict__ bits,
                                                       int32_t *__restrict__ payload, int32_t *__restrict__ status) {
    const int64_t slot = (int64_t)blockIdx.x * kT + threadIdx.x;
    const int64_t cnt = *count;
    if (slot == 0 && status && cnt > cap) atomicMax(status, (int32_t)min(cnt - cap, (int64_t)0x7fffffff));
    if (slot >= cap) return;
    int32_t *e = payload + slot * (1 + F);
    if (slot < cnt) {
        const int32_t row = rows[slot];
        e[0] = row;
        float *src = d_table + (int64_t)row * F;
        for (int f = 0; f < F; ++f) {
            e[1 + f] = __float_as_int(src[f]);
            src[f] = 0.0f;
        }
        atomicAnd(bits + ((uint32_t)row >> 5), ~(1u << ((uint32_t)row & 31u)));
    } else {
        e[0] = -1;
        for (int f = 0; f < F; ++f) e[1 + f] = 0;
    }
}

__global__ __launch_bounds__(kT) void rows_apply_kernel(float *__restrict__ d_table, int64_t total_rows, int F,
                                                        const int32_t *__restrict__ list, int64_t cap, float scale) {
    const int64_t slot = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (slot >= cap) return;
    const int32_t *e = list + slot * (1 + F);
    const int32_t row = e[0];
    if (row < 0 || (int64_t)row >= total_rows) return;
    float *dst = d_table + (int64_t)row * F;
    for (int f = 0; f < F; ++f) dst[f] = __fadd_rn(dst[f], __fmul_rn(__int_as_float(e[1 + f]), scale));
}

__global__ __launch_bounds__(kT) void rows_clear_kernel(float *__restrict__ d_table, int64_t total_rows, int F,
                                                        const int32_t *__restrict__ lists, int64_t cap,
                                                        int64_t list_stride, int n_lists, int32_t *__restrict__ count) {
    const int64_t gid = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (gid == 0 && count) *count = 0;
    if (gid >= cap * n_lists) return;
    const int64_t r = gid / cap, slot = gid - r * cap;
    const int32_t row = lists[r * list_stride + slot * (1 + F)];
    if (row < 0 || (int64_t)row >= total_rows) return;
    float *dst = d_table + (int64_t)row * F;
    for (int f = 0; f < F; ++f) dst[f] = 0.0f;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int hm_multi_copy_f32(const hm_copy_item *items, int n_items, void *stream) {
    HM_CHECK_ARG(n_items >= 0 && (n_items == 0 || items), "hm_multi_copy_f32: bad argument");
    for (int i = 0; i < n_items; ++i)
        HM_CHECK_ARG(items[i].numel >= 0 && (items[i].numel == 0 || (items[i].src && items[i].dst)),
                     "hm_multi_copy_f32: bad item");
    CopyTable tb;
    for (int first = 0; first < n_items;) {
        tb.n = 0;
        tb.chunk_start[0] = 0;
        int i = first;
        for (; i < n_items && tb.n < kMaxItems; ++i) {
            if (items[i].numel == 0) continue;
            const int64_t chunks = (items[i].numel + kChunk - 1) / kChunk;
            if ((int64_t)tb.chunk_start[tb.n] + chunks > (int64_t)0x7fffffff) break;
            tb.it[tb.n] = items[i];
            tb.chunk_start[tb.n + 1] = tb.chunk_start[tb.n] + (int32_t)chunks;
            ++tb.n;
        }
        HM_CHECK_ARG(i > first, "hm_multi_copy_f32: tensor too large for one launch");
        if (tb.n > 0)
            hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)tb.chunk_start[tb.n]), dim3(kT), 0, as_stream(stream), tb);
        first = i;
    }
    HM_CHECK_LAUNCH("hm_multi_copy_f32");
    return HM_OK;
}

int hm_rows_pack(float *d_table, int n_features, const int32_t *touched_rows, const int32_t *touched_count, int64_t cap,
                 uint32_t *touched_bits, int32_t *payload, int32_t *status, void *stream) {
    HM_CHECK_ARG(n_features >= 1 && n_features <= 8 && cap >= 0, "hm_rows_pack: bad argument");
    if (cap == 0) return HM_OK;
    HM_CHECK_ARG(d_table && touched_rows && touched_count && touched_bits && payload, "hm_rows_pack: NULL pointer");
    const int64_t grid = (cap + kT - 1) / kT;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_rows_pack: cap too large");
    hipLaunchKernelGGL(rows_pack_kernel, dim3((unsigned)grid), dim3(kT), 0, as_stream(stream), d_table, n_features,
                       touched_rows, touched_count, cap, touched_bits, payload, status);
    HM_CHECK_LAUNCH("hm_rows_pack");
    return HM_OK;
}

int hm_rows_apply(float *d_table, int64_t total_rows, int n_features, const int32_t *lists, int64_t cap,
                  int64_t list_stride, int n_lists, float scale, void *stream) {
    HM_CHECK_ARG(n_features >= 1 && n_features <= 8 && cap >= 0 && n_lists >= 0 && total_rows >= 0,
                 "hm_rows_apply: bad argument");
    if (cap == 0 || n_lists == 0) return HM_OK;
    HM_CHECK_ARG(d_table && lists && list_stride >= cap * (1 + n_features), "hm_rows_apply: bad list layout");
    const int64_t grid = (cap + kT - 1) / kT;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_rows_apply: cap too large");
    for (int r = 0; r < n_lists; ++r)    // one launch per rank: stream order IS the summation order of every row
        hipLaunchKernelGGL(rows_apply_kernel, dim3((unsigned)grid), dim3(kT), 0, as_stream(stream), d_table, total_rows,
                           n_features, lists + (int64_t)r * list_stride, cap, scale);
    HM_CHECK_LAUNCH("hm_rows_apply");
    return HM_OK;
}

int hm_rows_clear(float *d_table, int64_t total_rows, int n_features, const int32_t *lists, int64_t cap,
                  int64_t list_stride, int n_lists, int32_t *touched_count, void *stream) {
    HM_CHECK_ARG(n_features >= 1 && n_features <= 8 && cap >= 0 && n_lists >= 0 && total_rows >= 0,
                 "hm_rows_clear: bad argument");
    const int64_t total = cap * n_lists;
    if (total == 0 && !touched_count) return HM_OK;
    HM_CHECK_ARG(total == 0 || (d_table && lists && list_stride >= cap * (1 + n_features)), "hm_rows_clear: bad list layout");
    const int64_t grid = (max(total, (int64_t)1) + kT - 1) / kT;
    HM_CHECK_ARG(grid <= 0x7fffffffLL, "hm_rows_clear: too many entries");
    hipLaunchKernelGGL(rows_clear_kernel, dim3((unsigned)grid), dim3(kT), 0, as_stream(stream), d_table, total_rows,
                       n_features, lists, cap, list_stride, n_lists, touched_count);
    HM_CHECK_LAUNCH("hm_rows_clear");
    return HM_OK;
}

}  // extern "C"
