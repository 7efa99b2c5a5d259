// hm_pack.hip - builds the MFMA operand images of one folded MLP layer (layout contract: include/hashmod.h,
// hm_mlp_layer) in ONE pass: both packed images (32-row x 8-k tiles for the 64-point kernel, 16-row x 16-k
// tiles for the 16-point kernel) and the zero-padded bias.  Runs once per optimizer step per layer; the torch
// expression of the same permutation costs ~10 small kernels per layer.
#include "hm_common.h"

namespace {

struct PackArgs {
    const float *W;     // [out_dim, k_real] folded weights (row stride ldw)
    const float *bias;  // [out_dim]
    float *img8, *img16, *bias_out;
    int64_t ldw;
    int32_t out_dim, n_tiles, w0, w1;  // real widths of the two K segments (w1 may be 0)
    int32_t p8_0, n_oct;               // padded width of segment 0 / total octets   (8-k image)
    int32_t p16_0, nb;                 // padded width of segment 0 / total 16-blocks (16-k image)
};

__device__ __forceinline__ float fetch(const PackArgs &a, int row, int kpos, int p0) {
    int col;
    if (kpos < p0)
        col = kpos < a.w0 ? kpos : -1;
    else {
        const int kk = kpos - p0;
        col = kk < a.w1 ? a.w0 + kk : -1;
    }
    return (row < a.out_dim && col >= 0) ? a.W[(int64_t)row * a.ldw + col] : 0.0f;
}

__device__ __forceinline__ void pack_layer_body(const PackArgs &a, int64_t d) {
    const int64_t n8 = (int64_t)a.n_tiles * a.n_oct * 256;      // floats in the 8-k image
    const int64_t n16 = (int64_t)a.n_tiles * 2 * a.nb * 256;    // floats in the 16-k image
    if (d < n8) {
        const int s = d & 3, l = (d >> 2) & 63;
        const int64_t blk = d >> 8;
        const int g = (int)(blk % a.n_oct), u = (int)(blk / a.n_oct);
        a.img8[d] = fetch(a, 32 * u + (l & 31), 8 * g + 4 * (l >> 5) + s, a.p8_0);
    } else if (d < n8 + n16) {
        const int64_t q = d - n8;
        const int e = q & 3, l = (q >> 2) & 63;
        const int64_t blk = q >> 8;
        const int t = (int)(blk % a.nb), u = (int)(blk / a.nb);
        a.img16[q] = fetch(a, 16 * u + (l & 15), 16 * t + 4 * (l >> 4) + e, a.p16_0);
    } else {
        const int64_t r = d - n8 - n16;
        if (r < (int64_t)a.n_tiles * 32) a.bias_out[r] = r < a.out_dim ? a.bias[r] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void pack_layer_kernel(PackArgs a) {
    pack_layer_body(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// every layer of a network in ONE launch (blockIdx.y = layer): the per-step re-pack of the 9 SDF layers was 9 launches
struct PackTable {
    PackArgs a[HM_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void pack_layer_multi_kernel(PackTable t) {
    pack_layer_body(t.a[blockIdx.y], (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// bf16 operand image for v_mfma_f32_32x32x16_bf16 (hm_sdf_bf16.hip): same K space as the 16-k image,
//   w_bf16[((u*nb + t)*64 + l)*8 + j] = bf16(W[32u + (l&31)][16t + 8(l>>5) + j]),  u < n_tiles, t < nb
__global__ __launch_bounds__(256) void pack_layer_bf16_kernel(PackArgs a, __bf16 *img) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.n_tiles * a.nb * 512;
    if (d >= total) return;
    const int jj = d & 7, l = (d >> 3) & 63;
    const int64_t blk = d >> 9;
    const int t = (int)(blk % a.nb), u = (int)(blk / a.nb);
    img[d] = (__bf16)fetch(a, 32 * u + (l & 31), 16 * t + 8 * (l >> 5) + jj, a.p16_0);   // round to nearest even
}

}  // namespace

extern "C" {

int hm_pack_mlp_layer_bf16(const float *W, int64_t ldw, int out_dim, int seg_width0, int seg_width1, void *w_packed_bf16,
                           void *stream) {
    HM_CHECK_ARG(W && w_packed_bf16, "hm_pack_mlp_layer_bf16: NULL pointer");
    HM_CHECK_ARG(out_dim >= 1 && seg_width0 >= 1 && seg_width1 >= 0 && ldw >= seg_width0 + seg_width1,
                 "hm_pack_mlp_layer_bf16: bad shape");
    PackArgs a = {};
    a.W = W; a.ldw = ldw; a.out_dim = out_dim;
    a.n_tiles = (out_dim + 31) / 32;
    a.w0 = seg_width0; a.w1 = seg_width1;
    a.p16_0 = (seg_width0 + 15) / 16 * 16;
    a.nb = a.p16_0 / 16 + (seg_width1 + 15) / 16;
    const int64_t total = (int64_t)a.n_tiles * a.nb * 512;
    hipLaunchKernelGGL(pack_layer_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, static_cast<__bf16 *>(w_packed_bf16));
    HM_CHECK_LAUNCH("hm_pack_mlp_layer_bf16");
    return HM_OK;
}

static int fill_pack_args(PackArgs &a, const float *W, int64_t ldw, const float *bias, int out_dim, int seg_width0,
                          int seg_width1, float *w_packed, float *w_packed_m16, float *bias_padded, int64_t &total) {
    HM_CHECK_ARG(W && bias && w_packed && w_packed_m16 && bias_padded, "hm_pack_mlp_layer: NULL pointer");
    HM_CHECK_ARG(out_dim >= 1 && seg_width0 >= 1 && seg_width1 >= 0 && ldw >= seg_width0 + seg_width1,
                 "hm_pack_mlp_layer: bad shape");
    a.W = W; a.bias = bias; a.img8 = w_packed; a.img16 = w_packed_m16; a.bias_out = bias_padded;
    a.ldw = ldw;
    a.out_dim = out_dim;
    a.n_tiles = (out_dim + 31) / 32;
    a.w0 = seg_width0; a.w1 = seg_width1;
    a.p8_0 = (seg_width0 + 7) / 8 * 8;
    a.n_oct = a.p8_0 / 8 + (seg_width1 + 7) / 8;
    a.p16_0 = (seg_width0 + 15) / 16 * 16;
    a.nb = a.p16_0 / 16 + (seg_width1 + 15) / 16;
    total = (int64_t)a.n_tiles * a.n_oct * 256 + (int64_t)a.n_tiles * 2 * a.nb * 256 + a.n_tiles * 32;
    return HM_OK;
}

int hm_pack_mlp_layer(const float *W, int64_t ldw, const float *bias, int out_dim, int seg_width0, int seg_width1,
                      float *w_packed, float *w_packed_m16, float *bias_padded, void *stream) {
    PackArgs a;
    int64_t total = 0;
    const int rc = fill_pack_args(a, W, ldw, bias, out_dim, seg_width0, seg_width1, w_packed, w_packed_m16, bias_padded, total);
    if (rc != HM_OK) return rc;
    hipLaunchKernelGGL(pack_layer_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    HM_CHECK_LAUNCH("hm_pack_mlp_layer");
    return HM_OK;
}

int hm_pack_mlp_layers(const hm_pack_item *items, int n_items, void *stream) {
    HM_CHECK_ARG(n_items >= 0 && n_items <= HM_MAX_LAYERS && (n_items == 0 || items), "hm_pack_mlp_layers: bad argument");
    if (n_items == 0) return HM_OK;
    PackTable t;
    int64_t max_total = 0;
    for (int i = 0; i < n_items; ++i) {
        int64_t total = 0;
        const hm_pack_item &I = items[i];
        const int rc = fill_pack_args(t.a[i], I.W, I.ldw, I.bias, I.out_dim, I.seg_width0, I.seg_width1, I.w_packed,
                                      I.w_packed_m16, I.bias_padded, total);
        if (rc != HM_OK) return rc;
        max_total = total > max_total ? total : max_total;
    }
    for (int i = n_items; i < HM_MAX_LAYERS; ++i) t.a[i] = t.a[0];
    hipLaunchKernelGGL(pack_layer_multi_kernel, dim3((unsigned)((max_total + 255) / 256), (unsigned)n_items), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), t);
    HM_CHECK_LAUNCH("hm_pack_mlp_layers");
    return HM_OK;
}

}  // extern "C"
